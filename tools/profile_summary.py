#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small summaries committed under profiles/.
  python tools/profile_summary.py stats  <kernel_stats.csv> <steps> <out.md>
  python tools/profile_summary.py pmc    <fetch_counter.csv> <write_counter.csv> <lines_per_gpu> <out.json>"""
import csv
import json
import sys


def stats(path, steps, out):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary ({path.split('/')[-1]}), {steps} profiled steps\n\n")
        f.write("| kernel | calls/step | avg us | ms/step | % |\n|---|---|---|---|---|\n")
        for r in rows[:40]:
            f.write(f"| `{r['Name'][:90]}` | {int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | "
                    f"{float(r['TotalDurationNs']) / steps / 1e6:.3f} | {float(r['Percentage']):.1f} |\n")
        f.write(f"\ntotal kernel time per step: {tot / steps / 1e6:.3f} ms\n")


def pmc(fetch_csv, write_csv, lines, out):
    def per_kernel(path, counter):
        acc = {}
        for r in csv.DictReader(open(path)):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"].split("(")[0]
            a = acc.setdefault(name, [0.0, 0])
            a[0] += float(r["Counter_Value"])
            a[1] += 1
        return acc
    fe, wr = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    res = {}
    extra = 0.0   # the split-K reduce kernel belongs to its product's launch: its bytes count, its launches do not
    for name in fe:
        if name.startswith("pero_splitk_reduce_k"):
            extra += 2 * fe[name][0] * 1024 + wr.get(name, [0.0, 1])[0] * 1024
            continue
        if not name.startswith("void gemm_bf16"):
            continue
        n = fe[name][1]
        fetch_kb = fe[name][0] / n
        write_kb = wr.get(name, [0.0, 1])[0] / max(wr.get(name, [0.0, 1])[1], 1)
        # guide (MI355X_MICROARCH.md, HBM): on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced
        # streaming read -> doubled; WRITE_SIZE is exact for 16-byte-per-lane stores and float atomics; both in KiB
        res[name] = {"launches": n, "fetch_bytes": 2 * fetch_kb * 1024, "write_bytes": write_kb * 1024}
    tot_l = sum(v["launches"] for v in res.values()) or 1
    avg = (sum((v["fetch_bytes"] + v["write_bytes"]) * v["launches"] for v in res.values()) + extra) / tot_l
    if extra:
        res["pero_splitk_reduce_k (all launches, counted into the products)"] = {"bytes_total": extra}
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench   # the hash of the kernel sources this summary describes: bench.py quotes it only while they are unchanged
    json.dump({"lines_per_gpu": int(lines), "csrc_sha256": bench.csrc_hash(), "hbm_bytes_per_launch": round(avg), "by_kernel": res,
               "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 correction); separate --pmc passes"},
              open(out, "w"), indent=1)


def pmc_legs(base, out):
    """Per leg (gpurun_out/pmcl_<leg>_<COUNTER>/ of tools/pmc_legs.sh): average HBM bytes per launch of the leg's dominant kernel family -
    the codebook argmin kernel for config 3, the bf16 tile GEMMs (split-K reduce passes counted into their products) for configs 4 / 5."""
    import glob, os
    legs = {}
    for leg, pat in (("config3_vq_argmin", "vq_argmin"), ("config4_vicreg_step", "gemm_bf16"), ("config5_ntxent_step", "gemm_bf16")):
        tot = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            files = glob.glob(os.path.join(base, f"pmcl_{leg}_{counter}", "*counter_collection.csv"))
            if not files:
                break
            acc, n, extra = 0.0, 0, 0.0
            for r in csv.DictReader(open(files[0])):
                if r.get("Counter_Name") != counter:
                    continue
                name = r["Kernel_Name"]
                if pat in name:
                    acc += float(r["Counter_Value"]); n += 1
                elif pat == "gemm_bf16" and name.startswith("pero_splitk_reduce_k"):
                    extra += float(r["Counter_Value"])
            tot[counter] = (acc + extra, n)
        if len(tot) == 2 and tot["FETCH_SIZE"][1]:
            n = tot["FETCH_SIZE"][1]
            legs[leg] = {"kernels": pat, "launches": n, "fetch_bytes_per_launch": round(2 * tot["FETCH_SIZE"][0] * 1024 / n),
                         "write_bytes_per_launch": round(tot["WRITE_SIZE"][0] * 1024 / max(tot["WRITE_SIZE"][1], 1)),
                         "hbm_bytes_per_launch": round(2 * tot["FETCH_SIZE"][0] * 1024 / n + tot["WRITE_SIZE"][0] * 1024 / max(tot["WRITE_SIZE"][1], 1))}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    json.dump({"csrc_sha256": bench.csrc_hash(), "legs": legs,
               "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 correction); separate --pmc passes, one leg per run"}, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "pmc_legs":
        pmc_legs(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]), sys.argv[4])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
