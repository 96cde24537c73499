#!/usr/bin/env python3
"""Lint for the kernels whose vector-memory loads are issued by inline asm and waited for by COUNTED s_waitcnt vmcnt (gemm_e.hip): walks a
kernel's ISA in program order with the in-order vmcnt queue (loads, stores, LDS-DMA, atomics) and reports every instruction that READS or
OVERWRITES the destination registers of a load that no s_waitcnt has retired yet - the compiler is free to copy or re-home the result of an
asm load before the wait it cannot see (it did, under register pressure, in the first LayerNorm epilogue: values of not-yet-arrived loads
were copied and the arriving data later overwrote live registers).  Straight-line approximation: branches are ignored.
usage: python tools/check_async_loads.py <file.s> [kernel-name-substring] [--sgpr-only]
--sgpr-only: only the second check (vector-memory instruction reading an SGPR fewer than five wait states after a vector-ALU write of it): the
first one models straight-line code and gives false positives on compiler-scheduled kernels with loops (attention.hip)."""
import re, sys

def regs(tok):
    tok = tok.strip().rstrip(',')
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()

def main():
    sgpr_only = '--sgpr-only' in sys.argv
    argv = [a for a in sys.argv if a != '--sgpr-only']
    lines = open(argv[1]).read().split('\n')
    pat = argv[2] if len(argv) > 2 else ''
    names = [m.group(1) for l in lines for m in [re.match(r'^(_Z\w+):', l)] if m and pat in l]
    total = 0
    for nm in names:
        s = next(i for i, l in enumerate(lines) if l.startswith(nm + ':'))
        e = next(i for i in range(s, len(lines)) if lines[i].strip().startswith('s_endpgm'))
        queue = []   # outstanding vm ops in issue order: (kind, dst regs, text)
        issues = 0
        # second check: a vector-ALU write of an SGPR (v_readlane_b32 restoring a spilled SGPR, v_readfirstlane_b32) needs FIVE wait states
        # before a vector-memory instruction reads that SGPR; the compiler keeps them for its own instructions, not for inline-asm ones
        hist = []    # (wait states this instruction provides, SGPRs it writes from the vector ALU)
        def sregs(tok):
            tok = tok.strip().rstrip(',')
            m = re.match(r's\[(\d+):(\d+)\]', tok)
            if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
            m = re.match(r's(\d+)$', tok)
            return {int(m.group(1))} if m else set()
        for l in lines[s:e]:
            l2 = l.strip()
            if l2 and not l2.startswith((';', '.')) and not l2.endswith(':'):
                pp = re.split(r'[ ,]+', l2)
                if pp[0].startswith(('buffer_load', 'buffer_store', 'global_load_lds', 'buffer_atomic', 'global_load_dword', 'global_store_dword', 'global_atomic')):
                    need = set().union(*[sregs(t) for t in pp[1:]])
                    ws = 0
                    for w, wr in reversed(hist):
                        if ws >= 5: break
                        if wr & need:
                            print(f"{nm[:60]}: `{l2[:80]}` reads an SGPR a vector-ALU instruction wrote {ws} wait state(s) earlier")
                            issues += 1
                            break
                        ws += w
                m = re.match(r's_nop (\d+)', l2)
                hist.append((int(m.group(1)) + 1 if m else 1, sregs(pp[1]) if pp[0] in ('v_readlane_b32', 'v_readfirstlane_b32') and len(pp) > 1 else set()))
                hist = hist[-12:]
            l = l.strip()
            if not l or l.startswith((';', '.')) or l.endswith(':'): continue
            parts = re.split(r'[ ,]+', l)
            op = parts[0]
            m = re.match(r's_waitcnt.*vmcnt\((\d+)\)', l)
            if m:
                n = int(m.group(1))
                queue = queue[len(queue) - n:] if n < len(queue) else queue
                if n == 0: queue = []
                continue
            toks = [t for t in parts[1:]]
            used = set().union(*[regs(t) for t in toks]) if toks else set()
            pending = set().union(*[q[1] for q in queue]) if queue else set()
            if used & pending and not op.startswith('s_waitcnt') and not sgpr_only:
                bad = [q[2] for q in queue if q[1] & used]
                print(f"{nm[:60]}: `{l}` touches registers of a load still in flight: {bad[0][:70]}")
                issues += 1
            if op.startswith(('buffer_load', 'global_load_dword', 'scratch_load')) and 'lds' not in op:
                queue.append(('load', regs(toks[0]), l))
            elif op.startswith(('buffer_store', 'global_store', 'scratch_store', 'global_atomic', 'buffer_atomic', 'global_load_lds', 'buffer_load_dword')):
                queue.append(('other', set(), l))
        print(f"{nm[:70]}: {issues} suspicious instruction(s)")
        total += issues
    sys.exit(1 if total else 0)

if __name__ == '__main__':
    main()
