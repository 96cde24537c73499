"""CPU tests of the SURVEY.md section 8(f) "next" rows: the oracle's restatements against what the reference itself
produced (tests/golden g12-g15, oracle/make_golden_next.py) and the host-side logic of the product package."""
import os

import numpy as np
import torch

from oracle import pero_oracle as O


def test_oracle_topk_errors_match_reference_tester(golden):
    g = golden("g12_tester.npz")
    tot = {"errors_1": 0, "errors_3": 0, "errors_10": 0}
    length = 0
    rk = {"errors_1": 0, "errors_3": 0, "errors_10": 0}
    for i in range(int(g["n_batches"])):
        out, lab, msk = g[f"b{i}.output"], g[f"b{i}.labels"], g[f"b{i}.mask"]
        counts, n = O.topk_error_counts(out, lab, msk, (1, 3, 10))
        ranks = O.label_ranks(out.reshape(-1, out.shape[-1]), lab.reshape(-1), msk.reshape(-1))
        c2, n2 = O.errors_from_ranks(ranks, (1, 3, 10))
        assert n == n2 and counts == c2   # the rank formulation (what the kernel computes) == numpy argmax / argsort
        for k in tot:
            tot[k] += counts[k]; rk[k] += c2[k]
        length += n
    for k in tot:
        assert tot[k] / length == float(g[k]), (k, tot[k], length, float(g[k]))


def test_oracle_rank_tie_rules():
    out = np.array([[1.0, 3.0, 3.0, 2.0, 3.0], [0.0, 0.0, 0.0, 0.0, 0.0]], dtype=np.float32)
    labels = np.array([2, 4]); mask = np.array([1, 1])
    r = O.label_ranks(out, labels, mask)
    assert r.tolist() == [[0, 1, 1], [0, 4, 0]]
    # argmax picks index 1 (first max) -> label 2 is an error at k=1; stable top-2 = indices {2, 4} -> no error at k=2
    c, n = O.topk_error_counts(out[None], labels[None], mask[None], (1, 2))
    assert c == {"errors_1": 2, "errors_2": 0} and n == 2
    assert O.errors_from_ranks(r, (1, 2))[0] == c


def test_reference_checkpoint_fixture_is_a_plain_state_dict(golden):
    """The file the reference's `model.save()` wrote loads with the non-executing loader and lists the documented keys."""
    g = golden("g13_checkpoint.npz")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g13_reference_checkpoint.pth")
    sd = torch.load(path, map_location="cpu", weights_only=True)
    assert list(sd.keys()) == list(g["keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g["shapes"])
    assert "backbone.conv_layer.weight" in sd and "head.linear.bias" in sd
    assert not any(k.endswith(".pe") or "mask_pattern" in k for k in sd)   # non-persistent buffers
    # oracle forward on those weights reproduces the reference output stored beside the checkpoint
    x = O.prepare_images(torch.from_numpy(g["images"]))
    out, loss = O.masked_model_forward(sd, x, torch.from_numpy(g["labels"]), torch.from_numpy(g["mask"]), num_heads=4)
    assert np.abs(out.numpy() - g["output"]).max() < 1e-4
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * float(g["loss"])


def test_helper_paths_match_reference(golden):
    from pero_pretraining_amd.common import helpers as H
    g = golden("g13_checkpoint.npz")
    assert H.get_checkpoint_path("ckpts", 7) == str(g["checkpoint_path_7"])
    assert H.get_visualization_path("vis", 7, "trn") == str(g["visualization_path_7"])


def _g15_lines(g):
    widths = [int(w) for w in g["widths"]]
    out, out2 = [], []
    p = 0
    for w in widths:
        n = 10 * w * 3
        out.append(g["lines_flat"][p:p + n].reshape(10, w, 3)); out2.append(g["lines2_flat"][p:p + n].reshape(10, w, 3))
        p += n
    return out, out2


def test_oracle_collation_matches_reference_batch_creator(golden):
    g = golden("g15_batch_creator.npz")
    lines, lines2 = _g15_lines(g)
    wt = O.padded_width(max(l.shape[1] for l in lines))
    assert wt == g["pad.images"].shape[2]
    np.random.seed(int(g["pad.seed"]))
    lp1 = O.draw_left_paddings(lines, wt)
    lp2 = O.draw_left_paddings(lines2, wt)
    im1, m1 = O.collate_view(lines, lp1, wt)
    im2, m2 = O.collate_view(lines2, lp2, wt)
    assert np.array_equal(im1, g["pad.images"]) and np.array_equal(im2, g["pad.images2"])
    assert np.array_equal(m1, g["pad.image_masks"]) and np.array_equal(m2, g["pad.image_masks2"])
    shifts = [a - b for a, b in zip(lp1, lp2)]
    assert shifts == g["pad.shifts"].tolist()
    s1, s2 = O.shift_masks(shifts, m1, m2)
    assert np.array_equal(s1, g["pad.shift_masks"]) and np.array_equal(s2, g["pad.shift_masks2"])
    assert (s1 == 2).any() and (s1 == 0).any()   # the fixture exercises all three values


def test_oracle_vq_ema_update_matches_reference(golden):
    g = golden("g16_vq_ema.npz")
    cb, ew, cs = g["codebook0"], g["ema_w0"], g["ema_cluster_size0"]
    for step in range(2):
        feats = g[f"s{step}.features"]
        flat = np.ascontiguousarray(feats.transpose(0, 2, 3, 1)).reshape(-1, 32)
        idx, _ = O.vq_nearest(flat, cb)
        assert np.array_equal(idx, g[f"s{step}.indices"])
        q, _ = O.vq_quantize(feats, cb)
        assert np.abs(q - g[f"s{step}.quantized"]).max() < 1e-6
        cs, ew, cb = O.vq_ema_update(flat, idx, cs, ew, float(g["decay"]))
        assert np.abs(cs - g[f"s{step}.ema_cluster_size"]).max() < 1e-6
        assert np.abs(ew - g[f"s{step}.ema_w"]).max() < 1e-5
        assert np.abs(cb - g[f"s{step}.codebook"]).max() < 1e-4 * np.abs(g[f"s{step}.codebook"]).max()
        cb, ew, cs = g[f"s{step}.codebook"], g[f"s{step}.ema_w"], g[f"s{step}.ema_cluster_size"]  # next step: reference state


def test_label_text_and_record_formats(golden, tmp_path):
    """Host-side formats of the label pipeline: byte-identical to the file the reference's own save_labels wrote."""
    from pero_pretraining_amd.scripts import labels as L
    g = golden("g14_labels.npz")
    ids, lens, flat = g["ids"].tolist(), g["lens"].tolist(), g["labels"].tolist()
    data, p = {}, 0
    for i, n in zip(ids, lens):
        data[i] = flat[p:p + n]; p += n
    out = tmp_path / "labels.txt"
    L.save_labels(data, str(out))
    ref = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g14_labels.txt")
    assert out.read_bytes() == open(ref, "rb").read()
    # the file parses back (common/dataset.py:_parse_line) and converts to LMDB records (convert_gt_to_lmdb.py:28-39)
    parsed = [L.parse_line(l) for l in open(ref)]
    assert [p[0] for p in parsed] == ids and [len(p[1]) for p in parsed] == lens
    store = {}
    n = L.convert_labels_file(ref, lambda k, v: store.__setitem__(k, v), offset=5)
    assert n == 3 and sorted(store) == [b"         5", b"         6", b"         7"]   # the unlabeled 4th line is skipped
    assert store[b"         6"] == ('{"image": "b/line_1", "labels": [' + ", ".join(f'"{v}"' for v in data["b/line_1"]) + "]}").encode()
    assert L.parse_label_record(store[b"         7"]) == ("c-2.png", [str(data["c-2.png"][0])])
    assert L.parse_label_record(b'{"images": ["a", "b"], "labels": ["1"]}') == (["a", "b"], ["1"])


def test_collator_host_masks_follow_the_reference_slicing_semantics():
    """BatchCreator._host_masks (the numpy twin of csrc/collate.hip's mask kernel that rides along with the device masks): exhaustive small
    case against the reference's slice arithmetic (common/dataloader.py:92-96, 124-138), negative / clamped shifts included."""
    import numpy as np
    from pero_pretraining_amd.common.dataloader import BatchCreator
    S, sub = 6, 8
    cases = [(w1, w2, a, b, c) for w1 in (8, 17, 40) for w2 in (8, 33) for a in range(0, 5) for b in range(0, 5) for c in (0, -7, 3, 9)]
    im1, im2, sm1, sm2 = BatchCreator._host_masks([c[0] for c in cases], [c[2] for c in cases], [c[1] for c in cases], [c[3] for c in cases],
                                                  [c[4] for c in cases], S, sub)
    for i, (a_w, b_w, a, b, cs) in enumerate(cases):
        e1 = np.ones(S, np.uint8); e1[:a] = 0; e1[a + int(np.ceil(a_w / sub)):] = 0
        e2 = np.ones(S, np.uint8); e2[:b] = 0; e2[b + int(np.ceil(b_w / sub)):] = 0
        sh = cs + a - b
        s1 = np.zeros(S, np.uint8)
        if sh < 0:
            s1[:sh] = 1
        else:
            s1[sh:] = 1
        s2 = s1[::-1].copy()
        s1[(s1 == 1) & (e1 == 0)] = 2; s2[(s2 == 1) & (e2 == 0)] = 2
        assert np.array_equal(im1[i], e1) and np.array_equal(im2[i], e2), cases[i]
        assert np.array_equal(sm1[i], s1) and np.array_equal(sm2[i], s2), cases[i]
    only1 = BatchCreator._host_masks([16, 40], [1, 0], None, None, [0, 0], S, sub)
    assert only1[1] is None and only1[0].tolist() == [[0, 1, 1, 0, 0, 0], [1, 1, 1, 1, 1, 0]]
