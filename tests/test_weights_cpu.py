"""Weight files (host only): the seeded checkpoints hold fp16 values like an ultralytics checkpoint on disk, their folded
filters are exactly fp16 x the per-channel BatchNorm factor (the condition under which the fp16x3 context runs two passes
instead of three, csrc/conv_igemm.hip:x3_passes), and both CYW1 versions read back identically."""
import numpy as np
from caesar_yolo_amd import weights as W


def test_seeded_checkpoint_is_fp16_valued_and_folds_exactly():
    ck = W.seeded_checkpoint("n", 5, seed=3)
    for k, v in ck.items():
        assert np.array_equal(v.astype(np.float16).astype(np.float32), v), k
    for cs, w, b, sc in W.fold(ck, "n", 5, with_scale=True):
        w16 = (w / sc[:, None, None, None]).astype(np.float16)                     # what x3_passes() computes per weight
        assert np.array_equal(w16.astype(np.float32) * sc[:, None, None, None], w), cs.name
        if cs.bn:
            assert np.array_equal(w16.astype(np.float32), ck[cs.name + ".conv.weight"]), cs.name
    # the fp32 draw the checkpoint is rounded from does not qualify
    ck32 = W.seeded_checkpoint("n", 5, seed=3, half=False)
    cs, w, b, sc = W.fold(ck32, "n", 5, with_scale=True)[1]
    w16 = (w / sc[:, None, None, None]).astype(np.float16)
    assert not np.array_equal(w16.astype(np.float32) * sc[:, None, None, None], w)


def test_cyw1_versions_read_back_the_same(tmp_path):
    ck = W.seeded_checkpoint("n", 5, seed=4)
    names = {i: "c%d" % i for i in range(5)}
    p1, p2 = str(tmp_path / "v1.cyw"), str(tmp_path / "v2.cyw")
    W.write_cyw(p1, W.fold(ck, "n", 5), names, "n")
    W.write_cyw(p2, W.fold(ck, "n", 5, with_scale=True), names, "n")
    assert open(p1, "rb").read(8)[4:] == b"\x01\0\0\0" and open(p2, "rb").read(8)[4:] == b"\x02\0\0\0"
    s1, n1, w1, o1 = W.read_cyw(p1)
    s2, n2, w2, o2 = W.read_cyw(p2)
    assert (s1, n1, o1) == (s2, n2, o2) and W.read_cyw_header(p2) == W.read_cyw_header(p1)
    for k in w1:
        assert np.array_equal(w1[k][0], w2[k][0]) and np.array_equal(w1[k][1], w2[k][1])


def test_seeded_yolo11_weights_are_fp16_valued_and_calibrated():
    g, wd = W.seeded11_folded("n", 5)
    assert set(wd) == set(cs.name for cs in g.convs)
    for k, (w, b) in wd.items():
        assert np.array_equal(w.astype(np.float16).astype(np.float32), w), k
        assert np.isfinite(w).all() and 0.0 < float(np.abs(w).max()) < 100.0, k
    assert float(wd["model.23.cv3.0.2"][1][0]) == float(np.float16(W.SEEDED_CLS_BIAS))
    g2, wd2 = W.seeded11_folded("n", 5)                       # deterministic
    assert all(np.array_equal(wd[k][0], wd2[k][0]) for k in wd)
