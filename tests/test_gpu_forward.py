"""Network parity: cy_forward (the whole YOLOv8l graph on HIP kernels) against the torch-CPU fp32 oracle on the same
seeded weights and the same preprocessed tile.  f32 context: raw head output within 2e-4 (abs, logits are O(1-10));
intermediate conv outputs within 1e-4 of their scale.  f16 context: within 6e-2 (fp16 operands through 100+ layers)."""
import os
import numpy as np
import pytest
import torch
from gpu_common import detector, oracle_model, netin_from_chw, ROOT

pytestmark = pytest.mark.gpu


def _tile(name, h=None, w=None):
    from oracle import preprocessing_ref as P
    g = np.load(os.path.join(ROOT, "tests/golden/preproc.npz"))
    img = g["in/" + name]
    if h:
        img = img[:h, :w]
    dp = P.build_pipeline([("zscale", dict(contrasts=[0.25] * 3)), ("minmax", dict(norm_min=0, norm_max=255))])
    return dp(P.to_cube(img))


def _oracle_forward(images, imgsz):
    from oracle import yolov8_ref as Y
    m = oracle_model()
    xs = [Y.preprocess(im, imgsz)[0] for im in images]
    x = torch.cat(xs, 0)
    m.net.taps = {}
    with torch.no_grad():
        raw = m.net.forward(x)
    return x, raw.permute(0, 2, 1).contiguous(), m.net.taps


TAPS = ["model.0", "model.1", "model.2.cv1", "model.2.m.2.cv1", "model.12.m.0.cv2", "model.2.cv2", "model.4.cv2",
        "model.6.cv2", "model.8.cv2", "model.9.cv1", "model.9.cv2", "model.12.cv1", "model.12.cv2", "model.15.cv1",
        "model.15.cv2", "model.16", "model.18.cv2", "model.19", "model.21.cv2", "model.22.cv2.0.1", "model.22.cv3.2.1"]


@pytest.mark.parametrize("prec,tol_raw,tol_tap,env", [
    ("fp32", 2e-4, 1e-4, {}), ("fp16x3", 2e-4, 1e-4, {}), ("fp16", 6e-2, 3e-2, {}),
    # the kernels that only take over at benchmark-sized batches, forced on this small one: pixels-direct 1x1 (incl. the
    # upsample+concat inputs of layers 12/15), and with it off, the 256x128 ring kernel
    ("fp16", 6e-2, 3e-2, {"CY_DIRECT_MIN_BLOCKS": "1"}), ("fp16", 6e-2, 3e-2, {"CY_DIRECT_MIN_BLOCKS": "-1"}),
    # model.0 + model.1 as one kernel (default from 256 workgroups on); the model.0 tensor does not exist then
    ("fp16", 6e-2, 3e-2, {"CY_STEM_FUSE": "2"}),
    # the 64-channel bottlenecks of model.2 as ONE kernel each (cv1's output stays in LDS; opt-in, default is two launches)
    ("fp16", 6e-2, 3e-2, {"CY_BNECK_FUSE": "1"}),
    # the back-to-back pair model.3 -> model.4.cv1 as two launches (default: one kernel, the 256-channel map between them never stored)
    ("fp16", 6e-2, 3e-2, {"CY_FUSE_PW": "0"}),
    # kernel selection that sees the real batch of 2 (the default evaluates every threshold as for 256 tiles)
    ("fp16", 6e-2, 3e-2, {"CY_BATCH_INVARIANT": "0"}), ("fp16", 6e-2, 3e-2, {"CY_BATCH_INVARIANT": "0", "CY_DIRECT_MIN_BLOCKS": "1"})])
def test_forward_matches_oracle(prec, tol_raw, tol_tap, env, monkeypatch):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    det = detector(prec)
    imgs = [_tile("big512", 256, 256), _tile("big512", 256, 256)[::-1].copy()]
    x, raw, taps = _oracle_forward(imgs, 256)
    pred = det.forward(netin_from_chw(x, det.dtype))
    torch.cuda.synchronize()
    for name in TAPS:
        ref = taps[name]
        if name == "model.0" and prec == "fp16" and env.get("CY_BATCH_INVARIANT") != "0" and env.get("CY_STEM_FUSE") != "0":
            with pytest.raises(Exception, match="not materialised"):
                det.read_conv(name, ref.numel())
            continue
        if name == "model.2.m.2.cv1" and prec == "fp16" and env.get("CY_BNECK_FUSE") == "1":
            with pytest.raises(Exception, match="ran fused"):
                det.read_conv(name, ref.numel())
            continue
        got = torch.from_numpy(det.read_conv(name, ref.numel()))
        assert tuple(got.shape) == tuple(ref.shape), name
        sc = max(float(ref.abs().max()), 1.0)
        err = float((got - ref).abs().max())
        assert err <= tol_tap * sc, "%s: max abs err %.3e (scale %.2f)" % (name, err, sc)
    err = float((pred.cpu() - raw).abs().max())
    assert err <= tol_raw * max(1.0, float(raw.abs().max())), "raw head output: max abs err %.3e" % err


@pytest.mark.parametrize("prec", ["fp32", "fp16x3"])
def test_forward_ragged_letterboxed_shape_fp32(prec):
    """394-wide edge tiles of the 16k mosaic letterbox to 416x512: non-square grids, 3 batch entries."""
    det = detector(prec)
    base = _tile("big512")
    imgs = [base[:512, :394].copy(), base[:512, 100:494].copy(), base[:512, 118:512].copy()]
    x, raw, _ = _oracle_forward(imgs, 512)
    assert tuple(x.shape[2:]) == (512, 416)
    pred = det.forward(netin_from_chw(x, det.dtype))
    torch.cuda.synchronize()
    err = float((pred.cpu() - raw).abs().max())
    assert err <= 2e-4 * max(1.0, float(raw.abs().max())), err


@pytest.mark.parametrize("fuse", ["0", "2"])
def test_forward_ragged_letterboxed_shape_fp16(fuse, monkeypatch):
    """Same non-square grid in the fp16 context, with and without the fused stem kernel (104-px-wide output map: partial
    32-px tiles, stem pixels outside the map are the next layer's zero padding)."""
    monkeypatch.setenv("CY_STEM_FUSE", fuse)
    det = detector("fp16")
    base = _tile("big512")
    imgs = [base[:512, :394].copy(), base[:512, 100:494].copy(), base[:512, 118:512].copy()]
    x, raw, taps = _oracle_forward(imgs, 512)
    pred = det.forward(netin_from_chw(x, det.dtype))
    torch.cuda.synchronize()
    ref = taps["model.1"]
    got = torch.from_numpy(det.read_conv("model.1", ref.numel()))
    err = float((got - ref).abs().max())
    assert err <= 4e-3 * max(1.0, float(ref.abs().max())), "model.1: %.3e" % err
    err = float((pred.cpu() - raw).abs().max())
    assert err <= 6e-2 * max(1.0, float(raw.abs().max())), err


@pytest.mark.parametrize("imgsz,B", [(128, 3), (1024, 1)])
def test_forward_at_the_other_published_input_sizes(imgsz, B):
    """The reference publishes yolov8l weights trained at 128, 256, 512, 640 and 1024 px (README.md:190-207); 256 / 512 / 640 run in the
    other tests.  128 px: 16 / 8 / 4-px maps (narrower than every tuned patch); 1024 px: 128 / 64 / 32-px maps."""
    base = _tile("big512")
    big = np.concatenate([np.concatenate([base, base[:, ::-1]], 1), np.concatenate([base[::-1], base[::-1, ::-1]], 1)], 0)     # 1024 x 1024
    imgs = [big[i * 37:i * 37 + imgsz, i * 53:i * 53 + imgsz].copy() for i in range(B)] if imgsz < 1024 else [big]
    x, raw, _ = _oracle_forward(imgs, imgsz)
    assert tuple(x.shape[2:]) == (imgsz, imgsz)
    for prec, tol in (("fp16x3", 2e-4), ("fp32", 2e-4), ("fp16", 6e-2)):
        det = detector(prec, max_batch=3, max_imgsz=1024)
        pred = det.forward(netin_from_chw(x, det.dtype))
        torch.cuda.synchronize()
        err = float((pred.cpu() - raw).abs().max())
        print("imgsz %d %s: raw head output max abs err %.3e (scale %.2f)" % (imgsz, prec, err, float(raw.abs().max())))
        assert err <= tol * max(1.0, float(raw.abs().max())), (imgsz, prec, err)


@pytest.mark.parametrize("scale,nc", [("n", 5), ("s", 3)])
def test_other_scales_fp32(scale, nc):
    """yolov8n / yolov8s graphs (16..512 channels, slabs that are not multiples of 64) through the same kernels."""
    from caesar_yolo_amd import weights as W
    from caesar_yolo_amd.model import HipDetector
    from oracle import yolov8_ref as Y
    path = "/tmp/cy_test_seeded_%s_%d.cyw" % (scale, nc)
    if not os.path.exists(path):
        W.make_seeded_file(path, scale, nc)
    sc, names, wd, _ = W.read_cyw(path)
    for prec, tol in (("fp32", 3e-4), ("fp16x3", 3e-4), ("fp16", 8e-2)):
        det = HipDetector(path, device=0, precision=prec, max_batch=2, max_imgsz=256)
        om = Y.OracleYOLO(wd, names, sc)
        imgs = [_tile("big512", 192, 256), _tile("big512", 192, 256)[:, ::-1].copy()]
        xs = torch.cat([Y.preprocess(im, 256)[0] for im in imgs], 0)
        with torch.no_grad():
            raw = om.net.forward(xs).permute(0, 2, 1).contiguous()
        pred = det.forward(netin_from_chw(xs, det.dtype))
        torch.cuda.synchronize()
        err = float((pred.cpu() - raw).abs().max())
        assert err <= tol * max(1.0, float(raw.abs().max())), (scale, prec, err)
        det.close()


def test_batch_invariant_mode_is_bit_exact(monkeypatch):
    """Default kernel selection (CY_BATCH_INVARIANT unset or 1, fp16 context): every layer takes the kernel a 256-tile batch would take, whatever the batch, so
    a tile's head output is bit-identical alone, in a batch of 4, and through the split (two-stream) forward - the
    property that makes catalogs independent of the world size."""
    monkeypatch.delenv("CY_BATCH_INVARIANT", raising=False)          # the default
    det = detector("fp16")
    base = _tile("big512", 256, 256)
    imgs = [base, base[::-1].copy(), base[:, ::-1].copy(), base[::-1, ::-1].copy()]
    x, raw, _ = _oracle_forward(imgs, 256)
    xin = netin_from_chw(x, det.dtype)
    p4 = det.forward(xin).cpu()
    assert float((p4 - raw).abs().max()) <= 6e-2 * max(1.0, float(raw.abs().max()))
    for i in range(4):
        p1 = det.forward(xin[i:i + 1].contiguous()).cpu()
        assert torch.equal(p1[0], p4[i]), "tile %d differs between batch 1 and batch 4" % i
    p3 = det.forward(xin[1:4].contiguous()).cpu()
    assert torch.equal(p3, p4[1:4])


def test_fp16x3_results_do_not_depend_on_the_batch():
    """fp16x3 context: the kernel of a layer is chosen from its geometry alone, so a tile's head output is bit-identical alone
    and in any batch (catalogs independent of world size / batch split in the parity mode as well)."""
    det = detector("fp16x3")
    base = _tile("big512", 256, 256)
    imgs = [base, base[::-1].copy(), base[:, ::-1].copy(), base[::-1, ::-1].copy()]
    x, raw, _ = _oracle_forward(imgs, 256)
    xin = netin_from_chw(x, det.dtype)
    p4 = det.forward(xin).cpu()
    err = float((p4 - raw).abs().max())
    print("fp16x3 raw head output vs oracle: max abs err %.3e (scale %.2f)" % (err, float(raw.abs().max())))
    assert err <= 2e-4 * max(1.0, float(raw.abs().max()))
    for i in range(4):
        p1 = det.forward(xin[i:i + 1].contiguous()).cpu()
        assert torch.equal(p1[0], p4[i]), "tile %d differs between batch 1 and batch 4" % i


def test_back_to_back_pair_runs_in_one_kernel(monkeypatch):
    """Round 4 layer fusion: model.3 (3x3 s2, 128 -> 256) and its only reader model.4.cv1 (1x1, 256 -> 256) as ONE launch of the
    pixels-direct kernel -- the first layer's accumulators (bias + SiLU, rounded to fp16 as the store would) are the B operand of the
    second GEMM in registers.  Against the two-launch form: model.4.cv1's output agrees to fp16 rounding (the second GEMM sums its K in a
    permuted order), both agree with the oracle, the fused run does not materialise model.3 (debug read refused), head output within 6e-2."""
    det = detector("fp16")
    imgs = [_tile("big512", 256, 256), _tile("big512", 256, 256)[:, ::-1].copy()]
    x, raw, taps = _oracle_forward(imgs, 256)
    xin = netin_from_chw(x, det.dtype)
    ref = taps["model.4.cv1"]
    monkeypatch.setenv("CY_FUSE_PW", "0")
    p0 = det.forward(xin).cpu()
    t0 = torch.from_numpy(det.read_conv("model.4.cv1", ref.numel()))
    m3 = torch.from_numpy(det.read_conv("model.3", taps["model.3"].numel()))
    assert float((m3 - taps["model.3"]).abs().max()) <= 3e-2 * max(1.0, float(taps["model.3"].abs().max()))
    monkeypatch.setenv("CY_FUSE_PW", "1")
    p1 = det.forward(xin).cpu()
    t1 = torch.from_numpy(det.read_conv("model.4.cv1", ref.numel()))
    with pytest.raises(Exception, match="not materialised"):
        det.read_conv("model.3", taps["model.3"].numel())
    sc = max(1.0, float(ref.abs().max()))
    d01, e0, e1 = float((t1 - t0).abs().max()), float((t0 - ref).abs().max()), float((t1 - ref).abs().max())
    print("model.4.cv1: fused vs two launches %.3e, vs oracle %.3e (two launches) / %.3e (fused), scale %.2f; head output fused vs oracle %.3e"
          % (d01, e0, e1, sc, float((p1 - raw).abs().max())))
    assert d01 <= 4e-3 * sc and e0 <= 3e-2 * sc and e1 <= 3e-2 * sc
    assert float((p1 - raw).abs().max()) <= 6e-2 * max(1.0, float(raw.abs().max())) and float((p0 - raw).abs().max()) <= 6e-2 * max(1.0, float(raw.abs().max()))


def test_fp16x3_two_pass_form_on_fp16_checkpoints(monkeypatch):
    """fp16x3 context, round 4: a checkpoint whose tensors are fp16 values (what ultralytics stores; the seeded checkpoints are
    written that way) gives folded filters that are EXACTLY an fp16 filter times the BatchNorm factor of their channel; the
    packer recognises it from the numbers (x3_passes) and every MFMA layer runs two passes [x_lo w | x_hi w] with the factor in the
    epilogue instead of three.  Both forms against the oracle (2e-4), and against each other."""
    from gpu_common import seeded_weights
    from caesar_yolo_amd.model import HipDetector
    det2 = detector("fp16x3")
    n2, n3 = det2.weight_passes()
    assert (n2, n3) == (102, 0), "seeded yolov8l: all 102 MFMA convolutions on the two-pass form, got %d / %d" % (n2, n3)
    monkeypatch.setenv("CY_X3_PASSES", "3")
    det3 = HipDetector(seeded_weights()[0], device=0, precision="fp16x3", max_batch=4, max_imgsz=640)
    monkeypatch.delenv("CY_X3_PASSES")
    assert det3.weight_passes() == (0, 102)
    base = _tile("big512", 256, 256)
    x, raw, _ = _oracle_forward([base, base[::-1].copy()], 256)
    xin = netin_from_chw(x, det2.dtype)
    p2, p3 = det2.forward(xin).cpu(), det3.forward(xin).cpu()
    det3.close()
    sc = max(1.0, float(raw.abs().max()))
    e2, e3, e23 = float((p2 - raw).abs().max()), float((p3 - raw).abs().max()), float((p2 - p3).abs().max())
    print("fp16x3 raw head output vs oracle: two-pass %.3e, three-pass %.3e, between them %.3e (scale %.2f)" % (e2, e3, e23, sc))
    assert e2 <= 2e-4 * sc and e3 <= 2e-4 * sc


@pytest.mark.parametrize("shape", [(3, 160, 192), (20, 256, 256), (1, 512, 512), (5, 96, 416)])
def test_two_group_stem_kernel_is_bit_identical(shape, monkeypatch):
    """The persistent two-group form of the fused stem + first down conv (CY_STEM_V=2: 8 x 16 output patches, one wave group
    fills its LDS patch while the other convolves its previous one) does the same arithmetic in the same order as the
    one-group kernel: the whole head output must be bit for bit the same, on ragged maps and on more patches than workgroups."""
    B, H, W = shape
    det = detector("fp16", max_batch=20, max_imgsz=512)
    g = torch.Generator().manual_seed(B * 1000 + H + W)
    x = torch.rand((B, H, W, 4), generator=g).half().cuda()
    monkeypatch.setenv("CY_STEM_FUSE", "2")
    monkeypatch.setenv("CY_STEM_V", "1")
    ref = det.forward(x).clone()
    monkeypatch.setenv("CY_STEM_V", "2")
    for _ in range(3):
        got = det.forward(x)
        torch.cuda.synchronize()
        assert torch.equal(ref, got)


@pytest.mark.parametrize("prec", ["fp16", "fp16x3"])
def test_head_output_kernel_is_bit_identical(prec, monkeypatch):
    """The weights-stationary kernels of the detect head's output 1x1s (cv2.x.2 64 -> 64, cv3.x.2 256 -> nc, fp32 prediction
    rows) against the generic implicit-GEMM kernel they replace (CY_HEAD_DIRECT=0): one launch per layer (head1x1_kernel,
    CY_HEAD_PAIR=0) and the default, both layers of a level in one launch that writes whole prediction rows
    (head1x1_pair_kernel).  The same MFMA chain per output value, so the head output is bit-identical -- on square maps and on the
    ragged 416 x 512 letterbox shape whose stride-32 map (13 x 16 pixels x 3 tiles = 624 pixels) ends inside a 32-pixel group and
    whose 16-row runs cross image boundaries."""
    det = detector(prec)
    base = _tile("big512")
    for imgs, size in (([_tile("big512", 256, 256), _tile("big512", 256, 256)[::-1].copy()], 256),
                       ([base[:512, :394].copy(), base[:512, 100:494].copy(), base[:512, 118:512].copy()], 512)):
        x, raw, _ = _oracle_forward(imgs, size)
        xin = netin_from_chw(x, det.dtype)
        monkeypatch.setenv("CY_HEAD_DIRECT", "0")
        p0 = det.forward(xin).cpu().clone()
        monkeypatch.setenv("CY_HEAD_DIRECT", "1")
        monkeypatch.setenv("CY_HEAD_PAIR", "0")
        p1 = det.forward(xin).cpu().clone()
        monkeypatch.delenv("CY_HEAD_PAIR")
        p2 = det.forward(xin).cpu().clone()
        assert torch.equal(p0, p1), "one launch per layer vs generic (%s, %d px): max %.3e" % (prec, size, float((p0 - p1).abs().max()))
        assert torch.equal(p0, p2), "one launch per level vs generic (%s, %d px): max %.3e" % (prec, size, float((p0 - p2).abs().max()))
        tol = 6e-2 if prec == "fp16" else 2e-4
        assert float((p2 - raw).abs().max()) <= tol * max(1.0, float(raw.abs().max()))


@pytest.mark.parametrize("prec", ["fp32", "fp16x3"])
def test_stem_four_pixel_form_is_bit_identical(prec, monkeypatch):
    """model.0 in the fp32 / fp16x3 contexts (exact fp32 FMA chain, K = 27): four output pixels per thread (stem_quad_kernel,
    default on maps whose width is a multiple of 4) against one pixel per thread (CY_STEM_QUAD=0) -- the same chain per output
    value, so model.0 and the head output are bit-identical; both within the context's tolerance of the oracle."""
    det = detector(prec)
    imgs = [_tile("big512", 256, 256), _tile("big512", 256, 256)[::-1].copy()]
    x, raw, taps = _oracle_forward(imgs, 256)
    xin = netin_from_chw(x, det.dtype)
    ref = taps["model.0"]
    monkeypatch.setenv("CY_STEM_QUAD", "0")
    p0 = det.forward(xin).cpu().clone()
    t0 = torch.from_numpy(det.read_conv("model.0", ref.numel()))
    monkeypatch.setenv("CY_STEM_QUAD", "1")
    p1 = det.forward(xin).cpu().clone()
    t1 = torch.from_numpy(det.read_conv("model.0", ref.numel()))
    assert torch.equal(t0, t1), "model.0 differs: max %.3e" % float((t0 - t1).abs().max())
    assert torch.equal(p0, p1)
    assert float((t1 - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    assert float((p1 - raw).abs().max()) <= 2e-4 * max(1.0, float(raw.abs().max()))
