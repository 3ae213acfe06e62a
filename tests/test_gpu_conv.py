"""Kernel-level parity: the fused conv implicit-GEMM (csrc/conv_igemm.hip) against a plain PyTorch fp32 reference of
the same op (F.conv2d + bias + SiLU + residual), through the C-ABI entry cy_conv_bn_silu.
Tolerances: f32 path (exact-fp32 MFMA) 2e-5 relative to the output scale; f16 path (fp16 operands, fp32 accumulate)
is compared on fp16-rounded inputs/weights with 4e-3 relative to the output scale (one fp16 rounding of the result);
fp16x3 path (fp16 high + low halves, three MFMA passes, fp32 accumulate): fp32 inputs, the f32 tolerance 2e-5."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from gpu_common import detector

pytestmark = pytest.mark.gpu

CASES = [  # B, H, W, Cin, Cout, k, s, act, residual
    (2, 32, 32, 64, 64, 3, 1, True, False),
    (3, 20, 24, 128, 128, 3, 1, True, True),
    (2, 32, 32, 64, 128, 3, 2, True, False),
    (2, 16, 16, 256, 512, 1, 1, True, False),
    (1, 8, 8, 320, 128, 1, 1, True, False),
    (2, 16, 16, 64, 5, 1, 1, False, False),
    (1, 9, 11, 72, 40, 3, 1, True, True),
    (1, 64, 64, 64, 64, 3, 1, True, True),
    (4, 16, 16, 512, 256, 3, 1, True, False),
    (1, 5, 7, 8, 16, 3, 2, True, False),
    (32, 64, 64, 64, 128, 3, 1, True, False),     # halo-reuse kernel, 16-row patches
    (2, 50, 70, 128, 256, 3, 1, True, True),      # halo-reuse kernel, ragged patches + residual
    (3, 33, 17, 192, 192, 3, 1, False, False),    # halo-reuse kernel, Cout not a multiple of 128, no activation
    (5, 40, 48, 64, 64, 3, 1, True, True),        # ping-pong kernel, 64-channel tiles, residual
    (2, 24, 24, 256, 64, 3, 1, True, False),      # ping-pong kernel, 4 input slabs
    (1, 16, 16, 128, 40, 3, 1, True, False),      # ping-pong kernel, ragged Cout
    (4, 128, 128, 128, 256, 1, 1, True, False),   # 256x128 three-slab ring kernel (fp16 context), 1x1
    (3, 120, 100, 192, 160, 1, 1, True, True),    # ring kernel: ragged pixel tail (M % 256 != 0), ragged Cout, residual
    (2, 256, 256, 64, 256, 3, 2, True, False),    # ring kernel: 3x3 stride 2 (im2col gather with padding masks)
    (16, 64, 64, 64, 128, 1, 1, False, False),    # ring kernel: a single K-slab (prologue shorter than the ring)
    (2, 40, 64, 256, 192, 3, 1, True, True),      # 512-px wide kernel: ragged rows, Cout 192 (half-empty channel tile), residual, 4 slab pairs
    (1, 16, 32, 128, 128, 3, 1, False, False),    # wide kernel: a single patch, two slab pairs, no activation
    (3, 30, 60, 64, 256, 3, 1, True, False),      # wide kernel: ragged columns (60 of 64) and rows, one slab pair
    (6, 64, 64, 32, 32, 3, 1, True, True),        # persistent kernel, 32-channel variant (64-byte LDS rows), residual
    (3, 40, 50, 32, 64, 3, 1, True, False),       # 32 -> 64 channels, ragged patches
    (40, 32, 64, 128, 64, 3, 1, True, False),     # 64-channel variant of the wide kernel
    (9, 16, 16, 256, 256, 3, 1, True, True),      # dual-image variant of the wide kernel (16-px maps), odd batch, residual
    (4, 14, 15, 128, 192, 3, 1, True, False),     # dual-image variant, ragged 14x15 maps, Cout 192 (box head), two slab pairs
    (16, 128, 128, 128, 128, 3, 2, True, False),  # pixels-direct kernel, 3x3 stride 2 (taps as uniform address shifts), 128-ch tile
    (5, 130, 126, 128, 320, 3, 2, True, False),   # pixels-direct 3x3 s2: odd rows/cols at the border, ragged 256-ch tiles, 18 K chunks
    (70, 64, 64, 64, 64, 3, 1, True, True),       # persistent 64-channel kernel: 4-5 patches per workgroup (both wave groups, odd and even counts), residual
    (33, 40, 72, 64, 64, 3, 1, True, False),      # persistent 64-channel kernel: ragged patches, 1-2 patches per workgroup
]


# fp16 twice: with the default batch-invariant kernel selection (thresholds see a 256-tile batch) and with the selection
# that sees the real (small) batch of the case: between them every kernel variant runs
@pytest.mark.parametrize("prec,inv", [("fp32", "1"), ("fp16", "1"), ("fp16", "0"), ("fp16x3", "1")])
@pytest.mark.parametrize("case", CASES)
def test_conv_bn_silu(prec, inv, case, monkeypatch):
    monkeypatch.setenv("CY_BATCH_INVARIANT", inv)
    monkeypatch.setenv("CY_WIDE_DUAL", "2")       # the dual-image kernel normally waits for benchmark-sized batches: force it here
    B, H, W, Cin, Cout, k, s, act, use_res = case
    det = detector(prec)
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn((B, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, k, k), generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn((Cout,), generator=g) * 0.1
    if prec == "fp16":
        x, w = x.half().float(), w.half().float()
    y = F.conv2d(x, w, b, stride=s, padding=k // 2)
    if act:
        y = F.silu(y)
    res = None
    if use_res:
        res = torch.randn(y.shape, generator=g)
        if prec == "fp16":
            res = res.half().float()
        y = y + res
    xd = x.permute(0, 2, 3, 1).contiguous().to(det.dtype).cuda()
    rd = res.permute(0, 2, 3, 1).contiguous().to(det.dtype).cuda() if use_res else None
    out = det.conv_bn_silu(xd, w.numpy(), b.numpy(), k, s, act, rd)
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert got.shape == y.shape
    scale = float(y.abs().max())
    tol = (4e-3 if prec == "fp16" else 2e-5) * max(scale, 1.0)
    err = float((got - y).abs().max())
    assert err <= tol, "max abs err %.3e > %.3e (scale %.2f)" % (err, tol, scale)


@pytest.mark.parametrize("case", [(70, 64, 64, 64, 64, True), (33, 40, 72, 64, 64, False), (48, 64, 64, 32, 32, True),
                                  (64, 64, 64, 128, 128, True), (40, 64, 64, 256, 256, False)])
def test_two_group_and_ring_kernels_are_repeatable(case):
    """Race screen for the kernels whose LDS hand-overs rest on counted waits and barriers alone (the two-group persistent
    64-channel kernel, the wide kernel's weight ring, the pixels-direct ring): the same launch twenty times must give the same
    bits every time -- a synchronisation slip shows as a rare wrong tile, not as a steady error."""
    B, H, W, Cin, Cout, use_res = case
    det = detector("fp16")
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    xd = torch.randn((B, H, W, Cin), generator=g).half().cuda()
    rd = torch.randn((B, H, W, Cout), generator=g).half().cuda() if use_res else None
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) / (Cin * 9) ** 0.5).numpy()
    b = (torch.randn((Cout,), generator=g) * 0.1).numpy()
    first = det.conv_bn_silu(xd, w, b, 3, 1, True, rd)
    for _ in range(20):
        again = det.conv_bn_silu(xd, w, b, 3, 1, True, rd)
        assert torch.equal(first, again)
    w1 = (torch.randn((Cout, Cin, 1, 1), generator=g) / Cin ** 0.5).numpy()
    first = det.conv_bn_silu(xd, w1, b, 1, 1, True, rd)
    for _ in range(20):
        assert torch.equal(first, det.conv_bn_silu(xd, w1, b, 1, 1, True, rd))


@pytest.mark.parametrize("w16", [False, True])
@pytest.mark.parametrize("seed", list(range(24)))
def test_conv_random_geometry_fp16x3(seed, w16):
    """The same seeded random geometries through the fp16x3 context (three-pass K walk of the wide / pixels-direct / generic
    kernels, scaled weights, high / low output halves) against F.conv2d in float64: 4e-6 of the output scale.
    w16: the weights are fp16 values (what an ultralytics checkpoint stores) -> the TWO-pass form [x_lo w | x_hi w] of every kernel."""
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.choice([1, 3]))
    s = int(rng.choice([1, 2])) if k == 3 else 1
    Cin = int(rng.choice([64, 128, 192, 256, 320]))
    Cout = int(rng.choice([40, 64, 96, 128, 160, 192, 256, 320]))
    H = int(rng.choice([15, 16, 31, 32, 33, 48, 64]))
    Wd = int(rng.choice([16, 30, 32, 34, 62, 64, 96]))
    B = int(rng.choice([1, 3, 8, 24]))
    act = bool(rng.integers(0, 2))
    use_res = bool(rng.integers(0, 2))
    det = detector("fp16x3")
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, Cin, H, Wd), generator=g)
    w = torch.randn((Cout, Cin, k, k), generator=g) / (Cin * k * k) ** 0.5
    w[0] *= 1e-3                                          # rows of very different magnitude: the per-channel weight scale
    w[-1] *= 50.0
    if w16:
        w = w.half().float()                              # (row 0 then holds fp16 subnormals: exact in the MFMA)
    b = torch.randn((Cout,), generator=g) * 0.1
    y = F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=k // 2)
    if act:
        y = F.silu(y)
    res = None
    if use_res:
        res = torch.randn(y.shape, generator=g)
        y = y + res.double()
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    rd = res.permute(0, 2, 3, 1).contiguous().cuda() if use_res else None
    out = det.conv_bn_silu(xd, w.numpy(), b.numpy(), k, s, act, rd)
    torch.cuda.synchronize()
    got = out.double().cpu().permute(0, 3, 1, 2)
    assert got.shape == y.shape
    scale = max(float(y.abs().max()), 1.0)
    err = float((got - y).abs().max())
    print("fp16x3%s B%d %dx%d %d->%d k%d s%d: max abs err %.3e of scale %.2f (%.2e)" % (" two-pass" if w16 else "", B, H, Wd, Cin, Cout, k, s, err, scale, err / scale))
    assert err <= 4e-6 * scale, "B%d %dx%d %d->%d k%d s%d act%d res%d: max abs err %.3e (scale %.2f)" % (
        B, H, Wd, Cin, Cout, k, s, act, use_res, err, scale)


@pytest.mark.parametrize("seed", list(range(24)))
def test_conv_random_geometry_fp16(seed, monkeypatch):
    """Seeded random layer geometries around the dispatch boundaries of launch_conv (channel counts that are / are not
    multiples of 64 and 128, map widths around 16/32, ragged pixel tails, stride 1/2, k 1/3, residual on/off), with the
    pixels-direct kernel forced on for half of the seeds so that it also sees shapes it would not pick by itself."""
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.choice([1, 3]))
    s = int(rng.choice([1, 2])) if k == 3 else 1
    Cin = int(rng.choice([64, 128, 192, 256, 320]))
    Cout = int(rng.choice([40, 64, 96, 128, 160, 192, 256, 320]))
    H = int(rng.choice([15, 16, 31, 32, 33, 48, 64]))
    Wd = int(rng.choice([16, 30, 32, 34, 62, 64, 96]))
    B = int(rng.choice([1, 3, 8, 24]))
    act = bool(rng.integers(0, 2))
    use_res = bool(rng.integers(0, 2))
    if seed % 2:
        monkeypatch.setenv("CY_DIRECT_MIN_BLOCKS", "1")
    monkeypatch.setenv("CY_BATCH_INVARIANT", str((seed // 2) % 2))
    det = detector("fp16")
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn((B, Cin, H, Wd), generator=g)).half().float()
    w = (torch.randn((Cout, Cin, k, k), generator=g) / (Cin * k * k) ** 0.5).half().float()
    b = torch.randn((Cout,), generator=g) * 0.1
    y = F.conv2d(x, w, b, stride=s, padding=k // 2)
    if act:
        y = F.silu(y)
    res = None
    if use_res:
        res = torch.randn(y.shape, generator=g).half().float()
        y = y + res
    xd = x.permute(0, 2, 3, 1).contiguous().half().cuda()
    rd = res.permute(0, 2, 3, 1).contiguous().half().cuda() if use_res else None
    out = det.conv_bn_silu(xd, w.numpy(), b.numpy(), k, s, act, rd)
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert got.shape == y.shape
    scale = max(float(y.abs().max()), 1.0)
    err = float((got - y).abs().max())
    assert err <= 4e-3 * scale, "B%d %dx%d %d->%d k%d s%d act%d res%d: max abs err %.3e (scale %.2f)" % (
        B, H, Wd, Cin, Cout, k, s, act, use_res, err, scale)


@pytest.mark.parametrize("B,H,W,shortcut", [(2, 16, 32, True), (3, 40, 70, True), (1, 8, 32, False), (5, 128, 128, True), (2, 13, 9, False),
                                            (300, 24, 40, True)])
def test_fused_bottleneck64(B, H, W, shortcut):
    """bneck64_kernel (two 3x3 64->64 convs + SiLU + shortcut in one launch, cv1's output kept in LDS as fp16) against the same
    two steps in torch: fp16-rounded input / weights, fp32 accumulation, the intermediate rounded to fp16 as the unfused path
    stores it.  Covers ragged patches (8 x 32 px), borders (zero padding of BOTH convs), batches above the persistent grid."""
    det = detector("fp16")
    g = torch.Generator().manual_seed(1000 * B + 10 * H + W)
    x = torch.randn((B, 64, H, W), generator=g).half().float()
    w1 = (torch.randn((64, 64, 3, 3), generator=g) / 24.0).half().float()
    w2 = (torch.randn((64, 64, 3, 3), generator=g) / 24.0).half().float()
    b1 = torch.randn((64,), generator=g) * 0.1
    b2 = torch.randn((64,), generator=g) * 0.1
    t = F.silu(F.conv2d(x, w1, b1, padding=1)).half().float()
    ref = F.silu(F.conv2d(t, w2, b2, padding=1))
    if shortcut:
        ref = ref + x
    xd = x.permute(0, 2, 3, 1).contiguous().half().cuda()
    got = det.bottleneck64(xd, w1.numpy(), b1.numpy(), w2.numpy(), b2.numpy(), shortcut)
    torch.cuda.synchronize()
    got = got.float().cpu().permute(0, 3, 1, 2)
    sc = max(1.0, float(ref.abs().max()))
    err = float((got - ref).abs().max())
    assert err <= 4e-3 * sc, "max abs err %.3e (scale %.2f)" % (err, sc)


@pytest.mark.parametrize("case", [(70, 64, 64, 128, 128, True), (40, 64, 64, 128, 128, False), (33, 40, 60, 128, 256, True), (21, 52, 64, 256, 192, True),
                                  (300, 32, 32, 256, 256, False), (9, 64, 96, 64, 128, True), (2, 16, 32, 128, 128, False)])
def test_wide_persistent_kernel_is_bit_identical(case, monkeypatch):
    """conv3x3_widep_kernel (one workgroup per CU walking the patches, the next patch's prologue requested behind the last stage
    barrier, counted vmcnt waits around the epilogue) does the arithmetic of the one-patch wide kernel in the same order: same bits,
    on 1..3 patches per workgroup, odd and even counts, ragged rows / columns, a half-empty second channel tile, with and without
    the residual input -- and repeatably (the hand-overs rest on counted waits)."""
    B, H, W, Cin, Cout, use_res = case
    det = detector("fp16", max_batch=4)
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    xd = torch.randn((B, H, W, Cin), generator=g).half().cuda()
    rd = torch.randn((B, H, W, Cout), generator=g).half().cuda() if use_res else None
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) / (Cin * 9) ** 0.5).numpy()
    b = (torch.randn((Cout,), generator=g) * 0.1).numpy()
    monkeypatch.setenv("CY_WIDE_PERSIST", "0")
    ref = det.conv_bn_silu(xd, w, b, 3, 1, True, rd).clone()
    monkeypatch.setenv("CY_WIDE_PERSIST", "2")
    for _ in range(6):
        got = det.conv_bn_silu(xd, w, b, 3, 1, True, rd)
        torch.cuda.synchronize()
        assert torch.equal(ref, got)


@pytest.mark.parametrize("mf", ["32", "33"])
@pytest.mark.parametrize("case", [(3, 64, 64, 128, 128, True), (2, 40, 64, 256, 192, True), (33, 40, 60, 128, 256, False), (1, 16, 32, 64, 128, False),
                                  (2, 30, 60, 64, 256, True), (5, 32, 32, 512, 256, False)])
def test_wide_kernel_on_32x32x16_mfma(case, mf, monkeypatch):
    """conv3x3_wide32_kernel (round 4 A/B: the wide kernel's K loop on v_mfma_f32_32x32x16_f16; CY_WIDE_MFMA=32 compiler-placed
    fragment reads, 33 one read per MFMA gap): against F.conv2d on fp16-rounded inputs (K is summed 16 channels per instruction
    instead of 32, so not bit-identical to the 16x16x32 kernel), ragged rows / columns / channel tiles, residual, repeatable."""
    monkeypatch.setenv("CY_WIDE_MFMA", mf)
    monkeypatch.setenv("CY_STRIP", "0")           # ragged maps would otherwise take the strip form
    B, H, W, Cin, Cout, use_res = case
    det = detector("fp16", max_batch=4)
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn((B, Cin, H, W), generator=g).half().float()
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) / (Cin * 9) ** 0.5).half().float()
    b = torch.randn((Cout,), generator=g) * 0.1
    y = F.silu(F.conv2d(x, w, b, padding=1))
    res = None
    if use_res:
        res = torch.randn(y.shape, generator=g).half().float()
        y = y + res
    xd = x.permute(0, 2, 3, 1).contiguous().half().cuda()
    rd = res.permute(0, 2, 3, 1).contiguous().half().cuda() if use_res else None
    out = det.conv_bn_silu(xd, w.numpy(), b.numpy(), 3, 1, True, rd)
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 3, 1, 2)
    scale = max(float(y.abs().max()), 1.0)
    err = float((got - y).abs().max())
    assert err <= 4e-3 * scale, "max abs err %.3e (scale %.2f)" % (err, scale)
    for _ in range(4):
        assert torch.equal(out, det.conv_bn_silu(xd, w.numpy(), b.numpy(), 3, 1, True, rd))


@pytest.mark.parametrize("case", [(5, 80, 80, 128, 128, True), (7, 40, 40, 256, 256, False), (9, 20, 20, 512, 512, True), (3, 52, 64, 128, 128, True),
                                  (4, 26, 32, 256, 192, False), (2, 33, 47, 64, 128, True), (1, 17, 95, 128, 144, False), (13, 20, 20, 128, 128, False),
                                  (2, 100, 126, 64, 128, True),
                                  # maps of a few pixels (the stride-16 / 32 levels of 128- and 256-px inputs): several rows per 16-entry
                                  # fragment, several IMAGES per wave (81 / 25 entries per image), ragged 6 x 10 and 3 x 3 maps
                                  (37, 8, 8, 256, 256, True), (70, 4, 4, 512, 512, False), (9, 6, 10, 128, 192, True), (50, 3, 3, 64, 128, False),
                                  (5, 12, 12, 128, 128, True),
                                  # 64 output channels (half of the channel tile empty): the box head on those maps
                                  (37, 8, 8, 256, 64, False), (70, 4, 4, 512, 64, False)])
def test_strip_form_of_the_wide_kernel(case, monkeypatch):
    """conv3x3_widep_kernel<strip>: the batch flattened to one dimension with one shared zero entry per row and one zero row per image
    (map sizes from 17 px on and the maps of 3-12 px; the 80 / 40 / 20-px maps of 640-px inputs, the 52 x 64 / 26 x 32 maps of ragged tiles, the 8 x 8 / 4 x 4 maps of 128-px inputs), forced on
    these small launches: against F.conv2d on fp16-rounded inputs, and bit-for-bit repeatable.  Cases: last strip partly past the
    batch, rows shorter and longer than a 16-entry fragment step, both halo sizes (10 / 12 pieces), half-empty channel tile."""
    monkeypatch.setenv("CY_STRIP", "2")
    B, H, W, Cin, Cout, use_res = case
    det = detector("fp16", max_batch=4)
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn((B, Cin, H, W), generator=g).half().float()
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) / (Cin * 9) ** 0.5).half().float()
    b = torch.randn((Cout,), generator=g) * 0.1
    y = F.silu(F.conv2d(x, w, b, padding=1))
    res = None
    if use_res:
        res = torch.randn(y.shape, generator=g).half().float()
        y = y + res
    xd = x.permute(0, 2, 3, 1).contiguous().half().cuda()
    rd = res.permute(0, 2, 3, 1).contiguous().half().cuda() if use_res else None
    out = det.conv_bn_silu(xd, w.numpy(), b.numpy(), 3, 1, True, rd)
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 3, 1, 2)
    scale = max(float(y.abs().max()), 1.0)
    err = float((got - y).abs().max())
    assert err <= 4e-3 * scale, "max abs err %.3e (scale %.2f)" % (err, scale)
    for _ in range(4):
        assert torch.equal(out, det.conv_bn_silu(xd, w.numpy(), b.numpy(), 3, 1, True, rd))


@pytest.mark.parametrize("w16", [False, True])
@pytest.mark.parametrize("case", [(5, 80, 80, 128, 128, True), (7, 40, 40, 256, 256, False), (9, 20, 20, 512, 256, True), (3, 52, 64, 128, 128, True),
                                  (2, 33, 47, 64, 192, True), (2, 100, 126, 64, 128, False), (40, 64, 64, 128, 128, True), (3, 32, 64, 256, 256, False),
                                  (37, 8, 8, 256, 256, True), (70, 4, 4, 512, 256, False), (9, 6, 10, 128, 192, True), (70, 4, 4, 512, 64, False)])
def test_fp16x3_persistent_and_strip_forms(case, w16, monkeypatch):
    """fp16x3 context: the strip form of the persistent wide kernel (three-pass K walk over the flattened batch, scaled / split epilogue,
    residual as high + low halves requested two fragments at a time) and, with CY_X3_PERSIST=1, its 2-D form -- against F.conv2d in
    float64 (4e-6 of the output scale) and bit-identical to the one-patch fp16x3 wide kernel where that one applies."""
    B, H, W, Cin, Cout, use_res = case
    det = detector("fp16x3")
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn((B, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, 3, 3), generator=g) / (Cin * 9) ** 0.5
    if w16:
        w = w.half().float()                  # fp16-exact filter: the two-pass form of the same kernels
    b = torch.randn((Cout,), generator=g) * 0.1
    y = F.silu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    res = torch.randn(y.shape, generator=g) if use_res else None
    if use_res:
        y = y + res.double()
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    rd = res.permute(0, 2, 3, 1).contiguous().cuda() if use_res else None
    outs = {}
    for strip, pers in (("0", "0"), ("1", "1")):
        monkeypatch.setenv("CY_STRIP", strip)
        monkeypatch.setenv("CY_X3_PERSIST", pers)
        out = det.conv_bn_silu(xd, w.numpy(), b.numpy(), 3, 1, True, rd)
        torch.cuda.synchronize()
        outs[strip] = out.clone()
        got = out.double().cpu().permute(0, 3, 1, 2)
        scale = max(float(y.abs().max()), 1.0)
        err = float((got - y).abs().max())
        assert err <= 4e-6 * scale, "strip %s: max abs err %.3e (scale %.2f)" % (strip, err, scale)
    if W % 32 == 0 and H % 16 == 0:          # both runs took a form of the wide kernel: same arithmetic in the same order
        assert torch.equal(outs["0"], outs["1"])
