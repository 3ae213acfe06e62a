import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from gpu_common import detector
det = detector("fp16", max_batch=4)
for (B, H, W, Cin, Cout, res) in [(39, 52, 64, 128, 128, True), (39, 26, 32, 256, 256, True), (39, 52, 64, 256, 256, False), (39, 64, 52, 128, 128, True), (7, 80, 80, 128, 128, True), (5, 40, 40, 256, 256, False), (1, 52, 64, 128, 128, True)]:
    g = torch.Generator().manual_seed(B * 1000 + H)
    xd = torch.randn((B, H, W, Cin), generator=g).half().cuda()
    rd = torch.randn((B, H, W, Cout), generator=g).half().cuda() if res else None
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) / (Cin * 9) ** 0.5).numpy()
    b = (torch.randn((Cout,), generator=g) * 0.1).numpy()
    os.environ["CY_STRIP"] = "0"; os.environ["CY_WIDE_PERSIST"] = "0"
    ref = det.conv_bn_silu(xd, w, b, 3, 1, True, rd).clone()
    os.environ["CY_STRIP"] = "2"
    got = det.conv_bn_silu(xd, w, b, 3, 1, True, rd)
    torch.cuda.synchronize()
    d = (ref.float() - got.float()).abs()
    print((B, H, W, Cin, Cout, res), "equal" if torch.equal(ref, got) else "DIFF max %.3e at %d positions" % (float(d.max()), int((d > 0).sum())))
