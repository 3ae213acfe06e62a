"""Where does the fp16x3 context's error come from?  One conv layer against float64: exact-fp32 context, fp16x3 context and
torch-CPU fp32, with / without SiLU, for several K."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, torch.nn.functional as F
import __graft_entry__ as ge
ge.build()
from gpu_common import detector
for (Cin, k, act) in [(64, 1, False), (128, 3, False), (128, 3, True), (512, 3, False), (512, 3, True), (768, 1, True)]:
    g = torch.Generator().manual_seed(Cin + k)
    x = torch.randn((4, Cin, 64, 64), generator=g)
    w = torch.randn((128, Cin, k, k), generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn((128,), generator=g) * 0.1
    y = F.conv2d(x.double(), w.double(), b.double(), padding=k // 2)
    if act:
        y = F.silu(y)
    yc = F.conv2d(x, w, b, padding=k // 2)
    if act:
        yc = F.silu(yc)
    row = "Cin %4d k%d act%d: rms|y| %.2f  cpu-fp32 rms %.2e max %.2e" % (Cin, k, act, float(y.pow(2).mean().sqrt()), float((yc.double() - y).pow(2).mean().sqrt()), float((yc.double() - y).abs().max()))
    for prec in ("fp32", "fp16x3"):
        det = detector(prec)
        out = det.conv_bn_silu(x.permute(0, 2, 3, 1).contiguous().cuda(), w.numpy(), b.numpy(), k, 1, act, None)
        torch.cuda.synchronize()
        e = out.double().cpu().permute(0, 3, 1, 2) - y
        row += " | %s rms %.2e max %.2e" % (prec, float(e.pow(2).mean().sqrt()), float(e.abs().max()))
    print(row)
