"""Per-layer timing of cy_forward with hipEvents (developer tool).  python tools/profile_layers.py [B] [H] [prec]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ge.build()
from caesar_yolo_amd.model import YOLO

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
prec = sys.argv[3] if len(sys.argv) > 3 else "fp16"
m = YOLO("seeded:l:5", precision=prec, max_batch=B, max_imgsz=H, device=0)
det = m.engine(0)
x = torch.rand((B, H, H, 4), device="cuda").to(det.dtype)
for _ in range(2):
    det.forward(x)
torch.cuda.synchronize()
det.profile(True)
R = 3
for _ in range(R):
    det.forward(x)
torch.cuda.synchronize()
rows = det.profile_layers()
tot = sum(r["ms"] for r in rows)
print("%-24s %9s %9s %8s %7s" % ("conv", "ms/fwd", "GFLOP", "TFLOP/s", "share"))
for r in rows:
    if r["launches"]:
        ms = r["ms"] / R
        print("%-24s %9.3f %9.2f %8.1f %6.2f%%" % (r["name"], ms, r["flops"] / R / 1e9, r["flops"] / r["ms"] / 1e9, 100 * r["ms"] / tot))
print("total conv+stem ms/fwd %.3f -> %.1f tiles/s forward-only, %.1f TFLOP/s" % (tot / R, B / (tot / R) * 1e3, sum(r["flops"] for r in rows) / tot / 1e9))
