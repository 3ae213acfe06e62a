"""Dev tool (test infrastructure: calibrates seeded weights with the CPU oracle): per-kernel time of a YOLO11 forward.
python tests/tools/profile_yolo11.py [scale] [B] [H]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
ge.build()
from caesar_yolo_amd import weights as W
from caesar_yolo_amd.model import HipDetector
from yolo11_common import seeded_folded
scale = sys.argv[1] if len(sys.argv) > 1 else "l"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H = int(sys.argv[3]) if len(sys.argv) > 3 else 512
g, wd = seeded_folded(scale, 5)
path = "/tmp/y11_%s.cyw" % scale
W.write_cyw2(path, g, [(cs, wd[cs.name][0], wd[cs.name][1]) for cs in g.convs], {i: "c%d" % i for i in range(5)})
det = HipDetector(path, device=0, precision="fp16", max_batch=B, max_imgsz=H)
x = torch.rand((B, H, H, 4), device="cuda").half()
for _ in range(2):
    det.forward(x)
torch.cuda.synchronize()
det.profile(True)
R = 3
for _ in range(R):
    det.forward(x)
torch.cuda.synchronize()
rows = [r for r in det.profile_summary() if r["launches"]]
tot = sum(r["ms"] for r in rows)
for r in sorted(rows, key=lambda r: -r["ms"]):
    print("%-62s %8.3f ms/fwd %6d launches %8.1f TFLOP/s %5.1f%%" % (r["kernel"][:62], r["ms"] / R, r["launches"] // R,
                                                                  r["flops"] / r["ms"] / 1e9 if r["ms"] else 0, 100 * r["ms"] / tot))
print("yolo11%s B=%d %dx%d: %.3f ms/fwd -> %.1f tiles/s forward-only, %.1f TFLOP/s" % (scale, B, H, H, tot / R, B / (tot / R) * 1e3,
                                                                                     sum(r["flops"] for r in rows) / tot / 1e9))
