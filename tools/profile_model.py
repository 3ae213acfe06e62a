"""Per-layer hipEvent timing of cy_forward for any supported graph (developer tool).
python tools/profile_model.py <v8n|v8s|v8m|v8l|v8x|11n|11l> [B] [H] [prec]
YOLO11 weights: seeded through tests/yolo11_common.py (uses the oracle to calibrate them: tool only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
ge.build()
from caesar_yolo_amd.model import HipDetector
from caesar_yolo_amd import weights as W

spec = sys.argv[1] if len(sys.argv) > 1 else "v8l"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
H = int(sys.argv[3]) if len(sys.argv) > 3 else 512
prec = sys.argv[4] if len(sys.argv) > 4 else "fp16"
path = "/tmp/cy_prof_%s.cyw" % spec
if not os.path.exists(path):
    if spec.startswith("11"):
        from yolo11_common import seeded_folded
        g, wd = seeded_folded(spec[2:], 5)
        W.write_cyw2(path, g, [(cs, wd[cs.name][0], wd[cs.name][1]) for cs in g.convs], {i: "c%d" % i for i in range(5)})
    else:
        W.make_seeded_file(path, spec[2:], 5)
det = HipDetector(path, device=0, precision=prec, max_batch=B, max_imgsz=H)
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
print("%-28s %9s %9s %8s %7s" % ("layer", "ms/fwd", "GFLOP", "TFLOP/s", "share"))
for r in rows:
    if r["launches"]:
        print("%-28s %9.3f %9.2f %8.1f %6.2f%%" % (r["name"], r["ms"] / R, r["flops"] / R / 1e9, r["flops"] / max(r["ms"], 1e-9) / 1e9, 100 * r["ms"] / tot))
print("%s B=%d H=%d %s: profiled ms/fwd %.3f -> %.1f tiles/s forward-only, %.1f TFLOP/s" % (spec, B, H, prec, tot / R, B / (tot / R) * 1e3, sum(r["flops"] for r in rows) / tot / 1e9))
torch.cuda.synchronize()
import time
det.profile(False)
t = time.time()
for _ in range(5):
    det.forward(x)
torch.cuda.synchronize()
dt = (time.time() - t) / 5
print("wall per forward (all kernels, profiling off): %.3f ms -> %.1f tiles/s" % (dt * 1e3, B / dt))
