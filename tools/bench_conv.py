"""Single-layer conv microbenchmark through the C-ABI test entry (developer tool).
python tools/bench_conv.py B H W Cin Cout k s [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from caesar_yolo_amd.model import HipDetector
from caesar_yolo_amd import weights as W
a = [int(x) for x in sys.argv[1:8]]
B, H, Wd, Cin, Cout, k, s = a
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 20
wp = "/tmp/cy_bench_seed.cyw"
if not os.path.exists(wp):
    W.make_seeded_file(wp, "l", 5)
det = HipDetector(wp, device=0, precision="fp16", max_batch=1, max_imgsz=64)
x = torch.randn((B, H, Wd, Cin), device="cuda").half()
w = (np.random.default_rng(0).standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
b = np.zeros(Cout, np.float32)
for _ in range(3):
    y = det.conv_bn_silu(x, w, b, k, s, True)
torch.cuda.synchronize()
# cy_conv_bn_silu synchronises and re-uploads weights per call: time with events around the launch only is not possible
# from here, so report wall per call for orientation; use rocprofv3 for kernel time.
t = time.time()
for _ in range(iters):
    y = det.conv_bn_silu(x, w, b, k, s, True)
torch.cuda.synchronize()
dt = (time.time() - t) / iters
fl = 2.0 * B * (H // s) * (Wd // s) * Cout * Cin * k * k
print("wall/call %.3f ms (includes weight upload); %.1f GFLOP" % (dt * 1e3, fl / 1e9))
