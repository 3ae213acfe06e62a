"""Phase anatomy of the pixels-direct 1x1 / 3x3-s2 kernel (developer tool; diagnostic build, run with CY_DBG=64).
python tools/stamp_direct.py B H W Cin Cout [k s]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from caesar_yolo_amd.model import HipDetector
from caesar_yolo_amd import weights as W, lib as L
B, H, Wd, Cin, Cout = [int(x) for x in sys.argv[1:6]]
k = int(sys.argv[6]) if len(sys.argv) > 6 else 1
s = int(sys.argv[7]) if len(sys.argv) > 7 else 1
wp = "/tmp/cy_bench_seed.cyw"
if not os.path.exists(wp):
    W.make_seeded_file(wp, "l", 5)
det = HipDetector(wp, device=0, precision="fp16", max_batch=1, max_imgsz=64)
x = torch.randn((B, H, Wd, Cin), device="cuda").half()
w = (np.random.default_rng(0).standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
b = np.zeros(Cout, np.float32)
st = (C.c_ulonglong * 8)()
for i in range(5):
    if i == 2:
        L.load().cy_debug_stamps(st, 1)
    det.conv_bn_silu(x, w, b, k, s, True)
L.load().cy_debug_stamps(st, 0)
pro, loop, epi, cyc, tot, _, n = [float(st[i]) for i in range(7)]
chunks = Cin // 64 * k * k
u = 0.01
print("workgroups sampled %d, %d chunks; us per workgroup: entry->loop %.2f | loop %.2f (%.3f per chunk) | epilogue %.2f | clock in the loop %.0f MHz"
      % (n, chunks, u * pro / n, u * loop / n, u * loop / n / chunks, u * epi / n, cyc / (u * loop)))
