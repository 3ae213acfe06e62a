"""In-kernel cycle anatomy of the 3x3 halo kernel's tap loop (developer tool; run with CY_DBG=64)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from caesar_yolo_amd.model import HipDetector
from caesar_yolo_amd import weights as W, lib as L
B, H, Wd, Cin, Cout = [int(x) for x in sys.argv[1:6]]
wp = "/tmp/cy_bench_seed.cyw"
if not os.path.exists(wp):
    W.make_seeded_file(wp, "l", 5)
det = HipDetector(wp, device=0, precision="fp16", max_batch=1, max_imgsz=64)
x = torch.randn((B, H, Wd, Cin), device="cuda").half()
w = (np.random.default_rng(0).standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
b = np.zeros(Cout, np.float32)
st = (C.c_ulonglong * 8)()
for i in range(4):
    if i == 1:
        L.load().cy_debug_stamps(st, 1)
    det.conv_bn_silu(x, w, b, 3, 1, True)
L.load().cy_debug_stamps(st, 0)
dma, cmp_, wait, bar, tot, taps, waves = [st[i] for i in range(7)]
print("waves %d, taps/wave %.0f; cycles per tap per wave: dma-issue %.0f | reads+mfma %.0f | waitcnt %.0f | barrier %.0f | loop total %.0f (ideal mfma 512)"
      % (waves, taps / waves, dma / taps, cmp_ / taps, wait / taps, bar / taps, tot / taps))
