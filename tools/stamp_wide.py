"""In-kernel phase anatomy of the wide 3x3 kernel (developer tool; diagnostic build `CY_STAMPS=1 python __graft_entry__.py --force`,
run with CY_DBG=64).  python tools/stamp_wide.py B H W Cin Cout [res]"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from caesar_yolo_amd.model import HipDetector
from caesar_yolo_amd import weights as W, lib as L
B, H, Wd, Cin, Cout = [int(x) for x in sys.argv[1:6]]
use_res = len(sys.argv) > 6 and sys.argv[6] == "1"
wp = "/tmp/cy_bench_seed.cyw"
if not os.path.exists(wp):
    W.make_seeded_file(wp, "l", 5)
det = HipDetector(wp, device=0, precision="fp16", max_batch=1, max_imgsz=64)
x = torch.randn((B, H, Wd, Cin), device="cuda").half()
res = torch.randn((B, H, Wd, Cout), device="cuda").half() if use_res else None
w = (np.random.default_rng(0).standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
b = np.zeros(Cout, np.float32)
st = (C.c_ulonglong * 8)()
for i in range(5):
    if i == 2:
        L.load().cy_debug_stamps(st, 1)
    det.conv_bn_silu(x, w, b, 3, 1, True, res)
L.load().cy_debug_stamps(st, 0)
pro, loop, epi, drain, tot, stages, waves = [float(st[i]) for i in range(7)]
stages = waves * (Cin // 64) * 9     # records are per workgroup (wave 0)
u = 0.01   # s_memrealtime: 100 MHz
print("workgroups sampled %d, stages %.0f; us per workgroup: entry->loop %.2f | loop %.2f (%.3f per stage) | epilogue issue %.2f | s_memtime ticks per us in the loop %.0f | total %.2f"
      % (waves, stages / waves, u * pro / waves, u * loop / waves, u * loop / stages, u * epi / waves, drain / (u * loop), u * (pro + loop + epi) / waves))
