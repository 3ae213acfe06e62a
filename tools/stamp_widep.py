"""Phase anatomy of the persistent wide 3x3 kernel (developer tool; diagnostic build `CY_STAMPS=1 python __graft_entry__.py --force`,
run with CY_DBG=64 CY_WIDE_PERSIST=2).  python tools/stamp_widep.py B H W Cin Cout [res]"""
import os, sys, ctypes as C
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
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(6):
    if i == 2:
        L.load().cy_debug_stamps(st, 1)
        ev0.record()
    det.conv_bn_silu(x, w, b, 3, 1, True, res)
L.load().cy_debug_stamps(st, 0)
req, loop, epi, wait, tot, _, n = [float(st[i]) for i in range(7)]
u = 0.01
print("workgroups %d; us per patch (second patch of each workgroup): requests + residual wait %.2f | stage loop %.2f (%.3f per stage) | epilogue %.2f | "
      "wait + barrier %.2f | total %.2f" % (n, u * req / n, u * loop / n, u * loop / n / ((Cin // 64) * 9), u * epi / n, u * wait / n, u * (req + loop + epi + wait) / n))
raw = (C.c_ulonglong * (256 * 4))()
L.load().cy_debug_stamps(raw, -256)
life = np.array([raw[i * 4 + 3] for i in range(256)], np.float64) * u
npatch = B * ((Wd + 31) // 32) * ((H + 15) // 16) * ((Cout + 127) // 128) / 256.0
print("workgroup life (us): min %.1f  median %.1f  p90 %.1f  max %.1f  (%.1f patches each -> %.2f us per patch at the median, %.2f at the max)"
      % (life.min(), np.median(life), np.percentile(life, 90), life.max(), npatch, np.median(life) / npatch, life.max() / npatch))
