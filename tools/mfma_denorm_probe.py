"""Does v_mfma_f32_16x16x32_f16 honour fp16 subnormal inputs on this chip?  (fp16x3 context: the low halves of small
activations are subnormal.)  Through the C-ABI conv entry of the fp16 context: x = 2^-20 (subnormal), w = 1024, 1x1 conv over
64 channels without activation -> 64 * 2^-10 = 0.0625 if honoured, 0 if flushed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
ge.build()
from caesar_yolo_amd.model import HipDetector
from caesar_yolo_amd import weights as W
wp = "/tmp/cy_probe_l5.cyw"
if not os.path.exists(wp):
    W.make_seeded_file(wp, "n", 5)
det = HipDetector(wp, device=0, precision="fp16", max_batch=1, max_imgsz=64)
for e in (-14, -16, -20, -24):
    x = torch.full((1, 16, 16, 64), 2.0 ** e, dtype=torch.float16, device="cuda")
    w = np.full((128, 64, 1, 1), 1024.0, np.float32)
    b = np.zeros((128,), np.float32)
    y = det.conv_bn_silu(x, w, b, 1, 1, act=False)
    torch.cuda.synchronize()
    print("x = 2^%d: out %.6g (expected %.6g)" % (e, float(y.float().max()), 64 * 1024.0 * 2.0 ** e))
