"""Post-processing kernels with the GPU to themselves (developer tool): decode + NMS + IoU merge of the head output of one
batch of real benchmark tiles, timed with events.  python tools/time_post.py [B]   (run under rocprofv3 --kernel-trace --stats
for the per-kernel split)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ge.build()
from caesar_yolo_amd.model import YOLO

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = YOLO("seeded:l:5", precision="fp16", max_batch=B, max_imgsz=512, device=0)
det = m.engine(0)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand((B, 512, 512, 4), device="cuda", generator=g).half()
pred = det.forward(x).clone()
torch.cuda.synchronize()
for conf in (0.7, 0.25):
    for _ in range(2):
        d, a, c = det.decode_nms(pred, 512, 512, 512, 512, conf, 0.5)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    for _ in range(10):
        d, a, c = det.decode_nms(pred, 512, 512, 512, 512, conf, 0.5)
    e1.record()
    for _ in range(10):
        o, oc, osrc = det.iou_merge(d, c, conf, 0.3, 0.8)
    e2.record()
    torch.cuda.synchronize()
    print("conf %.2f: decode+nms %.3f ms, iou merge %.3f ms per %d tiles (incl. output allocation); mean detections %.1f"
          % (conf, e0.elapsed_time(e1) / 10, e1.elapsed_time(e2) / 10, B, float(c.float().mean())))
