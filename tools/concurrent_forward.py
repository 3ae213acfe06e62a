"""Developer experiment: one forward of B tiles against two concurrent forwards of B/2 on two streams (two contexts,
each with its own workspace): do the tails of one stream's kernels fill with the other's?  python tools/concurrent_forward.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ge.build()
from caesar_yolo_amd.model import YOLO

BS = [int(v) for v in sys.argv[1:]] or [256]
BM = max(BS)
H = 512
m1 = YOLO("seeded:l:5", precision="fp16", max_batch=BM, max_imgsz=H, device=0)
m2 = YOLO("seeded:l:5", precision="fp16", max_batch=(BM + 1) // 2, max_imgsz=H, device=0)
d1, d2 = m1.engine(0), m2.engine(0)
xx = torch.rand((BM, H, H, 4), device="cuda").to(d1.dtype)
s3 = torch.cuda.Stream()
for B in BS:
    x = xx[:B]
    h = B - B // 2
    xa, xb = x[:h], x[h:]

    def one():
        d1.forward(x)

    def two():
        cur = torch.cuda.current_stream()
        s3.wait_stream(cur)
        d1.forward(xa)
        with torch.cuda.stream(s3):
            d2.forward(xb)
        cur.wait_stream(s3)

    res = []
    for f in (one, two, one, two):
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        t = time.time()
        R = 5
        for _ in range(R):
            f()
        torch.cuda.synchronize()
        res.append((time.time() - t) / R * 1e3)
    print("B %3d: one batch %.3f / %.3f ms, two concurrent halves %.3f / %.3f ms  (%+.1f %%)" % (
        B, res[0], res[2], res[1], res[3], 100.0 * (min(res[1], res[3]) / min(res[0], res[2]) - 1.0)))
