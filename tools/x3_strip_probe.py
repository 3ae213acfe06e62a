import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, torch.nn.functional as F
from gpu_common import detector
det = detector("fp16x3")
os.environ["CY_STRIP"] = "1"; os.environ["CY_X3_PERSIST"] = "1"
for case in [(9,20,20,512,256,True),(9,20,20,512,256,False),(9,20,20,128,256,True),(9,20,20,512,128,True),(9,40,40,512,256,True),(2,20,20,512,256,True),
             (40,64,64,128,128,True),(40,64,64,128,128,False),(70,64,64,128,256,False),(300,32,32,256,256,False),(300,20,20,128,128,False)]:
    B,H,W,Cin,Cout,use_res = case
    g = torch.Generator().manual_seed(1)
    x = torch.randn((B,Cin,H,W), generator=g); w = torch.randn((Cout,Cin,3,3), generator=g)/(Cin*9)**0.5; b = torch.randn((Cout,), generator=g)*0.1
    y = F.silu(F.conv2d(x, w, b, padding=1))
    res = torch.randn(y.shape, generator=g) if use_res else None
    if use_res: y = y + res
    out = det.conv_bn_silu(x.permute(0,2,3,1).contiguous().cuda(), w.numpy(), b.numpy(), 3, 1, True, res.permute(0,2,3,1).contiguous().cuda() if use_res else None)
    torch.cuda.synchronize()
    got = out.cpu().permute(0,3,1,2)
    bad = ~torch.isfinite(got)
    d = (got - y).abs(); d[bad] = 0
    wrong = (d > 1e-3)
    idx = torch.nonzero(bad | wrong)
    print(case, "nan/inf %d, wrong %d of %d, max err %.2e" % (int(bad.sum()), int(wrong.sum()), got.numel(), float(d.max())),
          ("first bad (b,c,y,x): %s ... last %s" % (idx[0].tolist(), idx[-1].tolist())) if len(idx) else "")
