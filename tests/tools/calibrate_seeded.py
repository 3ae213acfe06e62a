"""Dev tool (not product code): fixed-point calibration of the seeded checkpoint's BatchNorm variances so that
every pre-activation of the random-init network is ~unit variance on a synthetic radio tile.  Uses the CPU oracle's
forward; writes the per-conv variance table consumed by caesar_yolo_amd.weights.seeded_checkpoint.

    python tests/tools/calibrate_seeded.py [scale]
"""
import json, os, sys, warnings
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
from caesar_yolo_amd import weights as W          # noqa: E402
from caesar_yolo_amd import yolov8_spec as S      # noqa: E402
from oracle import yolov8_ref as Y                # noqa: E402
from oracle import preprocessing_ref as P         # noqa: E402


class TapNet(Y.Net):
    def conv(self, name, x, s=1, act=True):
        w, b = self.w[name]
        y = F.conv2d(x, w, b, stride=s, padding=w.shape[-1] // 2)
        bb = b.view(1, -1, 1, 1)
        # BN convs: total pre-activation std; bias-only head convs: std of the weight term alone
        sd = float(y.std()) if act else float((y - bb).std())
        self.pre[name] = sd
        if self.fix:                       # rescale on the fly: folded weights scale as 1/sqrt(table value)
            tot = 1.0
            for _ in range(3 if act else 1):
                sd = float(y.std()) if act else float((y - bb).std())
                y = (y - bb) / sd + bb
                tot *= sd
            self.scale[name] = tot
        return F.silu(y) if act else y


def main():
    scale = sys.argv[1] if len(sys.argv) > 1 else "l"
    nc = 5
    g = np.load(os.path.join(ROOT, "tests/golden/preproc.npz"))
    dp = P.build_pipeline([("zscale", dict(contrasts=[0.25] * 3)), ("minmax", dict(norm_min=0, norm_max=255))])
    x, _ = Y.preprocess(dp(P.to_cube(g["in/big512"])), 512)
    table = dict(W._input_moments(scale, nc))
    for it in range(8):
        ck = W.seeded_checkpoint(scale, nc, var_table=table)
        wd = {cs.name: (w, b) for cs, w, b in W.fold(ck, scale, nc)}
        net = TapNet(wd, scale, nc)
        net.pre, net.scale, net.fix = {}, {}, (it % 2 == 0)
        with torch.no_grad():
            raw = net.forward(x)
        worst = max(abs(np.log(v)) for v in net.pre.values())
        print("iter %d (fix=%d): worst |log std| = %.3f, cls logit std %.3f" % (it, net.fix, worst, float(raw[0, 64:].std())))
        if not net.fix and worst < 0.05:
            break
        for k, v in net.scale.items():
            table[k] *= v * v
    path = os.path.join(ROOT, "caesar_yolo_amd", "seeded_calibration.json")
    allt = {}
    if os.path.exists(path):
        allt = json.load(open(path))
    allt[scale] = {k: float("%.6g" % v) for k, v in table.items()}
    json.dump(allt, open(path, "w"), indent=0, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
