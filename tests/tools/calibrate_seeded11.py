"""Dev tool (not product code): per-convolution normalisation factors for the seeded YOLO11 weights
(caesar_yolo_amd.weights.seeded11_folded).  One sequential pass of the CPU oracle over a random probe image: every conv's
random draw is divided by the std of its pre-activation, so activations stay O(1) through 90-170 convolutions.  The factors
are plain data: caesar_yolo_amd/seeded11_calibration.json, keyed "<scale>:<nc>:<seed>".

    python tests/tools/calibrate_seeded11.py [scale] [nc] [seed]
"""
import json, os, sys, warnings
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
from caesar_yolo_amd import weights as W          # noqa: E402
from caesar_yolo_amd import yolo11_graph as G     # noqa: E402
from oracle import yolo11_ref as O                # noqa: E402


class _Calibrating(O.Net11):
    def conv(self, name, x, s=1, act=True):
        w, b = self.w[name]
        groups = x.shape[1] // w.shape[1]
        y0 = F.conv2d(x, w, None, stride=s, padding=w.shape[-1] // 2, groups=groups)
        sd = float(y0.std())
        sd = sd if sd > 0 else 1.0
        self.sd[name] = sd
        # exactly what seeded11_folded will hand out: the draw divided by sd in fp32, rounded to fp16
        w = torch.from_numpy((w.numpy() / np.float32(sd)).astype(np.float16).astype(np.float32))
        self.w[name] = (w, b)
        y = F.conv2d(x, w, b, stride=s, padding=w.shape[-1] // 2, groups=groups)
        return F.silu(y) if act else y


def main():
    scale = sys.argv[1] if len(sys.argv) > 1 else "l"
    nc = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 11
    g = G.build(scale, nc)
    wd = W.seeded11_draw(g, seed)
    net = _Calibrating({k: (torch.from_numpy(v[0]), torch.from_numpy(v[1])) for k, v in wd.items()}, scale, nc)
    net.sd = {}
    x = torch.from_numpy(np.random.default_rng(seed + 1000).uniform(0, 1, (1, 3, 128, 128)).astype(np.float32))
    with torch.no_grad():
        net.forward(x)
    path = os.path.join(ROOT, "caesar_yolo_amd", "seeded11_calibration.json")
    allt = json.load(open(path)) if os.path.exists(path) else {}
    allt["%s:%d:%d" % (scale, nc, seed)] = {k: float(np.float32(v)) for k, v in net.sd.items()}
    json.dump(allt, open(path, "w"), indent=0, sort_keys=True)
    print("wrote %d factors for yolo11%s nc=%d seed %d to %s" % (len(net.sd), scale, nc, seed, path))


if __name__ == "__main__":
    main()
