"""Which kept anchors / merged detections differ between a context and the oracle on a reduced BASELINE config, and how close
to a threshold (conf, NMS IoU, merge IoU) the oracle's own values are there."""
import os, sys, json
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import numpy as np, torch
import __graft_entry__ as ge
ge.build()
import config_common as CC
from gpu_common import seeded_weights, oracle_model
from caesar_yolo_amd.model import YOLO
name, prec = sys.argv[1], sys.argv[2]
ref = CC.oracle_run(name)
img, ts, step, imgsz, spec = CC.config_input(name)
model = YOLO(seeded_weights()[0], precision=prec, max_batch=32, max_imgsz=imgsz, device=0)
eng = model.engine()
mosaic = eng.mosaic_to_device(img)
cfg = CC.device_pipeline(spec).program()
classes = {}
for tid, t in enumerate(ref["grid"]):
    classes.setdefault((t[3] - t[2], t[1] - t[0]), []).append(tid)
for (th, tw), tids in classes.items():
    xy = [(ref["grid"][t][0], ref["grid"][t][2]) for t in tids]
    netin, status, lb = eng.preproc(mosaic, xy, th, tw, imgsz, cfg)
    pred = eng.forward(netin)
    d, anch, cnt = eng.decode_nms(pred, lb.H, lb.W, th, tw, CC.CONF, CC.IOU)
    m, mcnt, _ = eng.iou_merge(d, cnt, CC.CONF, CC.SOFT, CC.HARD)
    torch.cuda.synchronize()
    d, anch, cnt, m, mcnt, predh = (x.cpu().numpy() for x in (d, anch, cnt, m, mcnt, pred))
    for b, t in enumerate(tids):
        if ref["raw"][t] is None:
            continue
        rb, rs, rc, ra = ref["raw"][t]
        n = int(cnt[b])
        ga = anch[b, :n].tolist()
        k = int(mcnt[b])
        if ga != ra.tolist() or k != len(ref["dets"][t][1]):
            sg, sr = set(ga), set(ra.tolist())
            print("tile %d: kept %d vs oracle %d; only here %s only oracle %s; merged %d vs %d" % (t, n, len(ra), sorted(sg - sr), sorted(sr - sg), k, len(ref["dets"][t][1])))
            for a in sorted((sg ^ sr)):
                cls = predh[b, a, 64:]
                sc = 1 / (1 + np.exp(-cls.astype(np.float64)))
                print("   anchor %d: class scores here %s" % (a, np.array2string(sc, precision=7)))
            if ga == ra.tolist():
                # same NMS output, different merge: look at pair IoUs near the merge thresholds
                bb = rb.astype(np.float64)
                M = CC.iou_matrix(bb, bb)
                iu = M[np.triu_indices(len(bb), 1)]
                for thr in (CC.SOFT, CC.HARD):
                    j = np.argmin(np.abs(iu - thr))
                    print("   nearest pair IoU to %.1f: %.8f" % (thr, iu[j]))
                print("   scores nearest to conf: ", np.sort(np.abs(rs - CC.CONF))[:3])
print("done")
