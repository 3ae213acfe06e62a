"""A/B of the wide 3x3 kernel's MFMA shape inside the real forward pass (developer tool, round 4).
Interleaved rounds in ONE process on one device (cdna_hip_programming.md rule 24): per setting the hipEvent time of every launch of the
CONV_WIDE_128 family and of each of its layers.  python tools/ab_wide_mfma.py [B] [px] [rounds]
Settings: 16x16x32 as shipped (one-patch + persistent forms), 16x16x32 one-patch form only, 32x32x16 (compiler-placed fragment reads),
32x32x16 with one fragment read per MFMA gap."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ge.build()
from caesar_yolo_amd.model import YOLO

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 4
SETTINGS = [("16x16x32 shipped", {}), ("16x16x32 one-patch", {"CY_WIDE_PERSIST": "0"}),
            ("32x32x16", {"CY_WIDE_MFMA": "32"}), ("32x32x16 sched", {"CY_WIDE_MFMA": "33"}),
            ("16x16x32 one-patch sched", {"CY_WIDE_PERSIST": "0", "CY_WIDE_SCHED": "1"})]
KEYS = ["CY_WIDE_PERSIST", "CY_WIDE_MFMA", "CY_WIDE_SCHED"]
m = YOLO("seeded:l:5", precision="fp16", max_batch=B, max_imgsz=H, device=0)
det = m.engine(0)
x = torch.rand((B, H, H, 4), device="cuda").half()
for _ in range(3):
    det.forward(x)
torch.cuda.synchronize()
WIDE = "conv3x3_wide_kernel 3x3 s1 16x32px x128ch"
res = {name: dict(ms=[], tflops=[], fwd_ms=[], layers={}) for name, _ in SETTINGS}
for rnd in range(ROUNDS):
    for name, env in SETTINGS:
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        det.forward(x)
        torch.cuda.synchronize()
        det.profile(True)
        R = 2
        for _ in range(R):
            det.forward(x)
        torch.cuda.synchronize()
        fam = [e for e in det.profile_summary() if e["kernel"].startswith(WIDE)][0]
        allms = sum(e["ms"] for e in det.profile_summary())
        r = res[name]
        r["ms"].append(fam["ms"] / R); r["tflops"].append(fam["flops"] / fam["ms"] / 1e9); r["fwd_ms"].append(allms / R)
        for e in det.profile_layers():
            if e["launches"]:
                r["layers"].setdefault(e["name"], []).append(e["ms"] / R)
        det.profile(False)
for k in KEYS:
    os.environ.pop(k, None)
med = lambda v: sorted(v)[len(v) // 2]
base = res[SETTINGS[0][0]]
out = {"batch": B, "px": H, "rounds": ROUNDS, "settings": {}}
print("%-22s %10s %10s %10s %10s" % ("setting", "wide ms", "min", "TFLOP/s", "fwd ms"))
for name, _ in SETTINGS:
    r = res[name]
    print("%-22s %10.3f %10.3f %10.1f %10.3f" % (name, med(r["ms"]), min(r["ms"]), med(r["tflops"]), med(r["fwd_ms"])))
    out["settings"][name] = dict(wide_ms_median=med(r["ms"]), wide_ms_min=min(r["ms"]), wide_tflops_median=med(r["tflops"]), forward_ms_median=med(r["fwd_ms"]))
# per layer: only the layers whose time differs between the first and the third setting by > 2 % are wide-kernel layers or their neighbours
print("\nper layer (median ms): layer, " + ", ".join(n for n, _ in SETTINGS))
lay = {}
for ln in base["layers"]:
    v = [med(res[n]["layers"][ln]) for n, _ in SETTINGS]
    if abs(v[2] - v[0]) > 0.02 * v[0] or abs(v[1] - v[0]) > 0.02 * v[0]:
        print("%-22s " % ln + " ".join("%8.4f" % t for t in v))
        lay[ln] = v
out["layers_ms"] = lay
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ab_wide_mfma_B%d_%d%s.json" % (B, H, os.environ.get("AB_TAG", ""))), "w"), indent=1)
