"""SQ counters of the wide 3x3 kernel on both MFMA shapes (developer tool, round 4).
On the GPU box, under the profiler (counters in their own run, kernel trace only):
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
      SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r04_w32_pmc -- python3 tools/pmc_wide_mfma.py run
runs one 256-tile forward pass per setting (the same layers on conv3x3_wide_kernel<true,2,false,2,false> / conv3x3_wide32_kernel<0> / <1>).
Here:  python tools/pmc_wide_mfma.py summarize  ->  profiles/r04_wide_mfma_ab.md (+ the hipEvent A/B of tools/ab_wide_mfma.py)."""
import csv, glob, json, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
COUNTERS = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES",
            "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"]
KERNELS = [("16x16x32 one-patch", "conv3x3_wide_kernel<true,2,false,2,false>"), ("32x32x16", "conv3x3_wide32_kernel<0>"),
           ("32x32x16, one read per MFMA gap", "conv3x3_wide32_kernel<1>")]


def run():
    import torch
    from caesar_yolo_amd.model import YOLO
    B, H = 256, 512
    m = YOLO("seeded:l:5", precision="fp16", max_batch=B, max_imgsz=H, device=0)
    det = m.engine(0)
    x = torch.rand((B, H, H, 4), device="cuda").half()
    for env in ({"CY_WIDE_PERSIST": "0"}, {"CY_WIDE_MFMA": "32"}, {"CY_WIDE_MFMA": "33"}):
        for k in ("CY_WIDE_PERSIST", "CY_WIDE_MFMA"):
            os.environ.pop(k, None)
        os.environ.update(env)
        for _ in range(2):
            det.forward(x)
        torch.cuda.synchronize()


def summarize():
    G = os.path.join(ROOT, "gpurun_out")
    f = glob.glob(os.path.join(G, "r04_w32_pmc", "*", "*counter_collection.csv"))
    rows = list(csv.DictReader(open(max(f, key=os.path.getmtime))))
    # one row per (dispatch, counter); dispatches of >= 1000 workgroups only (the 256-tile launches)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    dur = collections.defaultdict(float)
    seen = set()
    for r in rows:
        kn = r["Kernel_Name"].replace(" ", "").replace("cy::", "").replace("void", "").split("(")[0]
        agg[kn][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (kn, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); n[kn] += 1
            if "Start_Timestamp" in r and "End_Timestamp" in r:
                dur[kn] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    ab = json.load(open(os.path.join(G, "ab_wide_mfma_B256_512.json")))
    out = os.path.join(ROOT, "profiles", "r04_wide_mfma_ab.md")
    with open(out, "w") as fp:
        fp.write("# Round 4: the wide 3x3 kernel's K loop on `v_mfma_f32_32x32x16_f16` vs `v_mfma_f32_16x16x32_f16` (same box, one process)\n\n")
        fp.write("Same workgroup (16 x 32 px x 128 ch), wave tile (128 px x 64 ch, 128 accumulator VGPRs), stage stream, packed weights and LDS bytes per MFMA-flop; "
                 "`conv3x3_wide32_kernel` (csrc/conv_igemm.hip) issues half as many MFMA instructions.  `tools/ab_wide_mfma.py 256 512 4`: "
                 "hipEvent time of every launch of the CONV_WIDE_128 family inside the real 256-tile forward pass, settings interleaved over %d rounds.\n\n" % ab["rounds"])
        fp.write("| setting | wide family ms per forward (median) | min | TFLOP/s | whole forward ms |\n|---|---|---|---|---|\n")
        for k, v in ab["settings"].items():
            fp.write("| %s | %.3f | %.3f | %.1f | %.3f |\n" % (k, v["wide_ms_median"], v["wide_ms_min"], v["wide_tflops_median"], v["forward_ms_median"]))
        fp.write("\nPer layer (median ms; columns as above):\n\n| layer | " + " | ".join(ab["settings"].keys()) + " |\n|---|" + "---|" * len(ab["settings"]) + "\n")
        for ln, v in ab["layers_ms"].items():
            if ".m." in ln or "cv3" in ln:
                fp.write("| %s | " % ln + " | ".join("%.4f" % t for t in v) + " |\n")
        fp.write("\n## SQ counters, `rocprofv3 --pmc` (own run, kernel trace only), sums over the same layers of two 256-tile forward passes per setting\n\n")
        fp.write("SQ_WAVE_CYCLES / WAIT / ACTIVE count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (MI355X_MICROARCH.md, cycle constants); "
                 "clock = GRBM_GUI_ACTIVE / 8 / kernel time (approximate below ~0.3 ms per dispatch, same bias for all three).\n\n")
        fp.write("| kernel | dispatches | time under the profiler ms | SQ_WAVE_CYCLES | MFMA busy cycles | matrix pipe busy share = MFMA busy / (4 x WAVE_CYCLES) x 2 waves per SIMD | WAIT_ANY / WAVE | WAIT_INST_ANY / WAVE | ACTIVE_INST / WAVE | LDS_IDX_ACTIVE | LDS conflict / active | clock GHz |\n")
        fp.write("|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for label, kn in KERNELS:
            v = agg.get(kn)
            if not v:
                continue
            wv = max(v["SQ_WAVE_CYCLES"], 1.0)
            clk = v["GRBM_GUI_ACTIVE"] / 8.0 / (dur[kn] * 1e3) if dur[kn] else 0.0        # cycles / ns = GHz
            fp.write("| `%s` (%s) | %d | %.2f | %.4g | %.4g | %.3f | %.3f | %.3f | %.3f | %.4g | %.3f | %.2f |\n" % (
                kn, label, n[kn], dur[kn] / 1e3, v["SQ_WAVE_CYCLES"], v["SQ_VALU_MFMA_BUSY_CYCLES"],
                2.0 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * wv), v["SQ_WAIT_ANY"] / wv, v["SQ_WAIT_INST_ANY"] / wv,
                v["SQ_ACTIVE_INST_ANY"] / wv, v["SQ_LDS_IDX_ACTIVE"], v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1.0), clk))
        fp.write("\nReading: the MFMA work is identical (same busy cycles) and the LDS images of both shapes are conflict-free.  With the fragment reads spread "
                 "between the MFMAs the 32x32x16 loop needs the SAME wave cycles as the 16x16x32 loop (compiler-placed reads: +3.6 %), yet takes 7-8 % longer in wall time: "
                 "the chip holds a lower clock on the 32x32x16 shape (MI355X_MICROARCH.md, DVFS give-back item 7: bare bf16 loops 1.12-1.15x in favour of 16x16x32 at equal cycles per flop).  "
                 "The freed vector-issue slots (ACTIVE_INST share 0.21 -> 0.17) do not turn into matrix time: WAIT_INST_ANY stays at 0.46-0.50.  Kept in the tree as an opt-in "
                 "(`CY_WIDE_MFMA=32|33`, `tests/test_gpu_conv.py::test_wide_kernel_on_32x32x16_mfma`); the shipped kernels stay on 16x16x32.\n")
    print(open(out).read())


if __name__ == "__main__":
    (run if (len(sys.argv) > 1 and sys.argv[1] == "run") else summarize)()
