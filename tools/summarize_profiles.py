"""Turn the raw rocprofv3 CSVs collected by tools/collect_profiles.sh into the tracked summaries under profiles/.
python tools/summarize_profiles.py <tag> [final|c5|y11]   (gpurun_out/<tag>_<name>_* -> profiles/<tag>_<name>_*)"""
import csv, glob, json, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
name = sys.argv[2] if len(sys.argv) > 2 else "final"
CMD = {"final": "python bench.py", "c5": "python bench.py --config c5", "y11": "python bench.py --weights seeded11:l:5"}[name]
raw, trk = tag + "_" + name, tag + "_" + name          # prefixes under gpurun_out/ and profiles/
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)


def short(n):
    return n.replace("void ", "").replace("cy::", "").split("(")[0][:70]


MAIN_WGS = 1000      # a launch of >= 1000 workgroups belongs to a full batch (main lane); persistent kernels (one workgroup per CU) cannot be told apart


def lane_of(r):
    wg = max(int(r.get("Workgroup_Size", 0) or 0), 1)
    return "main" if int(r.get("Grid_Size", 0) or 0) // wg >= MAIN_WGS else "small"


def pmc(dirname, counters, by_lane=False):
    f = glob.glob(os.path.join(G, dirname, "*", "*counter_collection.csv"))
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    if not f:
        return out, cnt
    for r in csv.DictReader(open(max(f, key=os.path.getmtime))):      # newest run if older ones are still around
        if r["Counter_Name"] in counters:
            for k in ([short(r["Kernel_Name"])] + ([short(r["Kernel_Name"]) + "|" + lane_of(r)] if by_lane else [])):
                out[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] == counters[0]:
                    cnt[k] += 1
    return out, cnt


stats = glob.glob(os.path.join(G, raw + "_ktrace", "*", "*kernel_stats.csv"))
stats = [max(stats, key=os.path.getmtime)] if stats else []
rows = list(csv.DictReader(open(stats[0]))) if stats else []
with open(os.path.join(P, trk + "_kernel_stats.csv"), "w") as fp:
    if stats:
        fp.write(open(stats[0]).read())
bench = json.loads(open(os.path.join(G, raw + "_bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(P, trk + "_bench_N1.json"), "w"), indent=1)
fetch, nf = pmc(raw + "_pmc_fetch", ["FETCH_SIZE"], by_lane=True)
write, nw = pmc(raw + "_pmc_write", ["WRITE_SIZE"], by_lane=True)


def traffic_entry(k):
    f_kb, w_kb = fetch[k]["FETCH_SIZE"] / nf[k], write[k]["WRITE_SIZE"] / nw[k]
    # MI355X_MICROARCH.md, HBM section: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-byte
    # requests as 64 bytes for wide coalesced reads -> doubled; WRITE_SIZE is exact for 16-byte-per-lane stores
    return {"launches": nf[k], "FETCH_SIZE_KiB_per_launch": f_kb, "WRITE_SIZE_KiB_per_launch": w_kb,
            "hbm_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0}


traffic = {}
for k in fetch:
    if "|" in k or not (nf[k] and k in write and nw[k]):
        continue
    traffic[k] = traffic_entry(k)
    # the same split by launch size (grid size of the dispatch): full batches ("main": >= 1000 workgroups) and everything else
    # ("small": the small-batch lane, and EVERY launch of a persistent kernel) -- bench.py pairs the "main" bytes with the hipEvent
    # time of the full-batch launches, so that bytes and time refer to the same launches
    for lane in ("main", "small"):
        kl = k + "|" + lane
        if nf.get(kl) and nw.get(kl):
            traffic[k][lane] = traffic_entry(kl)
json.dump(traffic, open(os.path.join(P, trk + "_hbm_traffic.json"), "w"), indent=1)
# side-stream kernels (preprocessing, decode / NMS / merge, record compaction): HBM bytes per launch (PMC) over the launch duration of
# the kernel trace -> GB/s against the 8 TB/s peak ("HBM GB/s for preprocessing / decode")
side = {}
for r_ in rows:
    k = short(r_["Name"])
    if any(t in k for t in ("pre_", "decode_kernel", "nms_kernel", "iou_merge_kernel", "compact_", "mosaic_prepare", "pool5")) and k in traffic:
        avg_us = float(r_["AverageNs"]) / 1e3
        side[k] = {"calls": int(r_["Calls"]), "avg_us": avg_us, "hbm_bytes_per_launch": traffic[k]["hbm_bytes_per_launch"],
                   "GB_per_s": traffic[k]["hbm_bytes_per_launch"] / (avg_us * 1e-6) / 1e9,
                   "frac_of_8TBps": traffic[k]["hbm_bytes_per_launch"] / (avg_us * 1e-6) / 8e12,
                   "share_of_gpu_time_pct": float(r_["Percentage"])}
json.dump(side, open(os.path.join(P, trk + "_side_kernels.json"), "w"), indent=1)
sq, nsq = pmc(raw + "_pmc_sq", ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                                "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"])
with open(os.path.join(P, trk + "_summary.md"), "w") as fp:
    fp.write("# %s %s profile (1x MI355X, `%s`)\n\n" % (tag, name, CMD))
    fp.write("Workload: %s\n\n" % bench["config"]["workload"])
    fp.write("bench line: %.1f tiles/s, %.1f ms/step, conv-stack MFMA fraction of the whole job %.3f; dominant kernel `%s` %.1f TFLOP/s (%.3f of 2.5 PFLOP/s), avg launch %.4f ms\n\n"
             % (bench["value"], bench["ms_per_step"], bench["conv_stack_mfma_frac_whole_job"], bench["roofline"]["kernel"], bench["roofline"]["achieved"],
                bench["roofline"]["frac"], bench["roofline"]["avg_launch_ms"]))
    if bench.get("parity_value"):
        fp.write("parity context (%s) on the same workload: **%.1f tiles/s** = %.2f of the line above (%s two-pass / %s three-pass layers)\n\n" % (
            bench.get("parity_dtype"), bench["parity_value"], bench.get("parity_vs_value") or 0.0,
            (bench.get("parity_mode") or {}).get("layers_two_pass", "?"), (bench.get("parity_mode") or {}).get("layers_three_pass", "?")))
    cb = bench.get("cpu_baseline") or {}
    if cb:
        fp.write("CPU oracle beside it: %.2f tiles/s with %s torch threads (%s)\n\n" % (cb["value"], cb.get("cores"), cb.get("sample")))
    r = bench["roofline"]
    if r.get("achieved_all_launches"):
        fp.write("`roofline.achieved` = the launches of the full batches (main lane: %d launches in the profiled step); with the launches of the "
                 "small-batch lane, which run on another stream beside them: %.1f TFLOP/s over %s launches; with the GPU to itself: %s TFLOP/s.\n\n"
                 % (r["launches"], r["achieved_all_launches"], r.get("launches_all"), ("%.1f" % r["achieved_exclusive"]) if r.get("achieved_exclusive") else "n/a"))
    # the dominant kernel's launches in the rocprofv3 trace, split the same way (full batches launch >= 1000 workgroups; the
    # persistent form of the same kernel -- one workgroup per CU -- runs the 18-stage layers of the full batches)
    tr = glob.glob(os.path.join(G, raw + "_ktrace", "*", "*kernel_trace.csv"))
    if tr:
        main, small, pers = [0, 0.0], [0, 0.0], [0, 0.0]
        for row in csv.DictReader(open(max(tr, key=os.path.getmtime))):
            n = row["Kernel_Name"].replace(" ", "")
            d = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
            if "conv3x3_widep_kernel<false,false,0>" in n or "conv3x3_widep_kernel<false,true,0>" in n:
                pers[0] += 1; pers[1] += d
                continue
            if "conv3x3_wide_kernel<true,2,false,2,false" not in n:
                continue
            wgs = int(row.get("Grid_Size_X", row.get("Grid_Size", 0))) // 512
            tgt = main if wgs >= 1000 else small
            tgt[0] += 1; tgt[1] += d
        if main[0]:
            fp.write("Same split in the rocprofv3 kernel trace below: `conv3x3_wide_kernel<true, 2, false, 2, false, 0>` full batches %d launches, avg %.1f us; "
                     "its persistent form `conv3x3_widep_kernel<false, *, 0>` (18-stage layers of the full batches) %d launches, avg %.1f us; both together "
                     "%d launches, avg %.1f us; small-batch lane %d launches, avg %.1f us (bench.py's hipEvent average over the full batches, both forms: %.1f us).\n\n"
                     % (main[0], main[1] / main[0], pers[0], (pers[1] / pers[0]) if pers[0] else 0.0, main[0] + pers[0],
                        (main[1] + pers[1]) / (main[0] + pers[0]), small[0], (small[1] / small[0]) if small[0] else 0.0, 1e3 * r["avg_launch_ms"]))
    fp.write("## rocprofv3 --kernel-trace --stats (%s --steps 1 --warmup 1: two passes over the tile grid)\n\n" % CMD)
    fp.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:16]:
        fp.write("| `%s` | %s | %.2f | %.1f | %s |\n" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                     float(r["AverageNs"]) / 1e3, r["Percentage"]))
    fp.write("\n## HBM traffic per launch (separate --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied)\n\n")
    fp.write("| kernel | launches | 2*FETCH_SIZE MiB | WRITE_SIZE MiB | total MiB/launch |\n|---|---|---|---|---|\n")
    for k, v in sorted(traffic.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
        fp.write("| `%s` | %d | %.1f | %.1f | %.1f |\n" % (k, v["launches"], 2 * v["FETCH_SIZE_KiB_per_launch"] / 1024,
                                                        v["WRITE_SIZE_KiB_per_launch"] / 1024, v["hbm_bytes_per_launch"] / 2 ** 20))
    fp.write("\n## Side-stream kernels against the HBM roof (PMC bytes per launch / kernel-trace duration; 8 TB/s peak)\n\n")
    fp.write("| kernel | calls | avg us | MiB per launch | GB/s | of 8 TB/s | % of GPU time |\n|---|---|---|---|---|---|---|\n")
    for k, v in sorted(side.items(), key=lambda kv: -kv[1]["share_of_gpu_time_pct"]):
        fp.write("| `%s` | %d | %.1f | %.2f | %.0f | %.3f | %.2f |\n" % (k, v["calls"], v["avg_us"], v["hbm_bytes_per_launch"] / 2 ** 20, v["GB_per_s"], v["frac_of_8TBps"], v["share_of_gpu_time_pct"]))
    fp.write("\n(durations are those INSIDE the pipelined pass: these kernels run on low-priority side streams beside the conv stack and are latency-bound by design -- "
             "a few dozen workgroups; alone on the GPU `decode_kernel` takes 59 us per 256 tiles of 512^2 = 3.2 TB/s over the cache lines it touches "
             "(`tools/time_post.py` under rocprofv3), seven times faster than its launches in the pass)\n")
    fp.write("\n## SQ counters (bench.py --size 8192, one pass; sums over the dispatches of each kernel)\n\n")
    fp.write("| kernel | n | MFMA busy / (4 x BUSY_CU-ish) | WAIT_ANY/WAVE | WAIT_INST/WAVE | ACTIVE/WAVE | LDS conflict / LDS active |\n|---|---|---|---|---|---|---|\n")
    for k, v in sorted(sq.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:10]:
        wv = max(v.get("SQ_WAVE_CYCLES", 0), 1)
        fp.write("| `%s` | %d | %.3g | %.2f | %.2f | %.2f | %.2f |\n" % (
            k, nsq[k], v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), v.get("SQ_WAIT_ANY", 0) / wv, v.get("SQ_WAIT_INST_ANY", 0) / wv,
            v.get("SQ_ACTIVE_INST_ANY", 0) / wv, v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 0), 1)))
    fp.write("\n## forward kernels as bench.py timed them (hipEvents around every main-lane launch of the last timed step)\n\n")
    fp.write("| kernel | ms total | launches | avg us | TFLOP/s | share |\n|---|---|---|---|---|---|\n")
    for k in sorted(bench.get("forward_kernels", []), key=lambda k: -k["ms_total"]):
        fp.write("| %s | %.2f | %d | %.1f | %.0f | %.3f |\n" % (k["kernel"], k["ms_total"], k["launches"], 1e3 * k["ms_total"] / k["launches"], k["TFLOP/s"], k["share_of_forward"]))
print(open(os.path.join(P, trk + "_summary.md")).read())
