"""Turn the raw rocprofv3 CSVs collected by tools/collect_profiles.sh into the tracked summaries under profiles/.
python tools/summarize_profiles.py r01"""
import csv, glob, json, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)


def short(n):
    return n.replace("void ", "").replace("cy::", "").split("(")[0][:70]


def pmc(dirname, counters):
    f = glob.glob(os.path.join(G, dirname, "*", "*counter_collection.csv"))
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    if not f:
        return out, cnt
    for r in csv.DictReader(open(max(f, key=os.path.getmtime))):      # newest run if older ones are still around
        if r["Counter_Name"] in counters:
            k = short(r["Kernel_Name"])
            out[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == counters[0]:
                cnt[k] += 1
    return out, cnt


stats = glob.glob(os.path.join(G, tag + "_ktrace", "*", "*kernel_stats.csv"))
stats = [max(stats, key=os.path.getmtime)] if stats else []
rows = list(csv.DictReader(open(stats[0]))) if stats else []
with open(os.path.join(P, tag + "_final_kernel_stats.csv"), "w") as fp:
    if stats:
        fp.write(open(stats[0]).read())
bench = json.load(open(os.path.join(G, tag + "_bench.json")))
json.dump(bench, open(os.path.join(P, tag + "_final_bench_N1.json"), "w"), indent=1)
fetch, nf = pmc(tag + "_pmc_fetch", ["FETCH_SIZE"])
write, nw = pmc(tag + "_pmc_write", ["WRITE_SIZE"])
traffic = {}
for k in fetch:
    if nf[k] and k in write and nw[k]:
        f_kb, w_kb = fetch[k]["FETCH_SIZE"] / nf[k], write[k]["WRITE_SIZE"] / nw[k]
        # MI355X_MICROARCH.md, HBM section: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-byte
        # requests as 64 bytes for wide coalesced reads -> doubled; WRITE_SIZE is exact for 16-byte-per-lane stores
        traffic[k] = {"launches": nf[k], "FETCH_SIZE_KiB_per_launch": f_kb, "WRITE_SIZE_KiB_per_launch": w_kb,
                      "hbm_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0}
json.dump(traffic, open(os.path.join(P, tag + "_final_hbm_traffic.json"), "w"), indent=1)
sq, nsq = pmc(tag + "_pmc_sq", ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                                "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"])
with open(os.path.join(P, tag + "_final_summary.md"), "w") as fp:
    fp.write("# %s final profile (1x MI355X, default `python bench.py`)\n\n" % tag)
    fp.write("bench line: %.1f tiles/s, %.1f ms/step; dominant kernel `%s` %.1f TFLOP/s (%.3f of 2.5 PFLOP/s), avg launch %.4f ms\n\n"
             % (bench["value"], bench["ms_per_step"], bench["roofline"]["kernel"], bench["roofline"]["achieved"],
                bench["roofline"]["frac"], bench["roofline"]["avg_launch_ms"]))
    r = bench["roofline"]
    if r.get("achieved_all_launches"):
        fp.write("`roofline.achieved` = the launches of the full batches (main lane: %d launches in the profiled step); with the launches of the "
                 "small-batch lane, which run on another stream beside them: %.1f TFLOP/s over %s launches; with the GPU to itself: %s TFLOP/s.\n\n"
                 % (r["launches"], r["achieved_all_launches"], r.get("launches_all"), ("%.1f" % r["achieved_exclusive"]) if r.get("achieved_exclusive") else "n/a"))
    # the dominant kernel's launches in the rocprofv3 trace, split the same way (full batches launch >= 1000 workgroups; the
    # persistent form of the same kernel -- one workgroup per CU -- runs the 18-stage layers of the full batches)
    tr = glob.glob(os.path.join(G, tag + "_ktrace", "*", "*kernel_trace.csv"))
    if tr:
        main, small, pers = [0, 0.0], [0, 0.0], [0, 0.0]
        for row in csv.DictReader(open(max(tr, key=os.path.getmtime))):
            n = row["Kernel_Name"].replace(" ", "")
            d = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
            if "conv3x3_widep_kernel<false,false,0>" in n or "conv3x3_widep_kernel<false,true,0>" in n:
                pers[0] += 1; pers[1] += d
                continue
            if "conv3x3_wide_kernel<true,2,false,2,false>" not in n:
                continue
            wgs = int(row.get("Grid_Size_X", row.get("Grid_Size", 0))) // 512
            tgt = main if wgs >= 1000 else small
            tgt[0] += 1; tgt[1] += d
        if main[0]:
            fp.write("Same split in the rocprofv3 kernel trace below: `conv3x3_wide_kernel<true, 2, false, 2, false>` full batches %d launches, avg %.1f us; "
                     "its persistent form `conv3x3_widep_kernel<false, *, 0>` (18-stage layers of the full batches) %d launches, avg %.1f us; both together "
                     "%d launches, avg %.1f us; small-batch lane %d launches, avg %.1f us (bench.py's hipEvent average over the full batches, both forms: %.1f us).\n\n"
                     % (main[0], main[1] / main[0], pers[0], (pers[1] / pers[0]) if pers[0] else 0.0, main[0] + pers[0],
                        (main[1] + pers[1]) / (main[0] + pers[0]), small[0], (small[1] / small[0]) if small[0] else 0.0, 1e3 * r["avg_launch_ms"]))
    fp.write("## rocprofv3 --kernel-trace --stats (bench.py --steps 1 --warmup 1: two passes over the 1600-tile grid)\n\n")
    fp.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:16]:
        fp.write("| `%s` | %s | %.2f | %.1f | %s |\n" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                     float(r["AverageNs"]) / 1e3, r["Percentage"]))
    fp.write("\n## HBM traffic per launch (separate --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied)\n\n")
    fp.write("| kernel | launches | 2*FETCH_SIZE MiB | WRITE_SIZE MiB | total MiB/launch |\n|---|---|---|---|---|\n")
    for k, v in sorted(traffic.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
        fp.write("| `%s` | %d | %.1f | %.1f | %.1f |\n" % (k, v["launches"], 2 * v["FETCH_SIZE_KiB_per_launch"] / 1024,
                                                        v["WRITE_SIZE_KiB_per_launch"] / 1024, v["hbm_bytes_per_launch"] / 2 ** 20))
    fp.write("\n## SQ counters (bench.py --size 8192, one pass; sums over the dispatches of each kernel)\n\n")
    fp.write("| kernel | n | MFMA busy / (4 x BUSY_CU-ish) | WAIT_ANY/WAVE | WAIT_INST/WAVE | ACTIVE/WAVE | LDS conflict / LDS active |\n|---|---|---|---|---|---|---|\n")
    for k, v in sorted(sq.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:10]:
        wv = max(v.get("SQ_WAVE_CYCLES", 0), 1)
        fp.write("| `%s` | %d | %.3g | %.2f | %.2f | %.2f | %.2f |\n" % (
            k, nsq[k], v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), v.get("SQ_WAIT_ANY", 0) / wv, v.get("SQ_WAIT_INST_ANY", 0) / wv,
            v.get("SQ_ACTIVE_INST_ANY", 0) / wv, v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 0), 1)))
print(open(os.path.join(P, tag + "_final_summary.md")).read())
