"""profiles/<tag>_c5_* from gpurun_out/<tag>c5_bench.json + gpurun_out/<tag>c5_ktrace (bench.py --config c5, see
tools/collect_profiles.sh for the commands).  python tools/summarize_c5.py r02"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
bench = json.loads(open(os.path.join(G, tag + "c5_bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(P, tag + "_c5_bench_N1.json"), "w"), indent=1)
stats = max(glob.glob(os.path.join(G, tag + "c5_ktrace", "*", "*kernel_stats.csv")), key=os.path.getmtime)
shutil.copy(stats, os.path.join(P, tag + "_c5_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
short = lambda n: n.replace("void ", "").replace("cy::", "").split("(")[0][:70]
cb = bench.get("cpu_baseline") or {}
with open(os.path.join(P, tag + "_c5_summary.md"), "w") as fp:
    fp.write("# %s profile of `python bench.py --config c5` (1x MI355X): BASELINE config 5 at one-GPU size\n\n" % tag)
    fp.write("Workload: %s\n\n" % bench["config"]["workload"])
    fp.write("bench line: **%.1f tiles/s** (640x640 tiles), %.1f ms per %d-tile pass, conv-stack MFMA fraction of the whole job %.3f; "
             "CPU oracle beside it: %s tiles/s with %s torch threads.\n\n" % (
                 bench["value"], bench["ms_per_step"], bench["config"]["tiles"], bench["conv_stack_mfma_frac_whole_job"],
                 ("%.2f" % cb["value"]) if cb else "n/a", cb.get("cores", "n/a")))
    fp.write("## rocprofv3 --kernel-trace --stats (bench.py --config c5 --steps 1 --warmup 1: two passes over the grid)\n\n")
    fp.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:16]:
        t = float(r["TotalDurationNs"])
        fp.write("| `%s` | %s | %.2f | %.1f | %.2f |\n" % (short(r["Name"]), r["Calls"], t / 1e6, t / 1e3 / int(r["Calls"]), 100 * t / tot))
    fp.write("\n## forward kernels as bench.py timed them (hipEvents around every main-lane launch of the last timed step)\n\n")
    if 64 <= bench["config"]["tile_batch"] < 240:
        fp.write("Batches of 64..239 tiles run their forward as two concurrent half-batches, so these per-launch times overlap: the\n"
                 "TFLOP/s column understates each kernel's exclusive rate by up to 2x (per-layer exclusive rates: `tools/profile_layers.py 128 640`).\n\n")
    else:
        r = bench.get("roofline", {})
        fp.write("Dominant kernel `%s`: %.0f TFLOP/s inside the pipelined pass, %s with the GPU to itself.\n\n" % (
            r.get("kernel", "?"), r.get("achieved", 0.0), ("%.0f" % r["achieved_exclusive"]) if r.get("achieved_exclusive") else "n/a"))
    fp.write("| kernel | ms total | launches | avg us | TFLOP/s | share |\n|---|---|---|---|---|---|\n")
    for k in sorted(bench.get("forward_kernels", []), key=lambda k: -k["ms_total"]):
        fp.write("| %s | %.2f | %d | %.1f | %.0f | %.3f |\n" % (k["kernel"], k["ms_total"], k["launches"], 1e3 * k["ms_total"] / k["launches"],
                                                              k["TFLOP/s"], k["share_of_forward"]))
print("written", tag)
