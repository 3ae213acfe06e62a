"""Summarise a rocprofv3 kernel_trace.csv: median/min duration per (kernel substring, grid).  python tools/ktrace_summary.py dir [substr]"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if sub in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:48], r["Grid_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    v = sorted(v)
    print(k, len(v), "median us %.1f min %.1f" % (v[len(v) // 2], v[0]))
