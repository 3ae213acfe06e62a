"""Timeline of one benchmark pass from a rocprofv3 --kernel-trace CSV (developer tool): GPU busy / idle time, time covered
by the conv stack vs only by side-stream kernels, the largest idle gaps, and per-batch forward spans.
python tools/trace_gaps.py <kernel_trace.csv> [skip_first_n_ms]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as fp:
    for r in csv.DictReader(fp):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Queue_Id"])))
rows.sort()
t0 = rows[0][0]
skip = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.0
rows = [r for r in rows if r[0] - t0 >= skip]
conv = lambda n: ("conv" in n or "stem" in n or "pool5" in n or "dwconv" in n or "attention" in n)


def union(iv):
    iv = sorted(iv)
    out, cs, ce = 0, None, None
    spans = []
    for s, e in iv:
        if cs is None or s > ce:
            if cs is not None:
                spans.append((cs, ce))
            cs, ce = s, e
        else:
            ce = max(ce, e)
    if cs is not None:
        spans.append((cs, ce))
    return spans


allsp = union([(r[0], r[1]) for r in rows])
convsp = union([(r[0], r[1]) for r in rows if conv(r[2])])
total = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e in allsp)
cbusy = sum(e - s for s, e in convsp)
print("window %.3f ms, %d dispatches; GPU busy %.3f ms (%.1f %%), conv-stack kernels cover %.3f ms (%.1f %%), idle %.3f ms" % (
    total / 1e6, len(rows), busy / 1e6, 100.0 * busy / total, cbusy / 1e6, 100.0 * cbusy / total, (total - busy) / 1e6))
gaps = sorted(((convsp[i + 1][0] - convsp[i][1], convsp[i][1]) for i in range(len(convsp) - 1)), reverse=True)
print("conv-stack gaps: %d, sum %.3f ms; > 20 us: %d (sum %.3f ms); largest:" % (
    len(gaps), sum(g for g, _ in gaps) / 1e6, sum(1 for g, _ in gaps if g > 20000), sum(g for g, _ in gaps if g > 20000) / 1e6))
for g, at in gaps[:12]:
    during = sorted({r[2][:40] for r in rows if not conv(r[2]) and r[0] < at + g and r[1] > at})
    print("  %.1f us at +%.3f ms  (meanwhile: %s)" % (g / 1e3, (at - rows[0][0]) / 1e6, ", ".join(during) or "nothing"))
by = {}
for s, e, n, q in rows:
    k = n.split("(")[0][:60]
    by.setdefault(k, [0, 0])
    by[k][0] += e - s
    by[k][1] += 1
print("per kernel:")
for k, (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:18]:
    print("  %-60s %9.3f ms %5d" % (k, t / 1e6, c))
