"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel average duration and the average idle gap between
consecutive kernels (end of one -> start of the next), over the steady-state tail of the run.
usage: python tools/trace_gaps.py <kernel_trace.csv> [n_tail]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 600
rows = rows[-tail:]
dur = collections.defaultdict(list)
gap = collections.defaultdict(list)
def short(n):
    n = n.split("(")[0]
    return n[-40:]
for a, b in zip(rows, rows[1:]):
    dur[short(a["Kernel_Name"])].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gap[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("span_us", span / 1e3, "kernels", len(rows))
for k, v in dur.items():
    print("dur  %-45s n=%4d avg=%8.2f us" % (k, len(v), sum(v) / len(v) / 1e3))
for k, v in gap.items():
    print("gap  %-40s -> %-40s n=%4d avg=%8.2f us" % (k[0], k[1], len(v), sum(v) / len(v) / 1e3))
