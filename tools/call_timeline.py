"""developer tool: print the kernel time line (start offset, duration, gap to the previous kernel) of the LAST mapping() call of a
rocprofv3 --kernel-trace of `bench.py --steps K --frame-calls 0 --no-cpu-baseline`.   usage: python tools/call_timeline.py trace.csv K"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
K = int(sys.argv[2])
# the timed call = the last k_mapping_prep followed by >= K decode launches
preps = [i for i, r in enumerate(rows) if "k_mapping_prep" in r["Kernel_Name"]]
best = None
for i in preps:
    n = sum(1 for r in rows[i:] if "k_decode_tile" in r["Kernel_Name"])
    if n >= K:
        best = i
start = best
t0 = int(rows[start]["Start_Timestamp"])
prev_end = t0
ndec = 0
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "")[:46]
    if "k_decode_tile" in name: ndec += 1
    if ndec <= 3 or ndec >= K - 1:
        print(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:7.2f}  gap {(s - prev_end) / 1e3:6.2f}  {name}")
    prev_end = e
    if ndec >= K and "k_local_to_global" in name: break
print("call span us", (prev_end - t0) / 1e3)
