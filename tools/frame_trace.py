"""Developer tool: GPU timeline of one steady-state frame of the sequence workload from a rocprofv3 trace.
   on the GPU box:  rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/ft -o ft --output-format csv -- python bench_sequence.py --frames 40 --quiet
                    python tools/frame_trace.py gpurun_out/ft"""
import csv, glob, sys
root = sys.argv[1]
kt = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
mc = glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True)
ev = []
for r in csv.DictReader(open(kt)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]))
if mc:
    for r in csv.DictReader(open(mc[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
# frames are separated by the hall_scan generation; find the last k_decode_tile run (mapping) and walk back to the previous one
dec = [i for i, e in enumerate(ev) if "k_decode_tile" in e[2]]
# group decode indices into mapping calls (consecutive within 200 us)
calls = []
for i in dec:
    if calls and ev[i][0] - ev[calls[-1][-1]][1] < 200000:
        calls[-1].append(i)
    else:
        calls.append([i])
a = calls[-3][-1] + 1   # after the mapping of frame n-2
b = calls[-2][0]        # first decode of frame n-1's mapping
seg = ev[a:b]
t0 = seg[0][0]
busy = sum(e[1] - e[0] for e in seg)
print("events between two mapping calls:", len(seg), "span_us", (seg[-1][1] - t0) / 1e3, "busy_us", busy / 1e3)
prev = None
for s, e, n in seg:
    gap = (s - prev) / 1e3 if prev else 0.0
    print("%9.1f  dur %8.2f  gap %8.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n))
    prev = e
