#!/bin/bash
# developer tool (GPU box): HBM traffic counters of the "next" rows' kernels -> gpurun_out/r04/next_rows_traffic.json
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r04/nextpmc; mkdir -p $out
A="python bench_next.py --no-cpu-baseline"
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out -o fetch --output-format csv -- $A > /dev/null 2>> $out/log.txt
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out -o write --output-format csv -- $A > /dev/null 2>> $out/log.txt
python - <<'PY'
import csv, collections, glob, json, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in sorted(glob.glob("gpurun_out/r04/nextpmc/**/*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(fn)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if any(n in k for n in ("k_track_model", "k_sample_frame", "k_sdf_query", "k_region_sdf", "k_compact")):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `python bench_next.py --no-cpu-baseline`, "
              "per-launch means; KiB counters; traffic_bytes = 2 x FETCH_SIZE + WRITE_SIZE (the gfx950 FETCH_SIZE correction of "
              "profiles/r01_pmc_calibration.txt, as profiles/r04_hbm_traffic.json)"}
for k, v in agg.items():
    f = sum(v.get("FETCH_SIZE", [0])) / max(len(v.get("FETCH_SIZE", [1])), 1)
    w = sum(v.get("WRITE_SIZE", [0])) / max(len(v.get("WRITE_SIZE", [1])), 1)
    out[k] = {"fetch_kb": f, "write_kb": w, "traffic_bytes": int((2 * f + w) * 1024), "launches": len(v.get("FETCH_SIZE", []))}
json.dump(out, open("gpurun_out/r04/next_rows_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
PY
