timeout 600 python -m pytest tests/test_mapops_gpu.py tests/test_sampler_gpu.py tests/test_sequence.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/ft
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/ft -o ft --output-format csv -- python bench_sequence.py --frames 80 --quiet > /dev/null 2>&1
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/ft/ft_kernel_stats.csv"))]
for r in rows:
    n=r["Name"]
    if "k_vox" in n: print(n[:40], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
rm -rf gpurun_out/ft
for r in 1 2 3; do for d in . _old; do (cd $d; python bench_sequence.py --frames 120 --quiet 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=j['steady_state']; print('$d', round(s['scans_per_s'],1), round(s['median_process_frame_ms'],4), round(s['median_mapping_ms'],4))
"); done; done
