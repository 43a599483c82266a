# developer tool (GPU box): pool chain check (tests, kernel times inside the sequence workload)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s6
timeout 600 python -m pytest tests/test_mapops_gpu.py tests/test_sampler_gpu.py -m gpu -x -q 2>&1 | tail -1
rocprofv3 --kernel-trace --stats -d gpurun_out/s6/sq -o s --output-format csv -- python bench_sequence.py --frames 100 --quiet > gpurun_out/s6/seq100.json 2>/dev/null
python - <<'PY'
import csv, glob, json
for f in glob.glob("gpurun_out/s6/sq/**/s_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("k_pool_",)):
            print(r["Name"][:40], r["Calls"], r["AverageNs"], r["MaxNs"])
PY
rm -rf gpurun_out/s6/sq
for i in 1 2; do timeout 600 python bench_sequence.py --frames 100 --quiet 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['steady_state'])"; done
