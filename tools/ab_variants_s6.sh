# developer tool (GPU box): ordering-launch check (tests, kernel time in the bench and in the sequence workload)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s6
timeout 600 python -m pytest tests -m gpu -x -q -k "prep or sort or order or batch or mapping_prep or draw or dist or sequence" 2>&1 | tail -2
bash tools/kstats_quick.sh 2>&1 | grep sort
rocprofv3 --kernel-trace --stats -d gpurun_out/s6/sq -o s --output-format csv -- python bench_sequence.py --frames 60 --quiet > /dev/null 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/s6/sq/**/s_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_batch_sort" in r["Name"]:
            print("sequence:", r["Name"][:30], r["Calls"], r["AverageNs"])
PY
rm -rf gpurun_out/s6/sq
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), d['per_frame_regime']['ms_per_step'])"; done
