cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s6
rocprofv3 --kernel-trace --stats -d gpurun_out/s6/sortp -o s --output-format csv -- python bench.py --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
f=$(find gpurun_out/s6/sortp -name 's_kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("k_batch_sort", "k_mapping_prep", "k_decode_tile", "k_adam_all", "k_search_tiles")):
        print(r["Name"][:40], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
rm -rf gpurun_out/s6/sortp
