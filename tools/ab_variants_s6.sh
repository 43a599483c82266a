# developer tool (GPU box): A/B of compile-time variants through tools/variant_bench.py (only the named source is recompiled)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s6; o=gpurun_out/s6/variants.txt; : > $o
export VARIANT_SRCS=mapops.hip
for b in 8 16 32; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/s6/sort$b -o s --output-format csv -- python tools/variant_bench.py "-DCLID_SORT_BUCKETS=$b" --steps 20 --warmup 5 >> $o 2>> gpurun_out/s6/variants.err
  f=$(find gpurun_out/s6/sort$b -name 's_kernel_stats.csv' | head -1)
  python - "$f" $b >> $o <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_batch_sort" in r["Name"] or "k_mapping_prep" in r["Name"]:
        print("buckets", sys.argv[2], r["Name"][:24], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
  rm -rf gpurun_out/s6/sort$b
done
cat $o
