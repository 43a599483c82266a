# developer tool (GPU box): pre-numbered tiles on large maps (A/B on the sequence workload's map)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s6; o=gpurun_out/s6/variants.txt; : > $o
export VARIANT_SRCS=train_tile.hip
for f in "-DCLID_PRE_BIG=0" "-DCLID_PRE_BIG=1" "-DCLID_PRE_BIG=0" "-DCLID_PRE_BIG=1"; do
  python tools/variant_bench.py "$f" --sequence 60 >> $o 2>> gpurun_out/s6/variants.err
done
cat $o
