# developer tool (GPU box): A/B of compile-time variants (only the named source is recompiled)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s6; o=gpurun_out/s6/variants.txt; : > $o
export VARIANT_SRCS=train_tile.hip
for f in "-DCLID_TILE_WAVES=2" "-DCLID_TILE_WAVES=2"; do
  python tools/variant_bench.py "$f" --steps 200 --warmup 20 --frame-calls 0 >> $o 2>> gpurun_out/s6/variants.err
  python tools/variant_bench.py "$f" --config cfg3 --steps 100 --warmup 10 --frame-calls 0 >> $o 2>> gpurun_out/s6/variants.err
done
git stash -q 2>/dev/null
cat $o
