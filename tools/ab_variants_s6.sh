# developer tool (GPU box): A/B of compile-time variants (only the named source is recompiled)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s6; o=gpurun_out/s6/variants.txt; : > $o
export VARIANT_SRCS=train_tile.hip
for f in "-DCLID_TILE_XMAP=0" "-DCLID_TILE_XMAP=1" "-DCLID_TILE_XMAP=0" "-DCLID_TILE_XMAP=1"; do
  python tools/variant_bench.py "$f" --sequence 60 >> $o 2>> gpurun_out/s6/variants.err
done
cat $o
