# developer tool (GPU box): waves per SIMD of the directory search with tile numbering, on the sequence workload's map and at cfg2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s6; o=gpurun_out/s6/variants.txt; : > $o
export VARIANT_SRCS="train.hip"
for f in "-DCLID_CD_WAVES_TILES=6" "-DCLID_CD_WAVES_TILES=8" "-DCLID_CD_WAVES_TILES=5" "-DCLID_CD_WAVES_TILES=4"; do
  python tools/variant_bench.py "$f" --sequence 60 >> $o 2>> gpurun_out/s6/variants.err
  python tools/variant_bench.py "$f" --steps 200 --warmup 20 --frame-calls 0 >> $o 2>> gpurun_out/s6/variants.err
done
cat $o
