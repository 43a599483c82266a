# developer tool (GPU box): A/B of compile-time variants (only the named source is recompiled)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s6; o=gpurun_out/s6/variants.txt; : > $o
export VARIANT_SRCS=train_analytic.hip
for f in "-DCLID_ANALYTIC_REGW=0" "-DCLID_ANALYTIC_REGW=1" "-DCLID_ANALYTIC_REGW=1 -DCLID_ANALYTIC_WAVES=2" "-DCLID_ANALYTIC_REGW=0" "-DCLID_ANALYTIC_REGW=1" "-DCLID_ANALYTIC_REGW=1 -DCLID_ANALYTIC_WAVES=2"; do
  python tools/variant_bench.py "$f" --analytic --steps 100 --warmup 10 --frame-calls 0 >> $o 2>> gpurun_out/s6/variants.err
done
cat $o
