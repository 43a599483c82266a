# developer tool (GPU box): A/B of compile-time variants (only the named source is recompiled)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s6; o=gpurun_out/s6/variants.txt; : > $o
for f in "-DCLID_QT_WAVES=4" "-DCLID_QT_WAVES=3" "-DCLID_QT_WAVES=2" "-DCLID_QT_WAVES=4" "-DCLID_QT_WAVES=3"; do
  python tools/time_sdf_query.py "$f" >> $o 2>> gpurun_out/s6/variants.err
done
cat $o
