cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pmc2
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d gpurun_out/pmc2 -o a --output-format csv -- python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/pmc2/loga.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d gpurun_out/pmc2 -o b --output-format csv -- python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/pmc2/logb.txt 2>&1
ls gpurun_out/pmc2
