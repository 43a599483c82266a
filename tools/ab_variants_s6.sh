# developer tool (GPU box): k_pool_scatter grid bound (blocks per CU) on the sequence workload
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s6
timeout 600 python -m pytest tests/test_mapops_gpu.py tests/test_sampler_gpu.py -m gpu -x -q 2>&1 | tail -1
for m in 0 4 6 2 0 4 6 2; do CLID_POOL_SCATTER_BPC=$m timeout 600 python bench_sequence.py --frames 100 --quiet 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'BPC': $m, **d['steady_state']}))"; done
