# developer tool (GPU box): A/B of the pool stream's priority on the sequence workload
cd $GRAFT_REPO_ROOT; python -c "
import torch; print('torch priority range', torch.cuda.Stream.priority_range())
import ctypes as C; hip=C.CDLL('libamdhip64.so'); a,b=C.c_int(),C.c_int(); print(hip.hipDeviceGetStreamPriorityRange(C.byref(a),C.byref(b)), a.value, b.value)"
for m in 0 1 0 1; do CLID_POOL_PRIO=$m timeout 600 python bench_sequence.py --frames 120 --quiet 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'CLID_POOL_PRIO': $m, **d['steady_state']}))"; done
