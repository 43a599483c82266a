#!/bin/bash
# developer tool (GPU box): the round-6 measurement set -> gpurun_out/r06/ (summaries copied to profiles/ afterwards)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r06/set; mkdir -p "$out"
B="python bench.py --no-cpu-baseline"
# 1. headline lines: driver's command shape (steps 20 / warmup 5), the long run, cfg3, cfg4 on one GPU
python bench.py --steps 20 --warmup 5 > $out/bench_cfg2_s20.json 2> $out/log.txt
$B --steps 200 --warmup 20 > $out/bench_cfg2_s200.json 2>> $out/log.txt
$B --config cfg3 --steps 100 --warmup 10 > $out/bench_cfg3.json 2>> $out/log.txt
$B --config cfg4 --steps 50 --warmup 5 > $out/bench_cfg4_1gpu.json 2>> $out/log.txt
$B --analytic --steps 100 --warmup 10 --frame-calls 0 > $out/bench_cfg2_analytic.json 2>> $out/log.txt
# the probing kernels (debug bit 3) next to the cell-directory search, same box
for c in cfg2 cfg3 cfg4; do CLID_DEBUG_FLAGS=8 $B --config $c --steps 50 --warmup 5 --frame-calls 0 > $out/bench_${c}_probing.json 2>> $out/log.txt; done
# 2. batch-size sweep x decode kernel
: > $out/sweep.jsonl
for bs in 16384 65536 262144; do for v in 0 1 2; do
  $B --bs $bs --decode $v --steps 100 --warmup 10 --frame-calls 0 >> $out/sweep.jsonl 2>> $out/log.txt
done; done
# 3. rocprofv3 kernel trace + stats of the driver's command and of cfg3
timeout 600 rocprofv3 --kernel-trace --stats -d $out/trace_cfg2 -o stats --output-format csv -- $B --steps 20 --warmup 5 --frame-calls 0 > $out/trace_cfg2.json 2>> $out/log.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $out/trace_cfg3 -o stats --output-format csv -- $B --config cfg3 --steps 100 --frame-calls 0 > $out/trace_cfg3.json 2>> $out/log.txt
# 4. PMC passes (separate runs, --kernel-trace only): HBM traffic, instruction mix, matrix-core busy cycles
for cfg in cfg2 cfg3; do
  A="$B --config $cfg --steps 64 --warmup 5 --frame-calls 0"
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_$cfg -o fetch --output-format csv -- $A > /dev/null 2>> $out/log.txt
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_$cfg -o write --output-format csv -- $A > /dev/null 2>> $out/log.txt
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES -d $out/pmc_$cfg -o sq --output-format csv -- $A > /dev/null 2>> $out/log.txt
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $out/pmc_$cfg -o mfma --output-format csv -- $A > /dev/null 2>> $out/log.txt
done
python - <<'PY' > gpurun_out/r06/set/pmc_summary.txt
import csv, collections, glob, re, os
out = "gpurun_out/r06/set"
for cfg in ("cfg2", "cfg3"):
    print("=====", cfg, "(bench.py --config %s --steps 64 --warmup 5; per-launch means over all launches of the kernel)" % cfg)
    for fn in sorted(glob.glob(f"{out}/pmc_{cfg}/**/*_counter_collection.csv", recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"]
            if "clid::" in k or k.startswith("k_"):
                agg[re.sub(r"\(.*", "", k).replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(os.path.basename(fn).split("_")[0], k, {c: round(sum(x) / len(x), 1) for c, x in v.items()}, "launches", len(next(iter(v.values()))))
for t in ("trace_cfg2", "trace_cfg3"):
    for fn in sorted(glob.glob(f"{out}/{t}/**/stats_kernel_stats.csv", recursive=True)):
        print("=====", t, "kernel stats (top rows)")
        print("\n".join(open(fn).read().splitlines()[:8])[:3000])
PY
for t in trace_cfg2 trace_cfg3; do f=$(find $out/$t -name 'stats_kernel_trace.csv' | head -1); echo "== $t gaps"; python tools/trace_gaps.py "$f" 900 | grep -E "clid|span|k_" | grep -v "n=   [0-9] "; done >> $out/pmc_summary.txt
# 5. the sequence workload (cfg5): one GPU with the oracle replay of the first frames, and the 2-rank dry run (gloo, one GPU)
timeout 900 python bench_sequence.py --frames 200 --check-frames 3 --profile-last --quiet > $out/sequence_200.json 2>> $out/log.txt
timeout 900 python bench_sequence.py --frames 60 --quiet --gpus 2 --backend gloo > $out/sequence_60_gloo2.json 2>> $out/log.txt
timeout 600 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-cpu-baseline --frame-calls 0 > $out/bench_cfg2_gloo2.json 2>> $out/log.txt
timeout 600 python bench_next.py > $out/next_rows.jsonl 2>> $out/log.txt
# 6. weighted_first: False (fused three-launch iteration vs the autograd loop), the long sequence, process_frame's device time line
timeout 600 python tools/time_wf0.py 20 2>> $out/log.txt | tail -1 > $out/wf0_iteration.json
timeout 900 python bench_sequence.py --frames 800 --quiet > $out/sequence_800.json 2>> $out/log.txt
for n in 0 10 50; do timeout 600 python bench_sequence.py --frames 200 --quiet --track $n 2>> $out/log.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'track_evaluations_per_frame':$n,'scans_per_s':round(d['value'],1),'steady_state':d['steady_state'],'ms_per_frame':d['ms_per_frame']}))"; done > $out/sequence_track.jsonl
timeout 300 python tools/track_eval_timing.py 2>> $out/log.txt | tail -3 > $out/track_eval_timing.jsonl
timeout 300 python tools/sort_timing.py 2>> $out/log.txt | tail -8 > $out/sort_phase_timing.txt
for m in 0 1 0 1; do CLID_ASYNC_VOXEL=$m timeout 600 python bench_sequence.py --frames 120 --quiet 2>> $out/log.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'CLID_ASYNC_VOXEL': $m, **d['steady_state']}))"; done > $out/process_frame_async_ab.jsonl
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $out/ft -o ft --output-format csv -- python bench_sequence.py --frames 60 --quiet > /dev/null 2>> $out/log.txt
python tools/frame_trace.py $out/ft > $out/frame_trace.txt; rm -rf $out/ft
for t in trace_cfg2 trace_cfg3; do f=$(find $out/$t -name 'stats_kernel_stats.csv' | head -1); cp "$f" $out/${t}_kernel_stats.csv; done
rm -rf $out/trace_cfg2 $out/trace_cfg3 $out/pmc_cfg2 $out/pmc_cfg3
for f in train_tile.hip train.hip query.hip query_tile.hip sampler.hip; do tools/resusage.sh $f; done > $out/resource_usage.txt 2>> $out/log.txt
(git rev-parse HEAD 2>/dev/null || echo unknown) > $out/commit.txt
ls -la $out | head -40; tail -5 $out/log.txt
