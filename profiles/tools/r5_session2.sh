#!/bin/bash
# round 5, session 2: the new tests (captured collectives, flag hand-offs, long-context shape, bench launcher, log-mel
# contract), the forced-collective A/B of the four exchange forms at world 1, the default bench line with the MR-MT3 /
# long-context workloads, and the step breakdowns of those two from rocprofv3 kernel traces
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python3 -m pytest tests/test_train_graph_gpu.py tests/test_bench_shape_gpu.py tests/test_ddp_gpu.py -m gpu -x -q -s 2>&1 | grep -v amdgpu.ids | tail -40 | tee $O/s2_pytest_a.log
timeout 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_train_infer_gpu.py -m gpu -x -q -k "logmel or bench_spawns or launch_structure" 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/s2_pytest_b.log
B="--steps 20 --warmup 3 --no-inference --no-cpu-baseline --no-roofline --no-extra-workloads --extra-batch 0"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'ms/step %.3f' % d['ms_per_step'], 'graphs/step', d['graph_segments'], '|', d['collectives'], '| captured:', d.get('collectives_captured'))"; }
for rep in 1 2; do
  timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s2_plain_$rep.json | show "plain (no collectives)      "
  MRMT3_DDP_FORCE_COLLECTIVES=1 timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s2_forced_torch_$rep.json | show "segments + torch.distributed"
  MRMT3_DDP_FORCE_COLLECTIVES=1 MRMT3_DDP_NATIVE=1 timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s2_forced_native_$rep.json | show "segments + mrmt3_allreduce  "
  MRMT3_DDP_FORCE_COLLECTIVES=1 MRMT3_DDP_GRAPH=1 timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s2_forced_twograph_$rep.json | show "two graphs (captured)       "
  MRMT3_DDP_FORCE_COLLECTIVES=1 MRMT3_DDP_GRAPH=inline timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s2_forced_inline_$rep.json | show "one graph, in-line          "
done 2>&1 | tee $O/s2_collectives_ab.log
unset MASTER_ADDR MASTER_PORT RANK WORLD_SIZE LOCAL_RANK
timeout 900 python3 bench.py > $O/s2_bench_default.json 2> $O/s2_bench_default.err; tail -c 600 $O/s2_bench_default.err
python3 -c "
import json; d=json.load(open('$O/s2_bench_default.json'))
print('default line: ms/step', d['ms_per_step'], 'value', d['value'])
for k in ('train_b12','train_mrmt3','train_mrmt3_b12','train_long_context'):
    print(k, json.dumps(d.get(k)))
"
for W in mrmt3 long; do
  if [ $W = mrmt3 ]; then F="--variant segmem_v2_with_prev"; BB=64; else F="--variant segmem_v2_with_prev --mel-frames 2048 --batch 12"; BB=12; fi
  rm -rf $O/prof_$W
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W -- python3 bench.py $F --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-roofline --no-extra-workloads --extra-batch 0 > $O/bench_${W}_under_rocprof.json 2> $O/bench_${W}_under_rocprof.err
  python3 profiles/tools/step_breakdown.py $O/prof_$W $BB "MR-MT3 (segmem_v2_with_prev)" "${F/ --batch 12/}" > $O/r05_step_breakdown_$W.txt 2>&1; head -24 $O/r05_step_breakdown_$W.txt
  find $O/prof_$W -name "*kernel_trace.csv" -delete; find $O/prof_$W -name "*.db" -delete
done
