#!/bin/bash
# One complete GPU-box session of round 2: all GPU tests, smoke, the default bench line (CPU baseline, whole-output
# sha256 parity, ABI end to end), rocprofv3 kernel stats, the other BASELINE configurations, RCCL path at world size 1.
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${TAG:-m}
( time timeout 1800 python -m pytest tests -x -q -m gpu ) > gpurun_out/${T}_pytest_gpu.log 2>&1
tail -4 gpurun_out/${T}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; tail -2 gpurun_out/${T}_smoke.log
( time timeout 900 python bench.py ) > gpurun_out/${T}_bench.log 2>&1
grep "^{" gpurun_out/${T}_bench.log | cut -c1-3800
rm -rf gpurun_out/${T}_prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/${T}_prof -o bench -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline ) > gpurun_out/${T}_prof.log 2>&1
python tools/pmc_summary.py gpurun_out/${T}_prof > gpurun_out/${T}_prof_summary.txt 2>&1
grep -E "KERNEL k_" gpurun_out/${T}_prof_summary.txt
find gpurun_out -name "*.db" -delete
run() {  # name, args, env...
  local name=$1; local args=$2; shift; shift
  ( env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args ) > gpurun_out/${T}_$name.log 2>&1
  grep "^{" gpurun_out/${T}_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['config'].get('ratio'), d['config']['stage_ms'])" || tail -3 gpurun_out/${T}_$name.log
}
run silesia "--workload silesia"
run q9_512k "--quality 9 --lgwin 24 --shard-kb 512 --steps 2"
run q1_random "--quality 1 --data random --steps 5"
run q1_text_lgwin18 "--quality 1 --data text --lgwin 18"
BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --size-mb 256 > gpurun_out/${T}_bench_dist1.log 2>&1; grep "^{" gpurun_out/${T}_bench_dist1.log | cut -c1-900
