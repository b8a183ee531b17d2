#!/bin/bash
# round 3, GPU session m: the final state — the whole `-m gpu` suite, the RCCL path at world size 1, kernel stats of the
# long-shard command
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1000 python -m pytest tests -x -q -m gpu ) > gpurun_out/r03_m_pytest_gpu.log 2>&1
tail -3 gpurun_out/r03_m_pytest_gpu.log
BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --size-mb 256 > gpurun_out/r03_m_bench_rccl_world1.log 2>&1
grep '^{' gpurun_out/r03_m_bench_rccl_world1.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rccl world 1', d['value'], d['config']['gathered_stream'])" || tail -5 gpurun_out/r03_m_bench_rccl_world1.log
rm -rf gpurun_out/prof_1024k
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_1024k -o bench -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --shard-kb 1024 ) > gpurun_out/r03_m_prof_1024k.log 2>&1
python tools/pmc_summary.py gpurun_out/prof_1024k > gpurun_out/r03_m_kernel_stats_1024k.txt 2>&1
grep -E "KERNEL k_" gpurun_out/r03_m_kernel_stats_1024k.txt | head -24
find gpurun_out -name "*.db" -size +20M -delete
