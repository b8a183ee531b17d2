#!/bin/bash
# round 2, session q: qualities 2-4 after the fence removal (table read through the L2)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi.py -x -q -k "qualities_2 or stream_sequences_at or (one_shot_deep and (2- or 3- or 4-))" > gpurun_out/q_pytest.log 2>&1
tail -3 gpurun_out/q_pytest.log
for q in 4 2 3; do
  timeout 200 python bench.py --quality $q --no-cpu-baseline --steps 3 > gpurun_out/q_bench_q$q.log 2>&1
  grep "^{" gpurun_out/q_bench_q$q.log | cut -c1-120
  grep -o '"stage_ms": {[^}]*}' gpurun_out/q_bench_q$q.log
done
