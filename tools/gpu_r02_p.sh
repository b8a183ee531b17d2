#!/bin/bash
# round 2, session p: qualities 2-4 on hardware, the zero-copy one-call ABI path
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi.py -x -q -k "qualities_2 or golden or unsupported or one_shot or stream_sequences_at" > gpurun_out/p_pytest.log 2>&1
tail -5 gpurun_out/p_pytest.log
timeout 200 python tools/gpu_e2e_sweep.py 4 > gpurun_out/p_e2e.log 2>&1; cat gpurun_out/p_e2e.log
for q in 4 2; do
  timeout 300 python bench.py --quality $q --no-cpu-baseline > gpurun_out/p_bench_q$q.log 2>&1
  grep "^{" gpurun_out/p_bench_q$q.log | cut -c1-900
done
