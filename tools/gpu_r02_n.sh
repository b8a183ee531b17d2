#!/bin/bash
# round 2, session n: the mixed-corpus pathology fix (carried store count in c_search_exact) and the host-copy knobs
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_zy_full_size.py -x -q -k "mixed or q5" > gpurun_out/n_pytest.log 2>&1
tail -3 gpurun_out/n_pytest.log
timeout 300 python bench.py --workload silesia --no-cpu-baseline > gpurun_out/n_silesia.log 2>&1
grep "^{" gpurun_out/n_silesia.log | cut -c1-1500
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/n_text.log 2>&1
grep "^{" gpurun_out/n_text.log | cut -c1-600
timeout 400 python tools/gpu_e2e_sweep.py > gpurun_out/n_e2e.log 2>&1
cat gpurun_out/n_e2e.log
