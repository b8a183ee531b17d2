#!/bin/bash
# round 3, GPU session p (the last minutes of the budget): the stock one-shot call on buffers longer than the
# window (tests/test_gpu_zzz_stream.py: ctypes only, no torch import)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 215 python -m pytest tests/test_gpu_zzz_stream.py -x -q -s -m gpu > gpurun_out/r03_p_stream_tests.log 2>&1
tail -15 gpurun_out/r03_p_stream_tests.log
cat gpurun_out/stock_call_1GiB.json 2>/dev/null
