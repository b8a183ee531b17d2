#!/bin/bash
# First GPU session of the next round (not run yet): the tiled stream on hardware.
#   1. the GPU tests of the stream path (ctypes only: no torch import)
#   2. one 1 GiB stock call with the stage times (BROTLI_AMD_TILE_LOG=1: a sync behind every stage)
#   3. rocprofv3 kernel statistics of the same call -> gpurun_out/stream_prof (copy the summary to profiles/)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_zzz_stream.py -x -q -m gpu > gpurun_out/stream_tests.log 2>&1
tail -5 gpurun_out/stream_tests.log
python - <<'PY'
import sys
sys.path.insert(0, "tests")
import gen_inputs as G
open("/dev/shm/stream_1g.bin", "wb").write(bytes(G.enwik_text(1 << 30)))
PY
BROTLI_AMD_TILE_LOG=1 timeout 600 python bench.py --stock-call-child /dev/shm/stream_1g.bin 5 22 > gpurun_out/stream_1g_stages.log 2>&1
tail -40 gpurun_out/stream_1g_stages.log
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/stream_prof -- python /root/repo/bench.py --stock-call-child /dev/shm/stream_1g.bin 5 22 > /root/repo/gpurun_out/stream_1g_prof.log 2>&1
rm -f /dev/shm/stream_1g.bin
