#!/bin/bash
# round 2, session o: host copies through pinned lanes (brotli_amd_encode_host)
mkdir -p gpurun_out
for m in plain 1 2 4 8; do timeout 200 python tools/gpu_e2e_sweep.py $m >> gpurun_out/o_e2e.log 2>&1; done
cat gpurun_out/o_e2e.log
