#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/gpu_q1_diag.py 2>&1 | grep Q1DIAG | tee gpurun_out/j_q1_diag.log
