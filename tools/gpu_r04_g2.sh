#!/bin/bash
# Round 4, session g2: where the host-side time of the 1 GiB stock call goes (laps of brotli_amd_encode_host next to the
# device stages), and the copy lanes.
ulimit -c 0
O=gpurun_out/r04g2
mkdir -p $O
export TMPDIR=/tmp
BROTLI_AMD_TILE_LOG=1 timeout 400 python tools/stock_call.py 1024 22 text 3 > $O/stock_1024.log 2>&1
echo "stock 1 GiB rc $?: $(grep '"stage": "done"' $O/stock_1024.log | tail -1)" | tee $O/summary.txt
grep -E "host:|stream stage" $O/stock_1024.log | tail -16 | tee -a $O/summary.txt
BROTLI_AMD_COPY_THREADS=8 timeout 400 python tools/stock_call.py 1024 22 text 3 > $O/stock_1024_t8.log 2>&1
echo "stock 1 GiB, 8 copy lanes rc $?: $(grep '"stage": "done"' $O/stock_1024_t8.log | tail -1)" | tee -a $O/summary.txt
BROTLI_AMD_COPY_THREADS=2 timeout 400 python tools/stock_call.py 1024 22 text 3 > $O/stock_1024_t2.log 2>&1
echo "stock 1 GiB, 2 copy lanes rc $?: $(grep '"stage": "done"' $O/stock_1024_t2.log | tail -1)" | tee -a $O/summary.txt
