#!/bin/bash
# Round 4, session b: the 512 MiB / 1 GiB stock call after the res[0] fix (stage times), the stream tests, the bench line.
ulimit -c 0
O=gpurun_out/r04b
mkdir -p $O
for mib in 512 1024; do
  BROTLI_AMD_TILE_LOG=1 timeout 500 python tools/stock_call.py $mib 22 text 2 --ref > $O/stock_$mib.log 2>&1
  echo "stock $mib MiB rc $?: $(grep '"stage": "done"' $O/stock_$mib.log | tail -1)" | tee -a $O/summary.txt
done
timeout 300 python tools/stock_call.py 1024 22 text 3 > $O/stock_1024_quiet.log 2>&1
echo "stock 1024 MiB (no stage log) rc $?: $(grep '"stage": "done"' $O/stock_1024_quiet.log | tail -1)" | tee -a $O/summary.txt
timeout 300 python tools/stock_call.py 1024 22 mix 2 --ref > $O/stock_1024_mix.log 2>&1
echo "stock 1024 MiB mix rc $?: $(grep '"stage": "done"' $O/stock_1024_mix.log | tail -1)" | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_zzz_stream.py -q -m gpu -p no:cacheprovider > $O/pytest_stream.log 2>&1
echo "pytest stream rc $?: $(tail -1 $O/pytest_stream.log)" | tee -a $O/summary.txt
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc $?" | tee -a $O/summary.txt
