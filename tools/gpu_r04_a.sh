#!/bin/bash
# Round 4, session a: where HEAD stands on the MI355X.  smoke, the stock call bisected by size (each size a process of
# its own), the whole GPU suite without -x, the default bench line, q9 and the mix for the baselines of this round.
ulimit -c 0
export PYTHONFAULTHANDLER=1
O=gpurun_out/r04a
mkdir -p $O
echo "== smoke" | tee $O/summary.txt
timeout 300 python -c "import __graft_entry__ as e; e.smoke()" > $O/smoke.log 2>&1
echo "smoke rc $?" | tee -a $O/summary.txt
echo "== stock call by size" | tee -a $O/summary.txt
for mib in 64 128 256 512 1024; do
  BROTLI_AMD_TILE_LOG=1 timeout 400 python tools/stock_call.py $mib 22 text 1 --ref > $O/stock_$mib.log 2>&1
  rc=$?
  echo "stock $mib MiB rc $rc: $(grep '"stage": "done"' $O/stock_$mib.log | tail -1)" | tee -a $O/summary.txt
  if [ $rc -ne 0 ]; then
    BROTLI_AMD_TILE_LOG=2 timeout 400 python tools/stock_call.py $mib 22 text 1 > $O/stock_${mib}_each.log 2>&1
    echo "  again with a sync behind every launch: rc $?; last lines:" | tee -a $O/summary.txt
    tail -5 $O/stock_${mib}_each.log | tee -a $O/summary.txt
    break
  fi
done
echo "== pytest -m gpu" | tee -a $O/summary.txt
rm -f gpurun_out/last_test.txt
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?: $(tail -1 $O/pytest.log)" | tee -a $O/summary.txt
echo "== bench" | tee -a $O/summary.txt
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc $?" | tee -a $O/summary.txt
timeout 300 python bench.py --quality 9 --lgwin 24 --shard-kb 512 --steps 2 --no-cpu-baseline > $O/bench_q9.json 2> $O/bench_q9.err
echo "bench q9 rc $?" | tee -a $O/summary.txt
timeout 300 python bench.py --workload silesia --steps 2 --no-cpu-baseline > $O/bench_mix.json 2> $O/bench_mix.err
echo "bench mix rc $?" | tee -a $O/summary.txt
cat $O/summary.txt
