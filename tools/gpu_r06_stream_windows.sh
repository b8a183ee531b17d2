mkdir -p gpurun_out/r06_q
for cfg in "72 24 mix" "72 22 mix" "72 23 mix" "256 24 text" "256 22 text"; do
  echo "== $cfg" 
  BROTLI_AMD_TILE_LOG=1 BROTLI_AMD_VERBOSE=1 timeout 300 python tools/stock_call.py $cfg 2 2>&1 | tail -n 30 | cut -c1-400
done
