#!/bin/bash
# Round 4, session j: the GPU suite and smoke() once more on a fresh box at the shipped tree (a second green run in a row).
ulimit -c 0
O=gpurun_out/r04j
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?: $(tail -1 $O/pytest.log)" | tee $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc $?: $(tail -1 $O/smoke.log)" | tee -a $O/summary.txt
