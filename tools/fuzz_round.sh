#!/bin/bash
# Random parity sweeps of the current build against the CPU oracle (the checker), one box: tools/fuzz_round.sh <tag> -> gpurun_out/<tag>/
TAG=${1:-fuzz}
O=gpurun_out/$TAG; mkdir -p $O
for seed in 4 5 6; do
  timeout -k 10 900 python tools/fuzz_forward.py 200 $seed > $O/forward_$seed.log 2>&1; echo "forward seed $seed rc=$?" | tee -a $O/progress.txt
done
for seed in 2 3; do
  timeout -k 10 900 python tools/fuzz_loops.py 150 $seed > $O/loops_$seed.log 2>&1; echo "loops seed $seed rc=$?" | tee -a $O/progress.txt
done
tail -n 4 $O/forward_*.log $O/loops_*.log
