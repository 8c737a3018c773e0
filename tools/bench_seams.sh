#!/bin/bash
# Drop-in seam comparison + one record per BASELINE config.  Usage: tools/bench_seams.sh OUT.jsonl
out=${1:-gpurun_out/seams.jsonl}
: > "$out"
for cfg in 2 genea; do
  for seam in philox torch stepwise; do
    python bench.py --config $cfg --seam $seam --steps 200 --warmup 20 --no-cpu-baseline >> "$out" 2>> "$out.err" || exit 1
  done
done
python bench.py --config 1 --warmup 10 >> "$out" 2>> "$out.err" || exit 1
python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline >> "$out" 2>> "$out.err" || exit 1
python bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline >> "$out" 2>> "$out.err" || exit 1
python bench.py --config 5 --steps 30 --warmup 5 --no-cpu-baseline >> "$out" 2>> "$out.err" || exit 1
