#!/bin/bash
# round 3, session 1: new tests, bf16 fp32-stream A/B (accuracy + time)
set -o pipefail
O=gpurun_out/r03_s1; mkdir -p $O
python -m pytest tests/test_gpu_round3.py -x -q -s > $O/tests_round3.log 2>&1; echo "round3 tests rc=$?" | tee -a $O/summary.txt
for st in 0 1; do
  for seed in 2 3; do
    GDX_STREAM32=$st python tools/fuzz_loops.py 150 $seed --only bf16 > $O/fuzz_bf16_stream${st}_seed${seed}.log 2>&1
    echo "fuzz bf16 stream32=$st seed=$seed rc=$?" | tee -a $O/summary.txt
  done
done
for st in 0 1; do
  GDX_STREAM32=$st python bench.py --config 5 --dtype bf16 --steps 60 --warmup 10 --no-cpu-baseline > $O/bench_c5_bf16_stream${st}.json 2> $O/bench_c5_bf16_stream${st}.log
  echo "bench c5 bf16 stream32=$st rc=$?" | tee -a $O/summary.txt
  GDX_STREAM32=$st python bench.py --config 5 --dtype bf16 --batch 16 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_c5b16_bf16_stream${st}.json 2> $O/bench_c5b16_bf16_stream${st}.log
done
python bench.py --config 5 --steps 60 --warmup 10 --no-cpu-baseline > $O/bench_c5_fp16.json 2> $O/bench_c5_fp16.log
python bench.py --config 5 --batch 16 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_c5b16_fp16.json 2> $O/bench_c5b16_fp16.log
grep -h ms_per_step $O/*.json | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['config']['workload'][:60], r['dtype'], r['ms_per_step'], r['roofline']['avg_launch_us'])" | tee -a $O/summary.txt
