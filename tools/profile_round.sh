#!/bin/bash
# Per-round profile set (run on the MI355X box): tools/profile_round.sh <tag>   -> gpurun_out/<tag>/
#   bench records (configs 2 and 5), rocprofv3 --kernel-trace --stats summaries of the same commands, and separate --pmc passes:
#   FETCH_SIZE, WRITE_SIZE (HBM traffic of the FFN-1 GEMM) and SQ_VALU_MFMA_BUSY_CYCLES (matrix-pipe utilisation).
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in 2 5; do
  if [ $c = 2 ]; then S="--steps 100 --warmup 10"; P="--steps 10 --warmup 2"; else S="--config 5 --steps 12 --warmup 3"; P="--config 5 --steps 5 --warmup 1"; fi
  timeout -k 10 300 python3 $R/bench.py $( [ $c = 5 ] && echo "--config 5 --steps 30 --warmup 5 --no-cpu-baseline" ) > $O/bench_c$c.json 2> $O/bench_c$c.log || exit 1
  rocprofv3 --kernel-trace --stats -d $O/kt_c$c -o k -- python3 $R/bench.py $S --no-cpu-baseline > $O/kt_c$c.log 2>&1 || exit 1
  python3 $R/tools/kstats_db.py $O/kt_c$c/k_results.db 24 > $O/kernel_stats_c$c.txt || exit 1
  for ctr in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    t=$(echo $ctr | cut -d' ' -f1)
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${t}_c$c -o p -- python3 $R/bench.py $P --no-cpu-baseline > $O/pmc_${t}_c$c.log 2>&1 || exit 1
  done
  echo "config $c done" >> $O/progress.txt
done
cd $R && python3 tools/pmc_ffn1.py $O > $O/pmc_summary.txt 2>&1
cat $O/pmc_summary.txt
