#!/bin/bash
# Per-round profile set (run on the MI355X box): tools/profile_round.sh <tag> [configs...]   -> gpurun_out/<tag>/
#   per config: the bench record, the rocprofv3 --kernel-trace --stats summary of the same command, and separate --pmc passes:
#   FETCH_SIZE, WRITE_SIZE (HBM traffic of the FFN-1 GEMM) and SQ_VALU_MFMA_BUSY_CYCLES (matrix-pipe utilisation).
TAG=${1:-rXX}; shift
CONFIGS=${@:-2 5}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in $CONFIGS; do
  case $c in
    1) B="--config 1";       S="--config 1 --steps 10 --warmup 10"; P="--config 1 --steps 10 --warmup 2";;
    2) B="";                 S="--steps 100 --warmup 10";           P="--steps 10 --warmup 2";;
    3) B="--config 3";       S="--config 3 --steps 20 --warmup 3";  P="--config 3 --steps 3 --warmup 1";;
    4) B="--config 4 --steps 8 --warmup 2"; S="--config 4 --steps 4 --warmup 1"; P="--config 4 --steps 2 --warmup 1";;
    5) B="--config 5 --steps 30 --warmup 5"; S="--config 5 --steps 12 --warmup 3"; P="--config 5 --steps 5 --warmup 1";;
    genea) B="--config genea"; S="--config genea --steps 100 --warmup 10"; P="--config genea --steps 10 --warmup 2";;
    5b16) B="--config 5 --batch 16 --steps 200 --warmup 20"; S="--config 5 --batch 16 --steps 60 --warmup 10"; P="--config 5 --batch 16 --steps 10 --warmup 2";;
  esac
  timeout -k 10 400 python3 $R/bench.py $B --no-cpu-baseline > $O/bench_c$c.json 2> $O/bench_c$c.log || exit 1
  rocprofv3 --kernel-trace --stats -d $O/kt_c$c -o k -- python3 $R/bench.py $S --no-cpu-baseline > $O/kt_c$c.log 2>&1 || exit 1
  python3 $R/tools/kstats_db.py $O/kt_c$c/k_results.db 24 > $O/kernel_stats_c$c.txt || exit 1
  cat $O/bench_c$c.json >> $O/kernel_stats_c$c.txt
  for ctr in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    t=$(echo $ctr | cut -d' ' -f1)
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${t}_c$c -o p -- python3 $R/bench.py $P --no-cpu-baseline > $O/pmc_${t}_c$c.log 2>&1 || exit 1
  done
  rm -rf $O/kt_c$c
  echo "config $c done" >> $O/progress.txt
done
cd $R && python3 tools/pmc_ffn1.py $O $CONFIGS > $O/pmc_summary.txt 2>&1
find $O -name "*.csv" -size +3M -delete
cat $O/pmc_summary.txt
