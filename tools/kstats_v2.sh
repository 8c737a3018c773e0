#!/bin/bash
# rocprofv3 kernel statistics of the V2 (local attention + RoPE) workloads: tools/kstats_v2.sh <tag> -> gpurun_out/<tag>/
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {   # name, bench arguments
  n=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/kt_$n -o k -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/kt_$n.log 2>&1 || exit 1
  python3 $R/tools/kstats_db.py $O/kt_$n/k_results.db 20 > $O/kernel_stats_$n.txt || exit 1
  tail -1 $O/kt_$n.log | cut -c1-300 >> $O/bench_lines.txt
  echo "$n done" >> $O/progress.txt
}
run genea --config genea --steps 100 --warmup 10
run c2v2 --arch mdm --frames 200 --steps 100 --warmup 10
run c1 --config 1 --steps 10 --warmup 10
