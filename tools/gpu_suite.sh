#!/bin/bash
# the round-end sequence on one box: GPU test suite, smoke, headline bench (short) -- outputs under gpurun_out/suite/
O=gpurun_out/suite; mkdir -p $O; rm -f $O/*
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log)"
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.log)"
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.log; echo "bench rc=$?"
python bench.py --config genea --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_genea.json 2> $O/bench_genea.log
python bench.py --config 5 --steps 60 --warmup 10 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.log
python bench.py --config 1 --no-cpu-baseline > $O/bench_c1.json 2> $O/bench_c1.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/suite/bench_*.json')):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], r['ms_per_step'], 'ms/step', r['value'], r['unit'], 'roofline', r['roofline']['frac'], r['roofline']['avg_launch_us'])
    except Exception as e: print(f, 'unreadable', e)
PY
