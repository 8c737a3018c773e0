python -m pytest tests/test_gpu_round2.py -m gpu -q -k "clip" > gpurun_out/t4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t4.log; tail -3 gpurun_out/t4.log
for w in 4 8 q 4 8 q; do GDX_ATTNH_WAVES=$w python tools/attnh_one.py 128 521 4 1024 2>&1 | grep "attention f16" | sed "s/^/[waves=$w] /"; done
for w in 4 q 4 q; do GDX_ATTNH_WAVES=$w python bench.py --config 5 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('[waves=$w] config5 ms/step', r['ms_per_step'], 'step_tflops', r['step_tflops'])"; done
