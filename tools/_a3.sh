set -e
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "fp32_attention" > gpurun_out/a3_test.log 2>&1
for v in 2 4; do timeout -k 10 60 python tools/attn_one.py 64 197 4 512 $v; done > gpurun_out/a3_time.log 2>&1
for v in 2 4; do timeout -k 10 60 python tools/attn_one.py 64 201 4 512 $v; done >> gpurun_out/a3_time.log 2>&1
for v in 2 4; do timeout -k 10 60 python tools/attn_one.py 256 197 4 512 $v; done >> gpurun_out/a3_time.log 2>&1
for v in 2 4; do timeout -k 10 60 python tools/attn_one.py 4 61 4 512 $v; done >> gpurun_out/a3_time.log 2>&1
