mkdir -p gpurun_out/r04j
timeout 1200 python -m pytest tests/test_hip_fp8.py -x -q -m gpu 2>&1 | tail -3
python tools/bench_attn.py 2>&1 | grep "qkv GEMM\|fp8-stored\|table+index" > gpurun_out/r04j/bench_attn.txt
cat gpurun_out/r04j/bench_attn.txt
for i in 1 2; do
STSWIN_FP8_ATTN=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r04j/bench_fp8_$i.log 2>&1
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r04j/bench_bf16_$i.log 2>&1
done
grep -H -o '"value": [0-9.]*' gpurun_out/r04j/bench_*.log | grep -v "\.[0-9]*e"
