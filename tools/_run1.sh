mkdir -p gpurun_out/r04f
timeout 900 python -m pytest tests/test_hip_gemm.py tests/test_hip_production_dispatch.py tests/test_hip_swin.py tests/test_hip_head.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_hip_configs.py -x -q -m gpu -k "reproducible or config2" 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r04f/bench_fused.log 2>&1
STSWIN_TN_FUSED=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r04f/bench_unfused.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r04f/bench_fused2.log 2>&1
grep -h -o '"value": [0-9.]*' gpurun_out/r04f/bench_fused.log gpurun_out/r04f/bench_unfused.log gpurun_out/r04f/bench_fused2.log | awk 'NR%1==0'
