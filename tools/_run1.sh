mkdir -p gpurun_out/r04d
python -m pytest tests/test_hip_gemm.py tests/test_hip_production_dispatch.py -x -q -m gpu 2>&1 | tail -2
python tools/epi_decomp.py > gpurun_out/r04d/epi_decomp.txt 2>&1
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r04d/bench.log 2>&1
grep -o '"value": [0-9.]*' gpurun_out/r04d/bench.log | head -1; grep -o '"frac": [0-9.]*' gpurun_out/r04d/bench.log
