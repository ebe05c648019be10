mkdir -p gpurun_out/r04c
python -m pytest tests/test_hip_attention.py tests/test_hip_swin.py -x -q -m gpu 2>&1 | tail -2
python tools/bench_attn.py 2>&1 | grep "stage1"
STSWIN_ATTN_BWD_QPF=0 python tools/bench_attn.py 2>&1 | grep "stage1 bwd"
python tools/attn_timeline8.py > gpurun_out/r04c/timeline_qpf.txt 2>&1
