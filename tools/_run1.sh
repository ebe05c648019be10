mkdir -p gpurun_out/r04b
python -m pytest tests/test_hip_bf16_stages.py::test_tswinplus_eval_mode_bf16_weight_gradients_vs_the_fp32_oracle -x -q -s -m gpu > gpurun_out/r04b/test_grad.log 2>&1
python -m pytest tests/test_hip_configs.py::test_config4_full_size_fp8_finetune_step_vs_the_oracle tests/test_hip_gemm.py tests/test_hip_model.py tests/test_hip_abi.py -q -s -m gpu > gpurun_out/r04b/tests2.log 2>&1
tail -5 gpurun_out/r04b/tests2.log
