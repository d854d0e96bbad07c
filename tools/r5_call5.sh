mkdir -p gpurun_out/r5c5
( timeout 900 python -m pytest tests/test_modules_gpu.py -x -q --durations=8 -k "weight_scope_step_bit_identical or two_models or allocator_churn or cfg2_full_size or mask_head_side_stream or gradient_reducer_streams or training_step_tiny or flat_sgd or train_loop_flat_sgd" 2>&1 | grep -v amdgpu.ids | tail -22 ) > gpurun_out/r5c5/modules_subset.log 2>&1
tail -16 gpurun_out/r5c5/modules_subset.log
bash tools/r5_ab_env.sh r5c5 "CFUN_WGRAD_STREAM=0" "CFUN_WGRAD_STREAM=1" "CFUN_WGRAD_STREAM=1"
