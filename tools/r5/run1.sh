#!/bin/bash
# round 5, GPU call 1: affected tests + cfg2 bench with the grouped weight gradients (4 / 8 / off)
mkdir -p gpurun_out/r5
python -m pytest tests/test_gemm_gpu.py -x -q -m gpu -k "grouped or tn_wgrad or weight_gradient" 2>&1 | tail -15 > gpurun_out/r5/t_gemm.log
python -m pytest tests/test_models_gpu.py tests/test_fullsize_gpu.py tests/test_distributed_gpu.py tests/test_train_gpu.py tests/test_frontend_gpu.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r5/t1.log
python bench.py --steps 10 --warmup 3 > gpurun_out/r5/bench_g4.json 2> gpurun_out/r5/bench_g4.err
FFVC_WGRAD_GROUP=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-dtype > gpurun_out/r5/bench_g0.json 2> gpurun_out/r5/bench_g0.err
FFVC_WGRAD_GROUP=8 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-dtype > gpurun_out/r5/bench_g8.json 2> gpurun_out/r5/bench_g8.err
python bench.py --steps 10 --warmup 3 --augment-fused --no-cpu-baseline --no-alt-dtype > gpurun_out/r5/bench_fused.json 2> gpurun_out/r5/bench_fused.err
tail -4 gpurun_out/r5/t_gemm.log gpurun_out/r5/t1.log
