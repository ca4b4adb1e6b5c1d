#!/bin/bash
mkdir -p gpurun_out/r5
FFVC_DP_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python tools/dp_overlap.py mlp_mixer 4 > gpurun_out/r5/dp_overlap_mixer.txt 2>&1
python -m pytest tests/test_distributed_gpu.py tests/test_models_gpu.py -x -q -m gpu -k "exchange or bare or own or grouped or partly" 2>&1 | tail -15 > gpurun_out/r5/t3.log
python tools/r5/grad_spread.py bf16 6 > gpurun_out/r5/grad_spread_bf16.txt 2>&1
python tools/r5/grad_spread.py f16 6 > gpurun_out/r5/grad_spread_f16.txt 2>&1
python tools/r5/grad_spread.py f16 6 1 > gpurun_out/r5/grad_spread_f16_ls1.txt 2>&1
tail -n 4 gpurun_out/r5/t3.log
