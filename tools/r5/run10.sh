#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests/test_frontend_gpu.py tests/test_train_gpu.py tests/test_augment_cpu.py -x -q -m gpu 2>&1 | grep -v -E "amdgpu.ids|socket.cpp|Gloo" | tail -8 > gpurun_out/r5/t10.log
bash tools/r05_evidence.sh > gpurun_out/r5/evidence.log 2>&1
tail -n 4 gpurun_out/r5/t10.log; tail -n 6 gpurun_out/r5/evidence.log
