#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests/test_distributed_gpu.py -x -q -m gpu -k grouped 2>&1 | tail -60 > gpurun_out/r5/t8.log
cat gpurun_out/r5/t8.log | grep -v -E "amdgpu.ids|socket.cpp|Gloo" | tail -40
