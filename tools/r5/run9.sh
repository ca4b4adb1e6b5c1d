#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests -x -q -m gpu 2>&1 | grep -v -E "amdgpu.ids|socket.cpp|Gloo" | tail -30 > gpurun_out/r5/t9.log
tail -n 8 gpurun_out/r5/t9.log
