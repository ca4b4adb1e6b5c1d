#!/bin/bash
mkdir -p gpurun_out/r5
python tools/step_ops.py > gpurun_out/r5/step_ops.txt 2>&1
tail -n 70 gpurun_out/r5/step_ops.txt | cut -c1-200
