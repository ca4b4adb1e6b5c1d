#!/bin/bash
mkdir -p gpurun_out/r5
python tools/r5/ln_bwd_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r5/ln_bwd_probe.txt; cat gpurun_out/r5/ln_bwd_probe.txt
