#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests -x -q -m gpu 2>&1 | grep -v -E "amdgpu.ids|socket.cpp|Gloo" | tail -8 > gpurun_out/r5/t16.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5/smoke16.log 2>&1
bash tools/r05_evidence.sh > gpurun_out/r5/evidence16.log 2>&1
tail -n 3 gpurun_out/r5/t16.log; tail -n 2 gpurun_out/r5/smoke16.log; tail -n 5 gpurun_out/r5/evidence16.log
