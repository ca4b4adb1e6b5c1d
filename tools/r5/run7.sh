#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r5/t7.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5/smoke.log 2>&1
tail -n 5 gpurun_out/r5/t7.log; tail -n 4 gpurun_out/r5/smoke.log
