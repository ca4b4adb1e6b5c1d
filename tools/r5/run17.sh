#!/bin/bash
mkdir -p gpurun_out/r5
t0=$(date +%s)
python bench.py > gpurun_out/r5/bench_default.json 2> gpurun_out/r5/bench_default.err
t1=$(date +%s)
echo "wall $((t1-t0)) s" > gpurun_out/r5/bench_default.time
cat gpurun_out/r5/bench_default.time
