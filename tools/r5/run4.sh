#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests/test_frontend_gpu.py tests/test_kernels_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r5/t4a.log
python tools/r5/grad_spread.py bf16 6 > gpurun_out/r5/grad_spread_bf16_det.txt 2>&1
python tools/r5/grad_spread.py f16 6 > gpurun_out/r5/grad_spread_f16_det.txt 2>&1
python bench.py --steps 10 --warmup 3 --isolated-table gpurun_out/r5/isolated_sum2.txt > gpurun_out/r5/bench_seq1.json 2> gpurun_out/r5/bench_seq1.err
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r5/t4b.log
tail -n 3 gpurun_out/r5/t4a.log gpurun_out/r5/t4b.log
