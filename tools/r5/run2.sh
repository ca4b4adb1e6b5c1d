#!/bin/bash
# round 5, GPU call 2: DP tests after the deferred-hook fix, the attainable table, side-stream A/B with grouped wgrads, bf16 gradient spread
mkdir -p gpurun_out/r5
python -m pytest tests/test_distributed_gpu.py tests/test_models_gpu.py -x -q -m gpu -k "exchange or bare or own or grouped or partly" 2>&1 | tail -15 > gpurun_out/r5/t2.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-dtype --isolated-table gpurun_out/r5/isolated_sum.txt --gemm-shapes 40 > gpurun_out/r5/bench_att.json 2> gpurun_out/r5/bench_att.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-dtype --no-attainable --no-side-stream > gpurun_out/r5/bench_noside.json 2> gpurun_out/r5/bench_noside.err
python tools/r5/grad_spread.py bf16 8 > gpurun_out/r5/grad_spread_bf16.txt 2>&1
python tools/r5/grad_spread.py f16 4 > gpurun_out/r5/grad_spread_f16.txt 2>&1
tail -n 4 gpurun_out/r5/t2.log
