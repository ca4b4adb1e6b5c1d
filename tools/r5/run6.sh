#!/bin/bash
mkdir -p gpurun_out/r5
python tools/r5/aug_seq_bench.py > gpurun_out/r5/aug_seq_bench.txt 2>&1
python -m pytest tests/test_frontend_gpu.py tests/test_models_gpu.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r5/t6.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-dtype --no-attainable > gpurun_out/r5/bench_aug2.json 2> gpurun_out/r5/bench_aug2.err
tail -n 3 gpurun_out/r5/t6.log; cat gpurun_out/r5/aug_seq_bench.txt
