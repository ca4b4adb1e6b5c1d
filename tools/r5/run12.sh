#!/bin/bash
mkdir -p gpurun_out/r5
for bm in 128 256 512; do
FFVC_GEMM2_BM=$bm python tools/r5/small_m_bench.py 512 2>&1 | grep -E "wgrad|^#" > gpurun_out/r5/smallm_wgrad_bm$bm.txt
done
python tools/r5/small_m_bench.py 512 2>&1 | grep -E "wgrad|^#" > gpurun_out/r5/smallm_wgrad_default.txt
tail -n 8 gpurun_out/r5/smallm_wgrad_*.txt
