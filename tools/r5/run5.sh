#!/bin/bash
# round 5, GPU call 5: small-M GEMM A/B (64x128 tile) in isolation and on cfg3 / cfg4
mkdir -p gpurun_out/r5
python -m pytest tests/test_gemm_gpu.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r5/t5.log
FFVC_SMALLM=0 python tools/r5/small_m_bench.py 512 > gpurun_out/r5/smallm_off.txt 2>&1
FFVC_SMALLM=1 python tools/r5/small_m_bench.py 512 > gpurun_out/r5/smallm_on.txt 2>&1
C="--steps 6 --warmup 2 --no-cpu-baseline --no-alt-dtype --no-attainable --gemm-shapes 14"
for s in 0 1; do
FFVC_SMALLM=$s python bench.py $C --model-type vitgan --batch 32 > gpurun_out/r5/cfg3_smallm$s.json 2> gpurun_out/r5/cfg3_smallm$s.err
FFVC_SMALLM=$s python bench.py $C --model-type xtransformer --dim 256 --depth 16 --vq-image-size 32 --batch 16 > gpurun_out/r5/cfg4_smallm$s.json 2> gpurun_out/r5/cfg4_smallm$s.err
done
tail -n 3 gpurun_out/r5/t5.log
for f in cfg3_smallm0 cfg3_smallm1 cfg4_smallm0 cfg4_smallm1; do tail -1 gpurun_out/r5/$f.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f', round(d['ms_per_step'],2))"; done
