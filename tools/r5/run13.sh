#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests/test_gemm_gpu.py tests/test_models_gpu.py -x -q -m gpu 2>&1 | grep -v -E "amdgpu.ids|socket.cpp|Gloo" | tail -6 > gpurun_out/r5/t13.log
python tools/r5/small_m_bench.py 512 2>&1 | grep -E "wgrad|^#" > gpurun_out/r5/smallm_wgrad_new.txt
C="--steps 10 --warmup 3 --no-cpu-baseline --no-alt-dtype --no-attainable --no-roofline"
for s in 0 1; do
FFVC_SMALLM=$s python bench.py $C --model-type vitgan --batch 32 > gpurun_out/r5/cfg3b_smallm$s.json 2> gpurun_out/r5/cfg3b_smallm$s.err
FFVC_SMALLM=$s python bench.py $C --model-type xtransformer --dim 256 --depth 16 --vq-image-size 32 --batch 16 > gpurun_out/r5/cfg4b_smallm$s.json 2> gpurun_out/r5/cfg4b_smallm$s.err
FFVC_SMALLM=$s python bench.py $C > gpurun_out/r5/cfg2b_smallm$s.json 2> gpurun_out/r5/cfg2b_smallm$s.err
done
tail -n 3 gpurun_out/r5/t13.log; cat gpurun_out/r5/smallm_wgrad_new.txt
for f in cfg3b_smallm0 cfg3b_smallm1 cfg4b_smallm0 cfg4b_smallm1 cfg2b_smallm0 cfg2b_smallm1; do tail -1 gpurun_out/r5/$f.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f', round(d['ms_per_step'],2), d['step_ms_main_stream'])"; done
