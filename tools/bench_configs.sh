#!/bin/bash
# The other BASELINE.json configurations on one MI355X (cfg2 is bench.py's default): cfg3 VitGAN, cfg4 x-transformer 512^2,
# cfg5 ViT-L/14 512^2 in f16, with the fp8 tower, and with the fp8 tower + fp8 decoder convolutions.  Lines land in gpurun_out/${FFVC_ROUND:-r04}_bench_<cfg>.json (+ .err: top GEMM shapes).
C="--steps 10 --warmup 4 --no-cpu-baseline --no-alt-dtype --gemm-shapes 14"
python bench.py $C --model-type vitgan --batch 32 > gpurun_out/${FFVC_ROUND:-r04}_bench_cfg3.json 2> gpurun_out/${FFVC_ROUND:-r04}_bench_cfg3.err
python bench.py $C --model-type xtransformer --dim 256 --depth 16 --vq-image-size 32 --batch 16 > gpurun_out/${FFVC_ROUND:-r04}_bench_cfg4.json 2> gpurun_out/${FFVC_ROUND:-r04}_bench_cfg4.err
python bench.py $C --depth 1 --vq-image-size 32 --batch 8 --clip-model openclip/ViT-L-14/laion2b_s32b_b82k > gpurun_out/${FFVC_ROUND:-r04}_bench_cfg5_f16.json 2> gpurun_out/${FFVC_ROUND:-r04}_bench_cfg5_f16.err
python bench.py $C --depth 1 --vq-image-size 32 --batch 8 --clip-model openclip/ViT-L-14/laion2b_s32b_b82k --clip-fp8 > gpurun_out/${FFVC_ROUND:-r04}_bench_cfg5_fp8.json 2> gpurun_out/${FFVC_ROUND:-r04}_bench_cfg5_fp8.err
python bench.py $C --depth 1 --vq-image-size 32 --batch 8 --clip-model openclip/ViT-L-14/laion2b_s32b_b82k --clip-fp8 --dec-fp8 > gpurun_out/${FFVC_ROUND:-r04}_bench_cfg5_fp8dec.json 2> gpurun_out/${FFVC_ROUND:-r04}_bench_cfg5_fp8dec.err
for f in cfg3 cfg4 cfg5_f16 cfg5_fp8 cfg5_fp8dec; do tail -1 gpurun_out/${FFVC_ROUND:-r04}_bench_$f.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f', round(d['ms_per_step'],2), {k:(round(v['ms'],2),round(v['tflops'])) for k,v in d['kernel_classes'].items()}, {k:round(v['ms'],2) for k,v in d['hbm_kernels'].items()})"; done
