#!/bin/bash
# round 6, GPU call 8: cfg3 launch diet — grouped MLP weight gradients of the VitGAN blocks, padded qkv / w_out weight gradients on the
# LDS-DMA kernel, SLN scalar gradients + shared modulation gradient written in place, pooled GroupNorm accumulators
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
python -m pytest tests/test_gemm_gpu.py tests/test_models_gpu.py tests/test_kernels_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|error" | tee $O/run8_pytest.txt
python tools/r6/cfg3_tiles.py 2>/dev/null | grep -v "^/opt" | tee $O/run8_cfg3_tiles.txt
B="python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt-dtype --no-roofline --model-type vitgan --batch 32"
$B > /dev/null 2>&1
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "%.2f ms loss %.5f" % (d["ms_per_step"], d["final_loss"]))'
for rep in 1 2 3; do
  $B 2>/dev/null | tail -1 | python -c "$P" "new      " | tee -a $O/run8_cfg3_ab.txt
  FFVC_VIT_WGRAD_GROUP=0 $B 2>/dev/null | tail -1 | python -c "$P" "nogroup  " | tee -a $O/run8_cfg3_ab.txt
  FFVC_SMALLM_TT=0 $B 2>/dev/null | tail -1 | python -c "$P" "tt_v1    " | tee -a $O/run8_cfg3_ab.txt
  FFVC_SUMS_POOL=0 $B 2>/dev/null | tail -1 | python -c "$P" "nopool   " | tee -a $O/run8_cfg3_ab.txt
done
B2="python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt-dtype --no-roofline"
for rep in 1 2; do
  $B2 2>/dev/null | tail -1 | python -c "$P" "cfg2 new   " | tee -a $O/run8_cfg2_ab.txt
  FFVC_SUMS_POOL=0 $B2 2>/dev/null | tail -1 | python -c "$P" "cfg2 nopool" | tee -a $O/run8_cfg2_ab.txt
done
