#!/bin/bash
# round 6, GPU call 9: cfg3 / cfg4 step with the launch diet on vs its three switches off (the SLN changes have no switch)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
B="python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt-dtype --no-roofline --model-type vitgan --batch 32"
$B > /dev/null 2>&1
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "%.2f ms loss %.5f alloc %s" % (d["ms_per_step"], d["final_loss"], d.get("allocator_in_timed_region")))'
for rep in 1 2 3 4; do
  $B 2>/dev/null | tail -1 | python -c "$P" "new      " | tee -a $O/run9_cfg3_ab.txt
  FFVC_VIT_WGRAD_GROUP=0 FFVC_SMALLM_TT=0 FFVC_SUMS_POOL=0 $B 2>/dev/null | tail -1 | python -c "$P" "all off  " | tee -a $O/run9_cfg3_ab.txt
done
B4="python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt-dtype --no-roofline --model-type xtransformer --dim 256 --depth 16 --vq-image-size 32 --batch 16"
for rep in 1 2 3; do
  $B4 2>/dev/null | tail -1 | python -c "$P" "cfg4 new    " | tee -a $O/run9_cfg4_ab.txt
  FFVC_SUMS_POOL=0 $B4 2>/dev/null | tail -1 | python -c "$P" "cfg4 nopool " | tee -a $O/run9_cfg4_ab.txt
done
