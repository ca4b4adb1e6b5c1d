#!/bin/bash
# round 6, GPU call 10: augmentation kernels — wave-level bounding-box reduction in the tiled backward, one cutout per blockIdx.y in the
# forward (library A = before) — isolated (tools/r6/aug_seq_bench.py) and in the cfg2 step
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
LA=$R/feed_forward_vqgan_clip_amd/lib/libffvc_hip_a.so
for i in 1 2; do python tools/r6/aug_seq_bench.py 2>/dev/null | tail -1 | tee -a $O/run10_aug.txt; FFVC_LIB=$LA python tools/r6/aug_seq_bench.py 2>/dev/null | tail -1 | tee -a $O/run10_aug.txt; done
B="python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt-dtype --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "%.2f ms loss %.5f" % (d["ms_per_step"], d["final_loss"]))'
$B > /dev/null 2>&1
for rep in 1 2 3; do
  $B 2>/dev/null | tail -1 | python -c "$P" "new" | tee -a $O/run10_step_ab.txt
  FFVC_LIB=$LA $B 2>/dev/null | tail -1 | python -c "$P" "A  " | tee -a $O/run10_step_ab.txt
done
