#!/bin/bash
# round 6, GPU call 11: the argmin inside the VQ distance GEMM (FFVC_F_VQ_ARGMIN) — tests, isolated table, step A/B (FFVC_VQ_FUSE=0 = two launches)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
python -m pytest tests/test_kernels_gpu.py -q -x -k "vq" 2>&1 | grep -E "passed|failed|^E " | head -5 | tee $O/run11_pytest.txt
python tools/r6/vq_bench.py 2>/dev/null | grep rows | tee $O/run11_vq.txt
python tools/r6/cutouts_bench.py 2>/dev/null | tail -1 | tee $O/run11_cutouts.txt
B="python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt-dtype --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "%.2f ms loss %.5f" % (d["ms_per_step"], d["final_loss"]))'
$B > /dev/null 2>&1
for rep in 1 2 3; do
  $B 2>/dev/null | tail -1 | python -c "$P" "fused      " | tee -a $O/run11_step_ab.txt
  FFVC_VQ_FUSE=0 $B 2>/dev/null | tail -1 | python -c "$P" "two launches" | tee -a $O/run11_step_ab.txt
done
