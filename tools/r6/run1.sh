#!/bin/bash
# round 6, GPU call 1: the register-exchange epilogue and the two-workgroups-per-CU kernel — correctness first, then isolated timings,
# then the step.  A = lib/libffvc_hip.so (perm epilogue + gemm3), B = lib/libffvc_hip_b.so (LDS-pad epilogue, no gemm3).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | tail -15 > $O/run1_pytest_gemm.txt
cat $O/run1_pytest_gemm.txt
python tools/g3_bench.py --modes=-1,1 > $O/run1_g3_bench_A.txt 2>&1
FFVC_G3_STAGGER=0 python tools/g3_bench.py --modes=1 > $O/run1_g3_bench_A_nostagger.txt 2>&1
FFVC_LIB=$R/feed_forward_vqgan_clip_amd/lib/libffvc_hip_b.so python tools/g3_bench.py --modes=-1 > $O/run1_g3_bench_B.txt 2>&1
tail -45 $O/run1_g3_bench_A.txt; tail -42 $O/run1_g3_bench_A_nostagger.txt; tail -42 $O/run1_g3_bench_B.txt
B="python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt-dtype --no-roofline"
for rep in 1 2; do
  $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('A g3=heur %.2f ms loss %.5f ovf %s' % (d['ms_per_step'], d['final_loss'], d.get('overflow_steps')))" | tee -a $O/run1_step_ab.txt
  FFVC_GEMM3=-1 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('A g3=off  %.2f ms loss %.5f' % (d['ms_per_step'], d['final_loss']))" | tee -a $O/run1_step_ab.txt
  FFVC_LIB=$R/feed_forward_vqgan_clip_amd/lib/libffvc_hip_b.so $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B (pads)  %.2f ms loss %.5f' % (d['ms_per_step'], d['final_loss']))" | tee -a $O/run1_step_ab.txt
done
