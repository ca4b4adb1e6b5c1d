#!/bin/bash
# round 6, GPU call 4: C = NT GEMMs on the LDS-pad epilogue, convolutions on the register exchange, conv3 on; what bounds conv3's K step
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
LB=$R/feed_forward_vqgan_clip_amd/lib/libffvc_hip_b.so
python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | tail -3 | tee $O/run4_pytest.txt
python -m pytest tests/test_fullsize_gpu.py -x -q -s -k "gradients or cfg3_cfg4_full_size_f16" 2>&1 | grep -E "^\[|passed|failed|Error|assert" | tee -a $O/run4_pytest.txt
for e in 0 1 2 3 4 8 15; do echo "== FFVC_C3_EXP=$e"; FFVC_C3_EXP=$e python tools/gemm_bench.py --dtype f16 --only conv --batch 64 2>/dev/null | grep "256^2"; done | tee $O/run4_conv3_exp.txt
echo "== FFVC_CONV_ROW=2 (row-tile kernels also for the 256 / 512-channel levels)" | tee $O/run4_conv_row2.txt
FFVC_CONV_ROW=2 python tools/gemm_bench.py --dtype f16 --only conv --batch 64 2>/dev/null | grep conv | tee -a $O/run4_conv_row2.txt
echo "== default" | tee -a $O/run4_conv_row2.txt
python tools/gemm_bench.py --dtype f16 --only conv --batch 64 2>/dev/null | grep conv | tee -a $O/run4_conv_row2.txt
B="python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt-dtype --no-roofline"
for rep in 1 2 3; do
  $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C default    %.2f ms loss %.5f ovf %s' % (d['ms_per_step'], d['final_loss'], d.get('overflow_steps')))" | tee -a $O/run4_step_ab.txt
  FFVC_CONV_ROW=2 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C conv_row=2 %.2f ms loss %.5f' % (d['ms_per_step'], d['final_loss']))" | tee -a $O/run4_step_ab.txt
  FFVC_LIB=$LB $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B (r5 code)  %.2f ms loss %.5f' % (d['ms_per_step'], d['final_loss']))" | tee -a $O/run4_step_ab.txt
done
