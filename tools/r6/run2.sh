#!/bin/bash
# round 6, GPU call 2: pipelined row-tile convolution (conv3.hip), hoisted register-exchange epilogue, pipelined LayerNorm forward
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
LB=$R/feed_forward_vqgan_clip_amd/lib/libffvc_hip_b.so
python -m pytest tests/test_gemm_gpu.py tests/test_kernels_gpu.py -x -q 2>&1 | tail -15 | tee $O/run2_pytest.txt
for g in 4096 2048 1024 512 256; do FFVC_LN_FWD_GRID=$g python tools/r6/ln_fwd_bench.py 2>/dev/null | tail -1; done | tee $O/run2_ln_fwd.txt
FFVC_LIB=$LB python tools/r6/ln_fwd_bench.py 2>/dev/null | tail -1 | sed 's/^/B (one row per wave): /' | tee -a $O/run2_ln_fwd.txt
echo "== conv, row3 on" | tee $O/run2_conv.txt; python tools/gemm_bench.py --dtype f16 --only conv --batch 64 2>/dev/null | grep conv | tee -a $O/run2_conv.txt
echo "== conv, row3 off (perm epilogue)" | tee -a $O/run2_conv.txt; FFVC_CONV_ROW3=0 python tools/gemm_bench.py --dtype f16 --only conv --batch 64 2>/dev/null | grep conv | tee -a $O/run2_conv.txt
echo "== conv, B (pads)" | tee -a $O/run2_conv.txt; FFVC_LIB=$LB python tools/gemm_bench.py --dtype f16 --only conv --batch 64 2>/dev/null | grep conv | tee -a $O/run2_conv.txt
python tools/g3_bench.py --modes=-1 > $O/run2_g3_bench_A.txt 2>&1; grep -v "^/opt\|device" $O/run2_g3_bench_A.txt
B="python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt-dtype --no-roofline"
for rep in 1 2; do
  $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('A default   %.2f ms loss %.5f ovf %s' % (d['ms_per_step'], d['final_loss'], d.get('overflow_steps')))" | tee -a $O/run2_step_ab.txt
  FFVC_CONV_ROW3=0 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('A row3=0    %.2f ms loss %.5f' % (d['ms_per_step'], d['final_loss']))" | tee -a $O/run2_step_ab.txt
  FFVC_LN_FWD_GRID=4096 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('A lngrid4096 %.2f ms loss %.5f' % (d['ms_per_step'], d['final_loss']))" | tee -a $O/run2_step_ab.txt
  FFVC_LIB=$LB $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B (r5 code) %.2f ms loss %.5f' % (d['ms_per_step'], d['final_loss']))" | tee -a $O/run2_step_ab.txt
done
