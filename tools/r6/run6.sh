#!/bin/bash
# round 6, GPU call 6: GroupNorm-backward statistics folded into the dgrad convolution (FFVC_GNB_FUSE), token-mix weight-gradient split sweep
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
python -m pytest tests/test_gemm_gpu.py -x -q -k "groupnorm_backward or conv" 2>&1 | tail -5 | tee $O/run6_pytest.txt
python -m pytest tests/test_models_gpu.py tests/test_fullsize_gpu.py -x -q -k "decoder or vqgan or cfg2_full_size or gradients" 2>&1 | tail -5 | tee -a $O/run6_pytest.txt
B="python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt-dtype --no-roofline"
for rep in 1 2 3; do
  $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gnb fuse on   %.2f ms loss %.5f ovf %s' % (d['ms_per_step'], d['final_loss'], d.get('overflow_steps')))" | tee -a $O/run6_step_ab.txt
  FFVC_GNB_FUSE=0 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gnb fuse off  %.2f ms loss %.5f' % (d['ms_per_step'], d['final_loss']))" | tee -a $O/run6_step_ab.txt
done
for sk in 384 256 128; do
  FFVC_SK_TARGET=$sk $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sk_target $sk %.2f ms loss %.5f' % (d['ms_per_step'], d['final_loss']))" | tee -a $O/run6_step_ab.txt
done
