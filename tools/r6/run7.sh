#!/bin/bash
# round 6, GPU call 7: register-exchange epilogue with a four-deep side-input prefetch on the residual / multiply kinds of the
# K-major x K-major launches (library "pk") vs the padded epilogue (default library)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
LB=$R/feed_forward_vqgan_clip_amd/lib/libffvc_hip_pk.so
FFVC_LIB=$LB python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error" | tee $O/run7_pytest.txt
python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error" | tee -a $O/run7_pytest.txt
for rep in 1 2; do
  python tools/g3_bench.py --modes=-1 2>/dev/null | grep -v "^/opt\|device" | sed 's/^/A /' | tee -a $O/run7_g3.txt
  FFVC_LIB=$LB python tools/g3_bench.py --modes=-1 2>/dev/null | grep -v "^/opt\|device" | sed 's/^/PK /' | tee -a $O/run7_g3.txt
done
B="python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt-dtype --no-roofline"
$B > /dev/null 2>&1
for rep in 1 2 3; do
  $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('A  %.2f ms loss %.5f' % (d['ms_per_step'], d['final_loss']))" | tee -a $O/run7_step_ab.txt
  FFVC_LIB=$LB $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PK %.2f ms loss %.5f' % (d['ms_per_step'], d['final_loss']))" | tee -a $O/run7_step_ab.txt
done
