#!/bin/bash
# rocprofv3 kernel traces of the other BASELINE configurations (cfg3 / cfg4 / cfg5 with the fp8 tower) -> gpurun_out/${FFVC_ROUND:-r04}/kernel_trace_<cfg>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${FFVC_ROUND:-r04}
mkdir -p $O
C="--steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-alt-dtype"
run() {   # name, bench args...
  name=$1; shift
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o kt -- python3 $R/bench.py $C "$@" > /tmp/kt_$name.log 2>&1
  ( python3 $R/tools/stamp.py; python3 $R/tools/rocpd_summary.py $(ls /tmp/prof_$name/*/*_results.db /tmp/prof_$name/*_results.db 2>/dev/null | head -1) --steps 5 --top 45 ) > $O/kernel_trace_$name.txt 2>&1
  head -4 $O/kernel_trace_$name.txt | cut -c1-160
}
if [ -z "$FFVC_TRACE_ONLY_CFG5" ]; then
run cfg3 --model-type vitgan --batch 32
run cfg4 --model-type xtransformer --dim 256 --depth 16 --vq-image-size 32 --batch 16
fi
run cfg5_fp8dec --depth 1 --vq-image-size 32 --batch 8 --clip-model openclip/ViT-L-14/laion2b_s32b_b82k --clip-fp8 --dec-fp8
run cfg5_fp8 --depth 1 --vq-image-size 32 --batch 8 --clip-model openclip/ViT-L-14/laion2b_s32b_b82k --clip-fp8
