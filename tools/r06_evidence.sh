#!/bin/bash
# Round-6 evidence from ONE library build (run on the GPU box: bash tools/r06_evidence.sh).  Everything lands in
# gpurun_out/r06/ with the library stamp; copy what is to be judged into profiles/r06_*.
export FFVC_ROUND=r06
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
# 1. the bench line (cpu baseline, both 16-bit formats, roofline with the per-launch events and the top-5 shapes)
python3 bench.py --steps 20 --warmup 4 --isolated-table $O/isolated_sum.txt 2> $O/bench_line.err | tail -1 > $O/bench_line.json
( python3 tools/stamp.py; cat $O/isolated_sum.txt ) > $O/isolated_sum_stamped.txt
# 2. in-step shape table, isolated per-shape table
( python3 tools/stamp.py; python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alt-dtype --no-attainable --gemm-shapes 40 2>&1 >/dev/null | grep "^#" ) > $O/gemm_shapes_instep.txt
( python3 tools/stamp.py; python3 tools/gemm_bench.py --dtype f16 2>&1 | grep -v amdgpu.ids ) > $O/gemm_shapes.txt
# 3. the other configurations
bash tools/bench_configs.sh > $O/bench_configs.txt 2>&1
for f in cfg3 cfg4 cfg5_f16 cfg5_fp8 cfg5_fp8dec; do cp gpurun_out/r06_bench_$f.json $O/bench_$f.json; done
# 4. profiler passes (kernel trace, HBM traffic, MFMA busy)
bash tools/pmc_step.sh > $O/pmc_step.log 2>&1
# 5. two-stream timeline of the step
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /tmp/prof_tl -o kt -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-dtype --no-roofline > /tmp/tl.log 2>&1 )
( python3 tools/stamp.py; python3 tools/rocpd_timeline.py $(find /tmp/prof_tl -name "*.db" | head -1) --last-ms 400 ) > $O/two_stream_timeline.txt 2>&1
tail -3 $O/bench_configs.txt
# 6. cfg3 kernel trace, data-parallel overlap table, small-M GEMM table
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_cfg3 -o kt -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-alt-dtype --model-type vitgan --batch 32 > /tmp/kt3.log 2>&1 )
( python3 tools/stamp.py; python3 tools/rocpd_summary.py $(ls /tmp/prof_cfg3/*/*_results.db /tmp/prof_cfg3/*_results.db 2>/dev/null | head -1) --steps 7 --top 60 ) > $O/kernel_trace_cfg3.txt 2>&1
( python3 tools/stamp.py; FFVC_DP_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python3 tools/dp_overlap.py mlp_mixer 4 2>&1 | grep -v -E "amdgpu.ids|^RCCL|^HIP|^ROCm|^Hostname|^Librccl|Warning" ) > $O/dp_overlap.txt
# 6b. cfg5: the captured step, and a two-stream timeline of the eager one
C5="--depth 1 --vq-image-size 32 --batch 8 --clip-model openclip/ViT-L-14/laion2b_s32b_b82k --steps 6 --warmup 2 --no-cpu-baseline --no-alt-dtype --no-roofline"
python3 bench.py $C5 2>/dev/null | tail -1 > $O/bench_cfg5_f16_eager_short.json
python3 bench.py $C5 --graph 2>$O/bench_cfg5_f16_graph.err | tail -1 > $O/bench_cfg5_f16_graph.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /tmp/prof_tl5 -o kt -- python3 $R/bench.py $C5 > /tmp/tl5.log 2>&1 )
( python3 tools/stamp.py; python3 tools/rocpd_timeline.py $(find /tmp/prof_tl5 -name "*.db" | head -1) --last-ms 300 ) > $O/two_stream_timeline_cfg5.txt 2>&1
# 6c. round-6 kernel tables: convolutions (conv3 on / off), GroupNorm-backward fusion
( python3 tools/stamp.py; echo "== conv3 on"; python3 tools/gemm_bench.py --dtype f16 --only conv --batch 64 2>/dev/null | grep conv; echo "== FFVC_CONV_ROW3=0"; FFVC_CONV_ROW3=0 python3 tools/gemm_bench.py --dtype f16 --only conv --batch 64 2>/dev/null | grep conv ) > $O/conv_shapes.txt
( python3 tools/stamp.py; python3 tools/r6/gnb_bench.py 2>&1 | grep -v amdgpu ) > $O/gnb_bench.txt
# 7. the bare multi-rank launch on one shared GPU (gloo exchange): the launch contract of `python bench.py --gpus 2`
FFVC_DP_BACKEND=gloo FFVC_SHARE_DEVICE=1 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-alt-dtype > $O/dp_bench_2rank_shared_device.log 2>&1
python3 -c "import json; d=json.load(open('$O/bench_line.json')); print(d['ms_per_step'], d['value'], d['alt_dtype'], d['roofline']['achieved'], d['roofline'].get('attainable_ms'), d['roofline'].get('frac_of_attainable'))"
