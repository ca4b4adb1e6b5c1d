#!/bin/bash
# Round-4 evidence from ONE library build (run on the GPU box: bash tools/r04_evidence.sh).  Everything lands in
# gpurun_out/r04/ with the library stamp; copy what is to be judged into profiles/r04_*.
export FFVC_ROUND=r04
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
# 1. the bench line (cpu baseline, both 16-bit formats, roofline with the per-launch events and the top-5 shapes)
python3 bench.py --steps 20 --warmup 3 2> $O/bench_line.err | tail -1 > $O/bench_line.json
# 2. in-step shape table, isolated per-shape table
( python3 tools/stamp.py; python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alt-dtype --gemm-shapes 40 2>&1 >/dev/null | grep "^#" ) > $O/gemm_shapes_instep.txt
( python3 tools/stamp.py; python3 tools/gemm_bench.py --dtype f16 2>&1 | grep -v amdgpu.ids ) > $O/gemm_shapes.txt
# 3. the other configurations
bash tools/bench_configs.sh > $O/bench_configs.txt 2>&1
for f in cfg3 cfg4 cfg5_f16 cfg5_fp8 cfg5_fp8dec; do cp gpurun_out/r04_bench_$f.json $O/bench_$f.json; done
# 4. profiler passes (kernel trace, HBM traffic, MFMA busy)
bash tools/pmc_step.sh > $O/pmc_step.log 2>&1
# 5. two-stream timeline of the step
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d /tmp/prof_tl -o kt -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-dtype --no-roofline > /tmp/tl.log 2>&1 )
( python3 tools/stamp.py; python3 tools/rocpd_timeline.py $(find /tmp/prof_tl -name "*.db" | head -1) --last-ms 400 ) > $O/two_stream_timeline.txt 2>&1
tail -3 $O/bench_configs.txt
python3 -c "import json; d=json.load(open('$O/bench_line.json')); print(d['ms_per_step'], d['value'], d['alt_dtype'], d['roofline']['achieved'], d['roofline'].get('shapes'))"
