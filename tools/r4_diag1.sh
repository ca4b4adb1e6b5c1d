#!/bin/bash
# round-4 diagnostics, call 1: baseline line on this box, two-stream timeline of the step, TN tile-order sweep
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-dtype --gemm-shapes 45 2> $O/d1_bench.err | tail -1 > $O/d1_bench.json
grep "^#" $O/d1_bench.err > $O/d1_shapes.txt
for g in 1 2 4 8 16; do FFVC_TILE_GM=$g python3 tools/tn_gm.py f16 2>&1 | grep "^gm"; done > $O/d1_tn_gm.txt
python3 tools/tn_gm.py f16 2>&1 | grep "^gm" >> $O/d1_tn_gm.txt
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace -d /tmp/prof_d1 -o kt -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-dtype --no-roofline > $O/d1_prof_bench.json 2> $O/d1_prof.err )
DB=$(find /tmp/prof_d1 -name "*.db" | head -1)
python3 tools/rocpd_timeline.py $DB --last-ms 400 > $O/d1_timeline.txt 2>&1
python3 -c "import json; d=json.load(open('$O/d1_bench.json')); print(d['ms_per_step'], d['kernel_classes'], d['hbm_kernels'])"
