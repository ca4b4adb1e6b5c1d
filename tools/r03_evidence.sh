#!/bin/bash
# Round-3 evidence from ONE library build (run on the GPU box: bash tools/r03_evidence.sh).  Everything lands in
# gpurun_out/r03/ with the library stamp; copy what is to be judged into profiles/r03_*.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
# 1. the bench line (cpu baseline, both 16-bit formats, roofline with the per-launch events)
python3 bench.py --steps 20 --warmup 3 2> $O/bench_line.err | tail -1 > $O/bench_line.json
# 2. isolated per-shape table, tile-rule sweep (previous rule vs the shipped one), in-step shape table
( python3 tools/stamp.py; python3 tools/gemm_bench.py --dtype f16 2>&1 | grep -v amdgpu.ids ) > $O/gemm_shapes.txt
( python3 tools/stamp.py; echo "## FFVC_TILE_RULE=0 (rule of rounds 1-2: largest tile that fills whole rounds)"; FFVC_TILE_RULE=0 python3 tools/gemm_tile_sweep.py 2>&1 | grep "^NT";
  echo "## shipped rule (256-row tiles once 96 of them exist)"; python3 tools/gemm_tile_sweep.py 2>&1 | grep "^NT\|^conv" ) > $O/gemm_tile_sweep.txt
( python3 tools/stamp.py; python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alt-dtype --gemm-shapes 40 2>&1 >/dev/null | grep "^#" ) > $O/gemm_shapes_instep.txt
# 3. the other configurations
bash tools/bench_configs.sh > $O/bench_configs.txt 2>&1
for f in cfg3 cfg4 cfg5_f16 cfg5_fp8; do cp gpurun_out/r03_bench_$f.json $O/bench_$f.json; done
# 4. profiler passes (kernel trace, HBM traffic, MFMA busy)
bash tools/pmc_step.sh > $O/pmc_step.log 2>&1
tail -3 $O/bench_configs.txt
python3 -c "import json; d=json.load(open('$O/bench_line.json')); print(d['ms_per_step'], d['value'], d['alt_dtype'], d['roofline']['achieved'])"
