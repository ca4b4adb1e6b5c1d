#!/bin/bash
# A/B of environment switches on ONE box: usage  bash tools/ab.sh <tag> "VAR=1 VAR2=0" "..." ; each variant = one bench.py run
# (cfg2, 10 steps after 3 warm-ups, no CPU baseline / alt dtype / roofline); prints ms/step per variant, twice (A B A B order).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
tag=$1; shift
: > $O/ab_$tag.txt
for rep in 1 2; do
  for v in "$@"; do
    ms=$(env $v python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-dtype --no-roofline $AB_ARGS 2>$O/ab_$tag.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f loss %.5f' % (d['ms_per_step'], d['final_loss']))")
    echo "rep$rep [$v] $ms" | tee -a $O/ab_$tag.txt
  done
done
