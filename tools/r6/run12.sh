#!/bin/bash
# round 6, GPU call 12: how often does a step stall?  100-step lines, per-step times, with and without the small-request pool reservation
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
B="python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-alt-dtype --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); s=d["step_ms_main_stream"]; import statistics as st; m=st.median(s); print(sys.argv[1], "mean %.2f median %.2f max %.2f; steps > median + 5 ms:" % (d["ms_per_step"], m, max(s)), [(i, round(v,1)) for i,v in enumerate(s) if v > m + 5], d.get("allocator_in_timed_region",{}).get("new_segments"))'
for rep in 1 2; do
  $B 2>/dev/null | tail -1 | python -c "$P" "default        " | tee -a $O/run12_stalls.txt
  FFVC_RESERVE_SMALL_MIB=0 $B 2>/dev/null | tail -1 | python -c "$P" "no small reserve" | tee -a $O/run12_stalls.txt
  FFVC_SUMS_POOL=0 $B 2>/dev/null | tail -1 | python -c "$P" "no sums pool    " | tee -a $O/run12_stalls.txt
done
