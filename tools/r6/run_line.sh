#!/bin/bash
# the bench line + isolated table again (bench.py changed after tools/r06_evidence.sh ran; the library did not)
export FFVC_ROUND=r06
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
python3 bench.py --steps 20 --warmup 5 --isolated-table $O/isolated_sum.txt 2> $O/bench_line.err | tail -1 > $O/bench_line.json
( python3 tools/stamp.py; cat $O/isolated_sum.txt ) > $O/isolated_sum_stamped.txt
FFVC_DP_BACKEND=gloo FFVC_SHARE_DEVICE=1 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-alt-dtype > $O/dp_bench_2rank_shared_device.log 2>&1
python3 -c "import json; d=json.load(open('$O/bench_line.json')); r=d['roofline']; print(d['ms_per_step'], d['value'], d['alt_dtype']['ms_per_step'], r['achieved'], r['frac'], r.get('traffic'), r.get('attainable_ms'), r.get('hw_bound_ms'), d['allocator_in_timed_region'], d['overflow_steps'], max(d['step_ms_main_stream']))"
tail -2 $O/dp_bench_2rank_shared_device.log | cut -c1-600
