#!/bin/bash
# round 6, GPU call 5: the full bench line with the hardware-bound table, and a kernel trace of the step
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
python3 bench.py --steps 20 --warmup 4 --isolated-table $O/isolated_sum.txt 2> $O/bench_line.err | tail -1 > $O/bench_line.json
python3 -c "
import json; d=json.load(open('$O/bench_line.json'))
print('ms', d['ms_per_step'], 'img/s', d['value'], 'ovf', d.get('overflow_steps'), 'alt', d.get('alt_dtype',{}).get('ms_per_step'))
r=d['roofline']; print({k:r.get(k) for k in ('kernel','achieved','frac','attainable_ms','frac_of_attainable','hw_bound_ms','step_over_hw_bound','hw_bound_parts_ms','hw_bound_gemm_rate_tflops','attainable_aligned')})
print('gaps', r.get('top_gaps_ms'))
print('grad', d.get('parity_full_size',{}).get('grad_parity'))
print('parity', {k:v for k,v in d.get('parity_full_size',{}).items() if k.startswith('rel_')})
print('cpu', d.get('cpu_baseline'))
"
tail -5 $O/bench_line.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-alt-dtype > /tmp/kt.log 2>&1 )
( python3 tools/stamp.py; python3 tools/rocpd_summary.py $(ls /tmp/prof_kt/*/*_results.db /tmp/prof_kt/*_results.db 2>/dev/null | head -1) --steps 7 --top 70 ) > $O/kernel_trace_bench_cfg2.txt 2>&1
head -45 $O/kernel_trace_bench_cfg2.txt
