#!/bin/bash
mkdir -p gpurun_out/r5
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o kt -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-alt-dtype --model-type vitgan --batch 32 > /tmp/kt3.log 2>&1
cd $R
grep '^{' /tmp/kt3.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms_main_stream'])" > gpurun_out/r5/cfg3_trace_steps.txt
python3 tools/rocpd_summary.py $(ls /tmp/prof_c3/*/*_results.db /tmp/prof_c3/*_results.db 2>/dev/null | head -1) --steps 15 --top 40 > gpurun_out/r5/cfg3_trace_smallm1.txt 2>&1
python3 tools/rocpd_timeline.py $(ls /tmp/prof_c3/*/*_results.db /tmp/prof_c3/*_results.db 2>/dev/null | head -1) --last-ms 1200 > gpurun_out/r5/cfg3_timeline_smallm1.txt 2>&1
cat gpurun_out/r5/cfg3_trace_steps.txt; sort -k5 -n -r gpurun_out/r5/cfg3_trace_smallm1.txt | head -12 | cut -c1-150
