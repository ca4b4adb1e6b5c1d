# Profiles of one bench step for profiles/${FFVC_ROUND:-r04}_* (run on the GPU box: bash tools/pmc_step.sh).  Every output carries the library
# stamp (tools/stamp.py).  Counter passes are their own runs: --kernel-trace + --pmc only (no other trace domains).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${FFVC_ROUND:-r04}
mkdir -p $O
B="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-alt-dtype"
# 1. kernel trace (per-kernel time) of 5 steps
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-alt-dtype > /tmp/kt.log 2>&1
( python3 $R/tools/stamp.py; python3 $R/tools/rocpd_summary.py $(ls /tmp/prof_kt/*/*_results.db /tmp/prof_kt/*_results.db 2>/dev/null | head -1) --steps 7 --top 70 ) > $O/kernel_trace_bench_cfg2.txt 2>&1
grep '^{' /tmp/kt.log | tail -1 > $O/bench_line_under_profiler.json
# 2. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- $B > /tmp/pmc_$c.log 2>&1
done
python3 $R/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $O/pmc_traffic.json
( python3 $R/tools/stamp.py; python3 $R/tools/pmc_summary.py /tmp/pmc_FETCH_SIZE ) > $O/pmc_fetch_size_bench_step.txt 2>&1
( python3 $R/tools/stamp.py; python3 $R/tools/pmc_summary.py /tmp/pmc_WRITE_SIZE ) > $O/pmc_write_size_bench_step.txt 2>&1
# 3. matrix-pipe utilisation of the step
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_mfma -o p -- $B > /tmp/pmc_mfma.log 2>&1
MS=$(grep '^{' /tmp/pmc_mfma.log | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null || echo 0)
( python3 $R/tools/stamp.py; echo "# step time used for the step-level line: ${FFVC_STEP_MS:-$MS} ms (ms_per_step of THIS profiled run unless FFVC_STEP_MS is set; profiled runs clock ~3 % lower)"; python3 $R/tools/pmc_mfma.py /tmp/pmc_mfma 2 ${FFVC_STEP_MS:-$MS} ) > $O/pmc_mfma_busy.txt 2>&1
head -40 $O/pmc_mfma_busy.txt
