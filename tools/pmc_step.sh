# PMC traffic passes over one bench step (run on the GPU box: bash tools/pmc_step.sh)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > gpurun_out/pmc_traffic.json   # copy to profiles/r02_pmc_traffic.json
python tools/pmc_summary.py /tmp/pmc_FETCH_SIZE > gpurun_out/pmc_FETCH_SIZE.txt 2>&1
python tools/pmc_summary.py /tmp/pmc_WRITE_SIZE > gpurun_out/pmc_WRITE_SIZE.txt 2>&1
cat gpurun_out/pmc_traffic.json
# matrix-pipe utilisation of the step (own pass: counters only, --kernel-trace, no other trace domains)
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_mfma -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/pmc_mfma.log 2>&1
cd $GRAFT_REPO_ROOT
MS=$(grep '^{' /tmp/pmc_FETCH_SIZE.log | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null || echo 150)
python tools/pmc_mfma.py /tmp/pmc_mfma 2 ${FFVC_STEP_MS:-147} > gpurun_out/pmc_mfma_busy.txt 2>&1   # warmup + timed step profiled
head -30 gpurun_out/pmc_mfma_busy.txt
