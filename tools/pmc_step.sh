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
