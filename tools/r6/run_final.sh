#!/bin/bash
# round 6, final evidence from the final library: every profiles/r06_* file that carries the library stamp, then the whole GPU suite
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/r06 gpurun_out/r6
bash tools/r06_evidence.sh > gpurun_out/r6/final_evidence.log 2>&1
cp gpurun_out/r06/pmc_traffic.json profiles/r06_pmc_traffic.json      # (on the box's copy: the line's roofline.traffic wants the passes of THIS library)
bash tools/r6/run_line.sh > gpurun_out/r6/final_line.log 2>&1
python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|^E  |^FAILED" | head -20 | tee gpurun_out/r6/final_pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 1 | tee -a gpurun_out/r6/final_pytest.txt
tail -n 3 gpurun_out/r6/final_line.log | cut -c1-400
