cd /tmp && export TMPDIR=/tmp
export MNK=8192,8192,8192 FFVC_GEMM2_BM=512
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $GRAFT_REPO_ROOT/gpurun_out/sq_counters.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmc1 -- python3 $GRAFT_REPO_ROOT/tools/gemm_one.py nt 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES --output-format csv -d /tmp/pmc2 -- python3 $GRAFT_REPO_ROOT/tools/gemm_one.py nt 5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py /tmp/pmc1 gemm2 > gpurun_out/pmc_sq1.txt 2>&1
python tools/pmc_summary.py /tmp/pmc2 gemm2 > gpurun_out/pmc_sq2.txt 2>&1
cat gpurun_out/pmc_sq1.txt gpurun_out/pmc_sq2.txt
