cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/pmc_a $O/pmc_b
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmc_a -- python $R/tools/pmc_chain.py > $O/c4_pmc_a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $O/pmc_b -- python $R/tools/pmc_chain.py > $O/c4_pmc_b.log 2>&1
cd $R && python tools/pmc_summary.py $O/pmc_a $O/pmc_b > $O/c4_pmc_summary.md
cat $O/c4_pmc_summary.md
find $O/pmc_a $O/pmc_b -name "*.csv" -size +3M -delete
