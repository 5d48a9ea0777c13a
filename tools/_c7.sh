cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout 200 python $R/tools/chain_ablation.py 2>&1 | tail -12 > $O/c7_ablation.log
rm -rf $O/pmc_f $O/pmc_w
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python $R/tools/pmc_chain.py > $O/c7_pmc_f.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python $R/tools/pmc_chain.py > $O/c7_pmc_w.log 2>&1
cd $R && python tools/pmc_summary.py $O/pmc_f $O/pmc_w > $O/c7_pmc_summary.md
cat $O/c7_ablation.log $O/c7_pmc_summary.md
find $O/pmc_f $O/pmc_w -name "*.csv" -size +3M -delete
