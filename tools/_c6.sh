cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout 200 python $R/tools/chain_ablation.py > $O/c6_ablation.log 2>&1
rm -rf $O/c6_prof
timeout 300 rocprofv3 --kernel-trace -d $O/c6_prof -o bench -- python $R/bench.py --no-cpu-baseline --lanes 1 > $O/c6_prof_bench.log 2>&1
cat $O/c6_ablation.log
DB=$(find $O/c6_prof -name "*.db" | head -1); echo $DB
cd $R && python tools/rocprof_db_summary.py $DB > $O/c6_kernel_summary.md 2>&1; head -45 $O/c6_kernel_summary.md
find $O/c6_prof -type f -size +8M -delete
tail -c 300 $O/c6_prof_bench.log
