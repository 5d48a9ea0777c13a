cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/c16_prof
timeout 100 rocprofv3 --kernel-trace -d $O/c16_prof -o bench -- python $R/bench.py --no-cpu-baseline --lanes 1 --steps 2 --warmup 1 > $O/c16_prof_bench.log 2>&1
DB=$(find $O/c16_prof -name "*.db" | head -1)
cd $R && python tools/rocprof_db_summary.py $DB > $O/c16_kernel_summary.md 2>&1
find $O/c16_prof -type f -size +8M -delete
head -30 $O/c16_kernel_summary.md; grep "^{" $O/c16_prof_bench.log | tail -1 | cut -c1-200
