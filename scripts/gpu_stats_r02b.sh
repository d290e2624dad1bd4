mkdir -p gpurun_out/r2y
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2y/prof_stats -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r2y/prof_stats.log 2>&1
cd $R
for f in $(find gpurun_out/r2y/prof_stats -name "*kernel_stats.csv"); do cp $f gpurun_out/r2y/kernel_stats_200M.csv; grep "pg::" $f | sed 's/"[^"]*"/K/'; done
find gpurun_out/r2y -name "*.db" -delete; find gpurun_out/r2y -name "*kernel_trace.csv" -delete; find gpurun_out/r2y -name "*agent_info.csv" -delete
tail -c 600 gpurun_out/r2y/prof_stats.log | grep -o '"ms_per_step": [0-9.]*\|"k2_count_ms_per_step": [0-9.]*'
