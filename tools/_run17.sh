cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_now; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o step -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-extras --no-overlap-teacher > $O/log.txt 2>&1
find $O -name "*_kernel_trace.csv" -delete
cp $(find $O -name "step_kernel_stats.csv" | head -1) gpurun_out/step_now_kernel_stats.csv
head -30 gpurun_out/step_now_kernel_stats.csv | cut -c1-200
