cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/stage
for st in deeplab_fwd maskclip_fwd teacher_fwd; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stage/$st -o p -- python tools/bench_stage.py $st --iters 10 > gpurun_out/stage/$st.txt 2>&1
  f=$(find gpurun_out/stage/$st -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/stage/${st}_kernel_stats.csv
  find gpurun_out/stage/$st -name "*kernel_trace.csv" -delete
done
tail -2 gpurun_out/stage/*.txt
