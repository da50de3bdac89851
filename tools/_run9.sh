cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r06_gpu_tests.log
timeout 900 python bench.py > gpurun_out/r06_bench_a.json 2> gpurun_out/r06_bench_a.err
cat gpurun_out/r06_gpu_tests.log; tail -c 3000 gpurun_out/r06_bench_a.json; tail -5 gpurun_out/r06_bench_a.err
