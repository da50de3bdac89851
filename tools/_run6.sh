cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_conv.py -x -q -k "w128" 2>&1 | tail -15 > gpurun_out/w128v2_test.log
timeout 300 python tools/bench_lstm_group.py --modes 1,4 --rounds 3 > gpurun_out/w128v2_bench.log 2>&1
cat gpurun_out/w128v2_test.log gpurun_out/w128v2_bench.log
