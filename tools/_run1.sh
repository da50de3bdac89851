set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OESS_LSTM256=3 timeout 600 python -m pytest tests/test_hip_conv.py -x -q -k "convlstm" 2>&1 | tail -15 > gpurun_out/w128_test.log
timeout 300 python tools/bench_lstm_group.py --modes 1,0,3 --rounds 3 > gpurun_out/w128_bench.log 2>&1
timeout 200 python tools/bench_lstm_group.py --modes 1,3 --rounds 2 --zero > gpurun_out/w128_bench_zero.log 2>&1
cat gpurun_out/w128_test.log gpurun_out/w128_bench.log gpurun_out/w128_bench_zero.log
