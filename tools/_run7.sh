cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_conv.py -x -q -k "w128" 2>&1 | tail -8 > gpurun_out/w128v2_test.log
OESS_LIB_PATH=openess_amd/liboess_W128_ABL_8192.so timeout 300 python tools/bench_lstm_group.py --modes 4 --stamps > gpurun_out/w128v2_stamps.log 2>&1
cat gpurun_out/w128v2_test.log gpurun_out/w128v2_stamps.log
