cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_misc1.log; : > $O
OESS_LIB_PATH=openess_amd/liboess_W128_ABL_8192.so timeout 300 python tools/bench_lstm_group.py --modes 4 --stamps 2>&1 | grep -A2 "w128 stamps" | tail -3 >> $O
timeout 900 python -m pytest tests/test_hip_nets.py -x -q -k "full_size" 2>&1 | tail -5 >> $O
timeout 900 python bench.py --steps 40 --no-extras --no-pmc > gpurun_out/r06_bench_b.json 2> gpurun_out/r06_bench_b.err; tail -3 gpurun_out/r06_bench_b.err >> $O
cat $O; tail -c 2500 gpurun_out/r06_bench_b.json
