cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_sched.log; : > $O
OESS_LIB_PATH=openess_amd/liboess_W128_ABL_8192.so timeout 300 python tools/bench_lstm_group.py --modes 4 --stamps 2>&1 | grep -A2 "w128 stamps" | tail -2 >> $O
for rep in 1 2; do timeout 300 python tools/bench_lstm_group.py --modes 1,4 --rounds 3 2>&1 | grep "us median" >> $O; done
cat $O
