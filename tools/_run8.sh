cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/w128v2_stamps2.log; : > $O
for lib in liboess_W128_ABL_8192.so liboess_W128_ABL_8192_early.so; do
echo "== $lib" >> $O
OESS_LIB_PATH=openess_amd/$lib timeout 300 python tools/bench_lstm_group.py --modes 4 --stamps 2>&1 | grep -A1 "w128 stamps" | tail -2 >> $O
done
for lib in liboess.so liboess_W128_ABL_0_early.so liboess.so liboess_W128_ABL_0_early.so; do
echo -n "$lib " >> $O
OESS_LIB_PATH=openess_amd/$lib timeout 300 python tools/bench_lstm_group.py --modes 4 --rounds 3 2>&1 | grep "us median" >> $O
done
OESS_LIB_PATH=openess_amd/liboess_W128_ABL_0_early.so timeout 600 python -m pytest tests/test_hip_conv.py -x -q -k "w128" 2>&1 | tail -2 >> $O
cat $O
