cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_e1block.log; : > $O
for lib in liboess_W128_ABL_8192.so liboess_W128_ABL_8192_e2.so liboess_W128_ABL_8192_e4.so; do
echo "== $lib" >> $O
OESS_LIB_PATH=openess_amd/$lib timeout 300 python tools/bench_lstm_group.py --modes 4 --stamps 2>&1 | grep -A2 "w128 stamps" | tail -2 >> $O
done
for rep in 1 2; do for lib in liboess.so liboess_W128_ABL_0_e2.so liboess_W128_ABL_0_e4.so; do
echo -n "$lib " >> $O
OESS_LIB_PATH=openess_amd/$lib timeout 300 python tools/bench_lstm_group.py --modes 4 --rounds 3 2>&1 | grep "us median" >> $O
done; done
cat $O
