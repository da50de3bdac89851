cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/w128_stamps.log; : > $O
for v in 8192 8224; do
  echo "== ABL $v" >> $O
  OESS_LIB_PATH=openess_amd/liboess_W128_ABL_$v.so timeout 200 python tools/bench_lstm_group.py --modes 3 --stamps 2>&1 | grep -v "amdgpu.ids" >> $O
  OESS_LIB_PATH=openess_amd/liboess_W128_ABL_$v.so timeout 200 python tools/bench_lstm_group.py --modes 3 --rounds 2 2>&1 | grep "us median" >> $O
done
cat $O
