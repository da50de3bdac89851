cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/w128_abl3.log; : > $O
OESS_LIB_PATH=openess_amd/liboess_W128_ABL_4096.so OESS_LSTM256=3 timeout 600 python -m pytest tests/test_hip_conv.py -x -q -k "convlstm_fused_group" 2>&1 | tail -3 >> $O
for v in "" 1024 2048 3072 4096 32; do
  if [ -z "$v" ]; then lib=""; else lib=openess_amd/liboess_W128_ABL_$v.so; fi
  echo -n "ABL ${v:-0}: " >> $O
  OESS_LIB_PATH=$lib timeout 200 python tools/bench_lstm_group.py --modes 3 --rounds 3 2>&1 | grep "us median" >> $O
done
cat $O
