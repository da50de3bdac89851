cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/w128_abl2.log; : > $O
for rep in 1 2; do
for v in "" 1 16 32 64 2 128 256 512 896; do
  if [ -z "$v" ]; then lib=""; else lib=openess_amd/liboess_W128_ABL_$v.so; fi
  echo -n "ABL ${v:-0}: " >> $O
  OESS_LIB_PATH=$lib timeout 200 python tools/bench_lstm_group.py --modes 3 --rounds 3 2>&1 | grep "us median" >> $O
done
done
cat $O
