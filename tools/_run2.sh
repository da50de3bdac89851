cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/w128_abl.log; : > $O
for v in "" 1 2 3 7 15; do
  if [ -z "$v" ]; then lib=""; else lib=openess_amd/liboess_W128_ABL_$v.so; fi
  echo "== ABL ${v:-0}" >> $O
  OESS_LIB_PATH=$lib timeout 200 python tools/bench_lstm_group.py --modes 3 --rounds 2 2>&1 | grep "us median" >> $O
done
echo "== ABL 3/7/15 zero data" >> $O
for v in 3 15; do OESS_LIB_PATH=openess_amd/liboess_W128_ABL_$v.so timeout 200 python tools/bench_lstm_group.py --modes 3 --rounds 2 --zero 2>&1 | grep "us median" >> $O; done
# PMC passes on the full kernel (mode 3) and the old one (mode 1)
cd /tmp
for m in 1 3; do
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmcA_$m -- python $GRAFT_REPO_ROOT/tools/bench_lstm_group.py --modes $m --rounds 1 --iters 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/pmcB_$m -- python $GRAFT_REPO_ROOT/tools/bench_lstm_group.py --modes $m --rounds 1 --iters 5 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
for m in 1 3; do
echo "== PMC mode $m" >> $O
python tools/mfma_util.py /tmp/pmcA_$m >> $O 2>&1
python tools/pmc_parse.py /tmp/pmcB_$m "conv3x3_[a-z0-9_]*group_kernel" >> $O 2>&1
done
cat $O
