cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_gemm.log; : > $O
timeout 900 python -m pytest tests/test_hip_conv.py -x -q -k "conv1x1_w128" 2>&1 | tail -12 >> $O
timeout 600 python tools/bench_conv1x1.py >> $O 2>&1
cat $O
