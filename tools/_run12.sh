cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_misc2.log; : > $O
timeout 600 python tools/bench_host_pools.py --json >> $O 2>&1
timeout 600 python tools/bench_host_pools.py --json --no-copy >> $O 2>&1
timeout 600 python tools/bench_host_pools.py --json --pools 1 >> $O 2>&1
timeout 900 python -m pytest tests/test_ring_loader.py tests/test_hip_nets.py -x -q -m gpu -k "ring or full_size" 2>&1 | tail -3 >> $O
cat $O
