cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_gemm2.log; : > $O
timeout 900 python -m pytest tests/test_hip_conv.py -x -q -k "conv1x1_w128" 2>&1 | tail -3 >> $O
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 >> $O
for rep in 1 2; do for g in 1 0; do
  echo -n "OESS_W128_GEMM=$g: " >> $O
  OESS_W128_GEMM=$g timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'].get('serial_event_frames_per_s'), d['roofline']['frac'], d['roofline'].get('family_frac'))" >> $O
done; done
for g in 1 0; do echo "teacher_fwd OESS_W128_GEMM=$g" >> $O; OESS_W128_GEMM=$g timeout 300 python tools/bench_stage.py teacher_fwd --iters 20 2>&1 | tail -2 >> $O; done
cat $O
