cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_grid_ab.log; : > $O
for rep in 1 2; do for g in 256 240 224 208; do
  echo -n "grid $g: " >> $O
  OESS_W128_GRID=$g timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'].get('serial_event_frames_per_s'), d['roofline']['frac'])" >> $O
done; done
cat $O
