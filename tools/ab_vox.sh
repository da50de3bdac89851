export TMPDIR=/tmp
for lib in "" openess_amd/liboess_b.so; do
  OESS_LIB_PATH=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$$_${lib##*/} -o p -- python tools/bench_voxelizer.py --raw 1 --iters 30 > /dev/null 2>&1
  python - <<PY
import csv,glob
f=glob.glob("/tmp/ab_$$_${lib##*/}/p_kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))[:2]
print("lib=${lib:-default}", [(r["Name"][27:44], round(float(r["AverageNs"])/1e3,1)) for r in rows])
PY
done
