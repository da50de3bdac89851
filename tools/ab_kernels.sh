# usage: bash tools/ab_kernels.sh "<command>" [lib ...]  -- runs the command under rocprofv3 with the default library and with
# each variant library (default: openess_amd/liboess_b.so, see tools/build_variant.sh) on the same box, and prints the
# average duration of the top kernels of each run (PAT=<regex> in the environment: the kernels whose name matches instead)
export TMPDIR=/tmp
cmd=$1; shift
libs=("$@"); [ ${#libs[@]} -eq 0 ] && libs=(openess_amd/liboess_b.so)
for lib in "" "${libs[@]}"; do
  d=/tmp/abk_$$_$(basename "${lib:-default}")
  OESS_LIB_PATH=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- $cmd > /dev/null 2>&1
  python - "$d" "${lib:-default}" <<'PY'
import csv, os, re, sys
rows = list(csv.DictReader(open(sys.argv[1] + "/p_kernel_stats.csv")))
pat = os.environ.get("PAT")
rows = [r for r in rows if re.search(pat, r["Name"])] if pat else rows[:4]
print(sys.argv[2], [(r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:28], round(float(r["AverageNs"]) / 1e3, 1)) for r in rows])
PY
done
