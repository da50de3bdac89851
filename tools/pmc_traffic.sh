# usage: bash tools/pmc_traffic.sh "<command>" <kernel-name-regex>   (on the GPU box)
# HBM bytes per launch of the matching kernels: separate FETCH_SIZE / WRITE_SIZE rocprofv3 passes (counters in KiB; on gfx950
# FETCH_SIZE reports half of a 16 B/lane coalesced read stream, see the guide's HBM section - the raw value is printed).
export TMPDIR=/tmp
CMD=$1; PAT=$2
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pt_$$_$C -o p -- $CMD > /dev/null 2>&1
done
python - "$PAT" /tmp/pt_$$_* <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
pat = re.compile(sys.argv[1])
acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[2:]:
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            k = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0]
            if pat.search(k):
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: round(sum(x) / len(x) * 1024 / 1e6, 1) for c, x in v.items()}, "MB per launch (raw counter values)")
PY
