# usage: bash tools/pmc_kernels.sh "<command>" <kernel-name-regex>   (on the GPU box)
# Four rocprofv3 --pmc passes (--kernel-trace only) over the command; prints per-launch averages of each counter for matching kernels.
export TMPDIR=/tmp
CMD=$1; PAT=$2
SETS=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
      "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_BRANCH"
      "SQ_IFETCH SQ_WAIT_ANY SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_SMEM SQ_LEVEL_WAVES")
i=0
for S in "${SETS[@]}"; do
  rocprofv3 --pmc $S --kernel-trace --output-format csv -d /tmp/pmck_$$_$i -o p -- $CMD > /dev/null 2>&1
  i=$((i+1))
done
python - "$PAT" /tmp/pmck_$$_* <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
pat = re.compile(sys.argv[1])
acc, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
for d in sys.argv[2:]:
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            k = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0]
            if pat.search(k):
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in acc:
    print("==", k)
    for c in sorted(acc[k]):
        print(f"   {c:28s} {acc[k][c] / max(cnt[k][c], 1):16.0f} per launch ({cnt[k][c]} launches)")
PY
