# Same-box alternating A/B of the pooled contrastive features (hip.PointwiseFeature / hip.UpsampledNormalizedFeature).
#   usage: bash tools/ab_pooled_features.sh [student|teacher] [workload]
cd /root/repo
which=${1:-student}; wl=${2:-frame2voxel_full}
for i in 1 2 3; do
  for flag in True False; do
    python - $flag $which $wl <<'PY' 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-150
import sys
flag, which, wl = sys.argv[1] == "True", sys.argv[2], sys.argv[3]
from openess_amd.training.pretrain_step import PretrainStep
setattr(PretrainStep, "pooled_%s_features" % which, flag)
import bench
sys.argv = ["bench.py", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-pmc", "--no-extras", "--workload", wl]
print(which, "pooled =", flag, end="  ")
bench.main()
PY
  done
done
