# Same-box alternating A/B of the pooled student features (hip.PointwiseFeature) on the frame2voxel_full step.
cd /root/repo
for i in 1 2 3; do
  for flag in True False; do
    python - $flag <<'PY' 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-140
import sys
sys.argv = [sys.argv[0], "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-pmc", "--no-extras", "--workload", "frame2voxel_full"] if False else sys.argv
flag = sys.argv[1] == "True"
from openess_amd.training.pretrain_step import PretrainStep
PretrainStep.pooled_student_features = flag
import bench
sys.argv = ["bench.py", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-pmc", "--no-extras", "--workload", "frame2voxel_full"]
print("pooled =", flag, end="  ")
bench.main()
PY
  done
done
