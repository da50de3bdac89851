# usage (on the GPU box, from the repo root): bash tools/step_sequence.sh [bench.py workload]
# Kernel trace of bench.py, then the LAST step in launch order: duration (us), idle time in front of the launch (us), kernel name
# -> gpurun_out/seq_last_step.txt; prints the sums.  Shows where the queue runs empty (host) and the gaps between large kernels.
# (one stream, --no-overlap-teacher: in the product schedule three streams interleave and "the step in launch order" is not a sequence)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
WL=${1:-frame2voxel_pixel_distill}
timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/seq -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-extras --no-overlap-teacher --workload $WL > gpurun_out/seq.txt 2>&1
python - <<'PY'
import csv, glob, re
fn = glob.glob('gpurun_out/seq/**/p_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adamw' in r['Kernel_Name']]
per = len(idx) // 3                                   # optimiser launches per step (3 steps traced)
a, b = idx[-per - 1], idx[-1]
prev_end, tot_d, tot_g = None, 0.0, 0.0
with open('gpurun_out/seq_last_step.txt', 'w') as out:
    for r in rows[a + 1:b + 1]:
        n = re.sub(r'\(anonymous namespace\)::|void |at::native::', '', r['Kernel_Name'])[:80]
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        out.write(f"{(e - s) / 1e3:8.1f} gap {gap:6.1f}  {n}\n")
        prev_end = e; tot_d += (e - s) / 1e3; tot_g += max(gap, 0.0)
print(f"last step: {b - a} launches, kernels {tot_d / 1e3:.2f} ms, idle {tot_g / 1e3:.2f} ms -> gpurun_out/seq_last_step.txt")
PY
find gpurun_out/seq -name "*_kernel_trace.csv" -delete
