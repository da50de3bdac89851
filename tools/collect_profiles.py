#!/usr/bin/env python3
"""Copy the judged summaries of tools/final_run.sh from gpurun_out/fin/ into profiles/ (tracked), named per round."""
import json, os, shutil, subprocess, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r06"
src, dst = "gpurun_out/fin", "profiles"
pairs = {f"prof_frame2voxel_pixel_distill/step_kernel_stats.csv": f"{R}_step_pixel_distill_kernel_stats.csv",
         f"prof_frame2voxel_full/step_kernel_stats.csv": f"{R}_step_frame2voxel_full_kernel_stats.csv",
         f"prof_frame2recon_full/step_kernel_stats.csv": f"{R}_step_frame2recon_full_kernel_stats.csv",
         f"stage_deeplab_fwd/p_kernel_stats.csv": f"{R}_stage_deeplab_fwd_kernel_stats.csv",
         f"stage_maskclip_fwd/p_kernel_stats.csv": f"{R}_stage_maskclip_fwd_kernel_stats.csv",
         f"stage_teacher_fwd/p_kernel_stats.csv": f"{R}_stage_teacher_fwd_kernel_stats.csv",
         f"vox_raw1/p_kernel_stats.csv": f"{R}_voxelizer_raw_kernel_stats.csv",
         f"vox_raw0/p_kernel_stats.csv": f"{R}_voxelizer_f32_kernel_stats.csv",
         f"prof_pipelined/step_kernel_stats.csv": f"{R}_step_pixel_distill_pipelined_kernel_stats.csv"}
for a, b in pairs.items():
    shutil.copyfile(os.path.join(src, a), os.path.join(dst, b))
line = [l for l in open(os.path.join(src, "bench.txt")).read().split("\n") if l.startswith("{")][-1]
json.dump(json.loads(line), open(os.path.join(dst, f"{R}_bench_line.json"), "w"), indent=1)
subprocess.run([sys.executable, "tools/mfma_util.py", os.path.join(src, "pmc_mfma"), os.path.join(dst, f"{R}_mfma_util.json")],
               check=True, stdout=subprocess.DEVNULL)
# plain-text outputs quoted in DESIGN / EXPERIMENTS (present from round 3 on)
with open(os.path.join(dst, f"{R}_misc_outputs.txt"), "w") as f:
    for name in ("pytest_gpu.txt", "pytest_gpu_run2.txt", "smoke.txt", "bench_schedule_ab.txt", "bench_no_skew.txt", "aten_probe_frame2recon_full.txt", "bench_torchrun_world1.txt", "train_loop.txt", "png.txt", "deeplab_breakdown.txt",
                 "segmean.txt", "voxelizer_pmc.txt", "stage_deeplab_fwd.txt", "stage_maskclip_fwd.txt", "stage_teacher_fwd.txt",
                 "vox_raw1.txt", "vox_raw0.txt", "step_sequence.txt", "lstm_group_ab.txt", "pmc_w128.txt", "pmc_w128_mfma.txt", "host_pools.txt", "conv1x1_ab.txt", "bench_gemm_ab.txt"):
        p = os.path.join(src, name)
        if os.path.exists(p):
            body = [l for l in open(p, errors="replace").read().split("\n") if l.strip() and "amdgpu.ids" not in l and not l.startswith("+")]
            f.write(f"==== {name}\n" + "\n".join(body[-40:]) + "\n")
for name, out in (("seq_last_step.txt", f"{R}_step_sequence_last_step.txt"),):
    p = os.path.join(src, name)
    if os.path.exists(p):
        body = [l for l in open(p, errors="replace").read().split("\n") if l.strip() and "amdgpu.ids" not in l]
        open(os.path.join(dst, out), "w").write("\n".join(body) + "\n")
print(sorted(f for f in os.listdir(dst) if f.startswith(R)))
