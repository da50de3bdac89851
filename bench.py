#!/usr/bin/env python3
"""Headline benchmark: event-frames/s of one full hot-path step (BASELINE.json metric).

One "step" = one batch of B=8 synthetic DSEC-shaped samples per GPU, ENTIRELY inside the timed region:
  raw events (uint16 x,y / int64 t / uint8 p, resident in HBM) -> rectify + tri-linear voxelizer
  (8 x 100 x 440 x 640) -> frozen dilated-ResNet-50 teacher forward -> 20 recurrent E2VID encoder steps
  (EventPreprocessor + ConvLSTM) -> SemSegE2VID forward + Dice/CE pixel distillation -> backward ->
  2 x AdamW      (= BASELINE configs[1]: DSEC 640x480, 5-bin voxel, pixel-distill only, batch 8;
                  reference: config/pretrain/DSEC/frame2voxel_*.yaml with if_spatial_contrastive: False).
Random-init weights of the reference architectures, synthetic data (no network / datasets here).
N > 1: one process per GPU (torch.distributed nccl = RCCL), per-rank batch 8 (weak scaling), gradient
all-reduce of the trainable parameters, max-over-ranks timing.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC for RCCL on this host driver (before torch loads HIP)

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C, H_SENSOR, W_SENSOR, CROP, NWIN, N_PER, B = 5, 480, 640, 40, 20, 100000, 8
H_NET = H_SENSOR - CROP
PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def make_inputs(rank, device, workload):
    from tests import synth
    xs, ys, ts, ps = [], [], [], []
    for b in range(B):
        x, y, t, p = synth.dsec_raw_events(NWIN * N_PER, H_SENSOR, W_SENSOR, seed=1205 + rank * B + b)
        xs.append(x); ys.append(y); ts.append(t); ps.append(p)
    ev = dict(x=torch.from_numpy(np.concatenate(xs)).to(device), y=torch.from_numpy(np.concatenate(ys)).to(device),
              t=torch.from_numpy(np.concatenate(ts)).to(device), p=torch.from_numpy(np.concatenate(ps)).to(device),
              maps=torch.from_numpy(synth.rectify_map(H_SENSOR, W_SENSOR)[None]).to(device),
              seg_map=torch.zeros(B * NWIN, dtype=torch.int32, device=device),
              seg=torch.arange(0, (B * NWIN + 1) * N_PER, N_PER, dtype=torch.int64))
    g = torch.Generator().manual_seed(99 + rank)
    frame = torch.rand(B, 3, H_NET, W_SENSOR, generator=g).to(device)
    pl = torch.randint(0, 11, (B, H_NET, W_SENSOR), generator=g)
    pl[torch.rand(B, H_NET, W_SENSOR, generator=g) < 0.05] = 255
    yy = (torch.arange(H_NET) * 10 // H_NET)[:, None]
    xx = (torch.arange(W_SENSOR) * 10 // W_SENSOR)[None, :]
    sp = (yy * 10 + xx)[None].repeat(B, 1, 1).long()
    return ev, frame, pl.to(device), sp.to(device), (B - 1) * 100 + 100


def cpu_baseline(sample_events, rectify_map):
    """Oracle (CPU port) timed on this host on ONE full-size sample (= one event-frame, about 10 s of CPU work):
    scalar C voxelizer (2M events -> 100x440x640) + fp32 PyTorch-CPU teacher forward, 20 recurrent E2VID encoder
    steps, SemSegE2VID forward + backward + AdamW at B=1.  64 torch threads (more threads oversubscribe: the same
    step took 345 s with 256 threads on this 256-core host)."""
    from oracle import losses as ol
    from oracle import nets as on
    from oracle.step import OracleStep, voxelize_sample
    ncores = os.cpu_count() or 1
    nthr = min(ncores, 64)
    torch.set_num_threads(nthr)
    torch.manual_seed(1205)
    x, y, t, p = sample_events
    t0 = time.perf_counter()
    try:       # scalar C port (oracle/voxel_oracle.c) when built, else the NumPy restatement
        from oracle import cport
        ev = torch.from_numpy(cport.dsec_event_tensor(x, y, t, p, rectify_map, NWIN, C, H_SENSOR, W_SENSOR, CROP))[None]
        vox_kind = "C port, 1 thread"
    except Exception:
        ev = voxelize_sample(x, y, t, p, rectify_map, NWIN, C, H_SENSOR, W_SENSOR, CROP)[None]
        vox_kind = "NumPy"
    t_vox = time.perf_counter() - t0
    step = OracleStep('frame2voxel', 11, NWIN, C, False)
    g = torch.Generator().manual_seed(5)
    frame = torch.rand(1, 3, H_NET, W_SENSOR, generator=g)
    pl = torch.randint(0, 11, (1, H_NET, W_SENSOR), generator=g)
    t1 = time.perf_counter()
    step.train_step((ev, None, frame, pl))
    t_net = time.perf_counter() - t1
    total = t_vox + t_net
    return {"value": round(1.0 / total, 5), "unit": "event-frames/s", "cores": nthr, "kind": "port",
            "sample": f"oracle on host CPU ({ncores} cores, {nthr} torch threads), ONE full-size event-frame: voxelizer "
                      f"({vox_kind}) {t_vox:.2f}s + fp32 teacher fwd / 20 E2VID steps / SemSegE2VID fwd+bwd+AdamW at B=1 "
                      f"{t_net:.2f}s = {total:.1f}s"}


def _pmc_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of this same command
    (profiles/r01_conv_hbm_traffic.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 read correction applied).
    PMC counters cannot be sampled from inside the timed process, so the number travels with the repo; None when the
    profile is absent or belongs to another workload."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_conv_hbm_traffic.json")
    if workload != "frame2voxel_pixel_distill" or not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            return round(json.load(f)["hbm_bytes_per_launch"])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="frame2voxel_pixel_distill",
                    choices=["frame2voxel_pixel_distill", "frame2voxel_full", "frame2recon_full",
                             "frame2voxel_pixel_distill_online"],
                    help="..._online: the pseudo-labels are argmax of the frozen MaskCLIP ViT-B/16 tower run inside the step "
                         "(SURVEY 8f rank 1) instead of the offline PNG labels the reference reads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from openess_amd import hip
    from openess_amd.training.ddp import GradAllReduce, broadcast_module_states
    from openess_amd.training.pretrain_step import PretrainStep

    contrastive = a.workload in ("frame2voxel_full", "frame2recon_full")
    online_teacher = None
    if a.workload.endswith("_online"):
        from openess_amd.models.maskclip_model import maskClipFeatureExtractor
        torch.manual_seed(1205)
        online_teacher = maskClipFeatureExtractor(text_categories=11).to(device).eval()
    option = "frame2recon" if a.workload.startswith("frame2recon") else "frame2voxel"
    step = PretrainStep(config_option=option, img_size=(H_NET, W_SENSOR), nr_events_data=NWIN, nr_temporal_bins=C,
                        if_spatial_contrastive=contrastive, superpixel_size=100, device=device)
    if world > 1:      # identical initial weights on every rank
        broadcast_module_states(step.models_dict.values())
    trainable = [p for m in step.models_dict.values() for p in m.parameters()]
    reducer = GradAllReduce(trainable, world)
    ev, frame, pl, sp, S = make_inputs(rank, device, a.workload)
    voxels = torch.empty((B, NWIN * C, H_NET, W_SENSOR), dtype=torch.float32, device=device)

    def one_step():
        hip.voxelize_dsec_raw(ev["x"], ev["y"], ev["t"], ev["p"], ev["maps"], ev["seg_map"], ev["seg"], C, H_SENSOR,
                              W_SENSOR, crop_rows=CROP, out=voxels.view(B * NWIN * C, H_NET, W_SENSOR))
        first = frame if option == "frame2recon" else voxels
        labels = pl if online_teacher is None else online_teacher(frame).argmax(dim=1)
        batch = (first, None, frame, labels, sp, S)
        for opt in step.optimizers_dict.values():
            opt.zero_grad()
        t_loss, losses, _ = step.task_train_step(batch)
        t_loss.backward()
        reducer()
        for opt in step.optimizers_dict.values():
            opt.step()
        return t_loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        one_step()
    hip.conv_timing_begin()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = one_step()
    fence()
    dt = time.perf_counter() - t0
    conv_stats = hip.conv_timing_end()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0 and os.environ.get("OESS_CONV_BREAKDOWN") and conv_stats:
        for k, (n, tm, fl) in sorted(conv_stats["by_shape"].items(), key=lambda kv: -kv[1][1]):
            print(f"# conv HxWxCin->Cout k,s,d {k}: {n // a.steps:3d}/step {tm / a.steps:7.3f} ms/step {fl / tm / 1e9:7.1f} TF/s", file=sys.stderr)
    if rank == 0:
        ms = dt / a.steps * 1e3
        value = world * B * a.steps / dt
        roof = None
        if conv_stats and conv_stats["ms"] > 0:
            ach = conv_stats["flops"] / (conv_stats["ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "conv_fwd_dma_kernel<{128|64},128,2> + short-K conv_fwd_dma32_kernel<128,..,3> (implicit-GEMM bf16 MFMA, LDS-DMA; all fwd + dgrad launches with Cout > 64, incl. the fused ConvLSTM-epilogue variant: conv FLOPs only)",
                    "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": _pmc_traffic(a.workload),
                    "launches_per_step": conv_stats["launches"] // a.steps,
                    "avg_launch_us": round(conv_stats["ms"] * 1e3 / max(conv_stats["launches"], 1), 2),
                    "share_of_step_time": round(conv_stats["ms"] / (dt * 1e3), 3)}
        out = {"metric": "event-frames/sec fwd+bwd @640x480 B=8", "value": round(value, 2), "unit": "event-frames/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
               "data": "synthetic", "loss": round(float(loss.detach()), 4),
               "config": {"workload": f"DSEC 640x480 5-bin x20 voxelizer + {a.workload} pre-train step "
                                      f"(E2VID-recurrent encoder x20, SemSegE2VID decoder, dilated-R50 teacher, Dice+CE), "
                                      f"batch {B}/GPU, random-init weights", "global_batch": world * B,
                          "parallelism": f"dp{world}"},
               "roofline": roof}
        if not a.no_cpu_baseline and world == 1:
            from tests import synth
            sample = synth.dsec_raw_events(NWIN * N_PER, H_SENSOR, W_SENSOR, seed=1205)
            try:
                out["cpu_baseline"] = cpu_baseline(sample, synth.rectify_map(H_SENSOR, W_SENSOR))
            except Exception as e:      # the baseline must never cost the GPU number
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
