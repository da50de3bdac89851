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

The ONE JSON line rank 0 prints carries, besides the headline `value` (inputs resident in HBM, as the contract says):
  roofline      dominant kernel family (MFMA implicit-GEMM conv), live HIP-event durations inside the timed region;
                `traffic` = HBM bytes per launch from two rocprofv3 PMC passes THIS invocation drives (N=1; --no-pmc skips)
  stages        per-stage rooflines measured live after the timed region (HIP events, N=1): voxelizer (uniform and
                structured-locality events), superpixel scatter-mean (fp32 / bf16, block / random ids), DeepLabv3 forward,
                teacher forward, MaskCLIP tower forward
  configs       the other single-GPU BASELINE configurations timed the same way: frame2voxel_full (configs[2]),
                frame2recon_full
  ingest        the same step with the raw events starting in PINNED HOST buffers: H2D copy + voxelizer of batch i+1 on a
                side HIP stream under step i (never `value`)
  train_loop    train.py's own loop at the same size (tools/bench_train_loop.py): DataLoader workers -> pin thread ->
                BaseTrainer.device_batches (side-stream ingest) -> train_step
  cpu_baseline  the oracle on this host's CPU (1 warm-up + 3 timed, median)
"""
import argparse
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
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
PEAK_HBM_GBS = 8000.0           # MI355X HBM3E (MI355X_MICROARCH.md)
# algorithmic work per batch of 8 (SURVEY 8d / BASELINE.md section 4)
VOX_BYTES = 16 * B * NWIN * N_PER + 4 * B * NWIN * C * H_SENSOR * W_SENSOR           # 1.239 GB
GFLOP_FWD = {"deeplabv3_resnet50": 106.8 * B, "dilated_r50_teacher": 845.1 * B}
DOMINANT_ONE = re.compile(r"conv3x3_lstm_w128_kernel")          # THE dominant kernel (roofline.frac); DOMINANT = its conv family
DOMINANT = re.compile(r"conv3x3_lstm_w128_kernel|conv3x3_halo_kernel<[01]>|conv3x3_halo_group_kernel<1>|conv3x3_halo256_group_kernel|conv5x5s2_halo(_group)?_kernel|"
                      r"conv_fwd_dma_kernel<(128|64), 128, [24], (true|false), [01](, 256)?>|conv_fwd_dma_kernel<256, 256, 2, true, 0(, 512)?>|"
                      r"conv_fwd_dma32_kernel<128, true, 0, 3")


def make_events(rank, structured=False):
    from openess_amd.datasets import _synth
    gen = _synth.dsec_structured_events if structured else _synth.dsec_raw_events
    cols = [gen(NWIN * N_PER, H_SENSOR, W_SENSOR, seed=1205 + rank * B + b) for b in range(B)]
    return {k: torch.from_numpy(np.concatenate([c[i] for c in cols])) for i, k in enumerate("xytp")}


def make_inputs(rank, device):
    from openess_amd.datasets import _synth
    host = make_events(rank)
    ev = {k: v.to(device) for k, v in host.items()}
    ev.update(maps=torch.from_numpy(_synth.rectify_map(H_SENSOR, W_SENSOR)[None]).to(device),
              seg_map=torch.zeros(B * NWIN, dtype=torch.int32, device=device),
              seg=torch.arange(0, (B * NWIN + 1) * N_PER, N_PER, dtype=torch.int64))
    g = torch.Generator().manual_seed(99 + rank)
    frame = torch.rand(B, 3, H_NET, W_SENSOR, generator=g).to(device)
    pl = torch.randint(0, 11, (B, H_NET, W_SENSOR), generator=g)
    pl[torch.rand(B, H_NET, W_SENSOR, generator=g) < 0.05] = 255
    yy = (torch.arange(H_NET) * 10 // H_NET)[:, None]
    xx = (torch.arange(W_SENSOR) * 10 // W_SENSOR)[None, :]
    sp = (yy * 10 + xx)[None].repeat(B, 1, 1).long()
    return host, ev, frame, pl.to(device), sp.to(device), (B - 1) * 100 + 100


# ------------------------------------------------------------------------------------------------ CPU baseline
def _median_time(fn, warm=1, n=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), ts


def _host_cpu():
    """(model string, physical cores, logical cpus) of this host from /proc/cpuinfo."""
    model, phys, logical = "unknown", set(), 0
    try:
        pid = cid = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                k, _, v = line.partition(":")
                k, v = k.strip(), v.strip()
                if k == "model name" and model == "unknown":
                    model = v
                elif k == "processor":
                    logical += 1
                elif k == "physical id":
                    pid = v
                elif k == "core id":
                    cid = v
                elif not k and pid is not None and cid is not None:
                    phys.add((pid, cid))
                    pid = cid = None
        if pid is not None and cid is not None:
            phys.add((pid, cid))
    except OSError:
        pass
    return model, (len(phys) or (os.cpu_count() or 1)), (logical or (os.cpu_count() or 1))


def cpu_baseline(sample_events, rectify_map):
    """The oracle (CPU port of the reference path, validated against the reference's golden vectors) on this host.
    SURVEY 8d: voxelizer legs 3 warm-up + 10 timed, the B=1 net step 1 warm-up + 5 timed, the DDD17 step 1 + 3; medians; CPU
    model string, physical core count and the thread cap are stated in `sample`.  Legs: (i) voxelizer of ONE full-size event-frame -- the reference-faithful
    8-pass masked-scatter algorithm (oracle/events.py = representations.py:15-54, sequential and with the reference's 8
    threads, sequence_ov.py:304-305) and the scalar C port; (ii) fp32 PyTorch-CPU teacher forward + 20 recurrent E2VID
    encoder steps + SemSegE2VID forward/backward + AdamW at B=1 (configs[1] shape); (iii) the CPU-runnable configs[0]:
    DDD17-shaped 200x352, K=6, B=2 step.  `value` = 1 / (fastest voxelizer + net step) event-frames/s."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import events as oe
    from oracle.step import OracleStep
    ncores = os.cpu_count() or 1
    nthr = min(ncores, 64)           # more torch threads oversubscribe: the same step took 345 s with 256 threads on a 256-core host
    torch.set_num_threads(nthr)
    torch.manual_seed(1205)
    x, y, t, p = sample_events
    legs = {}
    # (i) voxelizer
    xy = rectify_map[y.astype(np.int64), x.astype(np.int64)]
    n = x.shape[0] // NWIN

    def chunk(i):
        s, e = i * n, (i + 1) * n
        return oe.voxelgrid_trilinear(xy[s:e, 0], xy[s:e, 1], p[s:e].astype(np.float32), oe.dsec_time_normalise(t[s:e]), C, H_SENSOR, W_SENSOR)
    legs["voxelizer_8pass_numpy_1thread_s"], _ = _median_time(lambda: [chunk(i) for i in range(NWIN)], warm=3, n=10)
    with ThreadPoolExecutor(8) as pool:
        legs["voxelizer_8pass_numpy_8threads_s"], _ = _median_time(lambda: list(pool.map(chunk, range(NWIN))), warm=3, n=10)
    ev = None
    try:
        from oracle import cport
        fn = lambda: cport.dsec_event_tensor(x, y, t, p, rectify_map, NWIN, C, H_SENSOR, W_SENSOR, CROP)  # noqa: E731
        legs["voxelizer_c_port_1thread_s"], _ = _median_time(fn, warm=3, n=10)
        ev = torch.from_numpy(fn())[None]
    except Exception as e:
        legs["voxelizer_c_port_error"] = repr(e)
    if ev is None:
        ev = torch.from_numpy(oe.dsec_event_tensor(x, y, t, p, rectify_map, NWIN, C, H_SENSOR, W_SENSOR, CROP))[None]
    t_vox = min(v for k, v in legs.items() if k.startswith("voxelizer") and k.endswith("_s"))
    # (ii) configs[1]-shaped step at B=1
    step = OracleStep('frame2voxel', 11, NWIN, C, False)
    g = torch.Generator().manual_seed(5)
    frame = torch.rand(1, 3, H_NET, W_SENSOR, generator=g)
    pl = torch.randint(0, 11, (1, H_NET, W_SENSOR), generator=g)
    t_net, all_net = _median_time(lambda: step.train_step((ev, None, frame, pl)), warm=1, n=5)
    legs["net_step_B1_s"] = t_net
    # (iii) BASELINE configs[0]: DDD17-shaped CPU step (200x352, K=6, B=2, 20 sub-windows x 5 bins)
    try:
        step0 = OracleStep('frame2voxel', 6, NWIN, C, False)
        ev0 = (torch.randn(2, NWIN * C, 200, 352, generator=g) * (torch.rand(2, NWIN * C, 200, 352, generator=g) > 0.9)).contiguous()
        fr0, pl0 = torch.rand(2, 3, 200, 352, generator=g), torch.randint(0, 6, (2, 200, 352), generator=g)
        t0, _ = _median_time(lambda: step0.train_step((ev0, None, fr0, pl0)), warm=1, n=3)
        legs["ddd17_config0_step_B2_s"] = t0
        legs["ddd17_config0_event_frames_per_s"] = round(2.0 / t0, 4)
    except Exception as e:
        legs["ddd17_config0_error"] = repr(e)
    total = t_vox + t_net
    model, phys, logical = _host_cpu()
    # (iv) full-size parity for free: the SAME full-size sample through the HIP step and the oracle from identical weights
    parity = None
    try:
        from tests import full_size_parity as fsp
        yy = (torch.arange(H_NET) * 10 // H_NET)[:, None]
        xx = (torch.arange(W_SENSOR) * 10 // W_SENSOR)[None, :]
        parity = fsp.compare(ev, frame, pl, (yy * 10 + xx)[None].long(), nwin=NWIN, bins=C)
        parity["tolerances"] = {"rel": 0.02, "nce_rel": 0.05, "argmax_agree_clear_margin": 0.99, "logit_rel_rms_err": 0.2}
        parity["pass"] = bool(parity["rel"] <= 0.02 and parity["nce_rel"] <= 0.05 and parity["argmax_agree_clear_margin"] >= 0.99 and parity["logit_rel_rms_err"] <= 0.2)
        parity["what"] = ("B = 1, 2 M events -> 100 x 440 x 640, 20 sub-windows: Dice + CE, superpixel InfoNCE and per-pixel argmax of the student "
                          "logits (agreement on the pixels whose oracle top-2 margin exceeds 4 x the rms logit error: random-init logits are near-ties), "
                          "HIP step vs CPU oracle from identical weights (tests/full_size_parity.py)")
    except Exception as e:      # must never cost the throughput number
        parity = {"error": repr(e)[:300]}
    return {"parity_full_size": parity, "value": round(1.0 / total, 5), "unit": "event-frames/s", "cores": nthr, "kind": "port",
            "cpu_model": model, "physical_cores": phys, "logical_cpus": logical,
            "sample": f"oracle on the host CPU ({model}: {phys} physical cores, {logical} logical; {nthr} torch threads used -- more "
                      f"oversubscribe), ONE full-size event-frame (2 M events -> 100x440x640, B=1 step); voxelizer legs 3 warm-up + 10 "
                      f"timed, net step 1 + 5, medians: fastest CPU voxelizer {t_vox:.3f}s + fp32 teacher fwd / 20 E2VID steps / "
                      f"SemSegE2VID fwd+bwd+AdamW at B=1 {t_net:.2f}s = {total:.2f}s",
            "legs": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in legs.items()}}


# ------------------------------------------------------------------------------------------------ PMC traffic (live)
def _pmc_pass(counter, timeout_s):
    """One `rocprofv3 --pmc <counter> --kernel-trace` pass over a 1+1-step child of this script; returns
    (launches, sum of counter in KiB) over the dominant kernel family, or raises."""
    out = tempfile.mkdtemp(prefix=f"oess_pmc_{counter}_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--",
           sys.executable, os.path.abspath(__file__), "--child", "--steps", "1", "--warmup", "1", "--no-overlap-teacher"]
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout_s, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        import csv
        n, kib, n1, kib1 = 0, 0.0, 0, 0.0
        for fn in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            with open(fn) as f:
                for r in csv.DictReader(f):
                    if r.get("Counter_Name") == counter and DOMINANT.search(r["Kernel_Name"]):
                        n += 1
                        kib += float(r["Counter_Value"])
                        if DOMINANT_ONE.search(r["Kernel_Name"]):
                            n1 += 1
                            kib1 += float(r["Counter_Value"])
        if n == 0:
            raise RuntimeError("no dominant-kernel rows in the counter collection")
        return n, kib, n1, kib1
    finally:
        shutil.rmtree(out, ignore_errors=True)


def pmc_traffic(timeout_s=420):
    """HBM bytes per launch of the dominant kernel family: separate FETCH_SIZE / WRITE_SIZE passes (the guide's HBM section:
    counters in KiB; on gfx950 FETCH_SIZE reports half of a 16 B/lane coalesced read stream -> read = 2 * FETCH_SIZE * 1024)."""
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    try:
        nf, f_kib, nf1, f1 = _pmc_pass("FETCH_SIZE", timeout_s)
        nw, w_kib, nw1, w1 = _pmc_pass("WRITE_SIZE", timeout_s)
        fam = {"hbm_bytes_per_launch": round((2 * f_kib / nf + w_kib / nw) * 1024), "read_bytes_per_launch": round(2 * f_kib / nf * 1024),
               "write_bytes_per_launch": round(w_kib / nw * 1024), "launches_counted": nf}
        one = None
        if nf1 and nw1:
            one = {"hbm_bytes_per_launch": round((2 * f1 / nf1 + w1 / nw1) * 1024), "read_bytes_per_launch": round(2 * f1 / nf1 * 1024),
                   "write_bytes_per_launch": round(w1 / nw1 * 1024), "launches_counted": nf1}
        return {"family": fam, "dominant_kernel": one}, None
    except Exception as e:      # counters must never cost the throughput number
        return None, repr(e)[:300]


# ------------------------------------------------------------------------------------------------ the step
class Workload:
    def __init__(self, name, rank, world, device, inputs, wavefront=False, force_buckets=False, skew=True, pipeline=True):
        from openess_amd.training.ddp import GradAllReduce, broadcast_module_states
        from openess_amd.training.pretrain_step import PretrainStep
        self.name, self.device, self.world = name, device, world
        self.host, self.ev, self.frame, self.pl, self.sp, self.S = inputs
        self.contrastive = name in ("frame2voxel_full", "frame2recon_full")
        self.online_teacher = None
        if name.endswith("_online"):
            from openess_amd.models.maskclip_model import maskClipFeatureExtractor
            torch.manual_seed(1205)
            self.online_teacher = maskClipFeatureExtractor(text_categories=11).to(device).eval()
        self.option = "frame2recon" if name.startswith("frame2recon") else "frame2voxel"
        self.step = PretrainStep(config_option=self.option, img_size=(H_NET, W_SENSOR), nr_events_data=NWIN, nr_temporal_bins=C,
                                 if_spatial_contrastive=self.contrastive, superpixel_size=100, device=device,
                                 online_teacher=self.online_teacher, wavefront=wavefront)
        if not skew and getattr(self.step, "reconstructor", None) is not None:
            self.step.reconstructor.skew = False
        if world > 1:      # identical initial weights on every rank
            broadcast_module_states(self.step.models_dict.values())
        # force_buckets: launched by torch.distributed.run with ONE rank -> the whole bucket / hook / RCCL all-reduce path still runs
        self.reducer = GradAllReduce([p for m in self.step.models_dict.values() for p in m.parameters()], world, force_buckets=force_buckets)
        self.voxels = [torch.empty((B, NWIN * C, H_NET, W_SENSOR), dtype=torch.float32, device=device)]
        self.pipeline = pipeline

    def voxelize(self, ev, out):
        from openess_amd import hip
        hip.voxelize_dsec_raw(ev["x"], ev["y"], ev["t"], ev["p"], self.ev["maps"], self.ev["seg_map"], self.ev["seg"], C, H_SENSOR,
                              W_SENSOR, crop_rows=CROP, out=out.view(B * NWIN * C, H_NET, W_SENSOR))

    def batch_of(self, voxels):
        first = self.frame if self.option == "frame2recon" else voxels
        return (first, None, self.frame, self.pl, self.sp, self.S)   # labels: replaced inside the step by the online teacher's argmax when set

    def front_of(self, i):
        """Voxelizer + the frozen half of step i (teacher encoder, recurrent E2VID encoder): enqueued, not waited for."""
        if len(self.voxels) < 2:
            self.voxels.append(torch.empty_like(self.voxels[0]))
        vox = self.voxels[i & 1]             # step i + 1 is voxelized while step i's front may still read its tensor
        self.voxelize(self.ev, vox)
        batch = self.batch_of(vox)
        return batch, self.step.front(batch)

    def back(self, batch, front):
        """The trainable half: student forward, losses, backward, gradient all-reduce, 2 x AdamW."""
        for opt in self.step.optimizers_dict.values():
            opt.zero_grad()
        self.reducer.prepare()
        t_loss, losses, _ = self.step.task_train_step(batch, front=front)
        t_loss.backward()
        self.reducer()
        for opt in self.step.optimizers_dict.values():
            opt.step()
        return t_loss

    def train(self, voxels):
        return self.back(self.batch_of(voxels), None)

    def one_step(self):
        self.voxelize(self.ev, self.voxels[0])
        return self.train(self.voxels[0])

    def run(self, n):
        """n complete steps.  pipeline (default): the frozen half of step i + 1 is enqueued before the trainable half of step i
        (PretrainStep.pipeline_steps does the same for a trainer's loop); otherwise one step after the other."""
        loss = None
        if not self.pipeline:
            for _ in range(n):
                loss = self.one_step()
            return loss
        prev = self.front_of(0)
        for i in range(n):
            nxt = self.front_of(i + 1) if i + 1 < n else None
            loss = self.back(*prev)
            prev = nxt
        return loss

    def fence(self):
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(self, steps, warmup, conv_timing=False):
        from openess_amd import hip
        if warmup:
            self.run(warmup)
        if conv_timing:
            hip.conv_timing_begin(lstm_only=(conv_timing == "lstm"))
        self.fence()
        t0 = time.perf_counter()
        loss = self.run(steps)
        self.fence()
        dt = time.perf_counter() - t0
        stats = hip.conv_timing_end() if conv_timing else None
        if self.world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=self.device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, float(loss.detach()), stats

    def timed_with_ingest(self, steps, warmup):
        """The raw event columns of EVERY step start in pinned host memory (what DataLoader(pin_memory=True) hands over,
        13 B/event = 208 MB per batch): H2D copy + voxelizer of batch i+1 run on a side HIP stream under step i
        (double-buffered device columns and voxel tensors)."""
        side = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream(self.device)
        pinned = {k: v.pin_memory() for k, v in self.host.items()}
        dev = [{k: torch.empty_like(v, device=self.device) for k, v in self.host.items()} for _ in range(2)]
        if len(self.voxels) < 2:
            self.voxels.append(torch.empty_like(self.voxels[0]))
        ready, consumed = [None, None], [None, None]

        def produce(slot):
            with torch.cuda.stream(side):
                if consumed[slot] is not None:
                    side.wait_event(consumed[slot])            # step i-1 has finished reading voxels[slot]
                for k in pinned:
                    dev[slot][k].copy_(pinned[k], non_blocking=True)
                self.voxelize(dev[slot], self.voxels[slot])
                e = torch.cuda.Event()
                e.record(side)
                ready[slot] = e

        def loop(n):
            loss = None
            produce(0)
            for i in range(n):
                slot = i & 1
                main.wait_event(ready[slot])
                if i + 1 < n:
                    produce(1 - slot)
                loss = self.train(self.voxels[slot])
                e = torch.cuda.Event()
                e.record(main)
                consumed[slot] = e
            return loss
        loop(max(warmup, 2))
        self.fence()
        t0 = time.perf_counter()
        loop(steps)
        self.fence()
        dt = time.perf_counter() - t0
        if self.world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=self.device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt


# ------------------------------------------------------------------------------------------------ per-stage rooflines
def _ev_time(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def _hbm(ms, nbytes, note):
    a = nbytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "ms": round(ms, 4), "algorithmic_bytes": int(nbytes), "achieved": round(a, 1), "peak": PEAK_HBM_GBS,
            "unit": "GB/s", "frac": round(a / PEAK_HBM_GBS, 4), "note": note}


def _mfma(ms, gflop, note):
    a = gflop / ms              # GFLOP / ms = TFLOP/s
    return {"bound": "mfma", "ms": round(ms, 3), "algorithmic_gflop": round(gflop, 1), "achieved": round(a, 1), "peak": PEAK_BF16_TFLOPS,
            "unit": "TFLOP/s", "frac": round(a / PEAK_BF16_TFLOPS, 4), "note": note}


def stage_rooflines(wl, rank, device):
    """Live HIP-event timings (current stream = the launch stream of every kernel) of the hot-path stages on their own,
    at the BASELINE size, with the algorithmic figures of SURVEY 8d."""
    from openess_amd import hip
    st = {}
    out = wl.voxels[0]
    st["voxelizer"] = _hbm(_ev_time(lambda: wl.voxelize(wl.ev, out)), VOX_BYTES,
                           "raw u16/u16/i64/u8 columns + rectify gather + tri-linear splat + crop, B=8 x 20 x 100k uniform events; "
                           "algorithmic bytes = 16 B/event + 4 B/voxel at 480 rows (SURVEY 8d; the kernel ingests 13 B/event)")
    ev_s = {k: v.to(device) for k, v in make_events(rank, structured=True).items()}
    st["voxelizer_structured_events"] = _hbm(_ev_time(lambda: wl.voxelize(ev_s, out)), VOX_BYTES,
                                             "same, 70 % of the events on 200 moving edges (SURVEY 8d locality variant)")
    del ev_s
    # the same events through VoxelGrid.convert's OWN interface (fp32 x, y, p, t columns, rectification and time normalisation
    # done by the caller as in the reference): the 16 B/event the algorithmic figure of SURVEY 8d counts are what it reads
    try:
        n_ev = wl.ev["x"].numel()
        rm = wl.ev["maps"][0][wl.ev["y"].long(), wl.ev["x"].long()]
        xf, yf, pf = rm[:, 0].contiguous(), rm[:, 1].contiguous(), wl.ev["p"].float()
        tt = wl.ev["t"].view(-1, N_PER)
        d = (tt - tt[:, :1]).double().float()
        tf = (d / d[:, -1:]).reshape(-1).contiguous()
        st["voxelizer_f32_interface"] = _hbm(_ev_time(lambda: hip.voxelize_trilinear(xf, yf, pf, tf, wl.ev["seg"], C, H_SENSOR, W_SENSOR, crop_rows=CROP,
                                                                                     out=out.view(B * NWIN * C, H_NET, W_SENSOR))), VOX_BYTES,
                                                "VoxelGrid.convert's interface: fp32 x', y', p, t_norm columns (16 B/event) already rectified / normalised by the "
                                                "caller; no rectify gather in the kernel")
        del rm, xf, yf, pf, tf, d
    except Exception as e:
        st["voxelizer_f32_interface"] = {"error": repr(e)[:200]}
    # K7 superpixel scatter-mean, forward and backward, the reference's fp32 features and this pipeline's bf16 features
    ids_rand = torch.randint(0, 256, (B, H_NET // 8, W_SENSOR // 8), device=device).repeat_interleave(8, 1).repeat_interleave(8, 2)
    for dt, tag in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        feat = torch.randn(B, H_NET, W_SENSOR, 256, device=device).to(dt).permute(0, 3, 1, 2).requires_grad_(True)
        fb = feat.numel() * feat.element_size()
        for ids, name, S in ((wl.sp, "blocks", B * 100), (ids_rand, "random_ids_le_255", (B - 1) * 100 + 256)):
            ms_f = _ev_time(lambda: hip.superpixel_pool(feat.detach(), ids, 100, S=S))
            k = hip.superpixel_pool(feat, ids, 100, S=S)
            g = torch.randn_like(k)
            ms_fb = _ev_time(lambda: torch.autograd.grad(hip.superpixel_pool(feat, ids, 100, S=S), feat, g))
            st[f"scatter_mean_{tag}_{name}"] = _hbm(ms_fb, 2 * fb + 2 * ids.numel() * 8 + 2 * S * 256 * 4,
                                                    f"K7 forward + backward, {tag} features (SURVEY 8d counts fp32 = 4.63 GB); forward alone "
                                                    f"{ms_f:.3f} ms = {(fb + ids.numel() * 8) / ms_f / 1e6:.0f} GB/s")
        del feat
    # forward-only MFMA stages (train-mode BatchNorm = what the step runs; no autograd bookkeeping)
    from openess_amd.training.pretrain_step import PretrainStep
    recon = torch.rand(B, 3, H_NET, W_SENSOR, device=device)
    with torch.no_grad():
        teacher = wl.step.model_frame
        teacher.train()
        st["dilated_r50_teacher_forward"] = _mfma(_ev_time(lambda: teacher(wl.frame), iters=10), GFLOP_FWD["dilated_r50_teacher"],
                                                  "DilationFeatureExtractor forward incl. BatchNorm(train), x4 bilinear, L2 (a10)")
        dl = PretrainStep(config_option="frame2recon", img_size=(H_NET, W_SENSOR), nr_events_data=NWIN, device=device).model_recon
        dl.train()
        st["deeplabv3_forward"] = _mfma(_ev_time(lambda: dl(recon), iters=10), GFLOP_FWD["deeplabv3_resnet50"],
                                        "deeplabv3_resnet50 forward incl. ASPP 6/12/18, BatchNorm(train), 2 bilinear resizes (a12; north_star 'ASPP forward')")
        del dl
        try:
            from openess_amd.models.maskclip_model import maskClipFeatureExtractor
            torch.manual_seed(0)
            m = maskClipFeatureExtractor(text_categories=11).to(device).eval()
            L, Cw = 1121, 768
            gemm = 2.0 * B * L * (11 * (Cw * 3 * Cw + Cw * Cw) + (Cw * Cw + Cw * Cw) + 12 * 2 * Cw * 4 * Cw) + 2.0 * B * 1120 * (3 * 256 * Cw + Cw * 512 + 512 * 11)
            attn = 11 * 2 * 2.0 * B * 12 * L * L * 64
            st["maskclip_vit_b16_forward"] = _mfma(_ev_time(lambda: m(wl.frame), iters=10), (gemm + attn) / 1e9,
                                                   "MaskCLIP ViT-B/16 tower, 1121 tokens, value-path last block (a19; north_star 'CLIP forward')")
            del m
        except Exception as e:
            st["maskclip_vit_b16_forward"] = {"error": repr(e)[:200]}
    torch.cuda.empty_cache()
    return st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="frame2voxel_pixel_distill",
                    choices=["frame2voxel_pixel_distill", "frame2voxel_full", "frame2recon_full",
                             "frame2voxel_pixel_distill_online"],
                    help="..._online: the pseudo-labels are argmax of the frozen MaskCLIP ViT-B/16 tower run inside the step "
                         "(SURVEY 8f rank 1) instead of the offline PNG labels the reference reads")
    ap.add_argument("--wavefront", action="store_true", help="headline workload with the recurrent encoder on the wavefront schedule "
                    "(one HIP stream per ConvLSTM level; same results; per-launch durations of the roofline object overlap)")
    ap.add_argument("--no-skew", action="store_true", help="recurrent encoder in the plain order (one ConvLSTM launch per level and "
                    "sub-window) instead of the default skewed schedule with grouped launches; same results (A/B)")
    ap.add_argument("--no-s2-group", action="store_true", help="skewed schedule with one launch per stride-2 encoder conv (A/B of the "
                    "grouped launch of levels 1 and 2)")
    ap.add_argument("--no-overlap-teacher", action="store_true", help="everything on ONE stream, one step after the other: no teacher / "
                    "encoder side streams, no cross-step pipelining (A/B; same results)")
    ap.add_argument("--no-pipeline", action="store_true", help="frozen half of step i+1 NOT enqueued ahead of the trainable half of step i "
                    "(the teacher / encoder side streams inside a step stay; A/B; same results)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 PMC child passes (roofline.traffic = null)")
    ap.add_argument("--no-extras", action="store_true", help="headline + roofline only: no stages / configs / ingest blocks")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)        # profiled child of pmc_traffic(): steps only
    a = ap.parse_args()
    if a.no_s2_group:
        from openess_amd.e2vid.model.unet import UNetRecurrent
        UNetRecurrent.group_s2 = False
    if a.no_overlap_teacher:
        from openess_amd.training.pretrain_step import PretrainStep
        PretrainStep.overlap_teacher = False

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # a 1-rank launch through torch.distributed.run (RANK / MASTER_PORT in the environment) also initialises RCCL and keeps the
    # gradient reducer active: the collective path is exercised on a single GPU (tools/final_run.sh)
    launched = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    inputs = make_inputs(rank, device)
    wl = Workload(a.workload, rank, world, device, inputs, wavefront=a.wavefront, force_buckets=launched and world == 1, skew=not a.no_skew,
                  pipeline=not (a.no_pipeline or a.no_overlap_teacher))
    if a.child:
        wl.timed(a.steps, a.warmup)
        return
    # Event records are barrier packets of their own (~2.8 us of idle queue each: 2 x 115 per step = 1.3 ms of a ONE-stream step, which
    # is why the serial region reads ~3 % under an uninstrumented one-stream run).  OESS_BENCH_EVENTS=dominant brackets only the dominant
    # kernel's 22 launches in the timed region (A/B: 221.3 -> 221.9 event-frames/s, the other streams hide most of it); the default
    # keeps all 115 so that `timed_region` carries the family figure as well.
    lstm_only_timed = bool(getattr(wl.step, "overlap_teacher", False)) and os.environ.get("OESS_BENCH_EVENTS", "all") == "dominant" and \
        not os.environ.get("OESS_CONV_BREAKDOWN")
    dt, loss, conv_stats = wl.timed(a.steps, a.warmup, conv_timing="lstm" if lstm_only_timed else True)
    if rank == 0 and os.environ.get("OESS_CONV_BREAKDOWN") and conv_stats:
        for k, (n, tm, fl) in sorted(conv_stats["by_shape"].items(), key=lambda kv: -kv[1][1]):
            print(f"# conv HxWxCin->Cout k,s,d {k}: {n // a.steps:3d}/step {tm / a.steps:7.3f} ms/step {fl / tm / 1e9:7.1f} TF/s", file=sys.stderr)
    extras = not a.no_extras
    out = None
    # The product step runs the frozen teacher on a second HIP stream under the recurrent encoder, so the per-launch durations of
    # the timed region include the time two kernels share the CUs.  A short second region on ONE stream (all ranks; same process,
    # same box) gives the launches' own durations: `roofline.serial_reference`.
    serial = None
    if getattr(wl.step, "overlap_teacher", False):
        n_ser = max(10, a.steps // 5)
        keep = wl.pipeline
        wl.step.overlap_teacher, wl.pipeline = False, False
        dt_ser, _, serial = wl.timed(n_ser, 2, conv_timing=True)
        wl.step.overlap_teacher, wl.pipeline = True, keep
        if serial:
            serial["steps"], serial["dt"] = n_ser, dt_ser

    def dominant(cs, steps, dt_):
        grp = [(n, tm, fl) for k, (n, tm, fl) in cs["by_shape"].items()
               if k and k[0] == "group" and all(isinstance(s_, tuple) and s_[-1] == "lstm" for s_ in k[1:])]
        if not grp:
            return None
        gn, gms, gfl = (sum(g[i] for g in grp) for i in range(3))
        return {"name": "conv3x3_lstm_w128_kernel (the fused-ConvLSTM levels of a stage of the recurrent encoder in ONE persistent launch: 128 x 128 wave tiles, one wave per SIMD, w128-tiled cell state; csrc/conv_lstm_w128.h)",
                "launches_per_step": round(gn / steps, 2),
                "sum_gflop": round(gfl / steps / 1e9, 1), "sum_us": round(gms / steps * 1e3, 1),
                "achieved": round(gfl / (gms * 1e-3) / 1e12, 1), "unit": "TFLOP/s",
                "frac": round(gfl / (gms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), "share_of_step_time": round(gms / (dt_ * 1e3), 3)}
    if rank == 0:
        ms = dt / a.steps * 1e3
        value = world * B * a.steps / dt
        roof = None
        if conv_stats and conv_stats["ms"] > 0:
            ach = conv_stats["flops"] / (conv_stats["ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "conv3x3_lstm_w128_kernel (the three fused-ConvLSTM levels of a stage of the recurrent encoder in one persistent launch) + conv3x3_halo_kernel<{0|1}> (3x3 stride-1, row-halo reuse) + conv_fwd_dma_kernel<{128|64},128,2> + conv_fwd_dma_kernel<256,256,2> (large 1x1 layers) + short-K conv_fwd_dma32_kernel<128,..,3> + conv5x5s2_halo{,_group}_kernel (5x5 stride-2 encoders, 2-D input halo; levels 1 + 2 in one launch): implicit-GEMM bf16 MFMA with LDS-DMA operands, every forward / data-gradient launch with Cout > 64; conv FLOPs only",
                    "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": None,
                    "launches_per_step": conv_stats["launches"] // a.steps,
                    "avg_launch_us": round(conv_stats["ms"] * 1e3 / max(conv_stats["launches"], 1), 2),
                    "algorithmic_gflop_per_launch": round(conv_stats["flops"] / max(conv_stats["launches"], 1) / 1e9, 2),
                    "share_of_step_time": round(conv_stats["ms"] / (dt * 1e3), 3)}
            # the ONE dominant kernel on its own: every grouped ConvLSTM launch (conv3x3_halo_group_kernel<1>) with the FLOPs of
            # exactly the levels that launch carried (a stage's first / last launches carry two levels, not three)
            dk = dominant(conv_stats, a.steps, dt)
            if dk:
                roof["dominant_kernel"] = dk
            if serial and serial["ms"] > 0:
                # The roofline of a KERNEL is read where the kernel has the chip to itself: the serial region.  The timed region's
                # own per-launch figures (kernels of two or three streams sharing the CUs) are kept beside it.
                timed = {k: roof[k] for k in ("achieved", "frac", "launches_per_step", "avg_launch_us", "share_of_step_time", "dominant_kernel")
                         if k in roof}
                if lstm_only_timed:     # only the dominant kernel's launches were bracketed there
                    timed = {"dominant_kernel": roof.get("dominant_kernel"), "frac": None,
                             "note": "only the dominant kernel's launches carry event records in the timed region (an event record is a "
                                     "barrier packet: 2 x 115 per step cost ~1.3 ms of idle queue on one stream); the family is read in the serial region"}
                ach_s = serial["flops"] / (serial["ms"] * 1e-3) / 1e12
                roof.update({"achieved": round(ach_s, 1), "frac": round(ach_s / PEAK_BF16_TFLOPS, 4),
                             "launches_per_step": serial["launches"] // serial["steps"],
                             "avg_launch_us": round(serial["ms"] * 1e3 / max(serial["launches"], 1), 2),
                             "algorithmic_gflop_per_launch": round(serial["flops"] / max(serial["launches"], 1) / 1e9, 2),
                             "share_of_step_time": round(serial["ms"] / (serial["dt"] * 1e3), 3),
                             "dominant_kernel": dominant(serial, serial["steps"], serial["dt"])})
                roof["region"] = (f"serial region: {serial['steps']} steps of the same workload on ONE stream, one step after the other "
                                  f"(--no-overlap-teacher order; {round(world * B * serial['steps'] / serial['dt'], 2)} event-frames/s), run in "
                                  "this process right after the timed region -- a launch's duration there is the kernel's own (the region's rate carries the "
                                  "two event records around each of the 115 family launches: ~1.3 ms of idle queue per step).  In the "
                                  "timed region the frozen half of a step (teacher encoder, recurrent E2VID encoder) runs on its own HIP "
                                  "streams and, for step i+1, ahead of the trainable half of step i (PretrainStep.front / pipeline_steps): "
                                  "its per-launch durations include CU time-sharing and are kept in `timed_region`")
                roof["serial_event_frames_per_s"] = round(world * B * serial["steps"] / serial["dt"], 2)
                roof["timed_region"] = timed
            # Top level = THE dominant kernel (its own launches: per-launch algorithmic FLOPs / summed event durations, serial region
            # when there is one); the 115-launch conv family it belongs to sits beside it (family_*), the timed region's figures flat.
            dk = roof.get("dominant_kernel")
            if dk:
                roof["family"] = {k: roof[k] for k in ("kernel", "achieved", "frac", "launches_per_step", "avg_launch_us",
                                                       "algorithmic_gflop_per_launch", "share_of_step_time")}
                roof["family_frac"], roof["family_achieved"] = roof["frac"], roof["achieved"]
                roof.update({"kernel": dk["name"], "achieved": dk["achieved"], "frac": dk["frac"], "launches_per_step": dk["launches_per_step"],
                             "avg_launch_us": round(dk["sum_us"] / max(dk["launches_per_step"], 1e-9), 2),
                             "algorithmic_gflop_per_launch": round(dk["sum_gflop"] / max(dk["launches_per_step"], 1e-9), 2),
                             "share_of_step_time": dk["share_of_step_time"]})
                tr_ = roof.get("timed_region") or {}
                if tr_.get("dominant_kernel"):
                    roof["timed_region_frac"] = tr_["dominant_kernel"]["frac"]
                    roof["timed_region_family_frac"] = tr_.get("frac")
        out = {"metric": "event-frames/sec fwd+bwd @640x480 B=8", "value": round(value, 2), "unit": "event-frames/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
               "data": "synthetic", "loss": round(loss, 4),
               "config": {"workload": f"DSEC 640x480 5-bin x20 voxelizer + {a.workload} pre-train step "
                                      f"(E2VID-recurrent encoder x20, SemSegE2VID decoder, dilated-R50 teacher, Dice+CE), "
                                      f"batch {B}/GPU, random-init weights", "global_batch": world * B,
                          "parallelism": f"dp{world}"},
               "roofline": roof}
        if launched:
            out["rccl"] = {"initialised": True, "world": world, "reducer_active": bool(wl.reducer.active),
                           "buckets": len(wl.reducer.buckets), "bucket_bytes": wl.reducer.exposed_bytes()}
    # ---- ingest-inclusive rate of the same workload (all ranks take part: the reducer is a collective)
    if extras:
        n_in = max(10, a.steps // 4)
        dt_in = wl.timed_with_ingest(n_in, 2)
        if rank == 0:
            out["ingest"] = {"value": round(world * B * n_in / dt_in, 2), "unit": "event-frames/s", "ms_per_step": round(dt_in / n_in * 1e3, 3),
                             "steps": n_in, "h2d_bytes_per_step": int(sum(v.numel() * v.element_size() for v in wl.host.values())),
                             "note": "raw event columns start in PINNED HOST buffers every step; H2D + voxelizer of batch i+1 on a side "
                                     "HIP stream under step i (north_star 'straight from pinned host event buffers'); never `value`"}
        if world == 1:
            out["stages"] = stage_rooflines(wl, rank, device)
            if rank == 0 and out.get("roofline") is not None:
                # the stage fractions north_star sets targets for, inside the `roofline` object the driver's record keeps
                short = {"voxelizer": "voxelizer", "voxelizer_f32": "voxelizer_f32_interface", "scatter_mean_bf16": "scatter_mean_bf16_blocks",
                         "scatter_mean_fp32": "scatter_mean_fp32_blocks", "deeplabv3_fwd": "deeplabv3_forward",
                         "vit_fwd": "maskclip_vit_b16_forward", "teacher_fwd": "dilated_r50_teacher_forward"}
                out["roofline"]["stage_fracs"] = {k: out["stages"][v].get("frac") for k, v in short.items() if v in out["stages"]}
                out["roofline"]["stage_ms"] = {k: out["stages"][v].get("ms") for k, v in short.items() if v in out["stages"]}
                for k, v in out["roofline"]["stage_fracs"].items():        # flat copies: the driver's record keeps only the top level
                    out["roofline"][k + "_frac"] = v
    # ---- the other single-GPU BASELINE configurations, same timing protocol
    if extras and a.workload == "frame2voxel_pixel_distill":
        del wl
        torch.cuda.empty_cache()
        cfgs = {}
        for name in ("frame2voxel_full", "frame2recon_full", "frame2voxel_pixel_distill+wavefront"):
            n2 = max(10, a.steps // 4)
            if world == 1:
                # each configuration in its OWN process, as a user would run it: in this process, behind the headline / ingest /
                # stage blocks, frame2recon_full read 294 event-frames/s where the same command alone reads 327 (allocator state)
                cmd = [sys.executable, os.path.abspath(__file__), "--workload", name.split("+")[0], "--steps", str(n2), "--warmup", "3",
                       "--no-extras", "--no-pmc", "--no-cpu-baseline"] + (["--wavefront"] if name.endswith("+wavefront") else [])
                try:
                    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                    line = json.loads([l for l in r.stdout.split("\n") if l.startswith("{")][-1])
                    cfgs[name] = {"value": line["value"], "unit": "event-frames/s", "ms_per_step": line["ms_per_step"], "steps": n2,
                                  "loss": line["loss"]}
                except Exception as e:      # never cost the headline
                    cfgs[name] = {"error": repr(e)[:200]}
                continue
            w2 = Workload(name.split("+")[0], rank, world, device, inputs, wavefront=name.endswith("+wavefront"))
            dt2, loss2, _ = w2.timed(n2, 3)
            cfgs[name] = {"value": round(world * B * n2 / dt2, 2), "unit": "event-frames/s", "ms_per_step": round(dt2 / n2 * 1e3, 3),
                          "steps": n2, "loss": round(loss2, 4)}
            del w2
            torch.cuda.empty_cache()
        if rank == 0:
            cfgs["frame2voxel_full"]["what"] = "BASELINE configs[2]: configs[1] + superpixel scatter-mean + InfoNCE (differentiable teacher head)"
            cfgs["frame2recon_full"]["what"] = "frame2recon pre-training: DeepLabv3/ASPP student + teacher + superpixel InfoNCE + Dice/CE"
            cfgs["frame2voxel_pixel_distill+wavefront"]["what"] = ("the headline workload with the recurrent encoder scheduled as a wavefront over one "
                                                                   "HIP stream per ConvLSTM level, overlapping the teacher forward (e2vid/wavefront.py; "
                                                                   "opt-in: overlapped launches would blur the per-launch roofline timing)")
            out["configs"] = cfgs
    if rank == 0 and extras and world == 1 and a.workload == "frame2voxel_pixel_distill":
        # the PRODUCT loop (train.py dispatch, DataLoader workers, pin thread, side-stream ingest) beside the headline
        try:
            torch.cuda.empty_cache()
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import contextlib
            import bench_train_loop
            with contextlib.redirect_stdout(sys.stderr):        # the trainer logs to stdout; this script's stdout is ONE JSON line
                out["train_loop"] = bench_train_loop.measure(batches=24, workers=4, prefetch=True)
                old = bench_train_loop.measure(batches=24, workers=10, prefetch=True, ring=False)
            out["train_loop"]["vs_headline"] = round(out["train_loop"]["value"] / out["value"], 3)
            out["train_loop"]["torch_dataloader_10_workers"] = {"value": old["value"], "vs_headline": round(old["value"] / out["value"], 3)}
        except Exception as e:      # never cost the headline
            out["train_loop"] = {"error": repr(e)[:300]}
        # 8-GPU dry run of the host side: eight 2-worker loader pools drained by one process on this box, every
        # batch copied to the device out of its pinned slot; to be read against 8 x the headline (tools/bench_host_pools.py).
        # In its own process: the pools fork 16 workers, which must not inherit this process's 20 GB of device-side state.
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_host_pools.py"), "--json"], capture_output=True, text=True, timeout=420)
            pools = json.loads([l for l in r.stdout.split("\n") if l.startswith("{")][-1])
            pools["vs_8x_headline"] = round(pools["value"] / (8 * out["value"]), 3)
            if isinstance(out.get("train_loop"), dict):
                out["train_loop"]["eight_pools_one_process"] = pools
        except Exception as e:
            if isinstance(out.get("train_loop"), dict):
                out["train_loop"]["eight_pools_one_process"] = {"error": repr(e)[:300]}
    if rank == 0 and extras and world == 1 and a.workload == "frame2voxel_pixel_distill":
        # BASELINE configs[2] (openess_trainer full path) and configs[4] (fine-tune / linear-probe): the stage-2/3 trainers and
        # OpenESSModel built through train.py's dispatch at the BASELINE size, train_step on a resident batch (tools/bench_stage2.py)
        # Each trainer runs in its OWN process, as train.py would run it (in this process, after the headline / ingest / loader
        # blocks, the last case read 236 event-frames/s where the same command alone reads 309: allocator and registry state of
        # the workloads before it).
        try:
            torch.cuda.empty_cache()
            if os.path.join(ROOT, "tools") not in sys.path:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_stage2
            for case in [c[0] for c in bench_stage2.CASES]:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_stage2.py"), "--only", case, "--steps",
                                    str(max(10, a.steps // 5))], capture_output=True, text=True, timeout=600)
                lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
                if r.returncode == 0 and lines:
                    out["configs"].update({"stage2:" + k: v for k, v in json.loads(lines[-1]).items()})
                else:
                    out["configs"]["stage2:" + case] = {"error": (r.stderr or "")[-300:]}
        except Exception as e:      # never cost the headline
            out["configs"]["stage2_error"] = repr(e)[:300]
    if rank == 0:
        if world == 1 and not a.no_pmc and out["roofline"] is not None:
            torch.cuda.empty_cache()
            tr, err = pmc_traffic()
            if tr is not None:
                one = tr["dominant_kernel"] or tr["family"]
                out["roofline"]["traffic"] = one["hbm_bytes_per_launch"]
                out["roofline"]["traffic_detail"] = dict(tr, method="2 rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a 1+1-step child "
                                                                    "of this command, gfx950 read correction x2; `traffic` = the dominant "
                                                                    "kernel's launches, `family` = all launches of its conv family")
            else:
                out["roofline"]["traffic_error"] = err
        if not a.no_cpu_baseline and world == 1:
            from openess_amd.datasets import _synth
            sample = _synth.dsec_raw_events(NWIN * N_PER, H_SENSOR, W_SENSOR, seed=1205)
            try:
                out["cpu_baseline"] = cpu_baseline(sample, _synth.rectify_map(H_SENSOR, W_SENSOR))
                out["parity_full_size"] = out["cpu_baseline"].pop("parity_full_size", None)
            except Exception as e:      # the baseline must never cost the GPU number
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if launched:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
