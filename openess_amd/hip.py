"""Thin tensor-level wrappers over the liboess C-ABI (include/oess.h).

PyTorch is plumbing only here: device memory (tensors), the current HIP stream and autograd
bookkeeping.  Every function enqueues hand-written HIP kernels on the CURRENT torch stream and
raises if the library is missing or a tensor is not on the GPU -- there is no CPU/eager fallback.
"""
import ctypes

import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("openess_amd HIP path needs CUDA/HIP tensors (no CPU fallback)")


def h2d_async(t, device):
    """Host tensor -> device; small ones without a device synchronisation: an upload from PAGEABLE memory blocks the host until the
    stream has drained (a full synchronisation per call -- one at the top of every training step for the segment offsets of
    the voxelizer); staged through pinned memory it is an ordinary stream-ordered copy."""
    if t.is_cuda:
        return t
    if not t.is_pinned() and t.numel() * t.element_size() <= (1 << 20):
        t = t.pin_memory()            # large pageable tensors (a loader without pin_memory=True) keep the runtime's own staging
    return t.to(device, non_blocking=True)


def _bump(t):
    """A kernel wrote `t` through its raw pointer: tell autograd / version-keyed caches."""
    torch.autograd.graph.increment_version(t)


_WS_CACHE = {}


def _workspace(nbytes, device, tag="ws"):
    # stream-ordered reuse of one buffer is only safe on ONE stream: the current stream is always part of the key (the teacher
    # forward / backward runs on its own stream next to the student's, training/pretrain_step.py)
    key = (tag, device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS_CACHE.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _WS_CACHE[key] = ws
    return ws


def _seg_info(seg_offsets):
    """seg_offsets: 1-D int64 (CPU or GPU). Returns (device tensor, n_seg, max_len) -- max_len is
    computed on the host copy so that no device sync is needed when offsets come from the loader."""
    if seg_offsets.is_cuda:
        host = seg_offsets.cpu()
        dev = seg_offsets
    else:
        host = seg_offsets
        dev = None
    host = host.to(torch.int64)
    n_seg = host.numel() - 1
    lens = host[1:] - host[:-1]
    if n_seg < 1 or bool((lens < 0).any()):
        raise ValueError("seg_offsets must be non-decreasing with at least 2 entries")
    return host, dev, n_seg, int(lens.max().item()), int(host[-1].item())


# ------------------------------------------------------------------------------------------ K1
def voxelize_trilinear(x, y, p, t, seg_offsets, C, H, W, crop_rows=0, count_mode=False, out=None):
    """Batched VoxelGrid.convert: returns (n_seg*C) x (H-crop_rows) x W float32 on the GPU."""
    lib = _lib.load()
    _need_gpu(x, y, p, t)
    for a in (x, y, p, t):
        if a.dtype != torch.float32 or not a.is_contiguous() or a.ndim != 1:
            raise ValueError("x, y, p, t must be contiguous 1-D float32")
    host, dev, n_seg, max_len, n_ev = _seg_info(seg_offsets)
    if n_ev > x.numel():
        raise ValueError("seg_offsets exceed the event arrays")
    if dev is None:
        dev = h2d_async(host, x.device)
    if out is None:
        out = torch.empty((n_seg * C, H - crop_rows, W), dtype=torch.float32, device=x.device)
    nbytes = lib.oess_voxelize_workspace_bytes(x.numel(), n_seg, max_len, C, H, W, crop_rows)
    ws = _workspace(nbytes, x.device, tag=("vox", torch.cuda.current_stream(x.device).cuda_stream))
    _lib.check(lib.oess_voxelize_trilinear_f32(_ptr(x), _ptr(y), _ptr(p), _ptr(t), _ptr(dev), n_seg, max_len, C, H, W,
                                               crop_rows, int(count_mode), _ptr(out), _ptr(ws), ws.numel(), _stream()),
               "oess_voxelize_trilinear_f32")
    _bump(out)
    return out


def voxelize_dsec_raw(x, y, t_us, p, rectify_maps, seg_map, seg_offsets, C, H, W, crop_rows=0, count_mode=False,
                      out=None):
    """Raw DSEC columns (uint16 x,y; int64 t; uint8 p) + rectify maps [n_maps,H,W,2] -> voxel tensor."""
    lib = _lib.load()
    _need_gpu(x, y, t_us, p, rectify_maps, seg_map)
    if x.dtype != torch.uint16 or y.dtype != torch.uint16 or t_us.dtype != torch.int64 or p.dtype != torch.uint8:
        raise ValueError("raw DSEC dtypes are x,y:uint16  t:int64  p:uint8")
    if rectify_maps.dtype != torch.float32 or rectify_maps.ndim != 4 or tuple(rectify_maps.shape[1:]) != (H, W, 2):
        raise ValueError("rectify_maps must be float32 [n_maps, H, W, 2]")
    host, dev, n_seg, max_len, n_ev = _seg_info(seg_offsets)
    if n_ev > x.numel():
        raise ValueError("seg_offsets exceed the event arrays")
    if dev is None:
        dev = h2d_async(host, x.device)
    seg_map = seg_map.to(torch.int32)
    if seg_map.numel() != n_seg:
        raise ValueError("seg_map needs one entry per segment")
    if out is None:
        out = torch.empty((n_seg * C, H - crop_rows, W), dtype=torch.float32, device=x.device)
    nbytes = lib.oess_voxelize_workspace_bytes(x.numel(), n_seg, max_len, C, H, W, crop_rows)
    ws = _workspace(nbytes, x.device, tag=("vox", torch.cuda.current_stream(x.device).cuda_stream))
    _lib.check(lib.oess_voxelize_dsec_raw(_ptr(x), _ptr(y), _ptr(t_us), _ptr(p), _ptr(rectify_maps.contiguous()),
                                          _ptr(seg_map), rectify_maps.shape[0], _ptr(dev), n_seg, max_len, C, H, W,
                                          crop_rows, int(count_mode), _ptr(out), _ptr(ws), ws.numel(), _stream()),
               "oess_voxelize_dsec_raw")
    _bump(out)
    return out


def voxelize_nearest(events, seg_offsets, nbins, H, W, crop_rows=0, separate_pol=True, count_mode=False, out=None):
    """Batched generate_voxel_grid over [N x 4] (x, y, t, p) int64 or float64 events."""
    lib = _lib.load()
    _need_gpu(events)
    if events.ndim != 2 or events.shape[1] != 4 or not events.is_contiguous():
        raise ValueError("events must be contiguous [N x 4]")
    host, dev, n_seg, max_len, n_ev = _seg_info(seg_offsets)
    if n_ev > events.shape[0]:
        raise ValueError("seg_offsets exceed the event array")
    if dev is None:
        dev = h2d_async(host, events.device)
    ch = 2 * nbins if separate_pol else nbins
    if out is None:
        out = torch.empty((n_seg * ch, H - crop_rows, W), dtype=torch.float32, device=events.device)
    nbytes = lib.oess_voxelize_workspace_bytes(events.shape[0], n_seg, max_len, 2 * nbins, H, W, crop_rows)
    ws = _workspace(nbytes, events.device, tag=("vox", torch.cuda.current_stream(events.device).cuda_stream))
    if events.dtype == torch.int64:
        fn, name = lib.oess_voxelize_nearest_i64, "oess_voxelize_nearest_i64"
    elif events.dtype == torch.float64:
        fn, name = lib.oess_voxelize_nearest_f64, "oess_voxelize_nearest_f64"
    else:
        raise ValueError("events must be int64 or float64")
    _lib.check(fn(_ptr(events), _ptr(dev), n_seg, max_len, nbins, H, W, crop_rows, int(separate_pol), int(count_mode),
                  _ptr(out), _ptr(ws), ws.numel(), _stream()), name)
    return out


def event_histogram(events, seg_offsets, H, W):
    lib = _lib.load()
    _need_gpu(events)
    if events.dtype != torch.int64 or events.ndim != 2 or events.shape[1] != 4 or not events.is_contiguous():
        raise ValueError("events must be contiguous int64 [N x 4]")
    host, dev, n_seg, max_len, _ = _seg_info(seg_offsets)
    if dev is None:
        dev = h2d_async(host, events.device)
    out = torch.empty((n_seg * 2, H, W), dtype=torch.float32, device=events.device)
    _lib.check(lib.oess_event_histogram_i64(_ptr(events), _ptr(dev), n_seg, max_len, H, W, _ptr(out), _stream()),
               "oess_event_histogram_i64")
    return out


# ------------------------------------------------------------------------------------------ f3: PNG maps
def png_decode_gray8_batch(files, lengths, H, W, flips=None, out=None):
    """Decode a batch of 8-bit single-channel PNG files on the GPU.  files: uint8 1-D (device) = the files back to back;
    lengths: host sequence of their byte lengths; flips: optional host sequence of bools (horizontal mirror per image).
    Returns (maps int64 [n, H, W], status int32 [n] on the device: 0 = ok, else the map is all 255)."""
    lib = _lib.load()
    _need_gpu(files)
    if files.dtype != torch.uint8 or files.ndim != 1 or not files.is_contiguous():
        raise ValueError("files must be a contiguous 1-D uint8 tensor")
    n = len(lengths)
    if n == 0 or int(sum(lengths)) > files.numel():
        raise ValueError("lengths do not match the byte tensor")
    offs, scr, a, b = [0], [], 0, 0
    for L in lengths:
        a += int(L)
        offs.append(a)
        scr.append(b)
        b += ((int(L) + 15) // 16) * 16 + H * (W + 1) + 16
    meta = h2d_async(torch.tensor(offs + scr, dtype=torch.int64), files.device)
    need = lib.oess_png_decode_scratch_bytes(a, n, H, W)
    ws = _workspace(max(need, b + 256), files.device, tag=("png", torch.cuda.current_stream(files.device).cuda_stream))
    fl = None
    if flips is not None and any(flips):
        fl = h2d_async(torch.tensor([1 if f else 0 for f in flips], dtype=torch.uint8), files.device)
    if out is None:
        out = torch.empty((n, H, W), dtype=torch.int64, device=files.device)
    status = torch.empty(n, dtype=torch.int32, device=files.device)
    _lib.check(lib.oess_png_decode_gray8_batch(_ptr(files), _ptr(meta), a, n, H, W, _ptr(fl), _ptr(out), _ptr(ws), ws.numel(),
                                               meta[n + 1:].data_ptr(), _ptr(status), _stream()), "oess_png_decode_gray8_batch")
    return out, status


# ------------------------------------------------------------------------------------------ K2
def masked_normalize(x, out=None):
    """EventPreprocessor / normalize_voxel_grid on a dense float32 tensor (whole-tensor statistics)."""
    lib = _lib.load()
    _need_gpu(x)
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("masked_normalize needs a contiguous float32 tensor")
    if out is None:
        out = torch.empty_like(x)
    stats = torch.empty(lib.oess_masked_stats_doubles(1), dtype=torch.float64, device=x.device)
    _lib.check(lib.oess_masked_normalize_f32(_ptr(x), _ptr(out), x.numel(), _ptr(stats), _stream()),
               "oess_masked_normalize_f32")
    return out


def masked_normalize_slice(x, c0, cs, out=None):
    """Normalise the channel slice x[:, c0:c0+cs] of a contiguous [B, C, H, W] float32 tensor."""
    lib = _lib.load()
    _need_gpu(x)
    if x.dtype != torch.float32 or not x.is_contiguous() or x.ndim != 4:
        raise ValueError("needs contiguous float32 [B, C, H, W]")
    B, Ct, H, W = x.shape
    if out is None:
        out = torch.empty((B, cs, H, W), dtype=torch.float32, device=x.device)
    stats = torch.empty(lib.oess_masked_stats_doubles(1), dtype=torch.float64, device=x.device)
    _lib.check(lib.oess_masked_normalize_slice_f32(_ptr(x), _ptr(out), B, Ct, c0, cs, H * W, _ptr(stats), _stream()),
               "oess_masked_normalize_slice_f32")
    return out


# ------------------------------------------------------------------------------------------ K7
class _SegmentMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat_pm, ids, pps, sps, S):
        lib = _lib.load()
        P, Cf = feat_pm.shape
        k = torch.empty((S, Cf), dtype=torch.float32, device=feat_pm.device)
        cnt = torch.empty((S,), dtype=torch.float32, device=feat_pm.device)
        is_bf16 = int(feat_pm.dtype == torch.bfloat16)
        ws_bytes = lib.oess_segment_mean_fwd_workspace_bytes(S, Cf)           # 96-bit fixed-point accumulators (deterministic sums)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=feat_pm.device)
        _lib.check(lib.oess_segment_mean_fwd(_ptr(feat_pm), is_bf16, _ptr(ids), P, pps, sps, Cf, S, _ptr(k), _ptr(cnt),
                                             _ptr(ws), ws_bytes, _stream()), "oess_segment_mean_fwd")
        ctx.save_for_backward(ids, cnt)
        ctx.meta = (P, Cf, pps, sps, S, feat_pm.dtype)
        ctx.mark_non_differentiable(cnt)
        return k, cnt

    @staticmethod
    def backward(ctx, gk, _gcnt):
        lib = _lib.load()
        ids, cnt = ctx.saved_tensors
        P, Cf, pps, sps, S, dtype = ctx.meta
        gk = gk.contiguous().float()
        gfeat = torch.empty((P, Cf), dtype=dtype, device=gk.device)
        table = torch.empty((S, Cf), dtype=dtype, device=gk.device)          # gk / (count + 1e-6), formed once per call
        _lib.check(lib.oess_segment_mean_bwd(_ptr(gk), _ptr(cnt), _ptr(ids), P, pps, sps, Cf, S, _ptr(gfeat),
                                             int(dtype == torch.bfloat16), _ptr(table), table.numel() * table.element_size(),
                                             _stream()), "oess_segment_mean_bwd")
        return gfeat, None, None, None, None


def segment_mean(feat_pm, ids, pixels_per_sample, superpixel_size, S):
    """Superpixel scatter-mean.  feat_pm: pixel-major [P, Cf] (fp32/bf16); ids: [P] int64 raw ids."""
    _need_gpu(feat_pm, ids)
    if feat_pm.dtype not in (torch.float32, torch.bfloat16) or not feat_pm.is_contiguous() or feat_pm.ndim != 2:
        raise ValueError("feat must be contiguous [P, Cf] float32/bfloat16")
    if feat_pm.shape[1] % 4:
        raise ValueError("channel count must be a multiple of 4")
    ids = ids.reshape(-1).contiguous().to(torch.int64)
    return _SegmentMean.apply(feat_pm, ids, int(pixels_per_sample), int(superpixel_size), int(S))[0]


def superpixel_pool(feat_nchw, superpixels, superpixel_size, S=None, with_count=False):
    """Drop-in for the inline block of training/pretrain_trainer.py:445-465.  feat_nchw is logically
    B x C x H x W (any memory format; channels_last is free), superpixels B x H x W int64.
    with_count: also the fp32 pixel count of every row (not differentiable)."""
    if hasattr(feat_nchw, 'pool') and hasattr(feat_nchw, 'materialize'):      # PointwiseFeature / UpsampledNormalizedFeature
        return feat_nchw.pool(superpixels, superpixel_size, S)
    B, C, H, W = feat_nchw.shape
    if S is None:   # data-dependent size exactly like sparse_coo_tensor (costs one device sync)
        off = torch.arange(0, B * superpixel_size, superpixel_size, device=superpixels.device)[:, None, None]
        S = int((superpixels + off).max().item()) + 1
    pm = feat_nchw.permute(0, 2, 3, 1).contiguous().view(B * H * W, C)
    if not with_count:
        return segment_mean(pm, superpixels, H * W, superpixel_size, S)
    _need_gpu(pm, superpixels)
    ids = superpixels.reshape(-1).contiguous().to(torch.int64)
    return _SegmentMean.apply(pm, ids, H * W, int(superpixel_size), int(S))


class PointwiseFeature:
    """A full-resolution feature map that is a 1x1 convolution of `x` (SemSegE2VID's 256-channel `x_ch256 = decoder_ch256(x)`,
    models/style_networks.py:166), kept as the pair (x, conv) because its only consumer in the pre-training step is the
    superpixel mean of training/pretrain_trainer.py:445-465 -- and the mean commutes with the per-pixel affine map:
        sum_p (W x_p + b) / (n + 1e-6)  =  W (sum_p x_p / (n + 1e-6)) + b n / (n + 1e-6).
    Pooling the 32 input channels and applying the convolution to the S pooled rows gives the reference's result without the
    8 x 256 x 440 x 640 tensor (1.15 GB in bf16), its convolution forward / input- / weight-gradient passes and the 256-channel
    scatter / gather -- and without that tensor's bf16 rounding.  `materialize()` is the tensor itself for any other consumer."""

    def __init__(self, x, conv):
        self.x, self.conv = x, conv

    @property
    def shape(self):
        B, _, H, W = self.x.shape
        return torch.Size((B, self.conv.weight.shape[0], H, W))

    def materialize(self):
        return self.conv(self.x)

    def pool(self, superpixels, superpixel_size, S=None):
        k_in, cnt = superpixel_pool(self.x, superpixels, superpixel_size, S, with_count=True)
        w = self.conv.weight.flatten(1).float()                                   # Cout x Cin
        k = k_in @ w.t()
        if self.conv.bias is not None:
            k = k + (cnt / (cnt + 1e-6)).unsqueeze(1) * self.conv.bias.float()
        return k


# ------------------------------------------------------------------------------------------ K9
class _TaskLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, K, ignore_index, flags, strides):
        lib = _lib.load()
        P, pps, sb, sp, sc = strides
        sums = torch.empty(lib.oess_task_loss_sums_doubles(K), dtype=torch.float64, device=logits.device)   # totals + partial rows
        loss = torch.empty(3, dtype=torch.float32, device=logits.device)
        is_bf16 = int(logits.dtype == torch.bfloat16)
        _lib.check(lib.oess_task_loss_fwd(_ptr(logits), is_bf16, _ptr(target), P, pps, sb, sp, sc, K, ignore_index,
                                          flags, _ptr(sums), _ptr(loss), _stream()), "oess_task_loss_fwd")
        ctx.save_for_backward(logits, target, sums)
        ctx.meta = (K, ignore_index, flags, strides)
        return loss[0], loss[1:].clone()

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        lib = _lib.load()
        logits, target, sums = ctx.saved_tensors
        K, ignore_index, flags, (P, pps, sb, sp, sc) = ctx.meta
        grad = torch.empty_like(logits)
        gdev = g_total.reshape(1).float().contiguous()          # device scalar: no host sync
        _lib.check(lib.oess_task_loss_bwd(_ptr(logits), int(logits.dtype == torch.bfloat16), _ptr(target), P, pps, sb,
                                          sp, sc, K, ignore_index, flags, _ptr(sums), 1.0, _ptr(gdev), _ptr(grad),
                                          int(grad.dtype == torch.bfloat16), _stream()), "oess_task_loss_bwd")
        return grad, None, None, None, None, None


def task_loss(logits, target, num_classes, ignore_index=255, losses=("dice", "cross_entropy")):
    """TaskLoss (Dice + CE).  logits: logically B x K x H x W, contiguous either as NCHW or as
    channels_last (NHWC); float32 or bfloat16.  Returns (total, [dice, ce])."""
    _need_gpu(logits, target)
    B, K, H, W = logits.shape
    if K != num_classes:
        raise ValueError("logits channel count != num_classes")
    if logits.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("logits must be float32 or bfloat16")
    if logits.is_contiguous():
        strides = (B * H * W, H * W, K * H * W, 1, H * W)
    elif logits.is_contiguous(memory_format=torch.channels_last):
        strides = (B * H * W, H * W, H * W * K, K, 1)
    else:
        logits = logits.contiguous()
        strides = (B * H * W, H * W, K * H * W, 1, H * W)
    target = target.reshape(-1).contiguous().to(torch.int64)
    flags = (1 if "dice" in losses else 0) | (2 if "cross_entropy" in losses else 0)
    return _TaskLoss.apply(logits, target, K, int(ignore_index), flags, strides)


# ------------------------------------------------------------------------------------------ a16 consistency losses
def _loss_scratch(device):
    lib = _lib.load()
    return torch.empty(lib.oess_loss_partials_bytes() // 8, dtype=torch.float64, device=device)


class _L1Mean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        lib = _lib.load()
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        _lib.check(lib.oess_l1_mean_fwd(_ptr(a), _ptr(b), a.numel(), int(a.dtype == torch.bfloat16), _ptr(_loss_scratch(a.device)),
                                        _ptr(loss), _stream()), "oess_l1_mean_fwd")
        ctx.save_for_backward(a, b)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        a, b = ctx.saved_tensors
        ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(b) if b is not None and ctx.needs_input_grad[1] else None
        if ga is None and gb is None:
            return None, None
        gdev = g.reshape(1).float().contiguous()
        _lib.check(lib.oess_l1_mean_bwd(_ptr(a), _ptr(b), a.numel(), int(a.dtype == torch.bfloat16), _ptr(gdev),
                                        None if ga is None else _ptr(ga), None if gb is None else _ptr(gb), _stream()),
                   "oess_l1_mean_bwd")
        return ga, gb


def l1_mean(a, b):
    """nn.L1Loss()(a, b) (training/openess_trainer.py:497).  a, b: same shape, dtype (fp32 / bf16) and memory layout.
    Two UpsampledFeature operands of the same geometry (DeepLabV3.lazy_feats): bilinear upsampling is linear, so
    |up(a) - up(b)| = |up(a - b)| -- the difference is formed on the low-resolution maps in fp32 and ONE full-resolution tensor
    is written, read, and differentiated instead of two."""
    if isinstance(a, UpsampledFeature) and isinstance(b, UpsampledFeature):
        if a.size != b.size or a.align_corners != b.align_corners or a.x.shape != b.x.shape:
            raise ValueError("l1_mean: upsampled operands differ in geometry")
        d = (a.x.float() - b.x.float()).to(a.x.dtype)
        u = bilinear_resize(d, size=a.size, align_corners=a.align_corners)
        return _L1Mean.apply(u if (u.is_contiguous() or u.is_contiguous(memory_format=torch.channels_last)) else u.contiguous(), None)
    if isinstance(a, UpsampledFeature):
        a = a.materialize()
    if isinstance(b, UpsampledFeature):
        b = b.materialize()
    _need_gpu(a, b)
    if a.shape != b.shape or a.dtype != b.dtype or a.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("l1_mean: operands must share shape and dtype (float32 or bfloat16)")
    if a.stride() != b.stride() or not (a.is_contiguous() or a.is_contiguous(memory_format=torch.channels_last)):
        a, b = a.contiguous(), b.contiguous()
    return _L1Mean.apply(a, b)


class _CosMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, eps):
        lib = _lib.load()
        (B, H, W, C), sa, sb = a.shape, _pix_stride(a), _pix_stride(b)
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        _lib.check(lib.oess_cosine_mean_fwd(_ptr(a), sa, _ptr(b), sb, B * H * W, C, int(a.dtype == torch.bfloat16), eps,
                                            _ptr(_loss_scratch(a.device)), _ptr(loss), _stream()), "oess_cosine_mean_fwd")
        ctx.save_for_backward(a, b)
        ctx.eps = eps
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        a, b = ctx.saved_tensors
        (B, H, W, C), sa, sb = a.shape, _pix_stride(a), _pix_stride(b)
        ga = torch.empty((B, H, W, C), dtype=a.dtype, device=a.device) if ctx.needs_input_grad[0] else None
        gb = torch.empty((B, H, W, C), dtype=b.dtype, device=b.device) if ctx.needs_input_grad[1] else None
        gdev = g.reshape(1).float().contiguous()
        _lib.check(lib.oess_cosine_mean_bwd(_ptr(a), sa, _ptr(b), sb, B * H * W, C, int(a.dtype == torch.bfloat16), ctx.eps,
                                            _ptr(gdev), None if ga is None else _ptr(ga), C, None if gb is None else _ptr(gb), C,
                                            _stream()), "oess_cosine_mean_bwd")
        return ga, gb, None


def cosine_mean_loss(a, b, eps=1e-8):
    """mean(1 - F.cosine_similarity(a, b, dim=1)) (training/openess_trainer.py:501) for logical B x C x H x W tensors;
    computed on NHWC views (a channels_last input is used in place, an NCHW one is re-laid out once)."""
    _need_gpu(a, b)
    if a.shape != b.shape or a.dtype != b.dtype or a.dtype not in (torch.float32, torch.bfloat16) or a.ndim != 4:
        raise ValueError("cosine_mean_loss: operands must be 4-D with the same shape and dtype (float32 or bfloat16)")
    an, bn = a.permute(0, 2, 3, 1), b.permute(0, 2, 3, 1)
    if an.stride(3) != 1 or not _uniform_pix_stride(an):
        an = an.contiguous()
    if bn.stride(3) != 1 or not _uniform_pix_stride(bn):
        bn = bn.contiguous()
    return _CosMean.apply(an, bn, float(eps))


def _pix_stride(x):
    """Pixel stride (elements) of an NHWC view; size-1 dims carry arbitrary strides in torch, so skip them."""
    B, H, W, C = x.shape
    if W > 1:
        return x.stride(2)
    if H > 1:
        return x.stride(1)
    if B > 1:
        return x.stride(0)
    return C


def _uniform_pix_stride(x):
    B, H, W, C = x.shape
    ps = _pix_stride(x)
    return (H == 1 or W == 1 or x.stride(1) == W * ps) and (B == 1 or x.stride(0) == H * W * ps) and ps >= C


class _NCELoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, k, q, temperature):
        lib = _lib.load()
        S, C = k.shape
        G = torch.empty((S, S), dtype=torch.float32, device=k.device)
        part = torch.empty(S, dtype=torch.float64, device=k.device)
        loss = torch.empty(1, dtype=torch.float32, device=k.device)
        _lib.check(lib.oess_nce_loss_fwd(_ptr(k), _ptr(q), S, C, temperature, _ptr(G), _ptr(part), part.numel() * 8, _ptr(loss),
                                         _stream()), "oess_nce_loss_fwd")
        ctx.save_for_backward(G, k, q)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        G, k, q = ctx.saved_tensors
        S, C = k.shape
        gk = torch.empty_like(k) if ctx.needs_input_grad[0] else None
        gq = torch.empty_like(q) if ctx.needs_input_grad[1] else None
        gdev = g.reshape(1).float().contiguous()
        _lib.check(lib.oess_nce_loss_bwd(_ptr(G), _ptr(k), _ptr(q), S, C, _ptr(gdev), None if gk is None else _ptr(gk),
                                         None if gq is None else _ptr(gq), _stream()), "oess_nce_loss_bwd")
        return gk, gq, None


def nce_loss(k, q, temperature=0.07):
    """NCELoss.forward (utils/loss_functions.py:147-154) on fp32 [S x C] superpixel means."""
    _need_gpu(k, q)
    if k.shape != q.shape or k.ndim != 2:
        raise ValueError("nce_loss: k and q must both be [S, C]")
    return _NCELoss.apply(k.float().contiguous(), q.float().contiguous(), float(temperature))


# ------------------------------------------------------------------------------------------ K11
def confusion_accumulate(pred, label, num_classes, ignore_label, conf):
    lib = _lib.load()
    _need_gpu(pred, label, conf)
    pred = pred.reshape(-1).contiguous().to(torch.int64)
    label = label.reshape(-1).contiguous().to(torch.int64)
    if conf.dtype != torch.int64 or conf.numel() != num_classes * num_classes or not conf.is_contiguous():
        raise ValueError("conf must be contiguous int64 K*K")
    _lib.check(lib.oess_confusion_accumulate(_ptr(pred), _ptr(label), pred.numel(), num_classes, ignore_label,
                                             _ptr(conf), _stream()), "oess_confusion_accumulate")
    return conf


# ------------------------------------------------------------------------------------------ conv
def _nhwc_geom(x):
    """x: bf16 [B, H, W, C] view whose last dim is dense and whose pixels are equally spaced."""
    if x.dtype != torch.bfloat16 or x.ndim != 4 or x.stride(3) != 1:
        raise ValueError("expected bf16 NHWC tensor with dense channels")
    B, H, W, C = x.shape
    ps = x.stride(2)
    if x.stride(1) != W * ps or (B > 1 and x.stride(0) != H * W * ps):
        raise ValueError("NHWC tensor must have uniformly strided pixels")
    return B, H, W, C, ps


_CONV_WS_NEED = {}


def conv_packed_bytes(Cout, Cin, R, S, flip=False):
    return _lib.load().oess_conv2d_packed_bytes(Cout, Cin, R, S, int(flip))


def pack_conv_weight(w, flip=False):
    """Conv2d.weight (OIHW, any float dtype) -> packed bf16 operand for conv2d_nhwc.
    flip=True (1) packs the data-gradient operator of a stride-1 convolution; flip=2 packs the forward operator with
    ConvLSTM gate-interleaved rows (convlstm_fused)."""
    lib = _lib.load()
    _need_gpu(w)
    w = w.detach().float().contiguous()
    Cout, Cin, R, S = w.shape
    nbytes = lib.oess_conv2d_packed_bytes(Cout, Cin, R, S, int(flip))
    packed = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=w.device)
    _lib.check(lib.oess_conv2d_pack_weight(_ptr(w), Cout, Cin, R, S, int(flip), _ptr(packed), nbytes, _stream()),
               "oess_conv2d_pack_weight")
    return packed


def conv2d_nhwc(x, packed, bias, Cout, R, S, stride=1, pad=0, dil=1, relu=False, residual=None, out=None,
                out_f32=False, tile_stats=None, allow_splitk=True):
    """out = act(conv(x, w) + bias [+ residual]) on NHWC bf16 views (channel slices of wider buffers are fine).
    x's channel count must be a multiple of 8 (zero-padded channels; packed weights are zero there).
    allow_splitk=False withholds the scratch of the split-K form (small-M / long-K layers): one pass, for A/B tests."""
    lib = _lib.load()
    _need_gpu(x, packed)
    B, H, W, Cin, ps_in = _nhwc_geom(x)
    Ho = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (S - 1) - 1) // stride + 1
    if out is None:
        out = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    if out.dtype == torch.float32:
        if out.stride(3) != 1:
            raise ValueError("fp32 output must have dense channels")
        ps_out, o_bf16, o_f32 = out.stride(2), None, _ptr(out)
    else:
        _, _, _, _, ps_out = _nhwc_geom(out)
        o_bf16, o_f32 = _ptr(out), None
    if tuple(out.shape) != (B, Ho, Wo, Cout):
        raise ValueError(f"bad output shape {tuple(out.shape)} != {(B, Ho, Wo, Cout)}")
    ps_res = 0
    if residual is not None:
        _, _, _, _, ps_res = _nhwc_geom(residual)
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous()):
        bias = bias.float().contiguous()
    ws, wsn = None, 0
    key = (B, H, W, Cin, Cout, R, S, stride, pad, dil, tile_stats is not None, o_f32 is not None)
    need = _CONV_WS_NEED.get(key)
    if need is None:       # host-only query of the dispatch rules (split-K layers want fp32 slice scratch), cached per geometry
        need = _CONV_WS_NEED[key] = lib.oess_conv2d_fwd_workspace_bytes(*key[:10], int(key[10]), int(key[11]))
    if need and allow_splitk:
        ws = _workspace(need, x.device, tag=("conv_splitk", torch.cuda.current_stream(x.device).cuda_stream))
        wsn = ws.numel()
    _lib.check(lib.oess_conv2d_fwd_bf16(_ptr(x), ps_in, B, H, W, Cin, _ptr(packed), _ptr(bias), Cout, R, S, stride, pad,
                                        dil, int(relu), _ptr(residual), ps_res, o_bf16, o_f32, ps_out, _ptr(tile_stats),
                                        _ptr(ws), wsn, _stream()),
               "oess_conv2d_fwd_bf16")
    return out


def e2vid_head_enc0(x8, head_packed, head_bias, head_relu, enc_packed, enc_bias, enc_relu, out=None):
    """E2VID head (5x5, bins -> 32) + encoder 0's conv (5x5 stride 2, 32 -> 64) in one kernel; the head output stays in LDS.
    x8: NHWC bf16 [B, H, W, 8]; out: NHWC bf16 [B, Ho, Wo, 64] view (may be a channel slice)."""
    lib = _lib.load()
    _need_gpu(x8, head_packed, enc_packed)
    B, H, W, C, ps = _nhwc_geom(x8)
    if C != 8:
        raise ValueError("e2vid_head_enc0: x8 must have 8 channels")
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if out is None:
        out = torch.empty((B, Ho, Wo, 64), dtype=torch.bfloat16, device=x8.device)
    _, _, _, Co, ops = _nhwc_geom(out)
    if tuple(out.shape) != (B, Ho, Wo, 64):
        raise ValueError(f"bad output shape {tuple(out.shape)} != {(B, Ho, Wo, 64)}")
    for b_ in (head_bias, enc_bias):
        if b_ is not None and (b_.dtype != torch.float32 or not b_.is_contiguous()):
            raise ValueError("biases must be contiguous fp32")
    _lib.check(lib.oess_e2vid_head_enc0_bf16(_ptr(x8), ps, B, H, W, _ptr(head_packed), _ptr(head_bias), int(bool(head_relu)),
                                             _ptr(enc_packed), _ptr(enc_bias), int(bool(enc_relu)), _ptr(out), ops, _stream()),
               "oess_e2vid_head_enc0_bf16")
    return out


def e2vid_events_head_enc0(events, c0, cs, normalize, head_packed, head_bias, head_relu, enc_packed, enc_bias, enc_relu, out=None):
    """e2vid_head_enc0 fed from the fp32 event tensor [B, Ctot, H, W]: the slice's EventPreprocessor normalisation and the
    NHWC8 bf16 packing happen inside the kernel (no intermediate tensor at all between the voxel grid and encoder 0's output)."""
    lib = _lib.load()
    _need_gpu(events, head_packed, enc_packed)
    if events.dtype != torch.float32 or not events.is_contiguous() or events.ndim != 4:
        raise ValueError("needs contiguous float32 [B, C, H, W]")
    B, Ct, H, W = events.shape
    if cs > 5 or c0 < 0 or c0 + cs > Ct:
        raise ValueError("bad channel slice")
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if out is None:
        out = torch.empty((B, Ho, Wo, 64), dtype=torch.bfloat16, device=events.device)
    _, _, _, _, ops = _nhwc_geom(out)
    if tuple(out.shape) != (B, Ho, Wo, 64):
        raise ValueError(f"bad output shape {tuple(out.shape)} != {(B, Ho, Wo, 64)}")
    stats = None
    if normalize and Ct > cs and Ct % cs == 0 and c0 % cs == 0:
        stats = masked_stats_slices(events, cs, index=c0 // cs)
    elif normalize:
        stats = torch.empty(lib.oess_masked_stats_doubles(1), dtype=torch.float64, device=events.device)
        _lib.check(lib.oess_masked_stats_slice_f32(_ptr(events), B, Ct, c0, cs, H * W, _ptr(stats), _stream()),
                   "oess_masked_stats_slice_f32")
    _lib.check(lib.oess_e2vid_events_head_enc0_bf16(_ptr(events), B, Ct, c0, cs, H, W, _ptr(stats), int(bool(normalize)),
                                                    _ptr(head_packed), _ptr(head_bias), int(bool(head_relu)), _ptr(enc_packed),
                                                    _ptr(enc_bias), int(bool(enc_relu)), _ptr(out), ops, _stream()),
               "oess_e2vid_events_head_enc0_bf16")
    return out


# ------------------------------------------------------------------------------------------ pointwise
def convlstm_gates(gates, cell, hidden_out, prev_cell_is_zero=False):
    """gates: bf16 NHWC [B,H,W,4C]; cell: fp32 [B,H,W,C] (updated in place); hidden_out: bf16 NHWC view
    [B,H,W,C] (may be a channel slice)."""
    lib = _lib.load()
    _need_gpu(gates, cell, hidden_out)
    B, H, W, C4, gs = _nhwc_geom(gates)
    C = C4 // 4
    _, _, _, Ch, hs = _nhwc_geom(hidden_out)
    if Ch != C or cell.dtype != torch.float32 or not cell.is_contiguous() or cell.numel() != B * H * W * C:
        raise ValueError("convlstm_gates: shape mismatch")
    _lib.check(lib.oess_convlstm_gates_bf16(_ptr(gates), gs, None if prev_cell_is_zero else _ptr(cell), _ptr(cell),
                                            _ptr(hidden_out), hs, B * H * W, C, _stream()), "oess_convlstm_gates_bf16")
    return hidden_out


def convlstm_fused(xh, packed_gates, bias, cell, hidden_out, k, pad, prev_cell_is_zero=False):
    """One ConvLSTM step in one kernel (e2vid/model/submodules.py:199-214): Gates conv over xh = cat(x, h_prev)
    (NHWC bf16) + cell update; `packed_gates` = pack_conv_weight(Gates.weight, flip=2) (gate-interleaved rows).
    cell: fp32 [B,H,W,C] updated in place; hidden_out: bf16 NHWC view [B,H,W,C] that must NOT overlap xh."""
    lib = _lib.load()
    _need_gpu(xh, packed_gates, cell, hidden_out)
    B, H, W, Cin, ps = _nhwc_geom(xh)
    _, _, _, C, hs = _nhwc_geom(hidden_out)
    if cell.dtype != torch.float32 or not cell.is_contiguous() or cell.numel() != B * H * W * C:
        raise ValueError("convlstm_fused: cell must be contiguous fp32 [B,H,W,C]")
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous() or bias.numel() != 4 * C):
        raise ValueError("convlstm_fused: bias must be contiguous fp32 [4C]")
    _lib.check(lib.oess_convlstm_fused_bf16(_ptr(xh), ps, B, H, W, Cin, _ptr(packed_gates), None if bias is None else _ptr(bias),
                                            C, k, k, pad, None if prev_cell_is_zero else _ptr(cell), _ptr(cell),
                                            _ptr(hidden_out), hs, _stream()), "oess_convlstm_fused_bf16")
    return hidden_out


def convlstm_w128_cell_elems(pixels, C):
    """Number of fp32 elements of a w128-tiled cell state for `pixels` = B*H*W pixels and C hidden channels (0: C % 64 != 0)."""
    return int(_lib.load().oess_convlstm_w128_cell_bytes(int(pixels), int(C))) // 4


def convlstm_w128_cell_relayout(src, pixels, C, to_tiled):
    """Cell state between the reference's [pixels, C] fp32 order and the w128-tiled order of oess_convlstm_w128_group_bf16."""
    lib = _lib.load()
    _need_gpu(src)
    n = convlstm_w128_cell_elems(pixels, C)
    if n == 0 or src.dtype != torch.float32 or not src.is_contiguous() or src.numel() != (pixels * C if to_tiled else n):
        raise ValueError("convlstm_w128_cell_relayout: contiguous fp32, C % 64 == 0, matching size")
    dst = torch.empty(n if to_tiled else pixels * C, dtype=torch.float32, device=src.device)
    _lib.check(lib.oess_convlstm_w128_cell_relayout(_ptr(src), _ptr(dst), int(pixels), int(C), int(bool(to_tiled)), _stream()),
               "oess_convlstm_w128_cell_relayout")
    return dst


def convlstm_w128_group(problems):
    """oess_convlstm_w128_group_bf16: `problems` as for convlstm_fused_group, except that `cell` is a flat fp32 tensor of
    convlstm_w128_cell_elems(B*H*W, C) elements in the w128-tiled layout.  Returns False (nothing launched) when the kernel does not
    take one of the problems; the caller then uses convlstm_fused_group on NHWC cells."""
    lib = _lib.load()
    n = len(problems)
    if not 1 <= n <= 3:
        raise ValueError("convlstm_w128_group: 1..3 problems")
    descs = (_lib.ConvLstmDesc * n)()
    flops, keys = 0.0, []
    for d, (xh, packed_gates, bias, cell, hidden_out, k, pad, prev_zero) in zip(descs, problems):
        _need_gpu(xh, packed_gates, cell, hidden_out)
        B, H, W, Cin, ps = _nhwc_geom(xh)
        _, _, _, C, hs = _nhwc_geom(hidden_out)
        if cell.dtype != torch.float32 or not cell.is_contiguous() or cell.numel() != convlstm_w128_cell_elems(B * H * W, C) or C % 64:
            return False
        if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous() or bias.numel() != 4 * C):
            raise ValueError("convlstm_w128_group: bias must be contiguous fp32 [4C]")
        d.in_, d.in_pix_stride, d.B, d.H, d.W, d.Cin = _ptr(xh), ps, B, H, W, Cin
        d.w_packed_gates, d.bias, d.C_hidden, d.R, d.S, d.pad = _ptr(packed_gates), _ptr(bias), C, k, k, pad
        d.prev_cell, d.cell, d.hidden, d.hidden_pix_stride = (None if prev_zero else _ptr(cell)), _ptr(cell), _ptr(hidden_out), hs
        fl = 2.0 * B * H * W * 4 * C * Cin * k * k
        flops += fl
        keys.append(((H, W, Cin, 4 * C, k, 1, "lstm"), fl))
    t = _CONV_TIMING
    if t is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = lib.oess_convlstm_w128_group_bf16(ctypes.addressof(descs), n, _stream())
    if rc == -22:                     # OESS_EINVAL: geometry not taken, nothing launched
        return False
    _lib.check(rc, "oess_convlstm_w128_group_bf16")
    if t is not None:
        e1.record()
        t["events"].append((e0, e1))
        t["flops"] += flops
        t.setdefault("keys", []).append((("group",) + tuple(k_[0] for k_ in keys), flops))
    return True


def convlstm_fused_group(problems):
    """Up to three INDEPENDENT ConvLSTM steps in one launch (oess_convlstm_fused_group_bf16): `problems` = tuples
    (xh, packed_gates, bias, cell, hidden_out, k, pad, prev_cell_is_zero) as for convlstm_fused.  Same results as calling
    convlstm_fused on each (the levels of E2VID's recurrent encoder on the skewed schedule, e2vid/model/unet.py mirror)."""
    lib = _lib.load()
    n = len(problems)
    if not 1 <= n <= 3:
        raise ValueError("convlstm_fused_group: 1..3 problems")
    descs = (_lib.ConvLstmDesc * n)()
    flops, keys = 0.0, []
    for d, (xh, packed_gates, bias, cell, hidden_out, k, pad, prev_zero) in zip(descs, problems):
        _need_gpu(xh, packed_gates, cell, hidden_out)
        B, H, W, Cin, ps = _nhwc_geom(xh)
        _, _, _, C, hs = _nhwc_geom(hidden_out)
        if cell.dtype != torch.float32 or not cell.is_contiguous() or cell.numel() != B * H * W * C:
            raise ValueError("convlstm_fused_group: cell must be contiguous fp32 [B,H,W,C]")
        if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous() or bias.numel() != 4 * C):
            raise ValueError("convlstm_fused_group: bias must be contiguous fp32 [4C]")
        d.in_, d.in_pix_stride, d.B, d.H, d.W, d.Cin = _ptr(xh), ps, B, H, W, Cin
        d.w_packed_gates, d.bias, d.C_hidden, d.R, d.S, d.pad = _ptr(packed_gates), _ptr(bias), C, k, k, pad
        d.prev_cell, d.cell, d.hidden, d.hidden_pix_stride = (None if prev_zero else _ptr(cell)), _ptr(cell), _ptr(hidden_out), hs
        fl = 2.0 * B * H * W * 4 * C * Cin * k * k
        flops += fl
        keys.append(((H, W, Cin, 4 * C, k, 1, "lstm"), fl))
    t = _CONV_TIMING
    if t is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(lib.oess_convlstm_fused_group_bf16(ctypes.addressof(descs), n, _stream()), "oess_convlstm_fused_group_bf16")
    if t is not None:
        e1.record()
        t["events"].append((e0, e1))
        t["flops"] += flops
        t.setdefault("keys", []).append((("group",) + tuple(k_[0] for k_ in keys), flops))
    return [p[4] for p in problems]


def conv5x5s2_group(problems):
    """Up to two INDEPENDENT 5x5 / stride-2 / pad-2 convolutions in one launch (oess_conv5x5s2_group_bf16): `problems` = tuples
    (x NHWC bf16, packed weight, bias | None, Cout, relu, out NHWC bf16 view).  Same results as conv2d_nhwc on each."""
    lib = _lib.load()
    n = len(problems)
    if not 1 <= n <= 2:
        raise ValueError("conv5x5s2_group: 1..2 problems")
    descs = (_lib.ConvS2Desc * n)()
    flops, keys = 0.0, []
    for d, (x, packed, bias, Cout, relu, out) in zip(descs, problems):
        _need_gpu(x, packed, out)
        B, H, W, Cin, ps = _nhwc_geom(x)
        Bo, Ho, Wo, Co, ops = _nhwc_geom(out)
        if (Bo, Ho, Wo, Co) != (B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cout):
            raise ValueError("conv5x5s2_group: output view must be [B, (H-1)//2+1, (W-1)//2+1, Cout]")
        if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous() or bias.numel() < Cout):
            raise ValueError("conv5x5s2_group: bias must be contiguous fp32 [Cout]")
        d.in_, d.in_pix_stride, d.B, d.H, d.W, d.Cin = _ptr(x), ps, B, H, W, Cin
        d.w_packed, d.bias, d.Cout, d.relu, d.out, d.out_pix_stride = _ptr(packed), _ptr(bias), Cout, int(bool(relu)), _ptr(out), ops
        fl = 2.0 * B * Ho * Wo * Cout * Cin * 25
        if Cout > 64:
            flops += fl
            keys.append((H, W, Cin, Cout, 5, 2, 1))
    t = _family_timing() if flops > 0 else None
    if t is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(lib.oess_conv5x5s2_group_bf16(ctypes.addressof(descs), n, _stream()), "oess_conv5x5s2_group_bf16")
    if t is not None:
        e1.record()
        t["events"].append((e0, e1))
        t["flops"] += flops
        t.setdefault("keys", []).append((("group",) + tuple(keys), flops))
    return [p[5] for p in problems]


_SLICE_STATS = {}


def masked_stats_slices(events, cs, index=None):
    """{sum, sumsq, nnz, -} of every cs-channel slice of a contiguous fp32 [B, Ctot, H, W] tensor in ONE launch -> float64
    [Ctot // cs, 4]; with `index` the row of that slice.
    The table is cached ONLY for the access pattern it exists for -- a sub-window loop walking the slices of one tensor in order
    (0, 1, 2, ...): a request is served from the cache iff the tensor's (pointer, version, shape) are unchanged AND `index` is
    exactly the successor of the previous request.  Anything else -- slice 0 (a new loop: the voxelizer rewrites the same
    allocation every batch with the same version), a loop entered in the middle, a repeated or skipped slice, `index=None` --
    recomputes, so stale statistics of a previous batch cannot be served to a caller that does not start at slice 0."""
    lib = _lib.load()
    B, Ct, H, W = events.shape
    key = (events.device.index, torch.cuda.current_stream(events.device).cuda_stream)
    tag = (events.data_ptr(), events._version, tuple(events.shape), cs)
    hit = _SLICE_STATS.get(key)
    if hit is not None and hit[0] == tag and index is not None and index > 0 and index == hit[2]:
        hit[2] = index + 1
        return hit[1][index]
    n = Ct // cs
    buf = torch.empty(lib.oess_masked_stats_doubles(n), dtype=torch.float64, device=events.device)   # totals + partial rows
    _lib.check(lib.oess_masked_stats_slices_f32(_ptr(events), B, Ct, cs, n, H * W, _ptr(buf), _stream()),
               "oess_masked_stats_slices_f32")
    stats = buf[:4 * n].view(n, 4)
    _SLICE_STATS[key] = [tag, stats, (index + 1) if index is not None else -1]
    return stats if index is None else stats[index]


def event_slice_to_nhwc8(events, c0, cs, normalize=True, out=None):
    """events: contiguous fp32 [B, Ctot, H, W]; returns logical [B, 8, H, W] channels_last bf16 holding the
    (optionally EventPreprocessor-normalised) slice events[:, c0:c0+cs], zero padded to 8 channels.
    When the tensor is a whole number of cs-channel slices (the sub-windows of one sample) the statistics of ALL slices are
    formed by one launch at the first request and reused for the others."""
    lib = _lib.load()
    _need_gpu(events)
    if events.dtype != torch.float32 or not events.is_contiguous() or events.ndim != 4:
        raise ValueError("needs contiguous float32 [B, C, H, W]")
    B, Ct, H, W = events.shape
    if out is None:
        out = torch.empty((B, H, W, 8), dtype=torch.bfloat16, device=events.device)
    stats = None
    if normalize and Ct > cs and Ct % cs == 0 and c0 % cs == 0:
        stats = masked_stats_slices(events, cs, index=c0 // cs)
    elif normalize:
        stats = torch.empty(lib.oess_masked_stats_doubles(1), dtype=torch.float64, device=events.device)
        _lib.check(lib.oess_masked_stats_slice_f32(_ptr(events), B, Ct, c0, cs, H * W, _ptr(stats), _stream()),
                   "oess_masked_stats_slice_f32")
    _lib.check(lib.oess_event_slice_to_nhwc8_bf16(_ptr(events), B, Ct, c0, cs, H * W, _ptr(stats), int(normalize),
                                                  _ptr(out), _stream()), "oess_event_slice_to_nhwc8_bf16")
    return out.permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------------ timing hooks
_CONV_TIMING = None


def conv_timing_begin(lstm_only=False):
    """bench.py: bracket every launch of the conv family (Cout > 64) with HIP events on the launch stream; resolved after the
    timed region (no sync inside it).  lstm_only: only the fused-ConvLSTM launches (the dominant kernel) -- an event record is a
    barrier packet of its own, ~2.8 us of idle queue each: 2 x 115 of them are 1.3 ms of a one-stream step."""
    global _CONV_TIMING
    _CONV_TIMING = {"events": [], "flops": 0.0, "lstm_only": bool(lstm_only)}


def _family_timing():
    """the timing record for a non-ConvLSTM launch of the family (None when only the dominant kernel is bracketed)"""
    t = _CONV_TIMING
    return None if (t is None or t.get("lstm_only")) else t


def conv_timing_end():
    global _CONV_TIMING
    t = _CONV_TIMING
    _CONV_TIMING = None
    if not t:
        return None
    torch.cuda.synchronize()
    times = [a.elapsed_time(b) for a, b in t["events"]]
    ms = sum(times)
    by = {}
    for (k, fl), tm in zip(t.get("keys", []), times):
        e = by.setdefault(k, [0, 0.0, 0.0])
        e[0] += 1; e[1] += tm; e[2] += fl
    return {"ms": ms, "flops": t["flops"], "launches": len(t["events"]), "by_shape": by}


_conv2d_nhwc_raw = conv2d_nhwc


def conv2d_nhwc(x, packed, bias, Cout, R, S, stride=1, pad=0, dil=1, relu=False, residual=None, out=None,
                out_f32=False, tile_stats=None, allow_splitk=True):
    t = _family_timing()
    if t is None or Cout <= 64:
        return _conv2d_nhwc_raw(x, packed, bias, Cout, R, S, stride, pad, dil, relu, residual, out, out_f32, tile_stats, allow_splitk)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = _conv2d_nhwc_raw(x, packed, bias, Cout, R, S, stride, pad, dil, relu, residual, out, out_f32, tile_stats, allow_splitk)
    e1.record()
    fl = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * Cout * x.shape[3] * R * S
    t["events"].append((e0, e1))
    t["flops"] += fl
    t.setdefault("keys", []).append(((x.shape[1], x.shape[2], x.shape[3], Cout, R, stride, dil), fl))
    return y


_convlstm_fused_raw = convlstm_fused


def convlstm_fused(xh, packed_gates, bias, cell, hidden_out, k, pad, prev_cell_is_zero=False):
    t = _CONV_TIMING
    if t is None:
        return _convlstm_fused_raw(xh, packed_gates, bias, cell, hidden_out, k, pad, prev_cell_is_zero)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = _convlstm_fused_raw(xh, packed_gates, bias, cell, hidden_out, k, pad, prev_cell_is_zero)
    e1.record()
    Cout = 4 * hidden_out.shape[3]
    fl = 2.0 * xh.shape[0] * xh.shape[1] * xh.shape[2] * Cout * xh.shape[3] * k * k     # the gate convolution's FLOPs only
    t["events"].append((e0, e1))
    t["flops"] += fl
    t.setdefault("keys", []).append(((xh.shape[1], xh.shape[2], xh.shape[3], Cout, k, 1, "lstm"), fl))
    return y


# ------------------------------------------------------------------------------------------ ViT pieces (MaskCLIP tower)
def layer_norm_tokens(x, gamma, beta, eps=1e-6, out=None):
    """nn.LayerNorm over the last axis of a bf16 token matrix [rows, C] (row-strided views are fine)."""
    lib = _lib.load()
    _need_gpu(x)
    if x.dtype != torch.bfloat16 or x.ndim != 2 or x.stride(1) != 1:
        raise ValueError("layer_norm_tokens: bf16 [rows, C] with dense channels")
    rows, C = x.shape
    if out is None:
        out = torch.empty((rows, C), dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.oess_layernorm_bf16(_ptr(x), x.stride(0), rows, C, _ptr(gamma), _ptr(beta), float(eps), _ptr(out), out.stride(0),
                                       _stream()), "oess_layernorm_bf16")
    return out


def attention_d64(qkv, B, L, heads, out=None):
    """softmax(Q K^T / 8) V per head (head dim 64) from nn.MultiheadAttention's packed in_proj output [B*L, 3*heads*64]
    (bf16) -> [B*L, heads*64] (bf16), i.e. the attention output before out_proj."""
    lib = _lib.load()
    _need_gpu(qkv)
    C = heads * 64
    if qkv.dtype != torch.bfloat16 or qkv.ndim != 2 or qkv.shape != (B * L, 3 * C) or qkv.stride(1) != 1:
        raise ValueError("attention_d64: qkv must be bf16 [B*L, 3*heads*64]")
    if out is None:
        out = torch.empty((B * L, C), dtype=torch.bfloat16, device=qkv.device)
    _lib.check(lib.oess_attention_d64_bf16(_ptr(qkv), qkv.stride(0), B, L, heads, 0.125, _ptr(out), out.stride(0), _stream()),
               "oess_attention_d64_bf16")
    return out


# scratch of oess_norm_reduce_finalize_tile_stats (per-slice double sums + last-block tickets; tickets are kept zero by the kernel)
_STATS_SCRATCH = {}


class _StatsScratch:
    def __init__(self, n, device):
        self.buf64 = torch.zeros((64, n), dtype=torch.float64, device=device)      # [32 slices][2][C]
        self.tickets = torch.zeros(((n + 31) // 32,), dtype=torch.int32, device=device)


def _stats_scratch(n, device):
    # stream-ordered reuse is only safe on one stream: key by the current stream as well
    key = (device.index, n, torch.cuda.current_stream(device).cuda_stream)
    sc = _STATS_SCRATCH.get(key)
    if sc is None:
        sc = _STATS_SCRATCH[key] = _StatsScratch(n, device)
    return sc


def _norm_partials(G, ppg, C, device, backward=False):
    """Workspace of the deterministic statistics kernels (per-workgroup partial sums, reduced in a fixed order)."""
    lib = _lib.load()
    nbytes = lib.oess_norm_partials_bytes(G, ppg, C, int(backward))
    ws = _workspace(nbytes, device, tag=("norm_part", torch.cuda.current_stream(device).cuda_stream))
    return ws, ws.numel()


# ------------------------------------------------------------------------------------------ norms / resampling
def _norm_forward(x_nhwc, G, gamma, beta, eps, relu, residual, running=None, momentum=0.1, out=None):
    """Shared BatchNorm(train)/InstanceNorm forward on an NHWC bf16 view.  Returns (out, mean, rstd)."""
    lib = _lib.load()
    B, H, W, C, ps = _nhwc_geom(x_nhwc)
    ppg = (B * H * W) // G
    dev = x_nhwc.device
    stats = torch.empty((6, G, C), dtype=torch.float32, device=dev)        # (unused, unused), mean, rstd, scale, shift
    rm = rv = None
    if running is not None:
        rm, rv = running
    ws, wsn = _norm_partials(G, ppg, C, dev)
    _lib.check(lib.oess_norm_stats_finalize_nhwc_bf16(_ptr(x_nhwc), ps, G, ppg, C, float(eps), _ptr(gamma), _ptr(beta), _ptr(rm),
                                                      _ptr(rv), float(momentum), _ptr(stats[2]), _ptr(stats[3]), _ptr(stats[4]),
                                                      _ptr(stats[5]), _ptr(ws), wsn, _stream()),
               "oess_norm_stats_finalize_nhwc_bf16")
    if out is None:
        out = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=dev)
    _, _, _, _, ops = _nhwc_geom(out)
    rps = 0
    if residual is not None:
        _, _, _, _, rps = _nhwc_geom(residual)
    _lib.check(lib.oess_norm_apply_nhwc_bf16(_ptr(x_nhwc), ps, _ptr(stats[4]), _ptr(stats[5]), _ptr(residual), rps, int(relu),
                                             G, ppg, C, _ptr(out), ops, _stream()), "oess_norm_apply_nhwc_bf16")
    return out, stats[2], stats[3]


def batch_norm_train_nhwc(x_nhwc, bn, relu=False, residual=None):
    """nn.BatchNorm2d in TRAIN mode (batch statistics, running-stat update), inference-only (no autograd):
    the frozen teacher encoder (image_model.py:113-114 freezes it but leaves it in .train())."""
    out, _, _ = _norm_forward(x_nhwc, 1, bn.weight.detach(), bn.bias.detach(), bn.eps, relu, residual,
                              running=(bn.running_mean, bn.running_var),
                              momentum=0.0 if bn.momentum is None else bn.momentum)
    from . import engine as _engine
    _engine.bump_bn_counter(bn)
    return out


class _BatchNormTrainFn(torch.autograd.Function):
    """nn.BatchNorm2d in train mode [+ residual add] [+ ReLU], forward and backward on the norm kernels."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, relu, eps, momentum, running_mean, running_var):
        xn = x.permute(0, 2, 3, 1)
        rn = None if residual is None else residual.permute(0, 2, 3, 1)
        out, mean, rstd = _norm_forward(xn, 1, gamma.detach(), beta.detach(), eps, relu, rn,
                                        running=(running_mean, running_var), momentum=momentum)
        ctx.save_for_backward(x, gamma, mean, rstd, out if relu else None)
        ctx.relu = relu
        ctx.has_res = residual is not None
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, gamma, mean, rstd, out = ctx.saved_tensors
        gy = gy.to(torch.bfloat16)
        if gy.stride(1) != 1:
            gy = gy.contiguous(memory_format=torch.channels_last)
        xn, gn = x.permute(0, 2, 3, 1), gy.permute(0, 2, 3, 1)
        B, H, W, C, xps = _nhwc_geom(xn)
        _, _, _, _, gps = _nhwc_geom(gn)
        dx = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=x.device)
        dres = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=x.device) if (ctx.has_res and ctx.relu) else None
        # d(beta), d(gamma) as two separate allocations: AccumulateGrad steals a whole tensor but has to COPY a view (one tiny
        # copy kernel per BatchNorm parameter and step otherwise)
        dgb = (torch.empty(C, dtype=torch.float32, device=x.device), torch.empty(C, dtype=torch.float32, device=x.device))
        g32 = gamma.detach().float().contiguous()
        ws, wsn = _norm_partials(1, B * H * W, C, x.device, backward=True)
        _lib.check(lib.oess_batchnorm_bwd_nhwc_bf16(_ptr(xn), xps, _ptr(gn), gps, None if out is None else _ptr(out), C, _ptr(mean),
                                                    _ptr(rstd), _ptr(g32), int(ctx.relu), B * H * W, C, _ptr(dgb[0]), _ptr(dgb[1]),
                                                    _ptr(dx), C, None if dres is None else _ptr(dres), C, _ptr(ws), wsn, _stream()),
                   "oess_batchnorm_bwd_nhwc_bf16")
        gres = None
        if ctx.has_res:
            gres = dres.permute(0, 3, 1, 2) if dres is not None else gy       # no ReLU: the residual sees dy itself
        return dx.permute(0, 3, 1, 2), dgb[1].to(gamma.dtype), dgb[0].to(gamma.dtype), gres, None, None, None, None, None


def batch_norm_train(x, bn, relu=False, residual=None):
    """Differentiable nn.BatchNorm2d (train mode: batch statistics + running-stat update) [+ residual] [+ ReLU] on a
    logical-NCHW channels_last bf16 tensor (models/_resnet.py:96-114, models/deeplabv3.py:295-348)."""
    _need_gpu(x)
    if x.dtype != torch.bfloat16 or x.stride(1) != 1:
        x = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    if residual is not None and (residual.dtype != torch.bfloat16 or residual.stride(1) != 1):
        residual = residual.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = _BatchNormTrainFn.apply(x, bn.weight, bn.bias, residual, bool(relu), float(bn.eps),
                                0.0 if bn.momentum is None else float(bn.momentum), bn.running_mean, bn.running_var)
    from . import engine as _engine
    _engine.bump_bn_counter(bn)
    return y


class _InstanceNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, relu, residual, eps):
        xn = x.permute(0, 2, 3, 1)
        rn = None if residual is None else residual.permute(0, 2, 3, 1)
        out, mean, rstd = _norm_forward(xn, x.shape[0], None, None, eps, relu, rn)
        ctx.save_for_backward(x, mean, rstd)
        ctx.relu = relu
        ctx.has_res = residual is not None
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, mean, rstd = ctx.saved_tensors
        gy = gy.to(torch.bfloat16)
        if gy.stride(1) != 1:
            gy = gy.contiguous(memory_format=torch.channels_last)
        xn, gn = x.permute(0, 2, 3, 1), gy.permute(0, 2, 3, 1)
        B, H, W, C, xps = _nhwc_geom(xn)
        _, _, _, _, gps = _nhwc_geom(gn)
        dx = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=x.device)
        s = torch.empty((2, B, C), dtype=torch.float32, device=x.device)
        ws, wsn = _norm_partials(B, H * W, C, x.device, backward=True)
        _lib.check(lib.oess_instnorm_bwd_nhwc_bf16(_ptr(xn), xps, _ptr(gn), gps, _ptr(mean), _ptr(rstd), int(ctx.relu), B, H * W,
                                                   C, _ptr(s[0]), _ptr(s[1]), _ptr(dx), C, _ptr(ws), wsn, _stream()),
                   "oess_instnorm_bwd_nhwc_bf16")
        return dx.permute(0, 3, 1, 2), None, (gy if ctx.has_res else None), None


def instance_norm(x, relu=False, residual=None, eps=1e-5):
    """nn.InstanceNorm2d(affine=False) [+ residual add] [+ ReLU] on a logical-NCHW channels_last bf16 tensor."""
    _need_gpu(x)
    if relu and residual is not None:
        raise NotImplementedError("ReLU after the residual add is not a pattern of the reference (the backward mask "
                                  "would depend on the residual)")
    if x.dtype != torch.bfloat16 or x.stride(1) != 1:
        x = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    if residual is not None and (residual.dtype != torch.bfloat16 or residual.stride(1) != 1):
        residual = residual.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    return _InstanceNormFn.apply(x, bool(relu), residual, float(eps))


class _UpsampleConcatFn(torch.autograd.Function):
    """cat([F.interpolate(x, scale_factor=2, mode='nearest'), skip], 1) written straight into one NHWC buffer."""

    @staticmethod
    def forward(ctx, x, skip):
        lib = _lib.load()
        xn = x.permute(0, 2, 3, 1)
        B, H, W, C, ps = _nhwc_geom(xn)
        Cs = 0 if skip is None else skip.shape[1]
        buf = torch.empty((B, 2 * H, 2 * W, C + Cs), dtype=torch.bfloat16, device=x.device)
        _lib.check(lib.oess_upsample_nearest2x_nhwc_bf16(_ptr(xn), ps, B, H, W, C, _ptr(buf), C + Cs, _stream()),
                   "oess_upsample_nearest2x_nhwc_bf16")
        if skip is not None:
            buf[..., C:] = skip.permute(0, 2, 3, 1)
        ctx.meta = (B, H, W, C, Cs)
        return buf.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        B, H, W, C, Cs = ctx.meta
        g = g.to(torch.bfloat16)
        if g.stride(1) != 1:
            g = g.contiguous(memory_format=torch.channels_last)
        gn = g.permute(0, 2, 3, 1)
        _, _, _, _, gps = _nhwc_geom(gn)
        gx = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=g.device)
        _lib.check(lib.oess_downsample_sum2x_nhwc_bf16(_ptr(gn), gps, B, H, W, C, _ptr(gx), C, _stream()),
                   "oess_downsample_sum2x_nhwc_bf16")
        gskip = None
        if Cs and ctx.needs_input_grad[1]:
            gskip = g[:, C:]
        return gx.permute(0, 3, 1, 2), gskip


def upsample2x_concat(x, skip=None):
    _need_gpu(x)
    if x.dtype != torch.bfloat16 or x.stride(1) != 1:
        x = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    return _UpsampleConcatFn.apply(x, skip)


def zero_insert(x_nhwc, stride, Hz, Wz):
    """z[:, ::stride, ::stride] = x on a zero Hz x Wz grid (NHWC bf16 views): the dilate step of a strided conv's dgrad."""
    lib = _lib.load()
    _need_gpu(x_nhwc)
    B, H, W, C, ps = _nhwc_geom(x_nhwc)
    out = torch.empty((B, Hz, Wz, C), dtype=torch.bfloat16, device=x_nhwc.device)
    _lib.check(lib.oess_zero_insert_nhwc_bf16(_ptr(x_nhwc), ps, B, H, W, C, stride, Hz, Wz, _ptr(out), C, _stream()),
               "oess_zero_insert_nhwc_bf16")
    return out


def bilinear_l2norm(x, scale=4, normalize=True):
    """nn.Upsample(scale, bilinear, align_corners=True) + F.normalize(dim=1), inference form (no autograd)."""
    lib = _lib.load()
    _need_gpu(x)
    xn = x.permute(0, 2, 3, 1)
    B, H, W, C, ps = _nhwc_geom(xn)
    out = torch.empty((B, H * scale, W * scale, C), dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.oess_bilinear_l2norm_nhwc_bf16(_ptr(xn), ps, B, H, W, C, scale, int(normalize), _ptr(out), C, None, _stream()),
               "oess_bilinear_l2norm_nhwc_bf16")
    return out.permute(0, 3, 1, 2)


class _BilinearL2NormTrain(torch.autograd.Function):
    """Differentiable nn.Upsample(scale, bilinear, align_corners=True) + F.normalize(dim=1) as ONE node: the forward is the fused
    inference kernel (which also leaves 1 / |x| per output pixel), so the un-normalised full-resolution tensor -- 1.15 GB at the
    BASELINE size -- is never written or re-read; the backward is the L2 adjoint on (y, g, 1 / |x|) followed by the
    deterministic two-pass bilinear adjoint."""

    @staticmethod
    def forward(ctx, x, scale):
        lib = _lib.load()
        xn = x.permute(0, 2, 3, 1)
        B, H, W, C, ps = _nhwc_geom(xn)
        y = torch.empty((B, H * scale, W * scale, C), dtype=torch.bfloat16, device=x.device)
        inv = torch.empty(B * H * scale * W * scale, dtype=torch.float32, device=x.device)
        _lib.check(lib.oess_bilinear_l2norm_nhwc_bf16(_ptr(xn), ps, B, H, W, C, scale, 1, _ptr(y), C, _ptr(inv), _stream()),
                   "oess_bilinear_l2norm_nhwc_bf16")
        ctx.save_for_backward(y, inv)
        ctx.meta = (B, H, W, C, scale)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        y, inv = ctx.saved_tensors
        B, H, W, C, scale = ctx.meta
        Ho, Wo = H * scale, W * scale
        gn = _nhwc_any(g.to(torch.bfloat16))
        gup = torch.empty_like(y)
        _lib.check(lib.oess_l2norm_nhwc_bwd(_ptr(y), C, _ptr(gn), _pix_stride(gn), _ptr(inv), B * Ho * Wo, C, 1, 1e-12, _ptr(gup), C,
                                            _stream()), "oess_l2norm_nhwc_bwd")
        gin = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=g.device)
        nbytes = lib.oess_resize_bilinear_bwd_workspace_bytes(B, W, C, Ho)
        ws = _workspace(nbytes, g.device, tag="resize")
        _lib.check(lib.oess_resize_bilinear_nhwc_bwd(_ptr(gup), C, B, H, W, C, 1, Ho, Wo, 1, _ptr(ws), ws.numel(), _ptr(gin), C, _stream()),
                   "oess_resize_bilinear_nhwc_bwd")
        return gin.permute(0, 3, 1, 2), None


class _BilinearL2NormPool(torch.autograd.Function):
    """k = scatter_mean(F.normalize(nn.Upsample(scale, bilinear, align_corners=True)(x)), superpixels) as ONE node: forward = the
    fused upsample + normalise kernel followed by K7 on its output; backward = ONE pass over the saved normalised map that gathers
    gk / (n + 1e-6) rows from the S x C table, forms the L2 adjoint on the fly and does the x pass of the bilinear adjoint,
    then the y pass -- instead of row gather -> L2 adjoint -> bilinear adjoint through three 1.15 GB tensors."""

    @staticmethod
    def forward(ctx, x, scale, ids, sps, S):
        lib = _lib.load()
        xn = x.permute(0, 2, 3, 1)
        B, H, W, C, ps = _nhwc_geom(xn)
        Ho, Wo = H * scale, W * scale
        y = torch.empty((B, Ho, Wo, C), dtype=torch.bfloat16, device=x.device)
        inv = torch.empty(B * Ho * Wo, dtype=torch.float32, device=x.device)
        _lib.check(lib.oess_bilinear_l2norm_nhwc_bf16(_ptr(xn), ps, B, H, W, C, scale, 1, _ptr(y), C, _ptr(inv), _stream()),
                   "oess_bilinear_l2norm_nhwc_bf16")
        k = torch.empty((S, C), dtype=torch.float32, device=x.device)
        cnt = torch.empty((S,), dtype=torch.float32, device=x.device)
        ws_bytes = lib.oess_segment_mean_fwd_workspace_bytes(S, C)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        _lib.check(lib.oess_segment_mean_fwd(_ptr(y), 1, _ptr(ids), B * Ho * Wo, Ho * Wo, sps, C, S, _ptr(k), _ptr(cnt), _ptr(ws), ws_bytes,
                                             _stream()), "oess_segment_mean_fwd")
        ctx.save_for_backward(y, inv, ids, cnt)
        ctx.meta = (B, H, W, C, scale, sps, S)
        return k

    @staticmethod
    def backward(ctx, gk):
        lib = _lib.load()
        y, inv, ids, cnt = ctx.saved_tensors
        B, H, W, C, scale, sps, S = ctx.meta
        Ho, Wo = H * scale, W * scale
        gk = gk.contiguous().float()
        gin = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=gk.device)
        nbytes = lib.oess_bilinear_l2norm_pool_bwd_workspace_bytes(B, W, C, Ho, S)
        ws = _workspace(nbytes, gk.device, tag="resize")
        _lib.check(lib.oess_bilinear_l2norm_pool_bwd_bf16(_ptr(y), C, _ptr(inv), _ptr(ids), _ptr(gk), _ptr(cnt), sps, S, B, H, W, C, Ho, Wo, 1,
                                                          1e-12, _ptr(ws), ws.numel(), _ptr(gin), C, _stream()),
                   "oess_bilinear_l2norm_pool_bwd_bf16")
        return gin.permute(0, 3, 1, 2), None, None, None, None


class UpsampledNormalizedFeature:
    """DilationFeatureExtractor's output  F.normalize(nn.Upsample(x4, bilinear, align_corners=True)(x))  (models/image_model.py:
    121-143) kept as (x, scale) for a consumer that only pools it over superpixels (training/pretrain_trainer.py:445-465):
    `pool` is the one-node form above; `materialize()` is the full-resolution tensor for anything else."""

    def __init__(self, x, scale):
        self.x, self.scale = x, int(scale)

    @property
    def shape(self):
        B, C, H, W = self.x.shape
        return torch.Size((B, C, H * self.scale, W * self.scale))

    def materialize(self):
        return bilinear_l2norm_train(self.x, self.scale)

    def pool(self, superpixels, superpixel_size, S=None):
        B, C, Ho, Wo = self.shape
        _need_gpu(self.x, superpixels)
        if tuple(superpixels.shape) != (B, Ho, Wo):
            raise ValueError("superpixel map does not match the upsampled feature size")
        if S is None:
            off = torch.arange(0, B * superpixel_size, superpixel_size, device=superpixels.device)[:, None, None]
            S = int((superpixels + off).max().item()) + 1
        ids = superpixels.reshape(-1).contiguous().to(torch.int64)
        return _BilinearL2NormPool.apply(self.x, self.scale, ids, int(superpixel_size), int(S))


class _PoolMatrixMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, matrix, S):
        lib = _lib.load()
        yn = _nhwc_any(y)
        B, h, w, C = yn.shape
        k = torch.empty((S, C), dtype=torch.float32, device=y.device)
        cnt = torch.empty((S,), dtype=torch.float32, device=y.device)
        _lib.check(lib.oess_pool_matrix_fwd(_ptr(matrix), _ptr(yn), _pix_stride(yn), int(yn.dtype == torch.bfloat16), B, h, w, C, S,
                                            _ptr(k), _ptr(cnt), _stream()), "oess_pool_matrix_fwd")
        ctx.save_for_backward(matrix)
        ctx.meta = (B, h, w, C, S, yn.dtype)
        return k

    @staticmethod
    def backward(ctx, gk):
        lib = _lib.load()
        (matrix,) = ctx.saved_tensors
        B, h, w, C, S, dtype = ctx.meta
        gk = gk.contiguous().float()
        gy = torch.empty((B, h, w, C), dtype=dtype, device=gk.device)
        _lib.check(lib.oess_pool_matrix_bwd(_ptr(matrix), _ptr(gk), B, h, w, C, S, _ptr(gy), C, int(dtype == torch.bfloat16), _stream()),
                   "oess_pool_matrix_bwd")
        return gy.permute(0, 3, 1, 2), None, None


POOL_MATRIX_MAX_BYTES = 256 << 20       # B = 8 at OS16: 18-26 MB; B = 32 at OS8 would be ~1 GB (see pool_matrix)


def pool_matrix(superpixels, in_hw, superpixel_size, S, align_corners=False):
    """Pooling matrix of (bilinear upsample from in_hw to the size of `superpixels`) followed by the superpixel scatter-mean:
    an opaque device buffer for UpsampledFeature.pool / _PoolMatrixMean (2^-40 fixed-point weight sums + pixel counts)."""
    lib = _lib.load()
    _need_gpu(superpixels)
    B, Ho, Wo = superpixels.shape
    h, w = int(in_hw[0]), int(in_hw[1])
    ids = superpixels.reshape(-1).contiguous().to(torch.int64)
    nbytes = lib.oess_pool_matrix_bytes(int(S), B, h, w)
    if nbytes > POOL_MATRIX_MAX_BYTES:
        # the matrix is dense S x (B h w) and both S and the column count grow with B: O(B^2) memset + scan per step.  Above the
        # cap the caller pools the materialised tensor instead (UpsampledFeature.pool does that on a None matrix).
        return None
    m = torch.empty(nbytes, dtype=torch.uint8, device=superpixels.device)
    _lib.check(lib.oess_pool_matrix_build(_ptr(ids), B, Ho, Wo, h, w, int(bool(align_corners)), int(superpixel_size), int(S), _ptr(m), nbytes,
                                          _stream()), "oess_pool_matrix_build")
    return m


class UpsampledFeature:
    """DeepLabv3's `feats = F.interpolate(feats, size=input_shape, mode='bilinear', align_corners=False)` (models/deeplabv3.py:184)
    kept as (low-resolution map, size) for a consumer that only pools it over superpixels (training/pretrain_trainer.py:445-465,
    frame2recon): upsampling and pooling are both linear, so `pool` multiplies the low-resolution map with the pooling matrix of
    the superpixel map (pool_matrix) -- forward and backward without any full-resolution tensor.  `materialize()` is the tensor."""

    def __init__(self, x, size, align_corners=False):
        self.x, self.size, self.align_corners = x, (int(size[0]), int(size[1])), bool(align_corners)

    @property
    def shape(self):
        B, C = self.x.shape[:2]
        return torch.Size((B, C, self.size[0], self.size[1]))

    def materialize(self):
        return bilinear_resize(self.x, size=self.size, align_corners=self.align_corners)

    def pool(self, superpixels, superpixel_size, S=None, matrix=None):
        B, C, Ho, Wo = self.shape
        _need_gpu(self.x, superpixels)
        if tuple(superpixels.shape) != (B, Ho, Wo):
            raise ValueError("superpixel map does not match the upsampled feature size")
        if S is None:
            off = torch.arange(0, B * superpixel_size, superpixel_size, device=superpixels.device)[:, None, None]
            S = int((superpixels + off).max().item()) + 1
        if matrix is None:
            matrix = pool_matrix(superpixels, self.x.shape[2:], superpixel_size, S, self.align_corners)
        if matrix is None:              # above POOL_MATRIX_MAX_BYTES: the reference's own order, upsample then scatter-mean (K7)
            return superpixel_pool(self.materialize(), superpixels, superpixel_size, S=S)
        return _PoolMatrixMean.apply(self.x, matrix, int(S))


def bilinear_l2norm_train(x, scale=4):
    """Differentiable form of bilinear_l2norm (normalisation on) for a channels_last bf16 tensor with C % 64 == 0."""
    _need_gpu(x)
    if x.dtype != torch.bfloat16 or x.shape[1] % 64 or x.shape[1] > 512:
        raise ValueError("bilinear_l2norm_train needs bf16 with C % 64 == 0, C <= 512")
    return _BilinearL2NormTrain.apply(x, int(scale))


def _nhwc_any(x):
    """Logical NCHW fp32/bf16 tensor -> NHWC view with dense channels and uniformly strided pixels (copy if needed)."""
    if x.dtype not in (torch.float32, torch.bfloat16) or x.ndim != 4:
        raise ValueError("expected a 4-D float32 / bfloat16 tensor")
    xn = x.permute(0, 2, 3, 1)
    if xn.stride(3) != 1 or not _uniform_pix_stride(xn):
        xn = xn.contiguous()
    return xn


class _BilinearResize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo, align):
        lib = _lib.load()
        xn = _nhwc_any(x)
        B, H, W, C = xn.shape
        out = torch.empty((B, Ho, Wo, C), dtype=x.dtype, device=x.device)
        _lib.check(lib.oess_resize_bilinear_nhwc_fwd(_ptr(xn), _pix_stride(xn), B, H, W, C, int(x.dtype == torch.bfloat16), Ho, Wo,
                                                     int(align), _ptr(out), C, _stream()), "oess_resize_bilinear_nhwc_fwd")
        ctx.meta = (B, H, W, C, Ho, Wo, int(align))
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        B, H, W, C, Ho, Wo, align = ctx.meta
        gn = _nhwc_any(g)
        gin = torch.empty((B, H, W, C), dtype=g.dtype, device=g.device)
        nbytes = lib.oess_resize_bilinear_bwd_workspace_bytes(B, W, C, Ho)
        ws = _workspace(nbytes, g.device, tag="resize")
        _lib.check(lib.oess_resize_bilinear_nhwc_bwd(_ptr(gn), _pix_stride(gn), B, H, W, C, int(g.dtype == torch.bfloat16), Ho, Wo,
                                                     align, _ptr(ws), ws.numel(), _ptr(gin), C, _stream()),
                   "oess_resize_bilinear_nhwc_bwd")
        return gin.permute(0, 3, 1, 2), None, None, None


def bilinear_resize(x, size=None, scale_factor=None, align_corners=False):
    """F.interpolate(x, size | scale_factor, mode='bilinear', align_corners) for logical B x C x H x W fp32 / bf16
    tensors, differentiable.  Result is logically NCHW, physically channels_last."""
    _need_gpu(x)
    if (size is None) == (scale_factor is None):
        raise ValueError("give exactly one of size / scale_factor")
    if size is None:
        if int(scale_factor) != scale_factor:
            raise ValueError("integer scale factors only (a fractional factor changes ATen's index rule)")
        size = (x.shape[2] * int(scale_factor), x.shape[3] * int(scale_factor))
    return _BilinearResize.apply(x, int(size[0]), int(size[1]), bool(align_corners))


class _L2Normalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        lib = _lib.load()
        xn = _nhwc_any(x)
        B, H, W, C = xn.shape
        y = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
        inv = torch.empty(B * H * W, dtype=torch.float32, device=x.device)
        _lib.check(lib.oess_l2norm_nhwc_fwd(_ptr(xn), _pix_stride(xn), B * H * W, C, int(x.dtype == torch.bfloat16), eps, _ptr(y), C,
                                            _ptr(inv), _stream()), "oess_l2norm_nhwc_fwd")
        ctx.save_for_backward(y, inv)
        ctx.eps = eps
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        y, inv = ctx.saved_tensors
        B, H, W, C = y.shape
        gn = _nhwc_any(g.to(y.dtype))
        gx = torch.empty_like(y)
        _lib.check(lib.oess_l2norm_nhwc_bwd(_ptr(y), C, _ptr(gn), _pix_stride(gn), _ptr(inv), B * H * W, C,
                                            int(y.dtype == torch.bfloat16), ctx.eps, _ptr(gx), C, _stream()), "oess_l2norm_nhwc_bwd")
        return gx.permute(0, 3, 1, 2), None


def l2_normalize(x, eps=1e-12):
    """F.normalize(x, p=2, dim=1, eps) for logical B x C x H x W fp32 / bf16 tensors, differentiable."""
    _need_gpu(x)
    return _L2Normalize.apply(x, float(eps))


def conv2d_wgrad(x, gy, Cout, Cin, R, S, stride=1, pad=0, dil=1):
    """dW (OIHW fp32) of conv2d_nhwc from NHWC bf16 views x [B,H,W,Cin_x>=Cin] and gy [B,Ho,Wo,Cout]."""
    lib = _lib.load()
    _need_gpu(x, gy)
    B, H, W, Cin_x, xps = _nhwc_geom(x)
    _, Ho, Wo, Cg, gps = _nhwc_geom(gy)
    if Cg != Cout:
        raise ValueError("gy channel count != Cout")
    dw = torch.empty((Cout, Cin, R, S), dtype=torch.float32, device=x.device)
    ws = _workspace(256 << 20, x.device, tag="wgrad")
    _lib.check(lib.oess_conv2d_wgrad_bf16(_ptr(x), xps, B, H, W, Cin_x, _ptr(gy), gps, Cout, Cin, R, S, stride, pad, dil,
                                          _ptr(dw), _ptr(ws), ws.numel(), _stream()), "oess_conv2d_wgrad_bf16")
    return dw


class _GlobalAvgPool(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d(1) on a channels_last bf16 map -> fp32 [B, C, 1, 1] (models/deeplabv3.py ASPPPooling): per-sample
    channel sums straight from the bf16 map by the deterministic statistics kernel (G = B groups) -- no fp32 copy of the map;
    the backward hands autograd the [B, C, 1, 1] quotient EXPANDED over H x W (stride 0, channels-last compatible), so the sum
    with the other ASPP branches' gradients reads it as a broadcast instead of a materialised, NCHW-ordered tensor."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        xn = x.permute(0, 2, 3, 1)
        B, H, W, C, ps = _nhwc_geom(xn)
        st = torch.empty((2, B, C), dtype=torch.float32, device=x.device)
        ws, wsn = _norm_partials(B, H * W, C, x.device)
        _lib.check(lib.oess_norm_stats_nhwc_bf16(_ptr(xn), ps, B, H * W, C, _ptr(st[0]), _ptr(st[1]), _ptr(ws), wsn, _stream()),
                   "oess_norm_stats_nhwc_bf16")
        ctx.meta = (H, W, x.dtype)
        return (st[0] * (1.0 / (H * W))).view(B, C, 1, 1)

    @staticmethod
    def backward(ctx, g):
        H, W, dtype = ctx.meta
        return (g * (1.0 / (H * W))).to(dtype).expand(-1, -1, H, W)


def global_avg_pool(x):
    """Mean over H x W of a logical B x C x H x W channels_last bf16 tensor -> fp32 [B, C, 1, 1], differentiable."""
    _need_gpu(x)
    if x.dtype != torch.bfloat16 or x.stride(1) != 1:
        raise ValueError("global_avg_pool needs a channels_last bf16 tensor")
    return _GlobalAvgPool.apply(x)


class _LinearProbe(torch.autograd.Function):
    """nn.Conv2d(K, K, 1) on the fp32 channels_last logits (models/style_networks.py:169-170, models/deeplabv3.py:186-187):
    per-pixel K x K product; weight / bias gradients from fixed-order double partial rows (bit-repeatable)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        B, K, H, W = x.shape
        y = torch.empty_like(x, memory_format=torch.channels_last)
        w2 = weight.detach().reshape(K, K).contiguous()
        _lib.check(lib.oess_linear_probe_fwd_f32(_ptr(x), _ptr(w2), _ptr(bias.detach()) if bias is not None else None, B * H * W, K,
                                                 _ptr(y), _stream()), "oess_linear_probe_fwd_f32")
        ctx.save_for_backward(x, w2)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, w2 = ctx.saved_tensors
        B, K, H, W = x.shape
        if not _dense_cl(g):
            g = g.contiguous(memory_format=torch.channels_last)
        gx = torch.empty_like(x, memory_format=torch.channels_last) if ctx.needs_input_grad[0] else None
        gw = torch.empty((K, K, 1, 1), dtype=torch.float32, device=x.device)
        gb = torch.empty((K,), dtype=torch.float32, device=x.device)
        nb = lib.oess_linear_probe_partials_bytes(K)
        ws = _workspace(nb, x.device, tag=("linear_probe", torch.cuda.current_stream(x.device).cuda_stream))
        _lib.check(lib.oess_linear_probe_bwd_f32(_ptr(x), _ptr(g), _ptr(w2), B * H * W, K, _ptr(gx) if gx is not None else None,
                                                 _ptr(gw), _ptr(gb), _ptr(ws), nb, _stream()), "oess_linear_probe_bwd_f32")
        return gx, gw, (gb if ctx.has_bias else None)


def _dense_cl(t):
    B, C, H, W = t.shape
    return t.dtype == torch.float32 and t.stride() == (H * W * C, 1, W * C, C)


def linear_probe(x, conv):
    """`conv(x)` for the linear-probe nn.Conv2d(K, K, 1) on fp32 channels_last logits [B, K, H, W], differentiable."""
    _need_gpu(x)
    K = x.shape[1]
    if K > 32 or tuple(conv.weight.shape) != (K, K, 1, 1) or conv.weight.dtype != torch.float32:
        # outside the kernel's range (more than 32 classes, or not the K x K probe): the module's own convolution, said once
        global _PROBE_WARNED
        if not _PROBE_WARNED:
            _PROBE_WARNED = True
            import warnings
            warnings.warn(f"linear_probe: K = {K} / weight {tuple(conv.weight.shape)} is outside the HIP kernel's range "
                          "(K <= 32, K x K x 1 x 1 fp32); using nn.Conv2d for this layer")
        return conv(x)
    if not _dense_cl(x):        # any other layout / dtype of the logits: one re-layout, then the kernel
        x = x.float().contiguous(memory_format=torch.channels_last)
        if not _dense_cl(x):    # degenerate shapes (H * W == 1 ...) where channels_last strides are ambiguous
            x = torch.empty_strided(x.shape, (x.shape[1] * x.shape[2] * x.shape[3], 1, x.shape[3] * x.shape[1], x.shape[1]),
                                    dtype=torch.float32, device=x.device).copy_(x)
    return _LinearProbe.apply(x, conv.weight, conv.bias)


_PROBE_WARNED = False


def channel_sum(x_nhwc):
    """Per-channel sum over all pixels of an NHWC bf16 view (bias gradient) -> fp32 [C]."""
    lib = _lib.load()
    B, H, W, C, ps = _nhwc_geom(x_nhwc)
    st = torch.empty((2, C), dtype=torch.float32, device=x_nhwc.device)
    ws, wsn = _norm_partials(1, B * H * W, C, x_nhwc.device)
    _lib.check(lib.oess_norm_stats_nhwc_bf16(_ptr(x_nhwc), ps, 1, B * H * W, C, _ptr(st[0]), _ptr(st[1]), _ptr(ws), wsn, _stream()),
               "oess_norm_stats_nhwc_bf16")
    return st[0]


def conv_bn_train_nhwc(x, packed, Cout, R, S, stride, pad, dil, bn, relu=False, residual=None, out=None):
    """Bias-free conv + nn.BatchNorm2d in TRAIN mode [+ residual] [+ ReLU], inference form (frozen teacher): the batch
    statistics come from the conv epilogue's fp32 accumulators (no separate statistics pass over the activation).
    `out`: optional NHWC bf16 destination view (e.g. a channel slice of a concat buffer)."""
    lib = _lib.load()
    B, H, W, _, _ = _nhwc_geom(x)
    Ho = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (S - 1) - 1) // stride + 1
    M = B * Ho * Wo
    tiles = (M + 127) // 128
    part = torch.empty((tiles, 2, Cout), dtype=torch.float32, device=x.device)
    y = conv2d_nhwc(x, packed, None, Cout, R, S, stride, pad, dil, tile_stats=part, out=out)
    _, _, _, _, yps = _nhwc_geom(y)
    mom = 0.0 if bn.momentum is None else bn.momentum
    rps = 0
    if residual is not None:
        _, _, _, _, rps = _nhwc_geom(residual)
    from . import engine as _engine
    if tiles <= 320 and Cout % 64 == 0:
        # small map: statistics + apply in ONE launch (every workgroup reduces the partials of its 64 channels itself)
        _lib.check(lib.oess_norm_tile_stats_apply_nhwc_bf16(_ptr(part), tiles, Cout, float(M), float(bn.eps), _ptr(bn.weight.detach()),
                                                            _ptr(bn.bias.detach()), _ptr(bn.running_mean), _ptr(bn.running_var),
                                                            float(mom), None, None, _ptr(y), yps, _ptr(residual), rps, int(relu), M,
                                                            _ptr(y), yps, _stream()), "oess_norm_tile_stats_apply_nhwc_bf16")
        _engine.bump_bn_counter(bn)
        return y
    st = torch.empty((6, 1, Cout), dtype=torch.float32, device=x.device)
    sc = _stats_scratch(Cout, x.device)
    _lib.check(lib.oess_norm_reduce_finalize_tile_stats(_ptr(part), tiles, Cout, _ptr(sc.buf64), _ptr(sc.tickets),
                                                        float(M), float(bn.eps), _ptr(bn.weight.detach()), _ptr(bn.bias.detach()),
                                                        _ptr(bn.running_mean), _ptr(bn.running_var), float(mom), _ptr(st[2]), _ptr(st[3]),
                                                        _ptr(st[4]), _ptr(st[5]), _stream()), "oess_norm_reduce_finalize_tile_stats")
    _lib.check(lib.oess_norm_apply_nhwc_bf16(_ptr(y), yps, _ptr(st[4]), _ptr(st[5]), _ptr(residual), rps, int(relu), 1, M, Cout,
                                             _ptr(y), yps, _stream()), "oess_norm_apply_nhwc_bf16")
    _engine.bump_bn_counter(bn)
    return y


# ------------------------------------------------------------------------------------------ small ops of the DeepLabv3 path (round 5)
class _MaxPool3x3s2(torch.autograd.Function):
    """nn.MaxPool2d(3, stride=2, padding=1) of the ResNet stem on a channels_last bf16 map (models/_resnet.py:124,197 of the
    reference): forward keeps one byte per output element (the winning tap), backward is a gather (every input element written
    once, fixed order)."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        xn = x.permute(0, 2, 3, 1)
        B, H, W, C, ps = _nhwc_geom(xn)
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((B, Ho, Wo, C), dtype=torch.bfloat16, device=x.device)
        need = ctx.needs_input_grad[0]
        idx = torch.empty((B, Ho, Wo, C), dtype=torch.uint8, device=x.device) if need else None
        _lib.check(lib.oess_maxpool3x3s2_fwd_nhwc_bf16(_ptr(xn), ps, B, H, W, C, _ptr(y), C, _ptr(idx), _stream()),
                   "oess_maxpool3x3s2_fwd_nhwc_bf16")
        ctx.idx, ctx.geom = idx, (B, H, W, C)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        B, H, W, C = ctx.geom
        gn = g.permute(0, 2, 3, 1)
        if gn.dtype != torch.bfloat16 or gn.stride(3) != 1 or not _uniform_pix_stride(gn):
            gn = gn.to(torch.bfloat16).contiguous()
        _, _, _, _, gps = _nhwc_geom(gn)
        gx = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=g.device)
        _lib.check(lib.oess_maxpool3x3s2_bwd_nhwc_bf16(_ptr(gn), gps, _ptr(ctx.idx), B, H, W, C, _ptr(gx), C, _stream()),
                   "oess_maxpool3x3s2_bwd_nhwc_bf16")
        return gx.permute(0, 3, 1, 2)


def max_pool_3x3s2(x):
    """MaxPool2d(kernel_size=3, stride=2, padding=1) on a logical NCHW channels_last bf16 tensor with C % 8 == 0."""
    _need_gpu(x)
    if x.dtype != torch.bfloat16 or x.stride(1) != 1 or x.shape[1] % 8:
        raise ValueError("max_pool_3x3s2 needs a channels_last bf16 tensor with C % 8 == 0")
    return _MaxPool3x3s2.apply(x)


_DROPOUT_CALLS = 0


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, offset):
        lib = _lib.load()
        xn = x.permute(0, 2, 3, 1)
        B, H, W, C, ps = _nhwc_geom(xn)
        y = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=x.device)
        _lib.check(lib.oess_dropout_nhwc_bf16(_ptr(xn), ps, _ptr(y), C, B * H * W, C, float(p), seed, offset, _stream()),
                   "oess_dropout_nhwc_bf16")
        ctx.meta = (float(p), seed, offset)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        p, seed, offset = ctx.meta
        gn = g.permute(0, 2, 3, 1)
        if gn.dtype != torch.bfloat16 or gn.stride(3) != 1 or not _uniform_pix_stride(gn):
            gn = gn.to(torch.bfloat16).contiguous()
        B, H, W, C, gps = _nhwc_geom(gn)
        gx = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=g.device)
        _lib.check(lib.oess_dropout_nhwc_bf16(_ptr(gn), gps, _ptr(gx), C, B * H * W, C, p, seed, offset, _stream()),
                   "oess_dropout_nhwc_bf16")
        return gx.permute(0, 3, 1, 2), None, None, None


def dropout(x, p, training=True, owner=None):
    """nn.Dropout(p) on a channels_last bf16 map (models/deeplabv3.py:343): Philox mask keyed by (torch.initial_seed(), a call
    counter, element), recomputed in the backward pass instead of stored.  `owner` (the nn.Dropout module) carries its OWN counter:
    the mask sequence of a model then depends on that model's calls only, not on what other models of the process did;
    without an owner the counter is the process-wide one.  Reproducible under torch.manual_seed."""
    global _DROPOUT_CALLS
    if not training or p <= 0.0:
        return x
    _need_gpu(x)
    if x.dtype != torch.bfloat16 or x.stride(1) != 1 or x.shape[1] % 8:
        raise ValueError("dropout needs a channels_last bf16 tensor with C % 8 == 0")
    if owner is not None:
        calls = getattr(owner, "_oess_dropout_calls", 0) + 1
        owner._oess_dropout_calls = calls
        calls += 1 << 40                                         # own sequence: never collides with the process-wide counter
    else:
        _DROPOUT_CALLS += 1
        calls = _DROPOUT_CALLS
    return _Dropout.apply(x, float(p), int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF, calls)


class _ASPPPoolBranch(torch.autograd.Function):
    """ASPPPooling (models/deeplabv3.py:305-316) as one node: AdaptiveAvgPool2d(1) -> 1x1 conv -> BatchNorm2d(train) over the B
    pooled vectors -> ReLU -> the 1x1 map broadcast back over H x W (bilinear from 1x1 == broadcast).  Forward: the per-sample
    channel sums of the bf16 map (statistics kernel) + one B-row GEMV/BatchNorm kernel; backward: per-sample sums of the incoming
    gradient slice, BatchNorm + GEMV adjoints, and the map's gradient as a stride-0 expanded [B, C, 1, 1] quotient."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, run_mean, run_var, momentum, eps):
        lib = _lib.load()
        xn = x.permute(0, 2, 3, 1)
        B, H, W, C, ps = _nhwc_geom(xn)
        Cout = weight.shape[0]
        st = torch.empty((2, B, C), dtype=torch.float32, device=x.device)
        ws, wsn = _norm_partials(B, H * W, C, x.device)
        _lib.check(lib.oess_norm_stats_nhwc_bf16(_ptr(xn), ps, B, H * W, C, _ptr(st[0]), _ptr(st[1]), _ptr(ws), wsn, _stream()),
                   "oess_norm_stats_nhwc_bf16")
        sums = st[0]                                  # per-sample channel sums; pooled = sums / (H W) is applied inside the kernels
        w2 = weight.detach().reshape(Cout, C)
        if w2.dtype != torch.float32 or not w2.is_contiguous():
            w2 = w2.float().contiguous()
        buf = torch.empty((2 * B + 2, Cout), dtype=torch.float32, device=x.device)      # y_pre | z | mean, rstd
        y_pre, z, stat = buf[:B], buf[B:2 * B], buf[2 * B:]
        zb = torch.empty((B, Cout), dtype=torch.bfloat16, device=x.device)
        _lib.check(lib.oess_aspp_pool_fwd_f32(_ptr(sums), 1.0 / (H * W), _ptr(w2), _ptr(gamma.detach()), _ptr(beta.detach()), _ptr(run_mean),
                                              _ptr(run_var), float(momentum), float(eps), B, C, Cout, _ptr(y_pre), _ptr(stat), _ptr(z),
                                              _ptr(zb), _stream()), "oess_aspp_pool_fwd_f32")
        ctx.save_for_backward(sums, w2, gamma.detach(), buf)
        ctx.geom = (B, H, W, C, Cout)
        return zb.view(B, Cout, 1, 1).expand(B, Cout, H, W)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        sums, w2, gamma, buf = ctx.saved_tensors
        B, H, W, C, Cout = ctx.geom
        y_pre, z, stat = buf[:B], buf[B:2 * B], buf[2 * B:]
        gn = g.permute(0, 2, 3, 1)
        if gn.dtype != torch.bfloat16 or gn.stride(3) != 1 or not _uniform_pix_stride(gn):
            gn = gn.to(torch.bfloat16).contiguous()
        _, _, _, _, gps = _nhwc_geom(gn)
        gs = torch.empty((2, B, Cout), dtype=torch.float32, device=g.device)
        ws, wsn = _norm_partials(B, H * W, Cout, g.device)
        _lib.check(lib.oess_norm_stats_nhwc_bf16(_ptr(gn), gps, B, H * W, Cout, _ptr(gs[0]), _ptr(gs[1]), _ptr(ws), wsn, _stream()),
                   "oess_norm_stats_nhwc_bf16")
        out = torch.empty((B * Cout + Cout * C + 2 * Cout,), dtype=torch.float32, device=g.device)
        dy = out[:B * Cout]
        gw = out[B * Cout:B * Cout + Cout * C].view(Cout, C, 1, 1)
        gg = out[B * Cout + Cout * C:B * Cout + Cout * C + Cout]
        gb = out[B * Cout + Cout * C + Cout:]
        gp = torch.empty((B, C), dtype=torch.bfloat16, device=g.device) if ctx.needs_input_grad[0] else None
        _lib.check(lib.oess_aspp_pool_bwd_f32(_ptr(gs[0]), _ptr(sums), 1.0 / (H * W), _ptr(w2), _ptr(gamma), _ptr(y_pre), _ptr(stat), _ptr(z),
                                              B, C, Cout, _ptr(dy), _ptr(gw), _ptr(gg), _ptr(gb), _ptr(gp), _stream()), "oess_aspp_pool_bwd_f32")
        gx = gp.view(B, C, 1, 1).expand(B, C, H, W) if gp is not None else None
        return gx, gw, gg, gb, None, None, None, None


def aspp_pool_branch(x, conv, bn):
    """ASPPPooling.forward for a channels_last bf16 map in TRAIN mode (batch statistics over the B pooled vectors)."""
    _need_gpu(x)
    B = x.shape[0]
    if x.dtype != torch.bfloat16 or x.stride(1) != 1 or x.shape[1] % 8 or x.shape[1] > 2048 or not 2 <= B <= 16:
        raise ValueError("aspp_pool_branch needs a channels_last bf16 map with C % 8 == 0, C <= 2048 and 2 <= B <= 16")
    from . import engine as _engine
    y = _ASPPPoolBranch.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                              0.0 if bn.momentum is None else float(bn.momentum), float(bn.eps))
    _engine.bump_bn_counter(bn)
    return y
