"""NHWC / bf16 execution helpers shared by the host-side model mirrors.

Convention: inside the engine every activation is a torch tensor that is LOGICALLY B x C x H x W
(so module signatures read like the reference's) but PHYSICALLY channels_last bf16, i.e. NHWC with a
uniform pixel stride; `nhwc(x)` is a free view.  Convolutions run on the hand-written MFMA kernel
(openess_amd/csrc/conv_fwd.hip); master weights stay fp32 nn.Parameters and are packed to the
kernel's bf16 operand format lazily (re-packed when the parameter's version counter changes).
"""
import weakref

import torch
import torch.nn.functional as F

from . import hip


def nhwc(x):
    """Logical NCHW tensor (channels_last or a channel slice of one) -> [B, H, W, C] view."""
    return x.permute(0, 2, 3, 1)


def from_nhwc(y):
    return y.permute(0, 3, 1, 2)


def to_cl_bf16(x, pad_to=8):
    """Any NCHW float tensor -> channels_last bf16 with the channel count padded (zeros) to a multiple
    of `pad_to` (the conv kernel gathers 16-byte = 8-channel chunks)."""
    B, C, H, W = x.shape
    Cp = (C + pad_to - 1) // pad_to * pad_to
    if Cp == C and x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last):
        return x
    if Cp == C:
        return x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    if Cp == 8 and pad_to == 8 and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and not x.requires_grad:
        # images / event slices (C <= 8, NCHW fp32): ONE kernel for the NCHW -> NHWC8 re-layout + bf16 + zero padding
        return hip.event_slice_to_nhwc8(x, 0, C, normalize=False)
    out = torch.zeros((B, H, W, Cp), dtype=torch.bfloat16, device=x.device)
    out[..., :C] = x.permute(0, 2, 3, 1)
    return from_nhwc(out)


def empty_cl(B, C, H, W, device, dtype=torch.bfloat16):
    return from_nhwc(torch.empty((B, H, W, C), dtype=dtype, device=device))


def zeros_cl(B, C, H, W, device, dtype=torch.bfloat16):
    return from_nhwc(torch.zeros((B, H, W, C), dtype=dtype, device=device))


class PackedWeight:
    """bf16 packed operand of a conv weight (optionally with an eval-mode BatchNorm folded in),
    cached per (parameter versions).

    Group refresh: after an optimiser step EVERY trainable conv weight of a model is stale at once, and packing them one by one
    is ~125 launches of ~6 us per frame2recon step (53 forward operands, 53 data-gradient operands, ...).  Plain weights
    (fp32, contiguous, no folded BatchNorm, no channel padding) therefore register here; the first stale one found by get()
    repacks ALL stale registered weights -- forward operands and the data-gradient operands that exist -- with ONE launch
    (oess_conv2d_pack_weight_multi) into their existing buffers (stream-ordered: everything that read the old operands was
    enqueued before)."""
    _registry = weakref.WeakSet()
    group_enabled = True                    # False: every stale operand is packed by its own launch (A/B, tools/ab_pack_group.py)
    _table_cache = {}                       # (device index, forward / flip) -> (key tuple, device table)

    def __init__(self):
        self.key = None
        self.packed = None
        self.bias = None
        self.packed_flip = None
        self._wref = None                   # weakref to the plain weight this operand was packed from (groupable members only)
        self._bref = None

    def flip(self):
        """Packed data-gradient operator of the currently packed weight (stride-1 convs)."""
        if self.packed_flip is None:
            self.packed_flip = hip.pack_conv_weight(self._w_for_flip, flip=True)
        return self.packed_flip

    @classmethod
    def refresh_stale(cls, device=None):
        """Repack every registered operand ON `device` (default: the current device) whose weight has a new version: one launch
        for the forward operands and one for the data-gradient operands (tile transposes of the fresh forward operands).
        Operands of other devices are left stale for their own device's call: the pointer table is uploaded to, and the pack
        kernel launched on, ONE device's current stream.  Returns the number of operands packed."""
        lib = hip._lib.load()
        dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        if dev.index is None:
            dev = torch.device('cuda', torch.cuda.current_device())
        rows = ([], [])                      # forward rows, flip rows
        nblk = [0, 0]
        members = []
        for pw in list(cls._registry):
            w = pw._wref() if pw._wref is not None else None
            if w is None or pw.packed is None or not w.is_cuda or w._version == pw.key[0]:
                continue
            if w.dtype != torch.float32 or not w.is_contiguous() or w.device != pw.packed.device:
                pw._wref = None              # no longer a plain operand (module cast / moved): get() repacks it on its own
                cls._registry.discard(pw)
                continue
            if w.device != dev:
                continue                     # another device's model: stays stale until that device refreshes
            members.append((pw, w))
            Cout, Cin, R, S = w.shape
            rows[0].append((w.data_ptr(), pw.packed.data_ptr(), Cout, Cin, R, S, 0, nblk[0]))
            nblk[0] += lib.oess_conv2d_pack_multi_blocks(Cout, Cin, R, S, 0)
            if pw.packed_flip is not None:
                rows[1].append((pw.packed.data_ptr(), pw.packed_flip.data_ptr(), Cout, Cin, R, S, 0, nblk[1]))
                nblk[1] += lib.oess_conv2d_pack_multi_blocks(Cout, Cin, R, S, 1)
        if not members:
            return 0
        for which in (0, 1):
            if not rows[which]:
                continue
            key = tuple(rows[which])
            hit = cls._table_cache.get((dev.index, which))
            if hit is None or hit[0] != key:
                table = hip.h2d_async(torch.tensor(rows[which], dtype=torch.int64), dev)
                cls._table_cache[(dev.index, which)] = (key, table)
            else:
                table = hit[1]
            with torch.cuda.device(dev):
                hip._lib.check(lib.oess_conv2d_pack_weight_multi(table.data_ptr(), len(rows[which]), nblk[which], which,
                                                                 torch.cuda.current_stream(dev).cuda_stream), "oess_conv2d_pack_weight_multi")
        for pw, w in members:
            b = pw._bref() if pw._bref is not None else None
            pw.key = (w._version, None if b is None else b._version, None, pw.key[3])
            pw._w_for_flip = w.detach()
        return len(rows[0]) + len(rows[1])

    def get(self, weight, bias=None, bn=None, need_flip=False, cin_pad=None, ver=None):
        key = (weight._version if ver is None else ver, None if bias is None else bias._version,
               None if bn is None else (bn.weight._version, bn.bias._version, bn.running_mean._version,
                                        bn.running_var._version), cin_pad)
        if PackedWeight.group_enabled and key != self.key and self.packed is not None and self._wref is not None and self._wref() is weight and \
                self.key is not None and self.key[2:] == key[2:]:
            PackedWeight.refresh_stale(weight.device)   # this operand and every other stale registered one of its device, in one launch
            if self.key[0] == key[0]:
                self.key = key                      # (bias version: the fp32 bias tensor shares the parameter's storage)
        if key != self.key:
            with torch.no_grad():
                w = weight.detach().float()
                b = None if bias is None else bias.detach().float()
                if bn is not None:     # y = gamma*(conv(x)-mu)/sqrt(var+eps)+beta  (BatchNorm2d in eval mode)
                    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
                    w = w * scale[:, None, None, None]
                    b0 = torch.zeros_like(scale) if b is None else b
                    b = (b0 - bn.running_mean.detach().float()) * scale + bn.bias.detach().float()
                if cin_pad is not None and cin_pad != w.shape[1]:
                    w = F.pad(w, (0, 0, 0, 0, 0, cin_pad - w.shape[1]))
                self.packed = hip.pack_conv_weight(w)
                self.packed_flip = None
                self._w_for_flip = w
                self.bias = None if b is None else b.contiguous()
                # plain weights join the group refresh: the operand is a pure function of the parameter's own storage
                plain = (ver is None and bn is None and weight.dtype == torch.float32 and weight.is_contiguous() and weight.is_cuda
                         and w.data_ptr() == weight.data_ptr() and weight.requires_grad
                         and weight.shape[0] % 8 == 0 and weight.shape[1] % 8 == 0
                         and (bias is None or (bias.dtype == torch.float32 and bias.is_contiguous())))
                if plain:
                    self._wref = weakref.ref(weight)
                    self._bref = None if bias is None else weakref.ref(bias)
                    PackedWeight._registry.add(self)
                else:
                    self._wref = self._bref = None
            self.key = key
        if need_flip and self.packed_flip is None:
            self.packed_flip = hip.pack_conv_weight(self._w_for_flip, flip=True)
        return self


def conv2d_infer(x, pw, Cout, k, stride=1, pad=0, dil=1, relu=False, residual=None, out=None, out_f32=False):
    """Inference conv on logical-NCHW channels_last tensors.  Returns logical NCHW."""
    y = hip.conv2d_nhwc(nhwc(x), pw.packed, pw.bias, Cout, k, k, stride, pad, dil, relu=relu,
                        residual=None if residual is None else nhwc(residual),
                        out=None if out is None else nhwc(out), out_f32=out_f32)
    return from_nhwc(y)


def _conv_backward(x, weight, pw, k, stride, pad, dil, gy, need_gx, need_gw, need_gb, gx_add=None):
    """Data / weight / bias gradients of conv2d on the MFMA kernels.  gy: channels_last bf16, logical NCHW.
    gx_add: optional NHWC bf16 tensor of x's shape that the data-gradient kernel adds in its epilogue (another consumer's
    gradient of the same x: the skip connection of a bottleneck), so that autograd has nothing left to sum."""
    gx = gw = gb = None
    Cin_x = x.shape[1]
    gy8 = gy if gy.shape[1] % 8 == 0 else to_cl_bf16(gy)
    if need_gx:
        g_in = nhwc(gy8)
        if stride != 1:         # strided conv: dilate dY with zeros, then the same stride-1 product
            Hz = x.shape[2] - dil * (k - 1) + 2 * pad
            Wz = x.shape[3] - dil * (k - 1) + 2 * pad
            g_in = hip.zero_insert(g_in, stride, Hz, Wz)
        gx = from_nhwc(hip.conv2d_nhwc(g_in, pw.flip(), None, Cin_x, k, k, 1, dil * (k - 1) - pad, dil, residual=gx_add))
    if need_gw or need_gb:
        Cout, Cin = weight.shape[0], weight.shape[1]
        if need_gw:
            gw = hip.conv2d_wgrad(nhwc(x), nhwc(gy8), gy8.shape[1], Cin, k, k, stride, pad, dil)
            if gy8.shape[1] != Cout:
                gw = gw[:Cout]
        if need_gb:
            gb = hip.channel_sum(nhwc(gy8))[:Cout]
    return gx, gw, gb


class _ConvTrainFn(torch.autograd.Function):
    """Trainable conv, all three products on hand-written MFMA kernels: forward (conv_fwd.hip), data gradient
    (the same kernel on the rotated / transposed packed weight; strided convs first dilate dY with zeros) and
    weight gradient (conv_wgrad.hip, LDS transpose-read operands, split-K)."""

    @staticmethod
    def forward(ctx, x, weight, bias, pw, k, stride, pad, dil, out_f32):
        Cout = weight.shape[0]
        y = conv2d_infer(x, pw, Cout, k, stride, pad, dil, out_f32=out_f32)
        ctx.save_for_backward(x, weight)
        ctx.meta = (pw, k, stride, pad, dil, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        pw, k, stride, pad, dil, has_bias = ctx.meta
        gy = gy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gx, gw, gb = _conv_backward(x, weight, pw, k, stride, pad, dil, gy, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                    has_bias and ctx.needs_input_grad[2])
        return gx, gw, gb, None, None, None, None, None, None


class _ConvBNTrainFn(torch.autograd.Function):
    """Bias-free conv -> nn.BatchNorm2d(train) [-> + residual] [-> ReLU] as ONE autograd node (models/_resnet.py:96-114,
    models/deeplabv3.py:295-348).  Forward: the batch statistics come from the conv epilogue's fp32 accumulators (per-tile
    partials, reduced in a fixed order in double) -- no statistics pass over the activation; backward: the fused BatchNorm
    backward kernels (d(gamma), d(beta), dx, d(residual) with the ReLU mask from the stored output), then the conv's data and
    weight gradients."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, pw, k, stride, pad, dil, relu, eps, momentum, running_mean, running_var,
                skip_in=None, skip_out=None):
        # skip_in / skip_out: one dict shared by the FIRST and the LAST conv of a bottleneck whose identity is the block input x
        # (models/_resnet.py Bottleneck without downsample).  x then has two consumers -- conv1 and the residual add -- and
        # autograd would sum their gradients with an ATen add over the whole activation.  Instead the last node parks its
        # residual gradient in the dict (and reports None for it) and the first node, which by data dependence runs later in
        # the backward pass, lets its data-gradient kernel add it in the epilogue.
        lib = hip._lib.load()
        Cout = weight.shape[0]
        xn = nhwc(x)
        B, H, W, _, _ = hip._nhwc_geom(xn)
        Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
        M = B * Ho * Wo
        tiles = (M + 127) // 128
        part = torch.empty((tiles, 2, Cout), dtype=torch.float32, device=x.device)
        y = hip.conv2d_nhwc(xn, pw.packed, None, Cout, k, k, stride, pad, dil, tile_stats=part)          # raw conv output (kept)
        st = torch.empty((4, Cout), dtype=torch.float32, device=x.device)                                  # mean, rstd, scale, shift
        out = torch.empty((B, Ho, Wo, Cout), dtype=torch.bfloat16, device=x.device)
        rn = None if residual is None else nhwc(residual)
        rps = 0 if rn is None else hip._nhwc_geom(rn)[4]
        if tiles <= 320 and Cout % 64 == 0:
            # small map: statistics (fixed order, double) + apply in ONE launch; mean / rstd are kept for the backward
            hip._lib.check(lib.oess_norm_tile_stats_apply_nhwc_bf16(part.data_ptr(), tiles, Cout, float(M), float(eps), gamma.data_ptr(),
                                                                    beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(),
                                                                    float(momentum), st[0].data_ptr(), st[1].data_ptr(), y.data_ptr(), Cout,
                                                                    None if rn is None else rn.data_ptr(), rps, int(relu), M,
                                                                    out.data_ptr(), Cout, hip._stream()),
                           "oess_norm_tile_stats_apply_nhwc_bf16")
        else:
            sc = hip._stats_scratch(Cout, x.device)
            hip._lib.check(lib.oess_norm_reduce_finalize_tile_stats(part.data_ptr(), tiles, Cout, sc.buf64.data_ptr(), sc.tickets.data_ptr(),
                                                                    float(M), float(eps), gamma.data_ptr(), beta.data_ptr(),
                                                                    running_mean.data_ptr(), running_var.data_ptr(), float(momentum),
                                                                    st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(), st[3].data_ptr(),
                                                                    hip._stream()), "oess_norm_reduce_finalize_tile_stats")
            hip._lib.check(lib.oess_norm_apply_nhwc_bf16(y.data_ptr(), Cout, st[2].data_ptr(), st[3].data_ptr(),
                                                         None if rn is None else rn.data_ptr(), rps, int(relu), 1, M, Cout, out.data_ptr(),
                                                         Cout, hip._stream()), "oess_norm_apply_nhwc_bf16")
        ctx.save_for_backward(x, weight, gamma, y, st, out if relu else None)
        ctx.meta = (pw, k, stride, pad, dil, relu, residual is not None)
        ctx.skip_in, ctx.skip_out = skip_in, skip_out
        if skip_in is not None and ctx.needs_input_grad[0]:
            skip_in['armed'] = True            # the consumer of the hand-over exists: only now may the block's last node park its gradient
        return from_nhwc(out)

    @staticmethod
    def backward(ctx, gy):
        lib = hip._lib.load()
        x, weight, gamma, y, st, out = ctx.saved_tensors
        pw, k, stride, pad, dil, relu, has_res = ctx.meta
        gy = gy.to(torch.bfloat16)
        if gy.stride(1) != 1:
            gy = gy.contiguous(memory_format=torch.channels_last)
        gn = nhwc(gy)
        B, Ho, Wo, C = y.shape
        M = B * Ho * Wo
        gps = hip._nhwc_geom(gn)[4]
        dy = torch.empty((B, Ho, Wo, C), dtype=torch.bfloat16, device=y.device)                # gradient w.r.t. the raw conv output
        dres = torch.empty((B, Ho, Wo, C), dtype=torch.bfloat16, device=y.device) if (has_res and relu) else None
        # two separate allocations: AccumulateGrad steals a whole tensor, a view it would have to copy
        dgb = (torch.empty(C, dtype=torch.float32, device=y.device), torch.empty(C, dtype=torch.float32, device=y.device))
        g32 = gamma.detach().float().contiguous()
        ws, wsn = hip._norm_partials(1, M, C, y.device, backward=True)
        hip._lib.check(lib.oess_batchnorm_bwd_nhwc_bf16(y.data_ptr(), C, gn.data_ptr(), gps, None if out is None else out.data_ptr(), C,
                                                        st[0].data_ptr(), st[1].data_ptr(), g32.data_ptr(), int(relu), M, C,
                                                        dgb[0].data_ptr(), dgb[1].data_ptr(), dy.data_ptr(), C,
                                                        None if dres is None else dres.data_ptr(), C, ws.data_ptr(), wsn, hip._stream()),
                       "oess_batchnorm_bwd_nhwc_bf16")
        gres = None
        if has_res:
            gres = from_nhwc(dres) if dres is not None else gy                                  # no ReLU: the residual sees dy itself
        if gres is not None and ctx.skip_out is not None and ctx.skip_out.get('armed') and ctx.needs_input_grad[4]:
            ctx.skip_out['g'] = nhwc(gres)          # handed to the block's first conv; autograd sees no residual gradient here
            gres = None
        gx_add = None
        if ctx.skip_in is not None:
            gx_add = ctx.skip_in.pop('g', None)
            if gx_add is not None and (not ctx.needs_input_grad[0] or stride != 1 or tuple(gx_add.shape) != tuple(nhwc(x).shape)):
                raise RuntimeError("conv_bn skip-gradient hand-over: the block input needs a gradient of its own shape")
        gx, gw, _ = _conv_backward(x, weight, pw, k, stride, pad, dil, from_nhwc(dy), ctx.needs_input_grad[0],
                                   ctx.needs_input_grad[1], False, gx_add=gx_add)
        return (gx, gw, dgb[1].to(gamma.dtype), dgb[0].to(gamma.dtype), gres) + (None,) * 12


def conv_bn_train(x, conv, bn, pw, relu=False, residual=None, skip_in=None, skip_out=None):
    """Differentiable bias-free conv + train-mode BatchNorm2d [+ residual] [+ ReLU] (one autograd node, statistics from the conv
    epilogue).  x: logical NCHW channels_last bf16 with C % 8 == 0."""
    k, s, p, d = conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.dilation[0]
    pw.get(conv.weight, None, None, cin_pad=x.shape[1])
    if residual is not None and (residual.dtype != torch.bfloat16 or residual.stride(1) != 1):
        residual = residual.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = _ConvBNTrainFn.apply(x, conv.weight, bn.weight, bn.bias, residual, pw, k, s, p, d, bool(relu), float(bn.eps),
                             0.0 if bn.momentum is None else float(bn.momentum), bn.running_mean, bn.running_var, skip_in, skip_out)
    bump_bn_counter(bn)
    return y


def conv2d_train(x, weight, bias, pw, k, stride=1, pad=0, dil=1, out_f32=False, ver=None):
    """`ver`: explicit cache key for weights DERIVED from parameters (their own _version is always 0)."""
    cin_pad = x.shape[1]
    pw.get(weight, bias, None, cin_pad=cin_pad, ver=ver)
    if not x.is_contiguous(memory_format=torch.channels_last) and x.stride(1) != 1:
        x = x.contiguous(memory_format=torch.channels_last)
    return _ConvTrainFn.apply(x, weight, bias, pw, k, stride, pad, dil, out_f32)


# ---- BatchNorm2d.num_batches_tracked: one tiny kernel per `+= 1`.  Inside `defer_bn_counters()` (a whole backbone forward)
# the bumps are collected and applied with ONE multi-tensor add on exit (53 launches -> 1 for the ResNet-50 teacher).
_BN_COUNTER_STACK = []


class defer_bn_counters:
    def __enter__(self):
        _BN_COUNTER_STACK.append([])
        return self

    def __exit__(self, *exc):
        pending = _BN_COUNTER_STACK.pop()
        if pending:
            if _BN_COUNTER_STACK:                    # nested: hand over to the outer scope
                _BN_COUNTER_STACK[-1].extend(pending)
            else:
                torch._foreach_add_(pending, 1)
        return False


def bump_bn_counter(bn):
    if bn.num_batches_tracked is None:
        return
    if _BN_COUNTER_STACK:
        _BN_COUNTER_STACK[-1].append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked += 1


_EVAL_BN_LOGGED = False


def batch_norm_act(x, bn, relu=False, residual=None):
    """BatchNorm2d (+ residual add + ReLU) on a channels_last bf16 tensor, honouring bn.training exactly like
    nn.BatchNorm2d (batch statistics + running-stat update in train mode).  Train mode runs on the HIP norm kernels
    (forward and, when autograd is recording, backward); eval mode with un-folded statistics is a plain affine."""
    hip_ok = x.dtype == torch.bfloat16 and x.stride(1) == 1 and x.shape[1] % 8 == 0 and x.shape[1] <= 2048
    if bn.training and hip_ok:
        r = residual
        if r is not None and r.stride(1) != 1:
            r = r.contiguous(memory_format=torch.channels_last)
        if torch.is_grad_enabled() and (x.requires_grad or bn.weight.requires_grad or (r is not None and r.requires_grad)):
            return hip.batch_norm_train(x, bn, relu=relu, residual=r)
        return from_nhwc(hip.batch_norm_train_nhwc(nhwc(x), bn, relu=relu, residual=None if r is None else nhwc(r)))
    if bn.training:
        # the training path has ONE implementation (no eager fallback): a layout the HIP norm kernels do not take
        # is a caller bug, not a reason to switch to ATen / MIOpen silently
        raise ValueError(f"batch_norm_act: train-mode BatchNorm needs a channels_last bf16 tensor with C % 8 == 0 and C <= 2048, got "
                         f"{tuple(x.shape)} {x.dtype} strides {tuple(x.stride())}")
    # eval mode with un-folded running statistics (validation of a model whose BatchNorm was not folded at pack time): a plain affine
    global _EVAL_BN_LOGGED
    if not _EVAL_BN_LOGGED:
        _EVAL_BN_LOGGED = True
        import logging
        logging.getLogger("openess_amd").info("batch_norm_act: eval-mode BatchNorm with un-folded statistics runs as an ATen affine (not on the training path)")
    y = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False,
                     0.0 if bn.momentum is None else bn.momentum, bn.eps)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y
