"""Host-side mirror of e2vid/model/submodules.py (reference lines cited per class).  Parameter names
and shapes are identical to the reference so its checkpoints load unchanged; forward passes run on the
HIP kernels (MFMA conv + fused gate kernel).  Inference only: the reference keeps E2VID frozen and runs
it under no_grad (e2vid/image_reconstructor.py:81)."""

import torch
import torch.nn as nn

from ... import engine, hip


class ConvLayer(nn.Module):
    """e2vid/model/submodules.py:7-31.  conv -> (BN | IN) -> activation.  BN is folded into the packed
    weights (module is in eval mode on this path); ReLU is fused in the conv epilogue."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, activation='relu', norm=None):
        super().__init__()
        bias = False if norm == 'BN' else True
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=bias)
        self.activation_name = activation
        self.norm = norm
        if norm == 'BN':
            self.norm_layer = nn.BatchNorm2d(out_channels)
        elif norm == 'IN':
            self.norm_layer = nn.InstanceNorm2d(out_channels, track_running_stats=True)
        self._pw = engine.PackedWeight()

    def forward(self, x, out=None):
        if self.activation_name not in (None, 'relu'):
            raise NotImplementedError("only relu / None activations are on the hot path")
        if self.norm == 'IN':
            raise NotImplementedError("norm='IN' E2VID variants are not on the hot path")
        if self.norm == 'BN' and self.norm_layer.training:
            raise RuntimeError("E2VID front end is frozen/eval on this path (pretrain_trainer.py:370-373)")
        c = self.conv2d
        pw = self._pw.get(c.weight, c.bias, self.norm_layer if self.norm == 'BN' else None, cin_pad=x.shape[1])
        return engine.conv2d_infer(x, pw, c.out_channels, c.kernel_size[0], c.stride[0], c.padding[0], 1,
                                   relu=self.activation_name == 'relu', out=out)


class TransposedConvLayer(nn.Module):
    """e2vid/model/submodules.py:34-62: ConvTranspose2d(k, stride 2, padding, output_padding 1) -> BN -> relu.  Only the offline
    reconstruction path runs it (unet.py:165-166; the training path stops at the latents).  A transposed convolution IS the
    data gradient of the strided convolution with the same weight tensor, so it runs on the same two kernels as autograd's
    dgrad: zero insertion (oess_zero_insert_nhwc_bf16) + the MFMA conv on the rotated / transposed packing; the eval-mode
    BatchNorm is folded into that packing and ReLU is fused in the epilogue."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, activation='relu', norm=None):
        super().__init__()
        bias = False if norm == 'BN' else True
        self.transposed_conv2d = nn.ConvTranspose2d(in_channels, out_channels, kernel_size, stride=2, padding=padding,
                                                    output_padding=1, bias=bias)
        self.activation_name = activation
        self.norm = norm
        if norm == 'BN':
            self.norm_layer = nn.BatchNorm2d(out_channels)
        elif norm == 'IN':
            self.norm_layer = nn.InstanceNorm2d(out_channels, track_running_stats=True)
        self._cache = {}

    def forward(self, x):
        if self.norm == 'IN' or self.activation_name not in (None, 'relu'):
            raise NotImplementedError("only BN / no norm with relu / None are used by the shipped E2VID configurations")
        t = self.transposed_conv2d
        if self.norm == 'BN' and self.norm_layer.training:
            raise RuntimeError("E2VID runs in eval mode on this path")
        k, pad = t.kernel_size[0], t.padding[0]
        bn = self.norm_layer if self.norm == 'BN' else None
        key = (t.weight._version, None if t.bias is None else t.bias._version,
               None if bn is None else (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version))
        if self._cache.get('key') != key:
            with torch.no_grad():
                w = t.weight.detach().float()                        # [Cin, Cout, k, k] == Conv2d weight of the strided conv it transposes
                b = None if t.bias is None else t.bias.detach().float()
                if bn is not None:
                    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
                    w = w * scale[None, :, None, None]
                    b0 = torch.zeros_like(scale) if b is None else b
                    b = (b0 - bn.running_mean.detach().float()) * scale + bn.bias.detach().float()
                self._cache = {'key': key, 'packed': hip.pack_conv_weight(w.contiguous(), flip=True),
                               'bias': None if b is None else b.contiguous()}
        B, Cin, H, W = x.shape
        Ho, Wo = (H - 1) * 2 - 2 * pad + k + 1, (W - 1) * 2 - 2 * pad + k + 1
        z = hip.zero_insert(engine.nhwc(x), 2, Ho - (k - 1) + 2 * pad, Wo - (k - 1) + 2 * pad)
        y = hip.conv2d_nhwc(z, self._cache['packed'], self._cache['bias'], t.out_channels, k, k, 1, (k - 1) - pad, 1,
                            relu=self.activation_name == 'relu')
        return engine.from_nhwc(y)


class UpsampleConvLayer(nn.Module):
    """e2vid/model/submodules.py:65-93 -- parameter container only (see TransposedConvLayer)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, activation='relu', norm=None):
        super().__init__()
        bias = False if norm == 'BN' else True
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=bias)
        self.norm = norm
        if norm == 'BN':
            self.norm_layer = nn.BatchNorm2d(out_channels)
        elif norm == 'IN':
            self.norm_layer = nn.InstanceNorm2d(out_channels, track_running_stats=True)


class ResidualBlock(nn.Module):
    """e2vid/model/submodules.py:140-172.  Runs only on the offline reconstruction path (after the latents are taken)."""

    def __init__(self, in_channels, out_channels, stride=1, downsample=None, norm=None):
        super().__init__()
        bias = False if norm == 'BN' else True
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=stride, padding=1, bias=bias)
        self.norm = norm
        if norm == 'BN':
            self.bn1 = nn.BatchNorm2d(out_channels)
            self.bn2 = nn.BatchNorm2d(out_channels)
        elif norm == 'IN':
            self.bn1 = nn.InstanceNorm2d(out_channels)
            self.bn2 = nn.InstanceNorm2d(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=bias)
        self.downsample = downsample
        self._pw1, self._pw2 = engine.PackedWeight(), engine.PackedWeight()

    def forward(self, x):
        """conv-bn-relu-conv-bn + residual + relu (:154-172); eval-mode BatchNorm folded, residual add and ReLU in the conv epilogue."""
        if self.norm == 'IN' or self.downsample is not None:
            raise NotImplementedError("E2VID residual blocks use BN / no norm and no downsample")
        bn1 = self.bn1 if self.norm == 'BN' else None
        bn2 = self.bn2 if self.norm == 'BN' else None
        if bn1 is not None and bn1.training:
            raise RuntimeError("E2VID runs in eval mode on this path")
        C = self.conv1.out_channels
        p1 = self._pw1.get(self.conv1.weight, self.conv1.bias, bn1, cin_pad=x.shape[1])
        y = engine.conv2d_infer(x, p1, C, 3, 1, 1, 1, relu=True)
        p2 = self._pw2.get(self.conv2.weight, self.conv2.bias, bn2, cin_pad=C)
        return engine.conv2d_infer(y, p2, C, 3, 1, 1, 1, relu=True, residual=x)


class ConvLSTM(nn.Module):
    """e2vid/model/submodules.py:175-214.  State = (hidden, cell).  Here the state lives in two buffers:
    `xh` = the cat(x, h) NHWC bf16 buffer read by the Gates conv (x written by the encoder conv, h by the
    fused gate kernel -> no torch.cat), and `cell` in fp32."""

    def __init__(self, input_size, hidden_size, kernel_size):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        pad = kernel_size // 2
        self.Gates = nn.Conv2d(input_size + hidden_size, 4 * hidden_size, kernel_size, padding=pad)
        self._pw = engine.PackedWeight()
        self._pw_fused = {}

    def fused_args(self, state):
        """Arguments of hip.convlstm_fused (or one problem of hip.convlstm_fused_group) for this state: xh, packed gates, bias,
        cell, hidden view, k, pad, prev_cell_is_zero.  The caller flips state['cur'] / clears state['fresh'] after the launch."""
        g = self.Gates
        cur = state['cur']
        xh = state['xh'][cur]
        k, pad = g.kernel_size[0], g.padding[0]
        pw = self._pw_fused
        key = (g.weight._version, g.bias._version)
        if pw.get('key') != key:
            with torch.no_grad():
                pw['packed'] = hip.pack_conv_weight(g.weight, flip=2)
                pw['bias'] = g.bias.detach().float().contiguous()
            pw['key'] = key
        h_view = state['xh'][1 - cur][:, self.input_size:]
        if state['fresh']:
            # first sub-window: h_prev = 0 and c_prev = 0 (submodules.py:190-198), so the h half of the Gates
            # reduction contributes nothing -> convolve the x half only (half the K loop), same cell update
            if pw.get('packed_x') is None or pw.get('key_x') != key:
                with torch.no_grad():
                    pw['packed_x'] = hip.pack_conv_weight(g.weight[:, :self.input_size], flip=2)
                pw['key_x'] = key
            return (engine.nhwc(xh[:, :self.input_size]), pw['packed_x'], pw['bias'], state['cell'], engine.nhwc(h_view), k, pad, True)
        return (engine.nhwc(xh), pw['packed'], pw['bias'], state['cell'], engine.nhwc(h_view), k, pad, False)

    def w128_ok(self, B, H, W):
        """Geometry rule of oess_convlstm_w128_group_bf16 (include/oess.h): 3 x 3 / pad 1 Gates, 64-channel multiples, a 256-pixel tile
        spans no more rows than the map has, 32-bit extents."""
        g = self.Gates
        C = self.hidden_size
        return (g.kernel_size == (3, 3) and g.padding == (1, 1) and C % 64 == 0 and self.input_size % 64 == 0 and H >= 8
                and (256 + W - 2) // W + 1 <= H and B * H * W * max(2 * (self.input_size + C), 4 * C) < 2 ** 31 and B * H * W * W < 2 ** 32
                and ((B * H * W + 255) // 256) * (C // 64) <= 9000)          # per-workgroup tile lists hold 124 entries (three levels: <= 62 at this bound)

    @staticmethod
    def cell_nhwc(state):
        """The cell state in the reference's order, fp32 [B, H, W, C] (a copy when the state keeps it w128-tiled)."""
        if 'cell_tiled' not in state:
            return state['cell']
        B, H, W, C = state['cell_tiled']
        return hip.convlstm_w128_cell_relayout(state['cell'], B * H * W, C, False).reshape(B, H, W, C)

    def step(self, state):
        """state: dict(xh=[two cat(x, h) buffers, B x (Cin+Ch) x H x W cl bf16], cur=index of the buffer whose x half
        was just written and whose h half holds h_prev, cell=fp32 [B,H,W,Ch], fresh=bool).
        Fused path (hidden % 32 == 0): ONE kernel = Gates conv + cell update; the new h goes to the OTHER cat buffer
        (the conv still reads h_prev around every tile), which becomes current.  Otherwise conv + gate kernel."""
        g = self.Gates
        cur = state['cur']
        xh = state['xh'][cur]
        k, pad = g.kernel_size[0], g.padding[0]
        if 'cell_tiled' in state:
            if not hip.convlstm_w128_group([self.fused_args(state)]):
                raise RuntimeError("oess_convlstm_w128_group_bf16 refused a state that ConvLSTM.w128_ok accepted")
            h_view = state['xh'][1 - cur][:, self.input_size:]
            state['cur'] = 1 - cur
        elif self.hidden_size % 32 == 0:
            hip.convlstm_fused(*self.fused_args(state))
            h_view = state['xh'][1 - cur][:, self.input_size:]
            state['cur'] = 1 - cur
        else:
            pw = self._pw.get(g.weight, g.bias, None, cin_pad=xh.shape[1])
            gates = engine.conv2d_infer(xh, pw, g.out_channels, k, 1, pad, 1, out=state.get('gates'))
            state['gates'] = gates
            h_view = xh[:, self.input_size:]
            hip.convlstm_gates(engine.nhwc(gates), state['cell'], engine.nhwc(h_view), prev_cell_is_zero=state['fresh'])
        state['fresh'] = False
        return h_view


class RecurrentConvLayer(nn.Module):
    """e2vid/model/submodules.py:96-115."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0,
                 recurrent_block_type='convlstm', activation='relu', norm=None):
        super().__init__()
        assert recurrent_block_type == 'convlstm', "only ConvLSTM is used by E2VID_lightweight"
        self.recurrent_block_type = recurrent_block_type
        self.conv = ConvLayer(in_channels, out_channels, kernel_size, stride, padding, activation, norm)
        self.recurrent_block = ConvLSTM(input_size=out_channels, hidden_size=out_channels, kernel_size=3)

    def new_state(self, x):
        B, _, H, W = x.shape
        c = self.conv.conv2d
        Ho = (H + 2 * c.padding[0] - c.kernel_size[0]) // c.stride[0] + 1
        Wo = (W + 2 * c.padding[0] - c.kernel_size[0]) // c.stride[0] + 1
        Co = c.out_channels
        # fused ConvLSTM path: the first step convolves the x half only (zero state) and every later read of a cat(x, h) buffer
        # follows the encoder conv's write of its x half and the previous step's write of its h half -> no zero fill needed
        # (6 fills of up to 157 MB per pre-training step); the conv + gate-kernel path reads h_prev = 0 from the buffer itself
        make = engine.empty_cl if self.recurrent_block.hidden_size % 32 == 0 else engine.zeros_cl
        st = {'xh': [make(B, 2 * Co, Ho, Wo, x.device), make(B, 2 * Co, Ho, Wo, x.device)], 'cur': 0, 'fresh': True}
        if self.recurrent_block.w128_ok(B, Ho, Wo):
            # the cell state only ever feeds the next ConvLSTM step (submodules.py:205-212): it lives in the gate kernel's own
            # "w128-tiled" layout (hip.convlstm_w128_group); ConvLSTM.cell_nhwc(state) gives the reference's [B, H, W, C] order
            st['cell'] = torch.empty(hip.convlstm_w128_cell_elems(B * Ho * Wo, Co), dtype=torch.float32, device=x.device)
            st['cell_tiled'] = (B, Ho, Wo, Co)
        else:
            st['cell'] = torch.empty((B, Ho, Wo, Co), dtype=torch.float32, device=x.device)
        return st

    def conv_s2_args(self, x, state):
        """One problem of hip.conv5x5s2_group for this layer's encoder conv (x -> x half of the current cat(x, h) buffer), or None
        when the layer is not a 5x5 / stride-2 / pad-2 conv with a foldable norm and relu / no activation."""
        m, c = self.conv, self.conv.conv2d
        if not (c.kernel_size == (5, 5) and c.stride == (2, 2) and c.padding == (2, 2) and m.activation_name in (None, 'relu')
                and m.norm != 'IN' and not (m.norm == 'BN' and m.norm_layer.training) and x.shape[1] % 32 == 0
                and c.out_channels % 64 == 0 and x.dtype == torch.bfloat16 and x.stride(1) == 1):
            return None
        pw = m._pw.get(c.weight, c.bias, m.norm_layer if m.norm == 'BN' else None, cin_pad=x.shape[1])
        out = state['xh'][state['cur']][:, :c.out_channels]
        return (engine.nhwc(x), pw.packed, pw.bias, c.out_channels, m.activation_name == 'relu', engine.nhwc(out))

    def run_conv(self, x, prev_state):
        state = prev_state if prev_state is not None else self.new_state(x)
        Co = self.conv.conv2d.out_channels
        self.conv(x, out=state['xh'][state['cur']][:, :Co])         # x -> first half of the current cat(x, h) buffer
        return state

    def forward(self, x, prev_state):
        state = self.run_conv(x, prev_state)
        h = self.recurrent_block.step(state)
        return h, state
