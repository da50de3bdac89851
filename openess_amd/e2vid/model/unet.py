"""Mirror of e2vid/model/unet.py:UNetRecurrent (lines 118-170): head conv + recurrent encoders.
The residual blocks, decoders and prediction layer are constructed (identical state_dict) but not
executed: the hot path only consumes `latent` (unet.py:163), exactly as the reference's callers do
(`_, _, latent = update_reconstruction(...)`, pretrain_trainer.py:441)."""
import torch.nn as nn

from .submodules import ConvLayer, RecurrentConvLayer, ResidualBlock, TransposedConvLayer, UpsampleConvLayer


class _SkewStates(list):
    """Per-level ConvLSTM states of the skewed schedule (UNetRecurrent._forward_skew) plus what is in flight between levels:
    ready[l] = hidden state of level l not yet consumed by level l + 1's encoder conv, out[l] = level l's latest hidden state."""

    def __init__(self, states, n):
        super().__init__(states if states is not None else [None] * n)
        self.ready = [None] * n
        self.out = [None] * n

    def pending(self):
        return any(r is not None for r in self.ready[:-1])


class UNetRecurrent(nn.Module):
    group_s2 = True          # skewed schedule: the two deeper encoder convs of a call as one launch (False: one launch each; A/B switch)

    def __init__(self, num_input_channels, num_output_channels=1, skip_type='sum', recurrent_block_type='convlstm',
                 activation='sigmoid', num_encoders=4, base_num_channels=32, num_residual_blocks=2, norm=None,
                 use_upsample_conv=True):
        super().__init__()
        self.num_input_channels = num_input_channels
        self.num_output_channels = num_output_channels
        self.skip_type = skip_type
        self.norm = norm
        self.num_encoders = num_encoders
        self.base_num_channels = base_num_channels
        self.num_residual_blocks = num_residual_blocks
        self.max_num_channels = base_num_channels * pow(2, num_encoders)
        enc_in = [base_num_channels * pow(2, i) for i in range(num_encoders)]
        enc_out = [base_num_channels * pow(2, i + 1) for i in range(num_encoders)]
        self.head = ConvLayer(num_input_channels, base_num_channels, kernel_size=5, stride=1, padding=2)
        self.encoders = nn.ModuleList([
            RecurrentConvLayer(i, o, kernel_size=5, stride=2, padding=2, recurrent_block_type=recurrent_block_type, norm=norm)
            for i, o in zip(enc_in, enc_out)])
        self.resblocks = nn.ModuleList([ResidualBlock(self.max_num_channels, self.max_num_channels, norm=norm)
                                        for _ in range(num_residual_blocks)])
        Up = UpsampleConvLayer if use_upsample_conv else TransposedConvLayer
        self.decoders = nn.ModuleList([
            Up(s if skip_type == 'sum' else 2 * s, s // 2, kernel_size=5, padding=2, norm=norm) for s in reversed(enc_out)])
        self.pred = ConvLayer(base_num_channels if skip_type == 'sum' else 2 * base_num_channels, num_output_channels, 1,
                              activation=None, norm=norm)

    def _head_enc0_fusable(self, x):
        """The head + encoder-0 conv pair that oess_e2vid_head_enc0_bf16 takes: 8 padded input channels, 5x5 stride-1 head to 32
        channels, 5x5 stride-2 encoder conv 32 -> 64, ReLU / no activation, no InstanceNorm (BatchNorm is folded when in eval)."""
        h, e = self.head, self.encoders[0].conv
        ok = x.is_cuda and x.shape[1] == 8 and x.dtype == __import__('torch').bfloat16
        for m, cin, cout, st in ((h, None, 32, 1), (e, 32, 64, 2)):
            c = m.conv2d
            ok = ok and c.kernel_size == (5, 5) and c.stride == (st, st) and c.padding == (2, 2) and c.out_channels == cout \
                and (cin is None or c.in_channels == cin) and m.activation_name in (None, 'relu') and m.norm != 'IN' \
                and not (m.norm == 'BN' and m.norm_layer.training)
        return ok and h.conv2d.in_channels <= 8

    def _head_enc0(self, x, prev_state, raw=None):
        """Encoder 0's conv output straight from the voxel slice (the 32-channel head output is never written): fills the x half
        of the level-0 cat(x, h) buffer and returns the state.  raw = (events fp32 [B, Ctot, H, W], c0, cs, normalize): the slice is
        normalised and packed inside the kernel as well (x is not needed)."""
        from ... import engine, hip
        h, e = self.head, self.encoders[0]
        state = prev_state if prev_state is not None else e.new_state(raw[0] if raw is not None else x)
        pwh = h._pw.get(h.conv2d.weight, h.conv2d.bias, h.norm_layer if h.norm == 'BN' else None, cin_pad=8)
        ec = e.conv
        pwe = ec._pw.get(ec.conv2d.weight, ec.conv2d.bias, ec.norm_layer if ec.norm == 'BN' else None, cin_pad=32)
        out = engine.nhwc(state['xh'][state['cur']][:, :64])
        if raw is not None:
            ev, c0, cs, normalize = raw
            hip.e2vid_events_head_enc0(ev, c0, cs, normalize, pwh.packed, pwh.bias, h.activation_name == 'relu', pwe.packed, pwe.bias,
                                       ec.activation_name == 'relu', out=out)
        else:
            hip.e2vid_head_enc0(engine.nhwc(x), pwh.packed, pwh.bias, h.activation_name == 'relu', pwe.packed, pwe.bias,
                                ec.activation_name == 'relu', out=out)
        return state

    def events_fusable(self, events, cs):
        """True when `forward(None, ..., need_head=False, raw=(events, c0, cs, normalize))` may replace EventPreprocessor + NHWC8
        re-layout + head + encoder-0 conv by one kernel."""
        import torch
        if not (events.is_cuda and events.dtype == torch.float32 and events.is_contiguous() and events.ndim == 4 and 0 < cs <= 5):
            return False
        probe = torch.empty((1, 8, 1, 1), dtype=torch.bfloat16, device=events.device)
        return self._head_enc0_fusable(probe)

    def _lstm_stage(self, st, levels):
        """The ConvLSTM steps of `levels` (independent of each other on the skewed schedule) as ONE launch when they all take the
        fused kernel, else one by one; st.out[l] = the new hidden state, st.ready[l] = the same view as level l + 1's next input."""
        from ... import hip
        if len(levels) > 3:                                  # the grouped launch takes three problems (num_encoders = 4 variants)
            self._lstm_stage(st, levels[:3])
            self._lstm_stage(st, levels[3:])
            return
        blocks = [self.encoders[l].recurrent_block for l in levels]
        if all('cell_tiled' in st[l] for l in levels):
            # the round-6 kernel: persistent workgroups, 128 x 128 wave tiles, w128-tiled cell states (any number of levels <= 3)
            if not hip.convlstm_w128_group([b.fused_args(st[l]) for b, l in zip(blocks, levels)]):
                raise RuntimeError("oess_convlstm_w128_group_bf16 refused states that ConvLSTM.w128_ok accepted")
            hs = True
        elif len(levels) > 1 and all(b.hidden_size % 32 == 0 and 'cell_tiled' not in st[l] for b, l in zip(blocks, levels)):
            hs = hip.convlstm_fused_group([b.fused_args(st[l]) for b, l in zip(blocks, levels)])
        else:
            hs = None
        if hs is not None:
            for l, b in zip(levels, blocks):
                state = st[l]
                h = state['xh'][1 - state['cur']][:, b.input_size:]
                state['cur'] = 1 - state['cur']
                state['fresh'] = False
                st.out[l] = st.ready[l] = h
        else:
            for l, b in zip(levels, blocks):
                st.out[l] = st.ready[l] = b.step(st[l])

    def _forward_skew(self, x, prev_states, need_head, raw):
        """Skewed schedule of the recurrent encoder on ONE stream: call s runs the encoder conv + ConvLSTM of level l for
        sub-window s - l (level l needs level l of the sub-window before and level l - 1 of the same one, so the three levels of
        a call do not depend on each other), and the ConvLSTM steps of a call are one launch (oess_convlstm_fused_group_bf16).
        Same kernels on the same buffers in a dependency-respecting order: results are identical to the plain order.  The
        deeper levels trail by l sub-windows until a call with need_head=True (the caller wants latents) drains them."""
        n = self.num_encoders
        st = prev_states if isinstance(prev_states, _SkewStates) else _SkewStates(prev_states, n)
        fuse = raw is not None or (not need_head and self._head_enc0_fusable(x))

        levels = self._deeper_convs(st)
        head = None
        if fuse:
            st[0] = self._head_enc0(x, st[0], raw)
        else:
            head = self.head(x)
            st[0] = self.encoders[0].run_conv(head, st[0])
        self._lstm_stage(st, levels + [0])
        latent = {1: head}
        if need_head:
            self._drain(st)
            for i in range(n):
                latent[2 ** (i + 1)] = st.out[i]
        return None, st, latent

    def _deeper_convs(self, st):
        """Encoder convs of the levels whose input arrived from the level above in the last call (independent of each other: level
        l reads h_{l-1}, writes the x half of its own cat buffer); two stride-2 convs go out as ONE launch.  Returns the levels."""
        from ... import hip
        n = self.num_encoders
        levels = [l for l in range(n - 1, 0, -1) if st.ready[l - 1] is not None]
        for l in levels:
            if st[l] is None:
                st[l] = self.encoders[l].new_state(st.ready[l - 1])
        probs = [self.encoders[l].conv_s2_args(st.ready[l - 1], st[l]) for l in levels] if (len(levels) == 2 and self.group_s2) else []
        if len(probs) == 2 and all(p is not None for p in probs):
            hip.conv5x5s2_group(probs)
        else:
            for l in levels:
                st[l] = self.encoders[l].run_conv(st.ready[l - 1], st[l])
        for l in levels:
            st.ready[l - 1] = None
        return levels

    def _drain(self, st):
        """Bring the deeper levels of a skewed sequence up to the last sub-window."""
        while st.pending():
            self._lstm_stage(st, self._deeper_convs(st))

    def forward(self, x, prev_states, reconstruct=False, wavefront=None, need_head=True, raw=None, skew=False):
        """x: logical [B, 8, H, W] channels_last bf16 (bins zero-padded to 8).  Returns (img | None, states, latent):
        the training path stops at the latents; `reconstruct=True` also runs the residual blocks, decoders and the
        prediction layer (unet.py:160-170) and returns the [B, 1, H, W] fp32 image in [0, 1] (offline reconstruction).
        `wavefront` (e2vid/wavefront.py): run level l on its own HIP stream, ordered by events; same kernels, same results.
        `need_head=False` (the caller discards this call's latents: every sub-window but the last of a pre-training step,
        pretrain_trainer.py:437-441): latent[1] is None and head + encoder-0 conv run as ONE kernel.
        `skew=True`: the skewed single-stream schedule of _forward_skew (calls with need_head=False return no usable latents and
        leave the deeper levels one / two sub-windows behind; the need_head=True call that ends the sequence drains them)."""
        if skew and not reconstruct and wavefront is None:
            return self._forward_skew(x, prev_states, need_head, raw)
        if isinstance(prev_states, _SkewStates):
            self._drain(prev_states)                      # a skewed sequence that did not end with need_head=True
            prev_states = list(prev_states)
        if prev_states is None:
            prev_states = [None] * self.num_encoders
        blocks, states = [], []
        if wavefront is not None and not reconstruct:
            import torch
            fuse = raw is not None or (not need_head and self._head_enc0_fusable(x))
            head = None
            if not fuse:
                with torch.cuda.stream(wavefront.streams[0]):
                    x = self.head(x)
                head = x
            for i, encoder in enumerate(self.encoders):
                with torch.cuda.stream(wavefront.streams[i]):
                    wavefront.before_conv(i)
                    state = self._head_enc0(x, prev_states[0], raw) if (fuse and i == 0) else encoder.run_conv(x, prev_states[i])
                    wavefront.after_conv(i)
                    wavefront.before_lstm(i)
                    x = encoder.recurrent_block.step(state)
                    wavefront.after_lstm(i)
                blocks.append(x)
                states.append(state)
        else:
            fuse = raw is not None or (not need_head and not reconstruct and self._head_enc0_fusable(x))
            head = None
            if not fuse:
                x = self.head(x)
                head = x
            for i, encoder in enumerate(self.encoders):
                if fuse and i == 0:
                    state = self._head_enc0(x, prev_states[0], raw)
                    x = encoder.recurrent_block.step(state)
                else:
                    x, state = encoder(x, prev_states[i])
                blocks.append(x)
                states.append(state)
        latent = {1: head}
        for i, b in enumerate(blocks):
            latent[2 ** (i + 1)] = b
        img = None
        if reconstruct:
            if self.skip_type != 'sum':
                raise NotImplementedError("E2VID checkpoints use skip_type 'sum'")
            for resblock in self.resblocks:
                x = resblock(x)
            for i, decoder in enumerate(self.decoders):
                x = decoder(x + blocks[self.num_encoders - i - 1])           # apply_skip_connection = skip_sum
            import torch
            from ... import engine
            p = self.pred
            pw = p._pw.get(p.conv2d.weight, p.conv2d.bias, p.norm_layer if p.norm == 'BN' else None, cin_pad=x.shape[1])
            logits = engine.conv2d_infer(x + head, pw, 1, 1, 1, 0, 1, out_f32=True)     # 32 -> 1, fp32 output
            img = torch.sigmoid(logits.float())
        return img, states, latent
