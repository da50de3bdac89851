"""Wavefront schedule of the recurrent E2VID encoder over one HIP stream per ConvLSTM level.

Level l of sub-window t needs level l of t-1 (its own state) and level l-1 of t (its input), so at "time" s the levels can
work on sub-windows s, s-1, s-2 concurrently.  Every kernel fills the GPU on its own; what the overlap buys is the idle tail
of each launch (17.2 / 8.6 / 4.3 rounds of tiles per ConvLSTM level) and the memory-bound head / encoder-conv / statistics
kernels running under MFMA-bound ones.  Results are bit-identical to the single-stream order: the same kernels run on the same
buffers, only ordered by events instead of by one queue.

Hazards of the ping-pong cat(x, h) buffers (e2vid/model/submodules.py mirror): ConvLSTM_l(t) writes h_l(t) over h_l(t-2),
which level l+1's encoder conv of sub-window t-2 read on ANOTHER stream -> stream l waits for that conv's event first.
Everything else is either same-stream or a true dependency (encoder conv of level l waits for ConvLSTM_{l-1}(t)).

Per-launch durations are stretched by the overlap (concurrent kernels time-share the CUs), so bench.py measures the `roofline`
object in a separate region with `wavefront` off."""
import torch


class EncoderWavefront:
    def __init__(self, device, num_levels):
        self.device = torch.device(device)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(num_levels)]
        self.begin_called = False

    def begin(self):
        """Order the level streams after everything already queued on the current stream (inputs, previous step)."""
        main = torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(main)
        n = len(self.streams)
        self.lstm_done = [[] for _ in range(n)]          # per level: event after ConvLSTM of sub-window t
        self.conv_done = [[] for _ in range(n)]          # per level: event after the encoder conv of sub-window t
        self.begin_called = True

    def end(self, *consumed_on_main):
        """The current stream continues after all levels have finished their last sub-window.  `consumed_on_main`: tensors that
        were ALLOCATED on a level stream (latents are views of the ConvLSTM cat(x, h) buffers, the head output) and are read by
        kernels of the current stream from here on: the caching allocator is told so (record_stream), otherwise a block freed
        by the caller could be handed out again on the level stream while those kernels still read it."""
        main = torch.cuda.current_stream(self.device)
        for s in self.streams:
            main.wait_stream(s)
        for t in consumed_on_main:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(main)
        self.begin_called = False

    def _check(self):
        if not self.begin_called:
            raise RuntimeError("EncoderWavefront: begin() must bracket the recurrent loop (the level streams are not ordered "
                               "after the current stream otherwise)")

    # ---- used by UNetRecurrent.forward
    def before_conv(self, level):
        self._check()
        if level > 0:
            self.streams[level].wait_event(self.lstm_done[level - 1][-1])       # h_{l-1}(t) is this conv's input

    def after_conv(self, level):
        e = torch.cuda.Event()
        e.record(self.streams[level])
        self.conv_done[level].append(e)

    def before_lstm(self, level):
        self._check()
        t = len(self.conv_done[level]) - 1
        nxt = level + 1
        if nxt < len(self.streams) and t >= 2 and len(self.conv_done[nxt]) > t - 2:
            self.streams[level].wait_event(self.conv_done[nxt][t - 2])          # the reader of the h buffer about to be overwritten

    def after_lstm(self, level):
        e = torch.cuda.Event()
        e.record(self.streams[level])
        self.lstm_done[level].append(e)
