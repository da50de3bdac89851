"""Mirror of e2vid/image_reconstructor.py:ImageReconstructor (:18-123) for the training path:
preprocess -> pad -> recurrent model step -> keep state.  No CudaTimer syncs in the hot loop."""
from types import SimpleNamespace

import torch

from .utils.inference_utils import CropParameters, EventPreprocessor


class ImageReconstructor:
    def __init__(self, model, height, width, num_bins, device, options=None, augmentation=False, standardization=False):
        options = options if options is not None else SimpleNamespace()
        self.model = model
        self.device = device
        self.height, self.width, self.num_bins = height, width, num_bins
        if augmentation or standardization:
            raise NotImplementedError("augmentation / standardization act on the discarded image output")
        self.no_recurrent = bool(getattr(options, 'no_recurrent', False))
        self.crop = CropParameters(self.width, self.height, self.model.num_encoders)
        self.last_states_for_each_channel = {'grayscale': None}
        self.event_preprocessor = EventPreprocessor(options)
        # skewed schedule of the recurrent encoder (UNetRecurrent._forward_skew) for calls with need_latents=False; False = plain order
        self.skew = not bool(getattr(options, 'no_skew', False))

    def update_reconstruction(self, event_tensor, event_tensor_id=None, stamp=None, channel_slice=None, reconstruct=False, wavefront=None,
                              need_latents=True):
        """event_tensor: fp32 [B, num_bins, H, W] (reference contract), or -- fused form -- the whole
        [B, C_total, H, W] event tensor plus channel_slice=(c0, cs) so that the slice, the normalisation
        and the NHWC re-layout are one kernel.  Returns (img | None, states, latent): the trainers discard the image
        (`_, _, latent = update_reconstruction(...)`), so it is only computed with `reconstruct=True` (offline
        reconstruction, e2vid/run_reconstruction.py), cropped back from the padded size like the reference's CropParameters.
        need_latents=False: the caller drops this call's latents (all but the last sub-window of a step): latent[1] (the head
        output) is None and is never written to memory (head + encoder-0 conv in one kernel).  With `self.skew` (default) such
        calls also run the recurrent encoder on the skewed schedule: the returned states hold levels 1, 2 one / two sub-windows
        behind until the next call with need_latents=True (or reconstruct=True) drains them -- same results, fewer launches."""
        with torch.no_grad():
            if channel_slice is None:
                events = event_tensor.to(self.device).float().contiguous()
                c0, cs = 0, events.shape[1]
            else:
                events, (c0, cs) = event_tensor, channel_slice
            import contextlib
            kw = {} if wavefront is None else {'wavefront': wavefront}
            if not need_latents and not reconstruct:
                kw['need_head'] = False
            unet = getattr(self.model, 'unetrecurrent', None)
            if self.skew and wavefront is None and not reconstruct and not self.no_recurrent and events.is_cuda and unet is not None:
                kw['skew'] = True
            if (not need_latents and not reconstruct and not self.crop.needs_pad and unet is not None
                    and unet.events_fusable(events, cs)):
                # EventPreprocessor apply + NHWC8 re-layout + head + encoder-0 conv in ONE kernel, straight from the event tensor
                kw['raw'] = (events, c0, cs, not self.event_preprocessor.no_normalize)
                x = None
            else:
                first = torch.cuda.stream(wavefront.streams[0]) if wavefront is not None else contextlib.nullcontext()
                with first:                               # EventPreprocessor + re-layout belong to level 0's stream
                    x = self.event_preprocessor.slice_to_nhwc8(events, c0, cs)
                    if self.crop.needs_pad:
                        x = self.crop.pad(x.float()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            img, states, latent = self.model(x, self.last_states_for_each_channel['grayscale'], reconstruct=reconstruct, **kw)
            self.last_states_for_each_channel['grayscale'] = None if self.no_recurrent else states
        return img, states, latent
