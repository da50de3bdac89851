"""Stage-1 (pre-training) step of the `frame2voxel` and `frame2recon` options, restated from
training/pretrain_trainer.py: buildModels (:107-208), createOptimizerDict (:211-274), train_step
(:324-361) and task_train_step (:364-534).  This is the self-contained step object used by bench.py,
smoke() and the trainer classes; it owns the models_dict / optimizers_dict with the reference's key names.
"""
import contextlib
import math

import torch

from .. import hip
from ..e2vid.image_reconstructor import ImageReconstructor
from ..e2vid.model.model import E2VID_LIGHTWEIGHT_CONFIG, E2VIDRecurrent
from ..models.deeplabv3 import deeplabv3_resnet50
from ..models.image_model import DilationFeatureExtractor
from ..models.style_networks import SemSegE2VID
from ..utils.loss_functions import NCELoss, TaskLoss
from ..utils.optim import AdamW          # torch.optim.AdamW with its step on the multi-tensor HIP kernel


class _Front:
    """What the frozen half of a step leaves for the trainable half (PretrainStep.front)."""
    __slots__ = ('content', 'teacher_enc', 'teacher_out', 'online_pl', 'streams', 'done')

    def __init__(self):
        self.content = self.teacher_enc = self.teacher_out = self.online_pl = self.streams = self.done = None


class PretrainStep:
    # frame2voxel + contrastive: the student's 256-channel map is only pooled over superpixels, and the mean commutes with its 1x1
    # convolution (hip.PointwiseFeature); False = materialise the map as the reference does (A/B, tests)
    pooled_student_features = True
    # contrastive: the teacher's upsampled + normalised features are only pooled too -> hip.UpsampledNormalizedFeature (one-pass
    # backward); False = the full-resolution tensor goes through autograd as three separate adjoints
    pooled_teacher_features = True
    # FROZEN FRONT / TRAINABLE BACK.  The teacher's encoder (image_model.py:130-143) and the recurrent E2VID encoder are frozen:
    # nothing in them depends on a weight the optimiser touches.  `front(batch)` runs them on their own HIP streams (teacher on
    # one, E2VID on another: the teacher's BatchNorm-apply passes and short-K 1x1 layers are bound by HBM, the ConvLSTM by the
    # matrix / LDS side) and returns a handle; `task_train_step(batch, front=handle)` is the trainable rest (student decoder /
    # DeepLabv3, teacher head when the contrastive loss trains it, losses).  A loop may therefore enqueue front(batch i+1) BEFORE
    # the back half of batch i (`pipeline_steps`): the frozen front of the next batch then runs under the HBM-bound decoder
    # forward / backward / AdamW of the current one.  Same kernels on the same values in a dependency-respecting order: losses,
    # gradients, weights and the teacher's running statistics are bit-identical to the one-stream order (tests/test_hip_determinism).
    # overlap_teacher = False puts everything on the caller's stream (A/B, per-launch timing without CU sharing).
    overlap_teacher = True
    # if_spatial_contrastive: False -> nothing reads the teacher's output.  True (default) runs the whole forward anyway, as the
    # reference does (pretrain_trainer.py:434); False stops after the encoder, whose BatchNorm updates are the call's only effect.
    run_unused_teacher_head = True

    def __init__(self, config_option='frame2voxel', num_classes=11, img_size=(440, 640), nr_events_data=20,
                 nr_temporal_bins=5, if_spatial_contrastive=False, if_dense_clip_supervision=True, superpixel_size=100,
                 lr=5e-4, weight_task_loss=1.0, task_loss=('dice', 'cross_entropy'), output_stride=32, device='cuda',
                 e2vid_config=None, text_embeddings=None, seed=1205, online_teacher=None, wavefront=False):
        self.config_option = config_option
        # a frozen MaskCLIP tower as ONLINE teacher: pseudo-labels = argmax of its logits on the frame, computed inside
        # the step, instead of the offline `pl_*_rgb` PNGs (README.md:295).  None = the reference's behaviour (labels from the batch).
        self.online_teacher = online_teacher
        self.device = torch.device(device)
        self.nr_events_data, self.bins = nr_events_data, nr_temporal_bins
        self.if_spatial_contrastive = if_spatial_contrastive
        self.if_dense_clip_supervision = if_dense_clip_supervision
        self.superpixel_size = superpixel_size
        self.weight_task_loss = weight_task_loss
        torch.manual_seed(seed)
        self.models_dict = {}
        if config_option == 'frame2voxel':
            self.front_end_sensor_b = E2VIDRecurrent(e2vid_config or E2VID_LIGHTWEIGHT_CONFIG)
            for p in self.front_end_sensor_b.parameters():
                p.requires_grad = False
            self.front_end_sensor_b.eval()
            self.input_height = math.ceil(img_size[0] / 8.0) * 8
            self.input_width = math.ceil(img_size[1] / 8.0) * 8
            self.models_dict['front_sensor_b'] = self.front_end_sensor_b
            self.task_backend = SemSegE2VID(input_c=256, output_c=num_classes, skip_connect=True, skip_type='concat',
                                            text_embeddings_path='',
                                            materialize_ch256=('pooled' if self.pooled_student_features else True)
                                            if if_spatial_contrastive else False)
            self.models_dict['back_end'] = self.task_backend
        elif config_option == 'frame2recon':
            self.model_recon = deeplabv3_resnet50(num_classes=num_classes, text_embeddings_path='',
                                                  output_stride=output_stride, pretrained_backbone='')
            self.models_dict['model_recon'] = self.model_recon
            self.model_recon.lazy_feats = bool(if_spatial_contrastive and self.pooled_student_features)
        else:
            raise NotImplementedError(config_option)
        self.model_frame = DilationFeatureExtractor(image_weights=None)
        self.model_frame.lazy_features = bool(if_spatial_contrastive and self.pooled_teacher_features)
        self.models_dict['model_frame'] = self.model_frame
        if text_embeddings is not None:
            tgt = self.task_backend if config_option == 'frame2voxel' else self.model_recon.classifier
            tgt.text_embeddings.copy_(text_embeddings)
        for m in self.models_dict.values():
            m.to(self.device)
        if config_option == 'frame2voxel':
            self.reconstructor = ImageReconstructor(self.front_end_sensor_b, self.input_height, self.input_width,
                                                    nr_temporal_bins, self.device)
            # wavefront schedule of the 20 recurrent sub-windows over one HIP stream per ConvLSTM level (e2vid/wavefront.py);
            # `self.wavefront = None` switches back to the single-stream order at any time (same results)
            self.wavefront = None
            if wavefront and self.device.type == 'cuda':
                from ..e2vid.wavefront import EncoderWavefront
                self.wavefront = EncoderWavefront(self.device, self.front_end_sensor_b.num_encoders)
        self.task_loss = TaskLoss(losses=list(task_loss), gamma=2.0, num_classes=num_classes, ignore_index=255)
        self.nce_loss = NCELoss(temperature=0.07)
        # createOptimizerDict (pretrain_trainer.py:225-259)
        params_frame = [p for p in self.model_frame.parameters() if p.requires_grad]
        if config_option == 'frame2voxel':
            params_voxel = [p for p in self.task_backend.parameters() if p.requires_grad]
            params_voxel = [p for p in self.front_end_sensor_b.parameters() if p.requires_grad] + params_voxel
            self.optimizers_dict = {'optimizer_voxel': AdamW(params_voxel, lr=lr),
                                    'optimizer_frame': AdamW(params_frame, lr=lr)}
        else:
            params_recon = [p for p in self.model_recon.parameters() if p.requires_grad]
            self.optimizers_dict = {'optimizer_recon': AdamW(params_recon, lr=lr),
                                    'optimizer_frame': AdamW(params_frame, lr=lr)}

    # ------------------------------------------------------------------ pretrain_trainer.py:364-534
    def _set_modes(self):
        for name, m in self.models_dict.items():
            m.train()
            if name == 'front_sensor_b':
                m.eval()                       # unfrozen_e2vid: False in every pre-training YAML

    def front(self, batch):
        """Frozen half of the step for `batch` (see the class comment): teacher encoder (+ its head when nothing trains it) and,
        for frame2voxel, the 20 recurrent E2VID encoder steps.  Everything is enqueued, nothing is waited for; the handle goes to
        task_train_step(batch, front=handle)."""
        self._set_modes()
        h = _Front()
        frame = batch[2] if self.config_option == 'frame2voxel' else batch[0]
        cuda = self.overlap_teacher and self.device.type == 'cuda' and frame.is_cuda
        main = torch.cuda.current_stream(self.device) if cuda else None
        if cuda:
            if getattr(self, '_front_stream', None) is None:
                # (a higher HIP stream priority for the two front streams was measured: 190.9 / 191.2 vs 190.7 / 191.2 event-frames/s)
                self._front_stream = torch.cuda.Stream(device=self.device)
                self._teacher_stream = torch.cuda.Stream(device=self.device)
            F, T = self._front_stream, self._teacher_stream
            F.wait_stream(main)                  # the batch's tensors and everything queued so far (NOT what the caller enqueues later)
            T.wait_stream(main)
        ctx = (lambda st: torch.cuda.stream(st)) if cuda else (lambda st: contextlib.nullcontext())
        if self.online_teacher is not None:
            with ctx(T if cuda else None), torch.no_grad():
                h.online_pl = self.online_teacher(frame).argmax(dim=1)
        with ctx(T if cuda else None):
            # the teacher head is trained by the contrastive loss: then it belongs to the back half; otherwise nothing can reach it
            # and the whole forward runs here without autograd bookkeeping (the reference runs it too, pretrain_trainer.py:434,484)
            if self.if_spatial_contrastive:
                h.teacher_enc = self.model_frame.encode(frame)
            elif self.run_unused_teacher_head:
                with torch.no_grad():
                    h.teacher_out = self.model_frame(frame)
            else:
                # Pixel distillation: nothing reads the teacher's features (pretrain_trainer.py:434 computes them and drops them).
                # What the call leaves behind are the train-mode BatchNorm updates of the ENCODER; the head (1x1 conv, x4 bilinear,
                # L2 normalise: no state) is dead code here exactly like the E2VID decoder whose outputs every caller discards.
                h.teacher_enc = self.model_frame.encode(frame)
        if self.config_option == 'frame2voxel':
            event = batch[0]
            with ctx(F if cuda else None):
                wf = getattr(self, 'wavefront', None)
                if wf is not None:
                    wf.begin()
                self.reconstructor.last_states_for_each_channel = {'grayscale': None}
                for i in range(self.nr_events_data):
                    _, _, latent_real = self.reconstructor.update_reconstruction(
                        event, channel_slice=(i * self.bins, self.bins), wavefront=wf,
                        need_latents=(i == self.nr_events_data - 1))      # only the last sub-window's latents are used (:437-441)
                if wf is not None:
                    wf.end(*latent_real.values())
                self.reconstructor.last_states_for_each_channel = {'grayscale': None}    # the sequence ends with the step
                h.content = {k: v.detach() for k, v in latent_real.items()}              # trainTaskStepPretrain (:550-562)
        if cuda:
            h.streams = (F, T)
            h.done = (torch.cuda.Event(), torch.cuda.Event())
            h.done[0].record(F)
            h.done[1].record(T)
        return h

    def _join_front(self, h):
        """The current stream continues after the front half; tensors allocated on the front streams are handed to it."""
        if h.streams is None:
            return
        main = torch.cuda.current_stream(self.device)
        for e in h.done:
            main.wait_event(e)
        live = [h.teacher_enc, h.teacher_out, getattr(h.teacher_out, 'x', None), h.online_pl] + list((h.content or {}).values())
        for t in live:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(main)
        h.streams = None

    def pipeline_steps(self, batches, back):
        """Software pipeline over an iterable of device batches: front(batch i+1) is enqueued before `back(batch i, front i)` runs,
        so the frozen half of the next batch executes under the trainable half of the current one.  `back` is the caller's step
        body (zero_grad, task_train_step(batch, front=...), backward, optimiser); yields its results in batch order."""
        prev = None
        for batch in batches:
            fr = self.front(batch)
            if prev is not None:
                yield back(*prev)
            prev = (batch, fr)
        if prev is not None:
            yield back(*prev)

    def _pool(self, feat, superpixels, S):
        return hip.superpixel_pool(feat, superpixels, self.superpixel_size, S=S)

    def task_train_step(self, batch, front=None):
        """batch: (event | frame, label, frame | recon, pl, superpixels) already on the device.
        `superpixel_rows` (optional 6th item) = max offset id + 1 computed by the loader on the host.
        front: the handle of front(batch) when the caller pipelines the steps; None = the frozen half is enqueued here."""
        losses = {}
        t_loss = 0.
        self._set_modes()
        S = batch[5] if len(batch) > 5 else None
        h = front if front is not None else self.front(batch)
        self._join_front(h)
        if h.online_pl is not None:
            batch = (*batch[:3], h.online_pl, *batch[4:])
        feat_frame = self.model_frame.head(h.teacher_enc) if self.if_spatial_contrastive else h.teacher_out
        if self.config_option == 'frame2voxel':
            pl = batch[3]
            pred, feat_voxel = self.task_backend(h.content)
            loss_dense = self.task_loss(pred[1], pl) * self.weight_task_loss
            losses['dense_clip_loss'] = loss_dense.detach()
            if self.if_spatial_contrastive:
                k = self._pool(feat_voxel, batch[4], S)
                q = self._pool(feat_frame, batch[4], S)
                loss_nce = self.nce_loss(k, q)
                losses['contrastive_nce_loss'] = loss_nce.detach()
                t_loss = t_loss + loss_nce
            if self.if_dense_clip_supervision:
                t_loss = t_loss + loss_dense
        else:                                                                   # frame2recon (:475-529)
            recon, pl = batch[2], batch[3]
            logits_recon, feat_recon = self.model_recon(recon)
            if self.if_spatial_contrastive:
                k = self._pool(feat_recon, batch[4], S)
                q = self._pool(feat_frame, batch[4], S)
                loss_nce = self.nce_loss(k, q)
                losses['contrastive_nce_loss'] = loss_nce.detach()
                t_loss = t_loss + loss_nce
            if self.if_dense_clip_supervision:
                loss_dense = self.task_loss(logits_recon, pl) * self.weight_task_loss
                losses['dense_clip_loss'] = loss_dense.detach()
                t_loss = t_loss + loss_dense
        return t_loss, losses, {}

    # ------------------------------------------------------------------ pretrain_trainer.py:324-361
    def train_step(self, batch):
        for opt in self.optimizers_dict.values():
            opt.zero_grad()
        t_loss, losses, outputs = self.task_train_step(batch)
        t_loss.backward()
        for opt in self.optimizers_dict.values():
            opt.step()
        return losses, outputs, t_loss.detach()
