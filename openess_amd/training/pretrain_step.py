"""Stage-1 (pre-training) step of the `frame2voxel` and `frame2recon` options, restated from
training/pretrain_trainer.py: buildModels (:107-208), createOptimizerDict (:211-274), train_step
(:324-361) and task_train_step (:364-534).  This is the self-contained step object used by bench.py,
smoke() and the trainer classes; it owns the models_dict / optimizers_dict with the reference's key names.
"""
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


class PretrainStep:
    # frame2voxel + contrastive: the student's 256-channel map is only pooled over superpixels, and the mean commutes with its 1x1
    # convolution (hip.PointwiseFeature); False = materialise the map as the reference does (A/B, tests)
    pooled_student_features = True
    # contrastive: the teacher's upsampled + normalised features are only pooled too -> hip.UpsampledNormalizedFeature (one-pass
    # backward); False = the full-resolution tensor goes through autograd as three separate adjoints
    pooled_teacher_features = True
    # The frozen teacher's forward (image_model.py:130-143: a third of the step, BatchNorm-apply passes and short-K 1x1 layers
    # bound by HBM) shares no data with the recurrent E2VID encoder (matrix- / LDS-bound) until the losses meet: it runs on its
    # own HIP stream under the encoder and joins where its features are first read (the pooling of the contrastive loss) or, when
    # nothing reads them (pixel distillation: only its BatchNorm side effects exist), at the end of the forward.  Same kernels on
    # the same buffers, ordered by events: results are bit-identical.  False = one stream (A/B, per-launch timing without overlap).
    overlap_teacher = True

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

    def _teacher(self, frame):
        """The reference always runs the teacher forward (pretrain_trainer.py:434,484), including its train-mode
        BatchNorm side effects, even when the contrastive loss is off and its output is unused.  In that case no
        gradient can reach the teacher's decoder, so the forward runs without autograd bookkeeping."""
        if self.if_spatial_contrastive:
            return self.model_frame(frame)
        with torch.no_grad():
            return self.model_frame(frame)

    def _teacher_begin(self, frame):
        """Teacher forward, on the side stream when `overlap_teacher` (see the class attribute); `_teacher_join()` before its
        result is read on the current stream."""
        self._teacher_side = None
        if not (self.overlap_teacher and self.device.type == 'cuda' and frame.is_cuda):
            return self._teacher(frame)
        from .. import engine
        # every stale packed operand of the step (the trainable weights after the optimiser step) is refreshed HERE, on the current
        # stream: the group repack must not be triggered from the side stream while this stream's convolutions read the operands
        engine.PackedWeight.refresh_stale(self.device)
        if getattr(self, '_teacher_stream', None) is None:
            self._teacher_stream = torch.cuda.Stream(device=self.device)
        side, main = self._teacher_stream, torch.cuda.current_stream(self.device)
        side.wait_stream(main)                       # the frame, the weights, the previous step
        with torch.cuda.stream(side):
            feat = self._teacher(frame)
        self._teacher_side = side
        self._teacher_out = feat
        return feat

    def _teacher_join(self):
        side = getattr(self, '_teacher_side', None)
        if side is None:
            return
        main = torch.cuda.current_stream(self.device)
        main.wait_stream(side)
        feat = self._teacher_out
        for t in (feat, getattr(feat, 'x', None)):      # tensors allocated on the side stream and read on this one from here on
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(main)
        self._teacher_side = self._teacher_out = None

    def _pool(self, feat, superpixels, S):
        return hip.superpixel_pool(feat, superpixels, self.superpixel_size, S=S)

    def task_train_step(self, batch):
        """batch: (event | frame, label, frame | recon, pl, superpixels) already on the device.
        `superpixel_rows` (optional 6th item) = max offset id + 1 computed by the loader on the host."""
        losses = {}
        t_loss = 0.
        self._set_modes()
        S = batch[5] if len(batch) > 5 else None
        if self.online_teacher is not None:
            with torch.no_grad():
                online_pl = self.online_teacher(batch[2] if self.config_option == 'frame2voxel' else batch[0]).argmax(dim=1)
            batch = (*batch[:3], online_pl, *batch[4:])
        if self.config_option == 'frame2voxel':
            event, frame, pl = batch[0], batch[2], batch[3]
            wf = getattr(self, 'wavefront', None)
            if wf is not None:
                wf.begin()              # the level streams start here: the recurrent encoder also overlaps the teacher forward below
            feat_frame = self._teacher_begin(frame)
            self.reconstructor.last_states_for_each_channel = {'grayscale': None}
            for i in range(self.nr_events_data):
                _, _, latent_real = self.reconstructor.update_reconstruction(
                    event, channel_slice=(i * self.bins, self.bins), wavefront=wf,
                    need_latents=(i == self.nr_events_data - 1))      # only the last sub-window's latents are used (:437-441)
            if wf is not None:
                wf.end(*latent_real.values())
            content = {k: v.detach() for k, v in latent_real.items()}          # trainTaskStepPretrain (:550-562)
            pred, feat_voxel = self.task_backend(content)
            loss_dense = self.task_loss(pred[1], pl) * self.weight_task_loss
            losses['dense_clip_loss'] = loss_dense.detach()
            if self.if_spatial_contrastive:
                k = self._pool(feat_voxel, batch[4], S)
                self._teacher_join()
                q = self._pool(feat_frame, batch[4], S)
                loss_nce = self.nce_loss(k, q)
                losses['contrastive_nce_loss'] = loss_nce.detach()
                t_loss = t_loss + loss_nce
            if self.if_dense_clip_supervision:
                t_loss = t_loss + loss_dense
            self._teacher_join()
        else:                                                                   # frame2recon (:475-529)
            frame, recon, pl = batch[0], batch[2], batch[3]
            feat_frame = self._teacher_begin(frame)
            logits_recon, feat_recon = self.model_recon(recon)
            if self.if_spatial_contrastive:
                k = self._pool(feat_recon, batch[4], S)
                self._teacher_join()
                q = self._pool(feat_frame, batch[4], S)
                loss_nce = self.nce_loss(k, q)
                losses['contrastive_nce_loss'] = loss_nce.detach()
                t_loss = t_loss + loss_nce
            if self.if_dense_clip_supervision:
                loss_dense = self.task_loss(logits_recon, pl) * self.weight_task_loss
                losses['dense_clip_loss'] = loss_dense.detach()
                t_loss = t_loss + loss_dense
            self._teacher_join()
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
