"""Shared body of the stage-2/3 trainers (supervised Dice + CE on ground-truth labels, one AdamW):
training/finetune_trainer.py:81-492, training/linear_probe_trainer.py:79-491, training/sup_only_trainer.py:80-511 are three
near-copies in the reference; what differs between them lives in the three modules of the same names next to this file."""
import math

import torch
import torch.nn.functional as f

from ..e2vid.image_reconstructor import ImageReconstructor
from ..e2vid.model.model import E2VID_LIGHTWEIGHT_CONFIG, E2VIDRecurrent
from ..models.deeplabv3 import deeplabv3_resnet50
from ..models.style_networks import SemSegE2VID
from ..utils.loss_functions import TaskLoss
from ..utils.optim import AdamW          # torch.optim.AdamW with its step on the multi-tensor HIP kernel
from .base_trainer_ov import BaseTrainer


class SupervisedTrainer(BaseTrainer):
    """What the three stage-2/3 trainers share.  A subclass states its differences through three hooks:
    `backend_kwargs()` (extra SemSegE2VID constructor arguments), `deeplab_kwargs()` (extra deeplabv3_resnet50 constructor
    arguments) and `amp_requested()` (whether the reference would build a GradScaler for this trainer)."""

    def backend_kwargs(self):
        return {}

    def deeplab_kwargs(self):
        return {}

    def amp_requested(self):
        return False

    def init_fn(self):
        """finetune_trainer.py:86-88 (and its two siblings): models, then optimisers, then the loss object."""
        s = self.settings
        self.buildModels()
        self.createOptimizerDict()
        self.task_loss = TaskLoss(losses=list(s.task_loss), gamma=2.0, num_classes=s.semseg_num_classes, ignore_index=255)
        # One arithmetic mode (bf16 storage, fp32 accumulation) whatever use_amp says; bf16 has fp32's exponent range, so the
        # GradScaler the reference builds under use_amp (sup_only_trainer.py:247-252) has nothing to do and is None here.
        self.scaler = None
        if self.amp_requested():
            self.settings.logger.info("use_amp requested: bf16 storage / fp32 accumulation is always on; no GradScaler is built")

    def buildModels(self):
        """finetune_trainer.py:104-196: `models_dict` (front_sensor_b + back_end, or model_recon) and the reconstructor."""
        s = self.settings
        self.models_dict = {}
        text_path = '' if not s.text_embeddings_path else s.text_embeddings_path
        try:
            open(text_path).close() if text_path else None
        except OSError:
            text_path = ''
        if s.config_option in ('recon2voxel', 'frame2voxel'):
            self.front_end_sensor_b = E2VIDRecurrent(E2VID_LIGHTWEIGHT_CONFIG)
            if not s.unfrozen_e2vid:
                for p in self.front_end_sensor_b.parameters():
                    p.requires_grad = False
                self.front_end_sensor_b.eval()
            self.input_height = math.ceil(s.img_size_b[0] / 8.0) * 8
            self.input_width = math.ceil(s.img_size_b[1] / 8.0) * 8
            self.models_dict['front_sensor_b'] = self.front_end_sensor_b
            self.task_backend = SemSegE2VID(input_c=256, output_c=s.semseg_num_classes, skip_connect=s.skip_connect_task,
                                            skip_type=s.skip_connect_task_type, text_embeddings_path=text_path,
                                            materialize_ch256=False, **self.backend_kwargs())
            self.models_dict['back_end'] = self.task_backend
        elif s.config_option == 'frame2recon':
            self.model_recon = deeplabv3_resnet50(num_classes=s.semseg_num_classes, text_embeddings_path=text_path,
                                                  output_stride=s.output_stride, pretrained_backbone=s.pretrained_backbone,
                                                  **self.deeplab_kwargs())
            self.models_dict['model_recon'] = self.model_recon
        else:
            raise NotImplementedError(s.config_option)
        for m in self.models_dict.values():
            m.to(self.device)
        if 'front_sensor_b' in self.models_dict:
            self.reconstructor = ImageReconstructor(self.front_end_sensor_b, self.input_height, self.input_width,
                                                    s.nr_temporal_bins_b, self.device, s.e2vid_config)

    def createOptimizerDict(self):
        """finetune_trainer.py:198-238: one AdamW over the trainable parameters of the student (optimizer_voxel / optimizer_recon)."""
        if not self.is_training:
            self.optimizers_dict = {}
            return
        s = self.settings
        if 'back_end' in self.models_dict:
            trainable = [p for p in self.task_backend.parameters() if p.requires_grad]
            self.optimizers_dict = {'optimizer_voxel': AdamW(trainable, lr=s.lr_voxel)}
        else:
            trainable = [p for p in self.model_recon.parameters() if p.requires_grad]
            self.optimizers_dict = {'optimizer_recon': AdamW(trainable, lr=s.lr_recon)}

    def _latents(self, event):
        s = self.settings
        self.reconstructor.last_states_for_each_channel = {'grayscale': None}
        for i in range(s.nr_events_data_b):
            _, _, latent = self.reconstructor.update_reconstruction(event, channel_slice=(i * s.input_channels_b, s.input_channels_b),
                                                                    need_latents=(i == s.nr_events_data_b - 1))
        return latent

    def _set_modes(self):
        s = self.settings
        for name, m in self.models_dict.items():
            m.train()
            if name == 'front_sensor_b' and not s.unfrozen_e2vid:
                m.eval()

    def front_step(self, batch):
        """Frozen half of a step: the recurrent E2VID encoder (frozen in every fine-tune / linear-probe YAML) depends on no weight
        the optimiser touches, so BaseTrainer.trainEpoch enqueues it for batch i+1 on its own HIP stream BEFORE the trainable half
        of batch i (decoder forward / backward / AdamW, bound by HBM) and it runs under it.  Returns None when there is nothing
        frozen to run ahead (frame2recon, unfrozen_e2vid): the step then runs whole in train_step."""
        s = self.settings
        if s.config_option not in ('recon2voxel', 'frame2voxel') or s.unfrozen_e2vid or not batch[0].is_cuda:
            return None
        self._set_modes()
        if getattr(self, '_front_stream', None) is None:
            self._front_stream = torch.cuda.Stream(device=self.device)
        F, main = self._front_stream, torch.cuda.current_stream(self.device)
        F.wait_stream(main)
        with torch.cuda.stream(F):
            latent = {k: v.detach() for k, v in self._latents(batch[0]).items()}
            self.reconstructor.last_states_for_each_channel = {'grayscale': None}
            done = torch.cuda.Event()
            done.record(F)
        return latent, done

    def task_train_step(self, batch, front=None):
        s = self.settings
        losses, t_loss = {}, 0.
        self._set_modes()
        gt = batch[1]
        if s.config_option in ('recon2voxel', 'frame2voxel'):
            if front is not None:
                latent, done = front
                main = torch.cuda.current_stream(self.device)
                main.wait_event(done)
                for v in latent.values():
                    if torch.is_tensor(v):
                        v.record_stream(main)
            else:
                latent = {k: v.detach() for k, v in self._latents(batch[0]).items()}
            pred, _ = self.task_backend(latent)
            labels = f.interpolate(gt.float().unsqueeze(1), size=(self.input_height, self.input_width), mode='nearest').squeeze(1).long()
            loss = self.task_loss(pred[1], labels) * s.weight_task_loss
            losses['semseg_sensor_b_loss'] = loss.detach()
        else:
            logits, _ = self.model_recon(batch[2])
            loss = self.task_loss(logits, gt) * s.weight_task_loss
            losses['semseg_recon_loss'] = loss.detach()
        return t_loss + loss, losses, {}

    def train_step(self, batch, front=None):
        for opt in self.optimizers_dict.values():
            opt.zero_grad()
        self.grad_reducer.prepare()          # N > 1: gradients accumulate straight into the all-reduce buckets
        t_loss, losses, outputs = self.task_train_step(batch, front=front)
        t_loss.backward()
        self.grad_reducer()
        for opt in self.optimizers_dict.values():
            opt.step()
        return losses, outputs, t_loss.detach()

    def val_step(self, batch, sensor, i_batch, vis_reconstr_idx, file_path):
        s = self.settings
        gt = batch[1]
        if s.config_option in ('recon2voxel', 'frame2voxel'):
            pred, _ = self.models_dict['back_end'](self._latents(batch[0]))
            pred = pred[1]
        else:
            pred, _ = self.models_dict['model_recon'](batch[2])
        losses = {'semseg_' + sensor + '_loss': self.task_loss(pred, gt).detach()}
        self.metrics_semseg_b.update_batch(pred.argmax(dim=1), gt)
        return losses, None
