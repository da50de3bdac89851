"""OpenESSPretrainModel (training/pretrain_trainer.py:81-667): stage-1 trainer, F2E contrastive + T2E
pseudo-label distillation.  The step itself lives in pretrain_step.PretrainStep."""
import torch

from .base_trainer_ov import BaseTrainer
from .pretrain_step import PretrainStep


class OpenESSPretrainModel(BaseTrainer):
    def init_fn(self):
        """pretrain_trainer.py:87-89: models, then optimisers, then the loss objects."""
        self.buildModels()
        self.createOptimizerDict()
        self.task_loss, self.nce_loss = self.step.task_loss, self.step.nce_loss

    def buildModels(self):
        """pretrain_trainer.py:106-229: `models_dict` (front_sensor_b / back_end / model_frame / model_recon) and the reconstructor."""
        s = self.settings
        text = None
        if s.text_embeddings_path and torch.cuda.is_available():
            try:
                text = torch.load(s.text_embeddings_path, map_location='cpu')
            except (FileNotFoundError, OSError):
                s.logger.info("text embeddings '%s' not found: random unit-norm embeddings", s.text_embeddings_path)
        online = None
        if getattr(s, 'pl_sources', '') == 'online_maskclip':          # extension: labels from the frozen tower, in the step
            from ..models.maskclip_model import maskClipFeatureExtractor
            kw = {k: getattr(s, k) for k in ('text_embeddings_path', 'visual_projs_path', 'maskclip_checkpoint') if getattr(s, k, None)}
            online = maskClipFeatureExtractor(text_categories=s.semseg_num_classes, **kw).to(self.device).eval()
        self.step = PretrainStep(config_option=s.config_option, online_teacher=online, num_classes=s.semseg_num_classes, img_size=tuple(s.img_size_b),
                                 nr_events_data=s.nr_events_data_b, nr_temporal_bins=s.nr_temporal_bins_b,
                                 if_spatial_contrastive=s.if_spatial_contrastive,
                                 if_dense_clip_supervision=s.if_dense_clip_supervision, superpixel_size=s.superpixel_size,
                                 lr=s.lr_voxel, weight_task_loss=s.weight_task_loss, task_loss=tuple(s.task_loss),
                                 output_stride=s.output_stride, device=self.device, text_embeddings=text)
        self.models_dict = self.step.models_dict
        self.reconstructor = getattr(self.step, 'reconstructor', None)

    def createOptimizerDict(self):
        """pretrain_trainer.py:231-243: one AdamW per trained module, keys optimizer_voxel / optimizer_recon / optimizer_frame,
        learning rates from the settings."""
        if not self.is_training:
            self.optimizers_dict = {}
            return
        s = self.settings
        self.optimizers_dict = self.step.optimizers_dict
        for key, lr in (('optimizer_voxel', s.lr_voxel), ('optimizer_recon', s.lr_recon), ('optimizer_frame', s.lr_frame)):
            if key in self.optimizers_dict:
                for g in self.optimizers_dict[key].param_groups:
                    g['lr'] = lr

    def task_train_step(self, batch, front=None):
        return self.step.task_train_step(batch, front=front)

    def front_step(self, batch):
        """Frozen half of the step (teacher encoder, recurrent E2VID encoder) for `batch`, enqueued on its own HIP streams: trainEpoch
        calls it for batch i+1 before train_step(batch i, front=...) (PretrainStep.front / pipeline_steps)."""
        return self.step.front(batch)

    def train_step(self, batch, front=None):
        for opt in self.optimizers_dict.values():
            opt.zero_grad()
        self.grad_reducer.prepare()          # N > 1: gradients accumulate straight into the all-reduce buckets
        t_loss, losses, outputs = self.task_train_step((batch[0], batch[1], batch[2], batch[3], batch[4], batch[-1]), front=front)
        t_loss.backward()
        self.grad_reducer()
        for opt in self.optimizers_dict.values():
            opt.step()
        return losses, outputs, t_loss.detach()

    def val_step(self, batch, sensor, i_batch, vis_reconstr_idx, file_path):
        """pretrain_trainer.py:625-655."""
        s = self.settings
        losses = {}
        gt = batch[1]
        if s.config_option in ('recon2voxel', 'frame2voxel'):
            self.reconstructor.last_states_for_each_channel = {'grayscale': None}
            for i in range(s.nr_events_data_b):
                _, _, content = self.reconstructor.update_reconstruction(batch[0], channel_slice=(i * s.input_channels_b, s.input_channels_b),
                                                                         need_latents=(i == s.nr_events_data_b - 1))
            pred, _ = self.models_dict['back_end'](content)
            pred = pred[1]
        else:
            pred, _ = self.models_dict['model_recon'](batch[2])
        losses['semseg_' + sensor + '_loss'] = self.task_loss(pred, gt).detach()
        self.metrics_semseg_b.update_batch(pred.argmax(dim=1), gt)
        return losses, None
