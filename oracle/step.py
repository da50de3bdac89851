"""Oracle (test infrastructure): CPU restatement of one frame2voxel / frame2recon pre-training step
(training/pretrain_trainer.py:324-361, 364-534, 550-562) built from oracle.events / oracle.nets /
oracle.losses.  Used as the parity reference of the GPU step and as bench.py's `cpu_baseline`."""
import torch

from . import events as oe
from . import losses as ol
from . import nets as on

E2VID_LIGHTWEIGHT_CONFIG = {'num_bins': 5, 'skip_type': 'sum', 'recurrent_block_type': 'convlstm', 'num_encoders': 3,
                            'base_num_channels': 32, 'num_residual_blocks': 2, 'use_upsample_conv': False, 'norm': 'BN'}


class OracleStep:
    def __init__(self, config_option='frame2voxel', num_classes=11, nr_events_data=20, bins=5,
                 if_spatial_contrastive=False, superpixel_size=100, lr=5e-4, output_stride=32, e2vid_config=None):
        self.opt, self.K, self.nwin, self.bins = config_option, num_classes, nr_events_data, bins
        self.contr, self.sps = if_spatial_contrastive, superpixel_size
        self.model_frame = on.DilationFeatureExtractor()
        if config_option == 'frame2voxel':
            self.front = on.E2VIDRecurrent(e2vid_config or E2VID_LIGHTWEIGHT_CONFIG).eval()
            for p in self.front.parameters():
                p.requires_grad = False
            self.back_end = on.SemSegE2VID(256, num_classes)
            trainable = [p for p in self.back_end.parameters() if p.requires_grad]
        else:
            self.model_recon = on.DeepLabV3(num_classes, output_stride)
            trainable = [p for p in self.model_recon.parameters() if p.requires_grad]
        self.opt_a = torch.optim.AdamW(trainable, lr=lr)
        self.opt_b = torch.optim.AdamW([p for p in self.model_frame.parameters() if p.requires_grad], lr=lr)

    def modules(self):
        d = {'model_frame': self.model_frame}
        if self.opt == 'frame2voxel':
            d.update(front_sensor_b=self.front, back_end=self.back_end)
        else:
            d.update(model_recon=self.model_recon)
        return d

    def loss(self, batch):
        self.model_frame.train()
        t_loss, losses = 0., {}
        if self.opt == 'frame2voxel':
            event, frame, pl = batch[0], batch[2], batch[3]
            self.back_end.train()
            feat_frame = self.model_frame(frame)
            states = None
            with torch.no_grad():
                for i in range(self.nwin):
                    x = on.event_preprocess(event[:, i * self.bins:(i + 1) * self.bins])
                    _, states, latent = self.front(x, states)
            pred, feat_voxel = self.back_end({k: v.detach() for k, v in latent.items()})
            dense = ol.task_loss(pred[1], pl, self.K)
            losses['dense_clip_loss'] = dense.detach()
            if self.contr:
                nce = ol.nce_loss(ol.superpixel_pool(feat_voxel, batch[4], self.sps), ol.superpixel_pool(feat_frame, batch[4], self.sps))
                losses['contrastive_nce_loss'] = nce.detach()
                t_loss = t_loss + nce
            t_loss = t_loss + dense
        else:
            frame, recon, pl = batch[0], batch[2], batch[3]
            self.model_recon.train()
            feat_frame = self.model_frame(frame)
            logits, feat_recon = self.model_recon(recon)
            if self.contr:
                nce = ol.nce_loss(ol.superpixel_pool(feat_recon, batch[4], self.sps), ol.superpixel_pool(feat_frame, batch[4], self.sps))
                losses['contrastive_nce_loss'] = nce.detach()
                t_loss = t_loss + nce
            dense = ol.task_loss(logits, pl, self.K)
            losses['dense_clip_loss'] = dense.detach()
            t_loss = t_loss + dense
        return t_loss, losses

    def train_step(self, batch):
        self.opt_a.zero_grad()
        self.opt_b.zero_grad()
        t_loss, losses = self.loss(batch)
        t_loss.backward()
        self.opt_a.step()
        self.opt_b.step()
        return losses, t_loss.detach()


class OracleSupervisedStep:
    """Stage 2/3 step: OpenESSFineTuneModel (training/finetune_trainer.py:285-386) and OpenESSLinearProbeModel
    (training/linear_probe_trainer.py:276-371) -- supervised Dice + CE on ground truth, ONE AdamW.  Linear probing
    freezes everything but a K->K 1x1 conv on the logits (style_networks.py:113-133,169-170; deeplabv3.py:162-170,
    186-187); modules stay in .train() (BatchNorm keeps using batch statistics), E2VID in .eval()."""

    def __init__(self, config_option='frame2voxel', num_classes=11, nr_events_data=20, bins=5, linear_probing=False,
                 lr=5e-4, output_stride=32, weight_task_loss=1.0):
        self.opt, self.K, self.nwin, self.bins, self.w = config_option, num_classes, nr_events_data, bins, weight_task_loss
        self.linear_probing = linear_probing
        if config_option in ('frame2voxel', 'recon2voxel'):
            self.front = on.E2VIDRecurrent(E2VID_LIGHTWEIGHT_CONFIG).eval()
            for p in self.front.parameters():
                p.requires_grad = False
            self.net = on.SemSegE2VID(256, num_classes)
        else:
            self.net = on.DeepLabV3(num_classes, output_stride)
        if linear_probing:
            for name, p in self.net.named_parameters():
                # style_networks.py:113-131 freezes decoder_scale_1..4 + ch256 + ch512 but NOT the (unused) decoder_scale_5,
                # which therefore stays in the optimiser with grad None; deeplabv3.py:162-168 freezes backbone + classifier
                if not name.startswith('decoder_scale_5'):
                    p.requires_grad = False
            self.net.linear_probe = torch.nn.Conv2d(num_classes, num_classes, 1)
        self.optim = torch.optim.AdamW([p for p in self.net.parameters() if p.requires_grad], lr=lr)

    def modules(self):
        if self.opt in ('frame2voxel', 'recon2voxel'):
            return {'front_sensor_b': self.front, 'back_end': self.net}
        return {'model_recon': self.net}

    def logits(self, batch):
        if self.opt in ('frame2voxel', 'recon2voxel'):
            states = None
            with torch.no_grad():
                for i in range(self.nwin):
                    _, states, latent = self.front(on.event_preprocess(batch[0][:, i * self.bins:(i + 1) * self.bins]), states)
            pred, _ = self.net({k: v.detach() for k, v in latent.items()})
            lg = pred[1]
        else:
            lg, _ = self.net(batch[2])
        return self.net.linear_probe(lg) if self.linear_probing else lg

    def train_step(self, batch):
        self.net.train()
        self.optim.zero_grad()
        loss = ol.task_loss(self.logits(batch), batch[1], self.K) * self.w
        loss.backward()
        self.optim.step()
        key = 'semseg_sensor_b_loss' if self.opt in ('frame2voxel', 'recon2voxel') else 'semseg_recon_loss'
        return {key: loss.detach()}, loss.detach()


class OracleOpenESSStep:
    """OpenESSModel's runnable branch, `frame2recon` (training/openess_trainer.py:478-529; train_step :326-355): two DeepLabv3
    students (frame and reconstruction), T2E pseudo-label TaskLoss on each (:487-495), L1 feature consistency (:497), cosine
    logit consistency (:501), and -- under if_spatial_contrastive -- the superpixel InfoNCE with the pooling offset HARD-CODED to
    30 whatever `superpixel_size` the YAML holds (:506-509; k pools feat_recon, q pools feat_frame :521-527).  Two AdamW
    optimisers, `optimizer_recon` and `optimizer_frame`, zeroed together and stepped in that order (:337-353)."""
    POOL_OFFSET = 30

    def __init__(self, num_classes=11, if_spatial_contrastive=True, lr_recon=5e-4, lr_frame=5e-4, output_stride=32,
                 weight_task_loss=1.0):
        self.K, self.contr, self.w = num_classes, if_spatial_contrastive, weight_task_loss
        self.model_recon = on.DeepLabV3(num_classes, output_stride)
        self.model_frame = on.DeepLabV3(num_classes, output_stride)
        self.opt_recon = torch.optim.AdamW([p for p in self.model_recon.parameters() if p.requires_grad], lr=lr_recon)
        self.opt_frame = torch.optim.AdamW([p for p in self.model_frame.parameters() if p.requires_grad], lr=lr_frame)

    def modules(self):
        return {'model_recon': self.model_recon, 'model_frame': self.model_frame}

    def loss(self, batch):
        frame, recon, pl = batch[0], batch[2], batch[3]
        self.model_frame.train()
        self.model_recon.train()
        losses, t_loss = {}, 0.
        logits_frame, feat_frame = self.model_frame(frame)
        l = ol.task_loss(logits_frame, pl, self.K) * self.w
        losses['semseg_frame_loss'] = l.detach()
        t_loss = t_loss + l
        logits_recon, feat_recon = self.model_recon(recon)
        l = ol.task_loss(logits_recon, pl, self.K) * self.w
        losses['semseg_recon_loss'] = l.detach()
        t_loss = t_loss + l
        l = torch.nn.functional.l1_loss(feat_frame, feat_recon)
        losses['cons_feat_loss'] = l.detach()
        t_loss = t_loss + l
        l = torch.mean(1 - torch.nn.functional.cosine_similarity(logits_frame, logits_recon, dim=1))
        losses['cons_pred_loss'] = l.detach()
        t_loss = t_loss + l
        if self.contr:
            k = ol.superpixel_pool(feat_recon, batch[4], self.POOL_OFFSET)
            q = ol.superpixel_pool(feat_frame, batch[4], self.POOL_OFFSET)
            l = ol.nce_loss(k, q)
            losses['contrastive_nce_loss'] = l.detach()
            t_loss = t_loss + l
        return t_loss, losses

    def train_step(self, batch):
        self.opt_recon.zero_grad()
        self.opt_frame.zero_grad()
        t_loss, losses = self.loss(batch)
        t_loss.backward()
        self.opt_recon.step()
        self.opt_frame.step()
        return losses, t_loss.detach()


def voxelize_sample(x, y, t, p, rectify_map, nwin, C, H, W, crop):
    return torch.from_numpy(oe.dsec_event_tensor(x, y, t, p, rectify_map, nwin, C, H, W, crop))
