"""Checkpoint files in the reference's format (utils/saver.py:8-96): a dict of state_dicts keyed by the
models_dict names, `Epoch_<n>.pt` (pre-training: only front_sensor_b / model_recon / back_end, :31-42) and
`ckp.pt` (:44-55); partial, shape-filtered loading of pretrained weights (:73-96)."""
import os

import torch

_SAVED_KEYS = ('front_sensor_b', 'model_recon', 'back_end')


class CheckpointSaver(object):
    def __init__(self, save_dir):
        self.save_dir = os.path.abspath(save_dir)
        self.latest_checkpoint = None

    def _state(self, models, epoch, step_count, keys=None):
        ckpt = {name: models[name].state_dict() for name in models if keys is None or name in keys}
        ckpt['epoch'] = epoch
        ckpt['step_count'] = step_count
        return ckpt

    def save_checkpoint_model(self, models, epoch, step_count):
        path = os.path.join(self.save_dir, 'Epoch_{}.pt'.format(epoch))
        torch.save(self._state(models, epoch, step_count, _SAVED_KEYS), path)
        self.latest_checkpoint = path
        return path

    def save_checkpoint_model_single(self, models, epoch, step_count):
        path = os.path.join(self.save_dir, 'ckp.pt')
        torch.save(self._state(models, epoch, step_count, _SAVED_KEYS), path)
        self.latest_checkpoint = path
        return path

    def load_checkpoint(self, models, optimizers, checkpoint_file=None, load_optimizer=False):
        """Model keys that exist are loaded; `epoch` / `step_count` must be present (KeyError otherwise, as utils/saver.py:67-71).
        The reference additionally reads `batch_size_a/b`, which no saver writes -- its resume path is broken there; those
        two keys are not required here."""
        ckpt = torch.load(checkpoint_file, map_location='cpu')
        for name in models:
            if name in ckpt:
                models[name].load_state_dict(ckpt[name])
        if load_optimizer:
            for name in optimizers:
                if name in ckpt:
                    optimizers[name].load_state_dict(ckpt[name])
        return {'epoch': ckpt['epoch'], 'step_count': ckpt['step_count']}       # KeyError on a foreign file, as the reference (:67-71)

    def load_pretrained_weights(self, models, models_to_load, checkpoint_file=None, frozen_backbone=False):
        """Shape-filtered partial load; with frozen_backbone the `classifier*` tensors are skipped (:73-96)."""
        ckpt = torch.load(checkpoint_file, map_location='cpu')
        for name in models_to_load:
            if name in ('front_sensor_b', 'e2vid_decoder'):       # never taken from a stage-1 checkpoint (saver.py:78-79)
                continue
            if name not in ckpt or name not in models:
                continue
            own = models[name].state_dict()
            picked = {k: v for k, v in ckpt[name].items()
                      if k in own and own[k].shape == v.shape and not (frozen_backbone and k.startswith('classifier'))}
            own.update(picked)
            models[name].load_state_dict(own)
        return ckpt
