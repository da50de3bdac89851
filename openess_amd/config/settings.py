"""Settings: YAML -> flat attribute bag with the reference's attribute names (config/settings.py:14-260), so
that every file under the reference's config/ tree loads unchanged and the trainers read the same fields.

Differences, all opt-in: `dataset_path: 'synthetic'` (or a missing directory together with
OPENESS_ALLOW_MISSING_DATA=1) selects the synthetic provider instead of failing the isdir assertion
(settings.py:117); `generate_log=False` creates nothing on disk, exactly like the reference.
Quirk reproduced on purpose: `if_linear_probing` is read from the `clip:` block only (settings.py:258), so the
reference's config/linear_probe/** files, which put it at top level, dispatch to OpenESSModel.
"""
import logging
import os
import shutil
import time
from types import SimpleNamespace

import numpy as np
import torch
import yaml

_CLASSES = {
    6: (['flat', 'background', 'object', 'vegetation', 'human', 'vehicle'],
        [[128, 64, 128], [70, 70, 70], [220, 220, 0], [107, 142, 35], [220, 20, 60], [0, 0, 142]]),
    11: (['background', 'building', 'fence', 'person', 'pole', 'road', 'sidewalk', 'vegetation', 'car', 'wall',
          'traffic sign'],
         [[0, 150, 255], [118, 118, 118], [214, 220, 229], [4, 50, 255], [190, 153, 153], [155, 55, 255],
          [102, 102, 156], [0, 176, 80], [250, 188, 1], [152, 251, 152], [255, 0, 0]]),
    19: (['road', 'sidewalk', 'building', 'wall', 'fence', 'pole', 'traffic light', 'traffic sign', 'vegetation',
          'terrain', 'sky', 'person', 'rider', 'car', 'truck', 'bus', 'train', 'motorcycle', 'bicycle'],
         # the reference fills only the first 11 rows of the 19-class map (settings.py:169-179)
         [[0, 0, 0], [70, 70, 70], [190, 153, 153], [220, 20, 60], [153, 153, 153], [128, 64, 128], [244, 35, 232],
          [107, 142, 35], [0, 0, 142], [102, 102, 156], [220, 220, 0]]),
}

# e2vid/options/inference_options.py defaults that the training path reads
_E2VID_DEFAULTS = dict(use_gpu=True, no_normalize=False, no_recurrent=False, hot_pixels_file=None, flip=False, color=False,
                       auto_hdr=False)

_DATASET_BLOCK = {'DSEC_events': 'DSEC_events', 'DDD17_events': 'DDD17_events', 'E2VIDDriving_events': 'E2VIDDriving_events',
                  'EventScape_recurrent_events': 'eventscape_events'}

_OPTIM = (('batch_size_b', int), ('lr_voxel', float), ('lr_recon', float), ('lr_frame', float), ('lr_decay', float),
          ('num_epochs', int), ('val_epoch_step', int), ('weight_task_loss', float))


class Settings:
    def __init__(self, settings_yaml, generate_log=True):
        assert os.path.isfile(settings_yaml), settings_yaml
        with open(settings_yaml, 'r') as stream:
            cfg = yaml.load(stream, yaml.Loader)
        self._hardware(cfg['hardware'])
        self._model(cfg['model'])
        self._dataset(cfg['dataset'])
        self._task(cfg['task'])
        ck = cfg['checkpoint']
        self.save_checkpoint, self.resume_training, self.resume_ckpt_file = ck['save_checkpoint'], ck['resume_training'], ck['resume_file']
        assert isinstance(self.resume_training, bool)
        self._logs(cfg['dir']['log'], settings_yaml, generate_log)
        opt = cfg['optim']
        for key, cast in _OPTIM:
            setattr(self, key, cast(opt[key]))
        self.task_loss = opt['task_loss']
        self._clip(cfg['clip'])

    # ------------------------------------------------------------------
    def _hardware(self, hw):
        dev = hw['gpu_device']
        self.gpu_device = torch.device("cpu") if dev == "cpu" else torch.device("cuda:" + str(dev))
        self.num_cpu_workers = hw['num_cpu_workers'] if hw['num_cpu_workers'] >= 0 else os.cpu_count()
        self.path_to_model = 'e2vid/pretrained/E2VID_lightweight.pth.tar'
        self.e2vid_config = SimpleNamespace(path_to_model=self.path_to_model, **_E2VID_DEFAULTS)

    def _model(self, m):
        for key in ('model_name', 'skip_connect_encoder', 'skip_connect_task', 'skip_connect_task_type',
                    'data_augmentation_train', 'train_on_event_labels', 'unfrozen_e2vid'):
            setattr(self, key, m[key])

    def _dataset(self, ds):
        name = ds['name_b']
        if name not in _DATASET_BLOCK:
            raise ValueError("Specified Dataset Sensor B: %s is not implemented" % name)
        self.dataset_name_b = name
        self.sensor_b_name = name.split('_')[-1]
        spec = ds[_DATASET_BLOCK[name]]
        self.split_train_b = spec.get('split_train', 'train') if name == 'DDD17_events' else 'train'
        if name in ('DSEC_events', 'DDD17_events'):
            self.delta_t_per_data_b = spec['delta_t_per_data']
        if name == 'EventScape_recurrent_events':
            self.nr_events_files_b = spec['nr_events_files_per_data']
            self.towns_b = spec['towns']
        self.semseg_label_train_b = name not in ('DSEC_events', 'E2VIDDriving_events')
        self.semseg_label_val_b = name != 'E2VIDDriving_events'
        self.fixed_duration_b = spec['fixed_duration']
        self.nr_events_data_b = spec['nr_events_data']
        self.event_representation_b = spec['event_representation']
        self.nr_events_window_b = spec['nr_events_window']
        self.nr_temporal_bins_b = spec['nr_temporal_bins']
        self.separate_pol_b = False
        if self.event_representation_b == 'voxel_grid':
            self.separate_pol_b = spec['separate_pol']
            self.input_channels_b = spec['nr_temporal_bins'] * (2 if self.separate_pol_b else 1)
        else:
            self.input_channels_b = 6 if self.event_representation_b == 'ev_segnet' else 2
        self.normalize_event_b = spec['normalize_event']
        self.require_paired_data_train_b = spec['require_paired_data_train']
        self.require_paired_data_val_b = spec['require_paired_data_val']
        self.input_channels_b_paired = 3 if (self.require_paired_data_train_b or self.require_paired_data_val_b) else None
        self.read_two_imgs_b = None
        self.extension_dataset_path_b = None
        self.img_size_b = spec['shape']
        self.dataset_path_b = spec['dataset_path']
        self.synthetic_data = self.dataset_path_b == 'synthetic' or (
            not os.path.isdir(self.dataset_path_b) and os.environ.get('OPENESS_ALLOW_MISSING_DATA') == '1')
        assert self.synthetic_data or os.path.isdir(self.dataset_path_b), self.dataset_path_b

    def _task(self, task):
        k = task['semseg_num_classes']
        self.semseg_num_classes = k
        if k in _CLASSES:
            names, colours = _CLASSES[k]
            self.semseg_ignore_label = 255
            self.semseg_class_names = list(names)
            self.semseg_color_map = np.zeros((k, 3), dtype=np.uint8)
            self.semseg_color_map[:len(colours)] = np.asarray(colours, dtype=np.uint8)

    def _logs(self, log_dir, settings_yaml, generate_log):
        if generate_log:
            self.timestr = time.strftime("%Y%m%d-%H%M%S")
            log_dir = os.path.join(log_dir, self.timestr)
            os.makedirs(log_dir)
            shutil.copyfile(settings_yaml, os.path.join(log_dir, os.path.split(settings_yaml)[-1]))
            logging.basicConfig(level=logging.INFO, filename=os.path.join(log_dir, 'running.log'))
            self.logger = logging.getLogger()
        else:
            self.logger = logging.getLogger("openess_amd.nolog")
        self.ckpt_dir = os.path.join(log_dir, 'checkpoints')
        self.vis_dir = os.path.join(log_dir, 'visualization')
        if generate_log:
            os.mkdir(self.ckpt_dir)
            os.mkdir(self.vis_dir)

    def _clip(self, c):
        self.config_option = c['config_option']
        self.skip_ratio = c['skip_ratio']
        self.text_embeddings_path = c['text_embeddings_path']
        self.maskclip_checkpoint = c['maskclip_checkpoint']
        self.visual_projs_path = c['visual_projs_path']
        self.output_stride = int(c['output_stride'])
        self.pretrained_backbone = c['pre_trained_backbone']
        self.if_supervised_only = c['if_supervised_only']
        if c.get('if_pretraining') is not None:             # these attributes exist only then (settings.py:237-246)
            self.if_pretraining = c['if_pretraining']
            for key in ('image_weights', 'if_spatial_contrastive', 'superpixel_sources', 'superpixel_size',
                        'if_dense_clip_supervision', 'pl_sources', 'if_sam_distillation'):
                setattr(self, key, c[key])
        if c.get('if_finetuning') is not None:
            self.if_finetuning = c['if_finetuning']
            for key in ('load_pretrained_weights', 'pretrained_file', 'if_switchable_train'):
                setattr(self, key, c[key])
        self.frozen_backbone = c.get('frozen_backbone', False)
        self.if_linear_probing = c.get('if_linear_probing', False)
        self.use_amp = c.get('use_amp', False)
