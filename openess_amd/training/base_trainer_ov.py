"""BaseTrainer (training/base_trainer_ov.py:20-590): loader construction, epoch / validation loops, checkpoint
hook, cosine LR.  Same overridable methods and attribute names (models_dict, optimizers_dict, lr_schedulers,
train_loader_sensor_b, val_loader_sensor_b, epoch_count, step_count, metrics_semseg_b).

MI355X-specific: `prepare_batch` takes the loader's RAW event columns and runs the batched HIP voxelizer on
the device (the reference voxelizes per sample in CPU loader workers and ships 901 MB of voxels per batch over
an unpinned H2D copy, base_trainer_ov.py:166-173 / pretrain_trainer.py:428-432); when torch.distributed is
initialised the step all-reduces gradients (training/ddp.py) and the sampler shards by rank."""
import math

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, Subset

from .. import hip
from ..evaluation.metrics import MetricsSemseg
from ..utils.saver import CheckpointSaver
from .ddp import GradAllReduce, broadcast_module_states, shard_indices


class BaseTrainer(object):
    is_training = True

    def __init__(self, settings, train=True):
        self.settings = settings
        self.is_training = bool(train)           # reference: every trainer's __init__(settings, train=True) (pretrain_trainer.py:82-83)
        if not torch.cuda.is_available():
            raise RuntimeError("openess_amd trainers need the GPU (no CPU fallback in the product path)")
        self.device = torch.device('cuda', torch.cuda.current_device())
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.do_val_training_epoch = True
        self._png_pending = []                   # (slot, status tensor, file names) of device-decoded PNG maps not yet checked
        self._png_batches = 0                    # batches that queued at least one status vector since the last check
        # Precision contract (INTEGRATION.md "Precision"): the reference's `use_amp` switches torch.autocast(fp16) + GradScaler
        # (pretrain_trainer.py:344-353).  This build has ONE numeric mode whatever the flag says: bf16 storage of activations /
        # activation gradients / MFMA weight operands, fp32 accumulation, fp32 master weights, optimiser state and losses; no
        # loss scaling is needed (bf16 has fp32's exponent range).  Say so once instead of silently ignoring the key.
        msg = ("numeric mode: bf16 storage / fp32 accumulate (MFMA), fp32 master weights and losses; YAML use_amp={} is accepted "
               "and has no effect (no fp32-activation or fp16-autocast mode in this build)".format(getattr(settings, 'use_amp', False)))
        settings.logger.info(msg)
        if self.rank_hint() == 0:
            print(msg)
        self.metrics_semseg_b = MetricsSemseg(settings.semseg_num_classes, settings.semseg_ignore_label,
                                              settings.semseg_class_names)
        self.init_fn()
        self.createDataLoaders()
        self.models_dict = {k: v.to(self.device) for k, v in self.models_dict.items()}
        self.saver = CheckpointSaver(save_dir=settings.ckpt_dir)
        self.epoch_count = self.step_count = 0
        self.checkpoint = None
        if settings.resume_training:
            self.checkpoint = self.saver.load_checkpoint(self.models_dict, self.optimizers_dict,
                                                         checkpoint_file=settings.resume_ckpt_file, load_optimizer=False)
            self.epoch_count, self.step_count = self.checkpoint['epoch'], self.checkpoint['step_count']
        elif getattr(settings, 'load_pretrained_weights', False):
            self.saver.load_pretrained_weights(self.models_dict, self.models_dict.keys(), settings.pretrained_file,
                                               settings.frozen_backbone)
            settings.logger.info('Pretrained checkpoints loaded from {}'.format(settings.pretrained_file))
        if self.world > 1:
            broadcast_module_states(self.models_dict.values())
        self.grad_reducer = GradAllReduce([p for m in self.models_dict.values() for p in m.parameters()], self.world)
        self.epoch = self.epoch_count
        # cosine schedule with T_max = epochs * iterations but stepped once per EPOCH (base_trainer_ov.py:68-75,375-376)
        total_steps = settings.num_epochs * len(self.train_loader_sensor_b)
        self.lr_schedulers = {k: torch.optim.lr_scheduler.CosineAnnealingLR(v, T_max=max(total_steps, 1))
                              for k, v in self.optimizers_dict.items()}

    @staticmethod
    def rank_hint():
        return dist.get_rank() if dist.is_initialized() else 0

    # ------------------------------------------------------------------ hooks for subclasses
    def init_fn(self):
        pass

    def train_step(self, batch):
        raise NotImplementedError

    def val_step(self, batch, sensor, i_batch, vis_reconstr_idx, file_path):
        raise NotImplementedError

    def trainEpoch(self):
        for m in self.models_dict.values():
            m.train()
        n = len(self.train_loader_sensor_b)
        batches = self.device_batches(self.train_loader_sensor_b)
        front_step = getattr(self, 'front_step', None)
        if front_step is not None and getattr(self.settings, 'pipeline_steps', True):
            # trainers with a frozen front half (OpenESSPretrainModel): front(batch i+1) is enqueued before the trainable half of
            # batch i, so that it runs under it (same results; settings.pipeline_steps: False = one step after the other)
            def results():
                prev = None
                for batch in batches:
                    fr = front_step(batch)
                    if prev is not None:
                        yield self.train_step(prev[0], front=prev[1])
                    prev = (batch, fr)
                if prev is not None:
                    yield self.train_step(prev[0], front=prev[1])
            outs = results()
        else:
            outs = (self.train_step(batch) for batch in batches)
        for i_batch, out in enumerate(outs):
            if i_batch % 20 == 0 and self.rank == 0:
                self.log_train(i_batch, n, out[0])
            self.step_count += 1
            self.check_png_status()
        self.check_png_status(force=True)

    def device_batches(self, loader, split='train'):
        """Ingest pipeline (north_star: "straight from pinned host event buffers"): yields device batches whose host -> device
        copies AND voxelization were enqueued on a SIDE HIP stream.  The generator hands batch i to the caller, the caller
        enqueues step i on the current stream, and only then is batch i+1 pulled from the DataLoader (pinned by its pin
        thread) and its copies + voxelizer launches enqueued on the side stream -- the host runs ahead of the GPU, so they
        execute under step i.  The current stream waits on one event per batch; tensors produced on the side stream are
        handed to the consumer stream with record_stream, so the caching allocator cannot recycle them early.
        `settings.ingest_prefetch: False` (or a CPU-only debug run) falls back to the in-order path."""
        release = getattr(loader, 'consumed_after', None)       # PinnedRingLoader: the slot returns to the workers after our copies
        if not getattr(self.settings, 'ingest_prefetch', True):
            for sample_batched in loader:
                batch = self.prepare_batch(sample_batched, split)
                if release is not None:
                    done = torch.cuda.Event()
                    done.record(torch.cuda.current_stream(self.device))
                    release(done)
                yield batch
            return
        if getattr(self, '_ingest_stream', None) is None:
            self._ingest_stream = torch.cuda.Stream(device=self.device)
        side = self._ingest_stream
        side.wait_stream(torch.cuda.current_stream(self.device))       # once: everything set up so far (rectify maps, weights)
        it = iter(loader)

        def produce():
            try:
                sample_batched = next(it)
            except StopIteration:
                return None
            with torch.cuda.stream(side):           # NOT ordered after the step just enqueued on the current stream: that is the overlap
                batch = self.prepare_batch(sample_batched, split)
                ready = torch.cuda.Event()
                ready.record(side)
                if release is not None:
                    release(ready)
            return batch, ready

        try:
            nxt = produce()
            while nxt is not None:
                batch, ready = nxt
                main = torch.cuda.current_stream(self.device)
                main.wait_event(ready)
                for t in batch:
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(main)
                yield batch                              # the caller enqueues step i here ...
                nxt = produce()                          # ... and batch i+1 is copied + voxelized under it
        finally:
            # also on an early `break` of the consumer (GeneratorExit) or an exception in its step: a prefetched batch's copies and
            # voxelizer launches may still be in flight on the side stream, and the next user of the main stream must be ordered
            # after them
            torch.cuda.current_stream(self.device).wait_stream(side)

    _PNG_STATUS_TEXT = {1: 'bad signature', 2: 'bad IHDR', 3: 'unsupported (not 8-bit gray / palette, or interlaced)', 4: 'bad zlib header',
                        5: 'bad DEFLATE block', 6: 'bad Huffman code', 7: 'stream overrun', 8: 'size mismatch', 9: 'bad filter type'}

    def check_png_status(self, force=False):
        """device_png: a file the GPU decoder could not decode became an all-255 map (for a superpixel slot 255 is an ordinary id,
        not an ignore value), so training on it must not pass silently.  Every decoded slot's status vector is queued by
        prepare_batch; here they are reduced with ONE device sync per `settings.png_check_every` batches (default 50) and at epoch
        end, and a bad file raises with its slot, path and reason."""
        pend = self._png_pending
        every = int(getattr(self.settings, 'png_check_every', 50))
        if not pend or (not force and self._png_batches < every):      # counted in BATCHES, whatever the number of PNG slots per batch
            return
        self._png_pending = []
        self._png_batches = 0
        allst = torch.cat([st.reshape(-1) for _, st, _ in pend])
        if not bool((allst != 0).any()):                           # the one sync
            return
        bad = []
        for slot, st, paths in pend:
            for j, code in enumerate(st.tolist()):
                if code:
                    bad.append(f"batch slot {slot}, sample {paths[j] if paths else j}: {self._PNG_STATUS_TEXT.get(code, code)}")
        raise RuntimeError("device_png: undecodable label / pseudo-label / superpixel PNG(s) were replaced by all-255 maps:\n  "
                           + "\n  ".join(bad[:20]) + "\nRe-encode them as 8-bit grayscale or turn device_png off.")

    def log_train(self, i_batch, n, losses):
        msg = 'epoch: [{0}][{1}/{2}], '.format(self.epoch_count, i_batch, n) + ', '.join(
            "{}: {:.5f}".format(k, float(v)) for k, v in losses.items())
        print(msg)
        self.settings.logger.info(msg)

    def resetValidationStatistics(self):
        self.metrics_semseg_b.reset()

    # ------------------------------------------------------------------ data
    def getDataloader(self, dataset_name):
        """Dataset selector (base_trainer_ov.py:83-90)."""
        if getattr(self.settings, 'synthetic_data', False):
            from ..datasets.synthetic_events import SyntheticEvents
            return SyntheticEvents
        if dataset_name == 'DDD17_events':
            from ..datasets.ddd17_events_loader import DDD17Events
            return DDD17Events
        if dataset_name == 'DSEC_events':
            from ..datasets.DSEC_events_loader import DSECEvents
            return DSECEvents
        raise ValueError(dataset_name)

    def createDataLoaders(self):
        s = self.settings
        builder = self.getDataloader(s.dataset_name_b)
        from ..datasets.synthetic_events import SyntheticEvents, collate
        if builder is SyntheticEvents:
            sensor_hw = (s.img_size_b[0] + (40 if s.dataset_name_b == 'DSEC_events' else 60), s.img_size_b[1])
            crop = sensor_hw[0] - s.img_size_b[0]
            common = dict(sensor_hw=sensor_hw, crop_rows=crop, nr_events_data=s.nr_events_data_b,
                          nr_events_window=s.nr_events_window_b, nr_bins=s.nr_temporal_bins_b,
                          num_classes=s.semseg_num_classes, config_option=s.config_option,
                          superpixel_size=getattr(s, 'superpixel_size', 100))
            n_train = getattr(s, 'synthetic_length', 2 * s.batch_size_b * self.world)
            train_ds = builder(length=n_train, mode='train', pool=getattr(s, 'synthetic_pool', 0), **common)
            val_ds = builder(length=max(s.batch_size_b, 2), mode='val', **common)
        elif s.dataset_name_b == 'DSEC_events':
            train_ds, val_ds = self.createDSECDataset(
                s.dataset_name_b, s.dataset_path_b, s.batch_size_b, s.nr_events_data_b, s.delta_t_per_data_b, s.nr_events_window_b,
                s.data_augmentation_train, s.event_representation_b, s.nr_temporal_bins_b, s.require_paired_data_train_b,
                s.require_paired_data_val_b, s.separate_pol_b, s.normalize_event_b, s.semseg_num_classes, s.fixed_duration_b,
                s.config_option, getattr(s, 'pl_sources', ''), getattr(s, 'superpixel_sources', ''), s.skip_ratio,
                getattr(s, 'if_sam_distillation', False))
        else:
            train_ds, val_ds = self.createDDD17EventsDataset(
                s.dataset_name_b, s.dataset_path_b, s.split_train_b, s.batch_size_b, s.nr_events_data_b, s.delta_t_per_data_b,
                s.nr_events_window_b, s.data_augmentation_train, s.event_representation_b, s.nr_temporal_bins_b,
                s.require_paired_data_train_b, s.require_paired_data_val_b, s.separate_pol_b, s.normalize_event_b, s.fixed_duration_b,
                s.config_option, getattr(s, 'pl_sources', ''), getattr(s, 'superpixel_sources', ''), s.skip_ratio,
                getattr(s, 'if_sam_distillation', False))
        self.sensor_geometry = (train_ds.sensor_hw, train_ds.crop_rows) if hasattr(train_ds, 'sensor_hw') else None
        self.rectify_maps = torch.from_numpy(train_ds.rectify_map[None]).to(self.device) if hasattr(train_ds, 'rectify_map') else None
        self._voxel_ds = {'train': train_ds, 'val': val_ds}            # un-wrapped datasets: Subset has no voxelize_batch
        if self.world > 1:
            train_ds = Subset(train_ds, shard_indices(len(train_ds), self.rank, self.world).tolist())
        # settings.ring_loader (default on with worker processes): the workers collate straight into a pinned shared ring
        # (datasets/ring_loader.py: one host copy per byte instead of the DataLoader's three); False = torch's DataLoader
        if getattr(s, 'ring_loader', True) and s.num_cpu_workers > 0 and len(train_ds) >= s.batch_size_b:
            from ..datasets.ring_loader import PinnedRingLoader
            self.train_loader_sensor_b = PinnedRingLoader(train_ds, batch_size=s.batch_size_b, shuffle=True, drop_last=True,
                                                          num_workers=s.num_cpu_workers)
        else:
            self.train_loader_sensor_b = DataLoader(train_ds, batch_size=s.batch_size_b, num_workers=s.num_cpu_workers,
                                                    pin_memory=True, shuffle=True, drop_last=True, collate_fn=collate)
        self.val_loader_sensor_b = DataLoader(val_ds, batch_size=s.batch_size_b, num_workers=s.num_cpu_workers,
                                              pin_memory=True, shuffle=False, drop_last=False, collate_fn=collate)

    # The reference's two dataset factories (base_trainer_ov.py:93-183, 187-276) with their positional signatures.  They return the
    # (train, validation) DATASETS; createDataLoaders wraps them (ring loader / DataLoader, sharding), where the reference builds its
    # two torch DataLoaders inside the factory.
    def createDSECDataset(self, dataset_name, dsec_dir, batch_size, nr_events_data, delta_t_per_data, nr_events_window, augmentation,
                          event_representation, nr_bins_per_data, require_paired_data_train, require_paired_data_val, separate_pol,
                          normalize_event, semseg_num_classes, fixed_duration, config_option, pl_sources, superpixel_sources,
                          skip_ratio, if_sam_distillation):
        builder = self.getDataloader(dataset_name)
        kw = dict(dsec_dir=dsec_dir, nr_events_data=nr_events_data, delta_t_per_data=delta_t_per_data, nr_events_window=nr_events_window,
                  event_representation=event_representation, nr_bins_per_data=nr_bins_per_data, separate_pol=separate_pol,
                  normalize_event=normalize_event, semseg_num_classes=semseg_num_classes, fixed_duration=fixed_duration,
                  config_option=config_option, pl_sources=pl_sources, device_png=getattr(self.settings, 'device_png_decode', False))
        train = builder(augmentation=augmentation, mode='train', require_paired_data=require_paired_data_train,
                        superpixel_sources=superpixel_sources, skip_ratio=skip_ratio, if_sam_distillation=if_sam_distillation, **kw)
        val = builder(augmentation=False, mode='val', require_paired_data=require_paired_data_val, superpixel_sources='', skip_ratio=2,
                      if_sam_distillation=False, **kw)                                              # :137-157: never augmented, every 2nd sample
        return train, val

    def createDDD17EventsDataset(self, dataset_name, root, split_train, batch_size, nr_events_data, delta_t_per_data, nr_events_per_data,
                                 augmentation, event_representation, nr_bins_per_data, require_paired_data_train,
                                 require_paired_data_val, separate_pol, normalize_event, fixed_duration, config_option, pl_sources,
                                 superpixel_sources, skip_ratio, if_sam_distillation):
        builder = self.getDataloader(dataset_name)
        kw = dict(event_representation=event_representation, nr_events_data=nr_events_data, delta_t_per_data=delta_t_per_data,
                  nr_bins_per_data=nr_bins_per_data, separate_pol=separate_pol, normalize_event=normalize_event,
                  fixed_duration=fixed_duration, nr_events_per_data=nr_events_per_data, config_option=config_option,
                  pl_sources=pl_sources, superpixel_sources=superpixel_sources, skip_ratio=skip_ratio,
                  if_sam_distillation=if_sam_distillation)
        train = builder(root, split=split_train, augmentation=augmentation, require_paired_data=require_paired_data_train, **kw)
        val = builder(root, split='valid', augmentation=False, require_paired_data=require_paired_data_val, **kw)    # :232-250
        return train, val

    def prepare_batch(self, sample_batched, split='train'):
        """Host batch -> device batch.  A raw-event dict becomes the B x (nr_events_data*C) x H x W voxel tensor
        via ONE batched launch sequence of the HIP voxelizer (rectification, time normalisation, crop fused)."""
        s = self.settings
        first = sample_batched[0]
        if len(sample_batched) != 7:            # datasets.synthetic_events.collate emits the 7-slot layout for every dataset
            raise ValueError(f"prepare_batch expects collate's 7-slot batch, got {len(sample_batched)} items")
        rest = [hip.h2d_async(t, self.device) if torch.is_tensor(t) else t for t in sample_batched[1:]]
        if any(isinstance(t, dict) and 'png_bytes' in t for t in rest):
            self._png_batches += 1
        for i, t in enumerate(rest):            # undecoded 8-bit PNG maps (device_png): one batched GPU decode per slot, flips included
            if isinstance(t, dict) and 'png_bytes' in t:
                maps, status = hip.png_decode_gray8_batch(hip.h2d_async(t['png_bytes'], self.device), t['png_lengths'],
                                                          t['hw'][0], t['hw'][1], t['flip'])
                rest[i] = maps
                # per slot: device tensor, non-zero = that map was filled with the ignore index.  Kept (with the file names) for
                # check_png_status(), which runs off the hot path -- one .any() sync per `png_check_every` batches and at epoch end
                self._png_pending.append((i + 1, status, list(sample_batched[-1]) if isinstance(sample_batched[-1], (list, tuple)) else None))
        ds = self._voxel_ds[split]
        if isinstance(first, dict) and 'events_list' in first:            # DDD17: int64 [N,4] rows per sample
            first = ds.voxelize_batch(first['events_list'], self.device, flips=first.get('flip'))
        elif isinstance(first, dict) and 'seg_offsets' in first:          # DSEC: raw columns + explicit sub-window offsets
            first = ds.voxelize_batch(first, self.device)
        elif isinstance(first, dict):
            (H, W), crop = self.sensor_geometry
            C, nwin = s.nr_temporal_bins_b, s.nr_events_data_b
            counts = first['events_per_sample'].tolist()
            if any(n % nwin for n in counts):
                # the reference drops the remainder: nr_events_temp = nr_events_loaded // nr_events_data (sequence_ov.py:302)
                keep, base = [], 0
                for n in counts:
                    keep.append(torch.arange(base, base + (n // nwin) * nwin))
                    base += n
                keep = torch.cat(keep)
                first = {k: (v[keep] if k != 'events_per_sample' else v) for k, v in first.items()}
            offs = [0]
            for n in counts:
                per, base = n // nwin, offs[-1]
                offs.extend([base + per * (i + 1) for i in range(nwin)])
            seg = torch.tensor(offs, dtype=torch.int64)
            B = len(counts)
            dev = {k: hip.h2d_async(first[k], self.device) for k in ('x', 'y', 't', 'p')}
            seg_map = torch.zeros(B * nwin, dtype=torch.int32, device=self.device)
            vox = hip.voxelize_dsec_raw(dev['x'], dev['y'], dev['t'], dev['p'], self.rectify_maps, seg_map, seg, C, H, W,
                                        crop_rows=crop)
            first = vox.view(B, nwin * C, H - crop, W)
        else:
            first = hip.h2d_async(first, self.device)
        sp = rest[3] if len(rest) > 3 and torch.is_tensor(rest[3]) else None
        S = None
        if sp is not None and getattr(s, 'if_spatial_contrastive', False):
            # host-side row count: no device sync in the step (OpenESSModel pools with its own hard-coded size)
            sps = getattr(self, 'pool_superpixel_size', None) or getattr(s, 'superpixel_size', 100)
            if torch.is_tensor(sample_batched[4]):
                S = int((sample_batched[4] + torch.arange(sample_batched[4].shape[0])[:, None, None] * sps).max()) + 1
            else:
                # device_png: the ids only exist on the device.  The row count is data dependent (S = max id + 1 exactly as
                # sparse_coo_tensor sizes it, pretrain_trainer.py:450-453), so one read-back is unavoidable -- it happens HERE, on
                # the ingest stream under the previous step, instead of inside the step
                off = torch.arange(0, sp.shape[0] * sps, sps, device=sp.device)[:, None, None]
                S = int((sp + off).max().item()) + 1
        return (first, *rest, S)

    # ------------------------------------------------------------------ loops (base_trainer_ov.py:358-448)
    def _epoch_loop(self, validate, single_ckpt):
        s = self.settings
        for _ in range(self.epoch_count, s.num_epochs):
            self.trainEpoch()
            if (self.epoch_count % s.val_epoch_step) == 0:
                if validate:
                    self.valEpochs()
                if s.save_checkpoint and self.rank == 0:
                    save = self.saver.save_checkpoint_model_single if single_ckpt else self.saver.save_checkpoint_model
                    save(self.models_dict, self.epoch_count, self.step_count)
            for opt in self.optimizers_dict:
                self.lr_schedulers[opt].step()
            self.epoch_count += 1

    def training(self):
        self._epoch_loop(validate=True, single_ckpt=True)

    def pretraining(self):
        self._epoch_loop(validate=False, single_ckpt=False)

    def valEpochs(self):
        self.resetValidationStatistics()
        with torch.no_grad():
            for m in self.models_dict.values():
                m.eval()
            summary = self.valEpoch(self.val_loader_sensor_b, 'sensor_b')
            self.resetValidationStatistics()
        return summary

    def valEpoch(self, data_loader, sensor_name):
        cumulative = {}
        for i_batch, sample_batched in enumerate(data_loader):
            batch = self.prepare_batch(sample_batched, 'val')
            out = self.val_step(batch[:-4], sensor_name, i_batch, -1, sample_batched[-1])
            for k, v in out[0].items():
                cumulative[k] = cumulative.get(k, 0) + v
        self.check_png_status(force=True)          # device_png: undecodable maps of the validation set raise here
        if self.world > 1 and self.metrics_semseg_b.metrics_acc is not None:      # 968-byte confusion-matrix sum
            dist.all_reduce(self.metrics_semseg_b.metrics_acc)
        metrics = self.metrics_semseg_b.get_metrics_summary()
        self.settings.logger.info('')
        for k, v in metrics.items():
            if k != 'cm':
                self.settings.logger.info(" '{}': '{:.2f}'%.".format(k, v.item()))
        self.settings.logger.info('')
        self.last_val_metrics = metrics
        return metrics
