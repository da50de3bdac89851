"""Mirror of e2vid/model/model.py:E2VIDRecurrent (lines 69-100) and BaseE2VID config parsing (:9-44)."""
import torch.nn as nn

from .unet import UNetRecurrent


class BaseE2VID(nn.Module):
    def __init__(self, config):
        super().__init__()
        assert 'num_bins' in config
        self.num_bins = int(config['num_bins'])
        self.skip_type = str(config.get('skip_type', 'sum'))
        self.num_encoders = int(config.get('num_encoders', 4))
        self.base_num_channels = int(config.get('base_num_channels', 32))
        self.num_residual_blocks = int(config.get('num_residual_blocks', 2))
        self.norm = str(config['norm']) if 'norm' in config else None
        self.use_upsample_conv = bool(config.get('use_upsample_conv', True))


class E2VIDRecurrent(BaseE2VID):
    def __init__(self, config):
        super().__init__(config)
        self.recurrent_block_type = str(config.get('recurrent_block_type', 'convlstm'))
        self.unetrecurrent = UNetRecurrent(num_input_channels=self.num_bins, num_output_channels=1,
                                           skip_type=self.skip_type, recurrent_block_type=self.recurrent_block_type,
                                           activation='sigmoid', num_encoders=self.num_encoders,
                                           base_num_channels=self.base_num_channels,
                                           num_residual_blocks=self.num_residual_blocks, norm=self.norm,
                                           use_upsample_conv=self.use_upsample_conv)

    def forward(self, event_tensor, prev_states, reconstruct=False, wavefront=None, need_head=True, raw=None, skew=False):
        return self.unetrecurrent.forward(event_tensor, prev_states, reconstruct=reconstruct, wavefront=wavefront,
                                          need_head=need_head, raw=raw, skew=skew)


# architecture of E2VID_lightweight.pth.tar; used for random-init synthetic runs
E2VID_LIGHTWEIGHT_CONFIG = {'num_bins': 5, 'skip_type': 'sum', 'recurrent_block_type': 'convlstm', 'num_encoders': 3,
                            'base_num_channels': 32, 'num_residual_blocks': 2, 'use_upsample_conv': False, 'norm': 'BN'}
