"""Small policy networks (reference: src/agent/networks.py:4-26 — MLP built from a list of layer dicts)."""
import torch.nn as nn


class MLP(nn.Module):
    """``config``: [{'in':..,'out':..,'drop_out':p,'activation':'ReLU'|'None'|...}, ...]; sub-module names follow the
    reference (``net.layer{i}-linear``) so that exported state_dicts map one-to-one."""

    def __init__(self, config):
        super().__init__()
        self.net = nn.Sequential()
        self.net_config = config
        for i, layer in enumerate(config):
            self.net.add_module(f'layer{i}-linear', nn.Linear(layer['in'], layer['out']))
            self.net.add_module(f'layer{i}-drop_out', nn.Dropout(layer['drop_out']))
            if layer['activation'] != 'None':
                self.net.add_module(f'layer{i}-activation', getattr(nn, layer['activation'])())

    def forward(self, x):
        return self.net(x)
