"""Weight-init helpers with the reference's semantics (codes/models/modules/module_util.py:7-44).

Only ``nn.Conv2d`` / ``nn.Linear`` / ``nn.BatchNorm2d`` are touched: ``nn.Conv3d``
layers keep PyTorch's default init (SURVEY trap 4), which is why D2DTInput's
"INN_init" leaves conv5 non-zero while DenseBlock's conv5 starts at zero.
"""
import torch.nn as nn
import torch.nn.init as init


def _apply(net_l, fn, scale):
    nets = net_l if isinstance(net_l, list) else [net_l]
    for net in nets:
        for m in net.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                fn(m.weight)
                m.weight.data *= scale
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.BatchNorm2d):
                init.constant_(m.weight, 1)
                init.constant_(m.bias.data, 0.0)


def initialize_weights(net_l, scale=1):
    _apply(net_l, lambda w: init.kaiming_normal_(w, a=0, mode="fan_in"), scale)


def initialize_weights_xavier(net_l, scale=1):
    _apply(net_l, init.xavier_normal_, scale)
