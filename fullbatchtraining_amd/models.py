"""Parameter containers for the ResNet family of the hot path (drop-in for reference ``fullbatch/models``).

``construct_model(cfg_model, channels, classes)`` (reference ``models.py:14-22``) returns a ``torch.nn.Module``
whose ``state_dict()`` has exactly the reference's keys, shapes, dtypes and *initial values* for the same
``torch.manual_seed`` -- the checkpoint layout (reference ``training/utils.py:43-51``, ``hubconf.py:37``) depends on
it.  The module is a parameter container: the training arithmetic never runs through ``forward`` of these modules
but through the HIP engine (``fullbatchtraining_amd.engine``), which mirrors parameters into its flat arena.
``forward`` is provided for host-side inspection only and runs plain torch ops.

Init parity requires consuming the torch CPU RNG in the same order as the reference constructor
(``resnets.py:68-126,128-177``): stem conv, then per stage [shortcut conv, block convs ...], then fc, followed by
the Kaiming-normal (fan_out, relu) re-initialisation of every conv in ``modules()`` order and constant BN
affine parameters.  ``zero_init_residual`` is never active on the training path.
"""
import os
from contextlib import nullcontext

import torch
from torch import nn

_LAYOUT = {
    18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]),
    50: ("bottleneck", [3, 4, 6, 3]), 101: ("bottleneck", [3, 4, 23, 3]), 152: ("bottleneck", [3, 8, 36, 3]),
    20: ("basic", [3, 3, 3]), 32: ("basic", [5, 5, 5]), 56: ("basic", [9, 9, 9]), 110: ("basic", [18, 18, 18]),
}


def _conv(cin, cout, k, stride=1, pad=0):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=pad, bias=False)


class _Shortcut(nn.Sequential):
    """downsample 'C' (reference resnets.py:147-152): AvgPool2d(stride) -> 1x1 conv -> BN; indices 0,1,2."""

    def __init__(self, cin, cout, stride):
        super().__init__(nn.AvgPool2d(kernel_size=stride, stride=stride), _conv(cin, cout, 1), nn.BatchNorm2d(cout))


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride, shortcut):
        super().__init__()
        self.conv1 = _conv(cin, planes, 3, stride, 1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.nonlin = nn.ReLU(inplace=True)
        self.conv2 = _conv(planes, planes, 3, 1, 1)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = shortcut
        self.stride = stride

    def forward(self, x):
        out = self.nonlin(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.nonlin(out + (x if self.downsample is None else self.downsample(x)))


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride, shortcut):
        super().__init__()
        self.conv1 = _conv(cin, planes, 1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv(planes, planes, 3, stride, 1)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = _conv(planes, planes * 4, 1)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.nonlin = nn.ReLU(inplace=True)
        self.downsample = shortcut
        self.stride = stride

    def forward(self, x):
        out = self.nonlin(self.bn1(self.conv1(x)))
        out = self.nonlin(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.nonlin(out + (x if self.downsample is None else self.downsample(x)))


class ResNet(nn.Module):
    def __init__(self, depth, channels, classes, stem="CIFAR", downsample="C"):
        super().__init__()
        if downsample != "C":
            raise NotImplementedError(f"downsample={downsample!r}: the engine implements the reference default 'C' only.")
        kind, stages = _LAYOUT[depth]
        block = BasicBlock if kind == "basic" else Bottleneck
        self.depth, self.kind, self.stem_kind, self.classes, self.channels = depth, kind, stem, classes, channels
        inplanes = 64
        if stem == "CIFAR":
            self.stem = nn.Sequential(_conv(channels, inplanes, 3, 1, 1), nn.BatchNorm2d(inplanes), nn.ReLU(inplace=True))
        elif stem == "standard":
            self.stem = nn.Sequential(_conv(channels, inplanes, 7, 2, 3), nn.BatchNorm2d(inplanes), nn.ReLU(inplace=True),
                                      nn.MaxPool2d(kernel_size=3, stride=2, padding=1))
        else:
            raise ValueError(f"Invalid stem designation {stem}.")
        stage_modules, width = [], 64
        for si, nblocks in enumerate(stages):
            stride = 1 if si == 0 else 2
            shortcut = None
            if stride != 1 or inplanes != width * block.expansion:
                shortcut = _Shortcut(inplanes, width * block.expansion, stride)  # built before the block (RNG order)
            blocks = [block(inplanes, width, stride, shortcut)]
            inplanes = width * block.expansion
            blocks += [block(inplanes, width, 1, None) for _ in range(1, nblocks)]
            stage_modules.append(nn.Sequential(*blocks))
            width *= 2
        self.layers = nn.Sequential(*stage_modules)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(inplanes, classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        x = self.layers(self.stem(x))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def construct_model(cfg_model, channels, classes):
    """cfg_model: the ``model`` group of the cfg tree (keys name, depth, stem, convolution, nonlin_fn, normalization, downsample)."""
    if "resnet" not in cfg_model.name.lower():
        raise NotImplementedError(f"Model family {cfg_model.name!r} is outside the engine's scope (ResNet-style CNNs only).")
    if cfg_model.convolution.lower() not in ("standard", "default", "zeros"):
        raise NotImplementedError(f"convolution={cfg_model.convolution!r} not supported by the HIP conv kernels.")
    if cfg_model.normalization != "BatchNorm2d" or cfg_model.nonlin_fn.lower() != "relu":
        raise NotImplementedError("Only BatchNorm2d + ReLU networks are implemented by the fused BN-ReLU kernels.")
    return ResNet(cfg_model.depth, channels, classes, stem=cfg_model.stem, downsample=cfg_model.downsample)


def prepare_model(model, cfg, process_idx, setup):
    """Reference ``models.py:55-78``: move to device, broadcast rank-0 parameters, create ``checkpoints/``."""
    model.to(**setup)
    if cfg.impl.setup.dist and torch.distributed.is_initialized():
        for param in model.parameters():
            torch.distributed.broadcast(param.data, 0)
        for buf in model.buffers():
            torch.distributed.broadcast(buf.data, 0)
        torch.distributed.barrier()
    else:
        model.no_sync = nullcontext
    os.makedirs(os.path.join(cfg.original_cwd, "checkpoints"), exist_ok=True)
    return model
