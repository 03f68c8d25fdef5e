"""Counterpart of the reference's measure_floating_point_accuracy.py: evaluates the full-batch gradient twice from the same
checkpoint and prints the L-inf / L2 / L1 difference (0.0 here: every reduction of the engine has a fixed order).

    python measure_floating_point_accuracy.py [cfg overrides, e.g. hyp=gradreg impl.mixed_precision=True data.size=5120]

Synthetic CIFAR-shaped data (this environment has no dataset access); `data.size` images (default 5120)."""
import sys

import torch

from fullbatchtraining_amd.cfg import compose
from fullbatchtraining_amd.models import construct_model
from fullbatchtraining_amd.training import _measure_implementation_noise


def main():
    over = [a for a in sys.argv[1:] if not a.startswith("data.size=")]
    size = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("data.size=")), 5120)
    if not any(a.startswith("hyp=") for a in over):
        over = ["hyp=fb1"] + over                      # full-batch GD (the reference's train_stochastic=False branch)
    cfg = compose(over, name="fp_noise")
    torch.manual_seed(cfg.seed if getattr(cfg, "seed", None) is not None else 0)
    model = construct_model(cfg.model, 3, 10)
    gen = torch.Generator().manual_seed(1234)
    x, y = torch.randn(size, 3, 32, 32, generator=gen), torch.randint(0, 10, (size,), generator=gen)
    setup = dict(device=torch.device("cuda:0"), dtype=torch.float, memory_format=torch.contiguous_format)
    _measure_implementation_noise(model, (x, y), None, setup, cfg)


if __name__ == "__main__":
    main()
