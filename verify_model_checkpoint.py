"""Counterpart of the reference's verify_model_checkpoint.py: load a 5-list checkpoint (written by this engine or by the
reference: same layout, same state_dict keys) and evaluate it on the validation set.

    python verify_model_checkpoint.py impl.checkpoint.name=<file under ./checkpoints> [cfg overrides] [data.size=10000]

This environment has no dataset access: the validation set is synthetic CIFAR-shaped data (`data.size` images)."""
import os
import sys

import torch

from fullbatchtraining_amd.cfg import compose
from fullbatchtraining_amd.models import construct_model
from fullbatchtraining_amd.training import evaluate


def main():
    over = [a for a in sys.argv[1:] if not a.startswith("data.size=")]
    size = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("data.size=")), 10000)
    cfg = compose(over, name="evaluation")
    if cfg.impl.checkpoint.name is None:
        raise ValueError("Could not load checkpoint")
    model = construct_model(cfg.model, 3, 10)
    file = os.path.join(cfg.original_cwd, "checkpoints", cfg.impl.checkpoint.name)
    _, model_state, _, _, step = torch.load(file, map_location="cpu", weights_only=False)
    model.load_state_dict(model_state)
    print(f"Loaded model checkpoint from step {step} successfully.")
    gen = torch.Generator().manual_seed(4321)
    x, y = torch.randn(size, 3, 32, 32, generator=gen), torch.randint(0, 10, (size,), generator=gen)
    setup = dict(device=torch.device("cuda:0"), dtype=torch.float, memory_format=torch.contiguous_format)
    stats = evaluate(model, (x, y), None, setup, cfg.impl, cfg.hyp, dryrun=cfg.dryrun)
    print(f'VAL loss {stats["valid_loss"][-1]:7.4f} | VAL Acc: {stats["valid_acc"][-1]:7.2%} |')


if __name__ == "__main__":
    main()
