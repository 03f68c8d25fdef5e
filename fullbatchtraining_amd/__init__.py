"""MI355X-native full-batch gradient-descent engine (hot path of JonasGeiping/fullbatchtraining).

Sub-modules mirror the reference package layout for the path that is implemented:
``models`` (construct_model / prepare_model), ``training`` (train, evaluate, checkpoints, optim_interface),
``cfg`` (Hydra-free composer of the same config tree), ``engine`` (GPU schedule), ``lib`` (ctypes binding of libfbengine.so),
``parallel`` (chunk sharding + reduce-scatter/all-gather step).
"""
from . import cfg, models  # noqa: F401  (light imports; engine/training import torch.distributed lazily)

__all__ = ["cfg", "models", "training", "engine", "lib", "parallel"]


def __getattr__(name):
    if name in ("training", "engine", "lib", "parallel", "build"):
        import importlib

        return importlib.import_module(f".{name}", __name__)
    raise AttributeError(name)
