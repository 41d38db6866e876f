"""Checkpoint layout of the reference (/root/reference/openeat/utils/checkpoint.py):
a flat ``state_dict`` in ``<epoch>.pt`` plus a YAML side-car ``<epoch>.yaml``."""
import logging
import os
import re
from collections import OrderedDict

import torch
import yaml


def _unwrap(model):
    return model.module if hasattr(model, "module") and isinstance(model.module, torch.nn.Module) else model


def _sidecar(path):
    info = re.sub(r"\.pt$", ".yaml", path)
    if os.path.exists(info):
        with open(info, "r") as f:
            return yaml.load(f, Loader=yaml.FullLoader) or {}
    return {}


def load_checkpoint(model: torch.nn.Module, path: str) -> dict:
    """checkpoint.py:12-27: keys unknown to the model are dropped, strict=False."""
    logging.info("Checkpoint: loading from checkpoint %s", path)
    ckpt = torch.load(path, map_location="cpu")
    own = model.state_dict()
    model.load_state_dict({k: v for k, v in ckpt.items() if k in own}, strict=False)
    return _sidecar(path)


def save_checkpoint(model: torch.nn.Module, path: str, infos=None):
    """checkpoint.py:30-48."""
    logging.info("Checkpoint: save to checkpoint %s", path)
    torch.save({k: v.detach().cpu() for k, v in _unwrap(model).state_dict().items()}, path)
    with open(re.sub(r"\.pt$", ".yaml", path), "w") as f:
        f.write(yaml.dump(infos or {}))


def filter_modules(model_state_dict, modules):
    present = [m for m in modules if any(k.startswith(m) for k in model_state_dict)]
    missing = [m for m in modules if m not in present]
    if missing:
        logging.warning("module(s) %s don't match available modules in the checkpoint", missing)
    return present


def load_trained_modules(model: torch.nn.Module, path: str, select_modules: list):
    """checkpoint.py:71-96: partial initialisation by key prefix."""
    target = model.state_dict()
    if os.path.isfile(path):
        src = torch.load(path, map_location="cpu")
        mods = filter_modules(src, select_modules)
        picked = OrderedDict((k, v) for k, v in src.items()
                             if any(k.startswith(m) for m in mods) and "concat_linear" not in k)
        target.update(picked)
    else:
        logging.warning("model was not found : %s", path)
    model.load_state_dict(target)
    return _sidecar(path)
