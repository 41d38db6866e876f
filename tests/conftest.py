import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """-> dict group -> dict key -> torch tensor (CPU)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for full in z.files:
        g, k = full.split("/", 1)
        out.setdefault(g, {})[k] = torch.from_numpy(z[full])
    return out


def load_golden_json(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def draw_param(shape, g, scale):
    """Same draw as tests/golden/make_fixtures.py::draw_param."""
    shape = tuple(shape)
    t = torch.randn(shape, generator=g)
    if len(shape) > 1:
        return t * (3.0 * scale / max(1.0, shape[-1] ** 0.5))
    return t * scale + 0.5


def redraw_state_dict(meta):
    """Rebuild the parameters of a fixture that stores only (key order, seed)."""
    g = torch.Generator().manual_seed(meta["seed"])
    return {k: draw_param(shape, g, meta["scale"]) for k, shape in meta["param_order"]}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    return torch.cuda.is_available()
