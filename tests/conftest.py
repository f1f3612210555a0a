import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def sd_from(g, prefix):
    """state-dict-like ordered list of (key, array) stored with a prefix by make_golden.py."""
    out = {}
    for k, v in g.items():
        if k.startswith(prefix):
            out[k[len(prefix):].replace("__", ".")] = v
    return out


def params_in_order(sd):
    """enc_layers.0.weight, enc_layers.0.bias, ..., dec_layers.N.bias (the order of net.parameters())."""
    def key(k):
        part, idx, kind = k.split(".")
        return (0 if part == "enc_layers" else 1, int(idx), 0 if kind == "weight" else 1)
    return [sd[k] for k in sorted(sd.keys(), key=key)], sorted(sd.keys(), key=key)


@pytest.fixture(scope="session")
def golden():
    return load_golden
