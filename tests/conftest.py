import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def small_cfg(F=2, T=8, X=9, Y=7, n_levels=16):
    """config_nvp_s / config_nvp_l values (reference config/*.json) with a small sparse grid."""
    enc = {"otype": "DenseGrid", "n_levels": n_levels, "n_features_per_level": F, "log2_hashmap_size": 32,
           "base_resolution": 16, "per_level_scale": 1.35}
    return {
        "2d_encoding_xy": dict(enc), "2d_encoding_xt": dict(enc), "2d_encoding_yt": dict(enc),
        "3d_encoding": {"otype": "SparseGrid", "n_features_per_level": F, "x_resolution": X, "y_resolution": Y,
                        "t_resolution": T, "upsample": False},
        "network": {"n_neurons": 128, "n_hidden_layers": 3},
    }
