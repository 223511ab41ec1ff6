import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    from __graft_entry__ import load_package
    return load_package()


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def full28_dit(pkg, gpu):
    """The 28-block, 7.2 B-parameter inverse-renderer DiT with synthetic weights, built once per session (14.5 GB of HBM)."""
    import torch
    net = tiny_net(pkg, 4096, 28, 32)
    sd = pkg.synthetic_weights.synth_state_dict(net, torch.bfloat16, device=gpu)
    dit = pkg.dit_engine.HipDiT(net, sd, device=gpu)
    del sd
    torch.cuda.empty_cache()
    return dit


def load_golden(name):
    from safetensors import safe_open
    path = os.path.join(GOLDEN, name)
    tensors = {}
    with safe_open(path, framework="pt") as f:
        meta = f.metadata()
        for k in f.keys():
            tensors[k] = f.get_tensor(k)
    return tensors, meta


def tiny_net(pkg, D, L, heads, forward=False):
    cfgm = pkg.diffusion_renderer_config
    cfg = cfgm.get_forward_renderer_config() if forward else cfgm.get_inverse_renderer_config()
    net = dict(cfg["net"])
    net.update(model_channels=D, num_blocks=L, num_heads=heads)
    return net


def rel_l2(a, b):
    a = a.double()
    b = b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
