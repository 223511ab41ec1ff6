"""Pins the CPU oracle: oracle/dit_oracle.py must reproduce, bit for bit, the golden vectors that
tools/make_goldens.py captured by running the reference itself (CPU, F1 patch) in the build container.
Runs without a GPU.  Weights/inputs are regenerated from the hash generator (not stored in the fixtures).
"""
import json

import pytest
import torch

from conftest import load_golden, tiny_net
from oracle import dit_oracle as O
from oracle.ref_import import reference_available

CASES = [  # fixture, tag, D, L, heads, forward
    ("dit_tinyA.safetensors", "tinyA", 256, 1, 2, False),
    ("dit_tinyB.safetensors", "tinyB", 512, 2, 4, False),
    ("dit_tinyF_forward.safetensors", "tinyF", 256, 1, 2, True),
]


def _inputs(pkg, tag, meta, net, dtype):
    sw = pkg.synthetic_weights
    F_, h, w = json.loads(meta["latent"])
    x = sw.synth_tensor(tag + ".x", (1, 16, F_, h, w), torch.float32, scale=2.0).to(torch.bfloat16).to(dtype)
    cond = sw.synth_tensor(tag + ".cond", (1, net["additional_concat_ch"], F_, h, w), torch.float32,
                           scale=1.0).to(torch.bfloat16).to(dtype)
    t = torch.tensor(float(meta["sigma"]), dtype=torch.float32)
    ci = torch.full((1, 1), int(meta["context_index"]), dtype=torch.long)
    return x, t, cond, ci


@pytest.mark.parametrize("fixture,tag,D,L,heads,forward", CASES)
def test_oracle_matches_reference_goldens(pkg, fixture, tag, D, L, heads, forward):
    gold, meta = load_golden(fixture)
    net = tiny_net(pkg, D, L, heads, forward)
    sd = pkg.synthetic_weights.synth_state_dict(net, torch.bfloat16)
    for dn, dtype in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        orc = O.DitOracle(sd, net, dtype=dtype)
        orc.trace = {}
        with torch.no_grad():
            y = orc.forward(*_inputs(pkg, tag, meta, net, dtype))
        assert torch.equal(y, gold[f"out.{dn}"]), f"{tag}/{dn}"
        for i in range(L):
            for j in range(3):
                assert torch.equal(orc.trace[f"block{i}.{j}"].squeeze(1), gold[f"block{i}.{j}.{dn}"])
    orc = O.DitOracle(sd, net, dtype=torch.float32, tables_dtype=torch.bfloat16)
    with torch.no_grad():
        y = orc.forward(*_inputs(pkg, tag, meta, net, torch.bfloat16))
    assert torch.equal(y, gold["out.fp32_tables_bf16"])


def test_oracle_wide_single_block(pkg):
    """Full-width block (D=4096, 32 heads) at S=256 against the reference's output."""
    gold, meta = load_golden("dit_wide1.safetensors")
    net = tiny_net(pkg, 4096, 1, 32)
    sd = pkg.synthetic_weights.synth_state_dict(net, torch.bfloat16)
    orc = O.DitOracle(sd, net, dtype=torch.bfloat16)
    with torch.no_grad():
        y = orc.forward(*_inputs(pkg, "wide1", meta, net, torch.bfloat16))
    assert torch.equal(y, gold["out.bf16"])


def test_scheduler_goldens():
    gold, _ = load_golden("scheduler.safetensors")
    for n in (4, 15, 35):
        assert torch.equal(O.edm_sigmas(n), gold[f"sigmas.{n}"])


def test_scheduler_step_goldens(pkg):
    gold, _ = load_golden("scheduler.safetensors")
    sw = pkg.synthetic_weights
    x = sw.synth_tensor("sched.x", (1, 16, 2, 8, 8), torch.float32, scale=40.0).to(torch.bfloat16)
    mo = sw.synth_tensor("sched.mo", (1, 16, 2, 8, 8), torch.float32, scale=1.0).to(torch.bfloat16)
    sig = O.edm_sigmas(4)
    assert torch.equal(O.edm_scale_input(x, sig[1]), gold["scaled"])
    assert torch.equal(O.edm_step(mo, sig[1], sig[2], x), gold["stepped"])
    assert torch.equal(O.edm_step(mo, sig[3], sig[4], x), gold["stepped_last"])


@pytest.mark.parametrize("fixture,tag", [("sampler_tinyA_g0.safetensors", "samplerA"),
                                         ("sampler_tinyB_g2.safetensors", "samplerB")])
def test_sampler_trajectory_goldens(pkg, fixture, tag):
    """EDM loop + CFG + post-process with the injected x_T the reference drew (stub tokenizer latents stored)."""
    gold, meta = load_golden(fixture)
    net = tiny_net(pkg, int(meta["D"]), int(meta["L"]), int(meta["heads"]))
    sd = pkg.synthetic_weights.synth_state_dict(net, torch.bfloat16)
    orc = O.DitOracle(sd, net, dtype=torch.bfloat16)
    steps, g = int(meta["steps"]), float(meta["guidance"])
    ci = torch.full((1, 1), int(meta["context_index"]), dtype=torch.long).to(torch.bfloat16)
    traj = []
    with torch.no_grad():
        x0 = O.sample_loop(orc.forward, gold["xT"], gold["latent_condition"], ci, steps, g, record=traj)
    for i, s in enumerate(traj):
        assert torch.equal(s, gold[f"traj.{i}"]), f"step {i}"
    assert torch.equal(x0, gold["x0"])


@pytest.mark.skipif(not reference_available(), reason="reference not present (GPU box)")
def test_oracle_equals_live_reference(pkg):
    """Where /root/reference exists (build container), re-run it and compare live (not just via fixtures)."""
    from oracle.ref_import import load_reference
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    ref = load_reference()
    net = tiny_net(pkg, 256, 2, 2)
    sd = pkg.synthetic_weights.synth_state_dict(net, torch.bfloat16)
    cfg = ref.config.get_inverse_renderer_config()
    cfg["net"] = dict(net)
    model = ref.model.CleanDiffusionRendererModel(cfg).to(torch.bfloat16)
    model.load_state_dict(sd, strict=False)
    sw = pkg.synthetic_weights
    x = sw.synth_tensor("live.x", (1, 16, 1, 8, 12), torch.float32, scale=3.0).to(torch.bfloat16)
    cond = sw.synth_tensor("live.c", (1, 16, 1, 8, 12), torch.float32, scale=1.0).to(torch.bfloat16)
    t = torch.tensor(0.37)
    ci = torch.full((1, 1), 4, dtype=torch.long)
    with torch.no_grad():
        y_ref = model.net(x=x, timesteps=t, latent_condition=cond, context_index=ci)
        y = O.DitOracle(sd, net).forward(x, t, cond, ci)
    assert torch.equal(y, y_ref)
