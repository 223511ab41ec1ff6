"""Host-side launch planning (no GPU): how the attention wrapper covers the (q-block, head) grid with whole CU rounds."""


def test_attention_plan_fills_whole_rounds(pkg):
    plan = pkg.native.attention_plan
    # full clip on one GPU: 72 q-blocks x 32 heads = 2304 workgroups = 9 whole rounds -> one unsplit launch
    assert plan(1, 32, 18432, 18432) == [(0, 18432, 1)]
    # 8-way token bands (gather form): 288 workgroups = 1 round + 32 -> 8 q-blocks unsplit, the last one in 8 key chunks
    assert plan(1, 32, 2304, 18432) == [(0, 2048, 1), (2048, 2304, 8)]
    # 8-way head all-to-all: 4 heads x 72 q-blocks, same 288 workgroups
    assert plan(1, 4, 18432, 18432) == [(0, 16384, 1), (16384, 18432, 8)]
    # 2-way: 1152 workgroups = 4 rounds + 128 -> tail in 2 chunks
    assert plan(1, 16, 18432, 18432) == [(0, 16384, 1), (16384, 18432, 2)]
    # less than one round in total: split everything (flash-decoding style)
    (q0, q1, n), = plan(1, 2, 256, 8192)
    assert (q0, q1) == (0, 256) and n == 8
    # tiny: nothing to gain
    assert plan(1, 2, 128, 128) == [(0, 128, 1)]
    # every plan covers [0, Sq) exactly once, cuts on 256-row boundaries
    for args in [(1, 32, 4608, 18432), (5, 32, 256, 256), (2, 32, 2048, 2048), (1, 8, 18432, 18432), (3, 4, 1000, 5000)]:
        p = plan(*args)
        assert p[0][0] == 0 and p[-1][1] == args[2]
        for a, b in zip(p, p[1:]):
            assert a[1] == b[0] and a[1] % 256 == 0


# ---------------------------------------------------------------------------------------------------------------------
# condition assembly against goldens captured from the reference's own _get_conditions /
# prepare_diffusion_renderer_latent_conditions (model_diffusion_renderer.py:158-209; tools/make_goldens.py conditions_case)
import json

import pytest
import torch

from conftest import load_golden
from stub_vae import StubVAE

COND_KEYS8 = ["basecolor", "normal", "metallic", "roughness", "depth", "env_ldr", "env_log", "env_nrm"]


def _cond_model(pkg, forward, mask, T, H, W, drop_keys=False):
    cfgm = pkg.diffusion_renderer_config
    cfg = (cfgm.get_forward_renderer_config if forward else cfgm.get_inverse_renderer_config)(H, W, T)
    cfg = dict(cfg, append_condition_mask=mask)
    if drop_keys:
        cfg.pop("condition_keys")
    m = pkg.model_diffusion_renderer.CleanDiffusionRendererModel(cfg, device="cpu")   # host logic only: no weights, no GPU
    m.vae = StubVAE()
    return m


@pytest.mark.parametrize("case,forward,mask,drop", [
    ("forward.all8.mask", True, True, ()), ("forward.all8.nomask", True, False, ()),
    ("forward.no_roughness.mask", True, True, ("roughness",)), ("forward.no_env.mask", True, True, ("env_ldr", "env_log", "env_nrm")),
    ("inverse.rgb", False, None, ()), ("inverse.default_keys.mask", False, True, ())])
def test_condition_assembly_equals_reference(pkg, case, forward, mask, drop):
    gold, meta = load_golden("conditions.safetensors")
    T, (H, W) = int(meta["T"]), json.loads(meta["HW"])
    sw = pkg.synthetic_weights
    assert pkg.diffusion_renderer_config.get_forward_renderer_config(H, W, T)["condition_keys"] == json.loads(meta["forward_keys"])
    assert pkg.diffusion_renderer_config.get_inverse_renderer_config(H, W, T)["condition_keys"] == json.loads(meta["inverse_keys"])

    def clip(k):
        return sw.synth_tensor("cond8." + k, (1, 3, T, H, W), torch.float32, scale=1.0).to(torch.bfloat16)

    if forward:
        batch = {k: clip(k) for k in COND_KEYS8 if k not in drop}
    else:
        rgb = clip("rgb")
        batch = {"rgb": rgb, "video": rgb}
    if mask is None:
        mask = meta["inverse_mask"] == "True"
    m = _cond_model(pkg, forward, mask, T, H, W, drop_keys=(case == "inverse.default_keys.mask"))
    cond, uncond = m._get_conditions(batch)
    lc = cond.to_dict()["latent_condition"]
    assert lc.dtype == gold[case].dtype and torch.equal(lc, gold[case]), case
    assert batch["latent_condition"] is lc                          # the reference writes it into the batch too
    assert not uncond.to_dict()["latent_condition"].any()           # uncond = zeros_like (CleanConditioner :88-96)
    want_key = json.loads(meta["input_data_key"]).get(case.rsplit(".", 1)[0] if forward else case)
    if want_key:
        assert m.input_data_key == want_key


def test_condition_assembly_errors_like_reference(pkg):
    m = _cond_model(pkg, True, True, 9, 32, 48)
    with pytest.raises(ValueError):                                  # no condition key present -> no latent shape (:177-178)
        m.prepare_diffusion_renderer_latent_conditions({"video": torch.zeros(1, 3, 9, 32, 48)})
    m.vae = None
    with pytest.raises(RuntimeError):
        m.prepare_diffusion_renderer_latent_conditions({"basecolor": torch.zeros(1, 3, 9, 32, 48)})
