"""The tokenizer oracle has no reference-side pin (diffusers absent, SURVEY.md F2: PARITY UNPINNED).  What can be
checked offline is checked here: every building block against plain torch.nn.functional compositions written
independently of the oracle, plus structural invariants of the causal tokenizer.  CPU only.
"""
import torch
import torch.nn.functional as F

from oracle import vae_oracle as VO


def _oracle(pkg, dtype=torch.float32):
    sw = pkg.synthetic_weights
    return VO.VaeOracle(sw.synth_vae_state_dict(sw.COSMOS_CV8x8x8, torch.float32), sw.COSMOS_CV8x8x8, dtype)


def test_state_dict_shape_contract(pkg):
    shapes = pkg.synthetic_weights.vae_param_shapes()
    assert shapes["encoder.conv_in.conv_s.weight"] == (128, 192, 1, 3, 3)
    assert shapes["encoder.down_blocks.0.resnets.0.conv_shortcut.weight"] == (256, 128, 1, 1, 1)
    assert shapes["encoder.conv_out.conv_t.weight"] == (16, 16, 3, 1, 1)
    assert shapes["decoder.up_blocks.2.resnets.0.conv1.conv_s.weight"] == (256, 512, 1, 3, 3)
    assert shapes["decoder.conv_out.conv_s.weight"] == (192, 256, 1, 3, 3)
    assert "encoder.down_blocks.1.downsamplers.0.conv3.weight" not in shapes     # both flags off: identity
    assert "decoder.up_blocks.0.upsamplers.0.conv3.weight" not in shapes
    assert "decoder.up_blocks.1.upsamplers.0.conv1.weight" in shapes


def test_causal_conv_is_conv3d_with_replicated_first_frame():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, 5, 6, 7, generator=g)
    w = torch.randn(3, 4, 3, 1, 1, generator=g)
    b = torch.randn(3, generator=g)
    y = VO.causal_conv3d(x, w, b)
    xp = torch.cat([x[:, :, :1], x[:, :, :1], x], 2)
    assert torch.allclose(y, F.conv3d(xp, w, b), atol=1e-6)
    assert y.shape == x[:, :3].shape
    # no look-ahead
    x2 = x.clone()
    x2[:, :, 3:] += 1
    assert torch.equal(VO.causal_conv3d(x2, w, b)[:, :, :3], y[:, :, :3])
    # spatial kernel: zero padding 1
    ws = torch.randn(3, 4, 1, 3, 3, generator=g)
    assert torch.allclose(VO.causal_conv3d(x, ws, b, padding=1), F.conv3d(x, ws, b, padding=(0, 1, 1)), atol=1e-6)


def test_group_norm_is_per_frame():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 8, 3, 4, 5, generator=g)
    w, b = torch.randn(8, generator=g), torch.randn(8, generator=g)
    y = VO.group_norm_causal(x, w, b)
    for t in range(3):
        f = x[0, :, t]
        ref = (f - f.mean()) / torch.sqrt(f.var(unbiased=False) + 1e-6) * w[:, None, None] + b[:, None, None]
        assert torch.allclose(y[0, :, t], ref, atol=1e-5)


def test_haar_patch_roundtrip_and_layout(pkg):
    v = _oracle(pkg)
    x = pkg.synthetic_weights.synth_tensor("vo.x", (1, 3, 9, 16, 24), torch.float32)
    p = v.patch(x)
    assert p.shape == (1, 192, 3, 4, 6)
    assert torch.allclose(v.unpatch(p), x, atol=1e-5)
    # channel 0 (lll of lll, colour 0) is the 4x4x4 block mean: each level's low-pass is 2^-1.5 * sum8 / sqrt(8)
    xr = torch.cat([x[:, :, :1].repeat(1, 1, 4, 1, 1), x[:, :, 1:]], 2)
    blk = xr[0, 0].reshape(3, 4, 4, 4, 6, 4).mean((1, 3, 5))
    assert torch.allclose(p[0, 0], blk, atol=1e-5)


def test_encode_decode_shapes_causality_and_image_mode(pkg):
    v = _oracle(pkg)
    x = pkg.synthetic_weights.synth_tensor("vo.y", (1, 3, 9, 32, 32), torch.float32)
    with torch.no_grad():
        z = v.encode(x)
        assert z.shape == (1, 16, 2, 4, 4)
        y = v.decode(z)
        assert y.shape == x.shape
        z1 = v.encode(x[:, :, :1])
        assert z1.shape == (1, 16, 1, 4, 4)
        assert torch.allclose(z1, z[:, :, :1], atol=1e-5)          # T = 1 image path == first latent frame of the video
        x2 = x.clone()
        x2[:, :, 1:] += 0.5
        assert torch.allclose(v.encode(x2)[:, :, :1], z[:, :, :1], atol=1e-5)   # latent frame 0 sees pixel frame 0 only
        assert v.decode(z1).shape == (1, 3, 1, 32, 32)


def test_bf16_mode_tracks_fp32(pkg):
    x = pkg.synthetic_weights.synth_tensor("vo.z", (1, 3, 9, 32, 32), torch.float32)
    with torch.no_grad():
        z32 = _oracle(pkg).encode(x)
        z16 = _oracle(pkg, torch.bfloat16).encode(x)
    assert ((z16.float() - z32).norm() / z32.norm()).item() < 5e-2


def test_tokenizer_hyperparameters_equal_reference_config(pkg):
    """synthetic_weights.COSMOS_CV8x8x8 (what HipCosmosTokenizer and the oracle are built from) against the hyper-parameters of
    the reference's VAE_config.json:1-30, committed as tests/golden/vae_config_hparams.json by tools/make_goldens.py."""
    import json
    import os
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "vae_config_hparams.json")) as f:
        ref = json.load(f)
    ours = pkg.synthetic_weights.COSMOS_CV8x8x8
    for k, v in ours.items():
        assert k in ref, k
        assert (list(v) if isinstance(v, tuple) else v) == ref[k], (k, v, ref[k])
    assert ref["_class_name"] == "AutoencoderKLCosmos" and ref["latents_mean_count"] == 16 * 16     # unused by the reference (F10)
    # every structural key of the reference config is covered
    assert set(ref) - set(ours) == {"_class_name", "_diffusers_version", "latents_mean_count", "latents_mean_sum",
                                    "latents_std_count", "latents_std_sum"}
