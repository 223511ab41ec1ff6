"""Forward-renderer env-map conditions: plain-torch preprocessing, CPU-testable.
The nvdiffrast cube lookup of the reference is replaced by an own bilinear fetch (PARITY UNPINNED for that step)."""
import math

import numpy as np
import pytest
import torch


def _pe(pkg):
    import importlib
    return importlib.import_module(pkg.__name__ + ".preprocess_envmap")


def test_tone_maps_match_formulas(pkg):
    pe = _pe(pkg)
    x = torch.tensor([0.0, 0.001, 0.0031308, 0.01, 0.5, 1.0, 4.0, 100.0, 65504.0])
    srgb = pe.rgb2srgb_official(x.clamp(0, 1))
    assert torch.allclose(srgb[1], torch.tensor(12.92 * 0.001))
    assert torch.allclose(srgb[5], torch.tensor(1.0), atol=1e-6)
    m = pe.hdr_mapping_official(x)
    ev0 = pe.rgb2srgb_official((x / (x + 1) * 16.0).clamp(0, 1))
    assert torch.equal(m["env_ev0"], ev0)
    assert torch.allclose(m["env_log"], pe.rgb2srgb_official(torch.log1p(x) / math.log1p(10000.0)).clamp(0, 1))
    assert m["env_log"].max() <= 1.0 and m["env_log"][0] == 0.0


def test_latlong_vec_is_unit_and_oriented(pkg):
    pe = _pe(pkg)
    v = pe.latlong_vec((16, 32), device="cpu")
    assert v.shape == (16, 32, 3)
    assert torch.allclose(v.norm(dim=-1), torch.ones(16, 32), atol=1e-6)
    assert v[0, :, 1].min() > 0.9 and v[-1, :, 1].max() < -0.9           # top row looks up (+Y), bottom row down
    assert torch.allclose(pe.latlong_vec(resolution=(16, 32), device="cpu"), v)


def test_cube_lookup_inverts_cube_to_dir_and_tracks_panorama(pkg):
    pe = _pe(pkg)
    # exactness of the face/coordinate inversion: look up the texel centres of a cube map whose value encodes (face, row, col)
    R = 8
    cube = torch.zeros(6, R, R, 3)
    for s in range(6):
        cube[s, ..., 0] = s
        cube[s, ..., 1] = torch.arange(R).view(R, 1)
        cube[s, ..., 2] = torch.arange(R).view(1, R)
    lin = torch.linspace(-1 + 1 / R, 1 - 1 / R, R)
    gy, gx = torch.meshgrid(lin, lin, indexing="ij")
    for s in range(6):
        got = pe.cube_lookup(cube, pe.cube_to_dir(s, gx, gy) * 3.7)      # any positive scale of the direction
        assert torch.allclose(got, cube[s], atol=1e-4), s
    # a smooth panorama sampled through the 64^2 cube map agrees with sampling the panorama directly
    H, W = 64, 128
    d = pe.latlong_vec((H, W), device="cpu")
    pano = torch.stack([d[..., 0] * 0.5 + 0.5, d[..., 1] * 0.5 + 0.5, (d[..., 2] * d[..., 0]) * 0.5 + 0.5], -1)
    cm = pe.latlong_to_cubemap_official(pano, [64, 64])
    q = pe.safe_normalize(torch.randn(500, 3, generator=torch.Generator().manual_seed(0)))
    got = pe.cube_lookup(cm, q)
    tu = torch.atan2(q[:, 0], -q[:, 2]) / (2 * np.pi) + 0.5
    tv = torch.acos(q[:, 1].clamp(-1, 1)) / np.pi
    grid = (torch.stack([tu, tv], -1) * 2 - 1).view(1, 1, -1, 2)
    ref = torch.nn.functional.grid_sample(pano.permute(2, 0, 1)[None], grid, mode="bilinear", padding_mode="border",
                                          align_corners=False)[0, :, 0].t()
    assert (got - ref).abs().max() < 0.05


def test_conditions_shapes_cache_and_errors(pkg):
    pe = _pe(pkg)
    pe.clear_environment_cache()
    env = torch.rand(1, 32, 64, 3) * 5.0
    c = pe.envmap_conditions(env, (16, 24), 9, "proj", 1.5, True, 90.0, device="cpu")
    assert c["env_ldr"].shape == (1, 3, 9, 16, 24) and c["env_log"].shape == (1, 3, 9, 16, 24)
    assert c["env_nrm"].shape == (1, 3, 1, 16, 24)
    assert -1.0 <= c["env_ldr"].min() and c["env_ldr"].max() <= 1.0
    assert pe.get_cache_stats()["cache_size"] == 1
    c2 = pe.envmap_conditions(env, (16, 24), 9, "proj", 1.5, True, 90.0, device="cpu")
    assert c2["env_ldr"].data_ptr() == c["env_ldr"].data_ptr() or torch.equal(c2["env_ldr"], c["env_ldr"])
    b = pe.envmap_conditions(torch.rand(20, 20, 3), (16, 24), 1, "ball", device="cpu")
    assert b["env_ldr"].shape == (1, 3, 1, 16, 24)
    with pytest.raises(ValueError):
        pe.envmap_conditions(env, (16, 24), 1, "sphere", device="cpu")
    with pytest.raises(ValueError):
        pe.tonemap_image_direct(123, (4, 4), device="cpu")
    pe.clear_environment_cache()
    assert pe.get_cache_stats()["cache_size"] == 0
