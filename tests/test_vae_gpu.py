"""Tokenizer kernels (through the C ABI) against torch.nn.functional / the CPU oracle.  PARITY UNPINNED at the
reference boundary (diffusers absent): what is asserted is HIP == oracle/vae_oracle.py (same roundings) and
kernel == torch.nn.functional on the same inputs.
"""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from oracle import vae_oracle as VO

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rnd(shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(BF)


def to_cl(pkg, x, gpu, halo=1):
    """[C,T,H,W] cpu bf16 -> CL on gpu."""
    V = pkg.native_vae
    C, T, H, W = x.shape
    cl = V.CL(T, H, W, C, halo, gpu)
    cl.interior().copy_(x.permute(1, 2, 3, 0).to(gpu))
    return cl


def from_cl(cl):
    return cl.interior().permute(3, 0, 1, 2).contiguous().cpu()


def repack(w, gpu):
    co, ci, kt, kh, kw = w.shape
    return w.permute(0, 2, 3, 4, 1).reshape(co, kt * kh * kw * ci).contiguous().to(gpu)


CONVS = [  # Cin, Cout, k, stride, pad, T, H, W
    (64, 128, (1, 3, 3), (1, 1, 1), 1, 3, 10, 12),
    (128, 64, (3, 1, 1), (1, 1, 1), 0, 5, 6, 9),
    (64, 192, (1, 1, 1), (1, 1, 1), 0, 2, 7, 5),
    (192, 128, (1, 3, 3), (1, 1, 1), 1, 1, 16, 16),
    (64, 16, (1, 3, 3), (1, 1, 1), 1, 2, 8, 8),
]


@pytest.mark.parametrize("Cin,Cout,k,stride,pad,T,H,W", CONVS)
def test_conv3d_igemm_matches_functional(pkg, gpu, Cin, Cout, k, stride, pad, T, H, W):
    V = pkg.native_vae
    x = rnd((Cin, T, H, W), seed=1)
    w = rnd((Cout, Cin) + k, 1.0 / (Cin * k[0] * k[1] * k[2]) ** 0.5, seed=2)
    b = rnd((Cout,), 0.1, seed=3)
    res = rnd((Cout, T, H, W), seed=4)
    y = V.conv3d(to_cl(pkg, x, gpu), repack(w, gpu), b.to(gpu), Cout, k, stride, pad, residual=to_cl(pkg, res, gpu))
    conv = VO.causal_conv3d(x.float()[None], w.float(), b.float(), stride, pad)[0].to(BF)
    ref = conv + res
    got = from_cl(y)
    assert got.shape == ref.shape
    assert rel_l2(got, ref.float()) < 3e-3
    assert (got.float() - ref.float()).abs().max() < 0.05
    assert (y.t[:, 0] == 0).all() and (y.t[:, :, 0] == 0).all() and (y.t[:, -1] == 0).all()     # halo untouched


def test_conv3d_strided_downsample_convs(pkg, gpu):
    """The hybrid down-sampler's two strided causal convs (CosmosDownsample3d)."""
    V = pkg.native_vae
    C, T, H, W = 64, 7, 8, 12
    x = rnd((C, T, H, W), seed=5)
    w1, b1 = rnd((C, C, 1, 3, 3), 0.04, seed=6), rnd((C,), 0.1, seed=7)
    xc = to_cl(pkg, x, gpu)
    y = V.conv3d(xc, repack(w1, gpu), b1.to(gpu), C, (1, 3, 3), (1, 2, 2), 0, out_dims=(T, H // 2, W // 2))
    xp = F.pad(x.float()[None], (0, 1, 0, 1, 0, 0))
    ref = VO.causal_conv3d(xp, w1.float(), b1.float(), (1, 2, 2), 0)[0]
    assert rel_l2(from_cl(y), ref) < 3e-3
    w2, b2 = rnd((C, C, 3, 1, 1), 0.07, seed=8), rnd((C,), 0.1, seed=9)
    y2 = V.conv3d(xc, repack(w2, gpu), b2.to(gpu), C, (3, 1, 1), (2, 1, 1), 0, t_off=2)
    xt = torch.cat([x.float()[None][:, :, :1], x.float()[None]], 2)
    ref2 = VO.causal_conv3d(xt, w2.float(), b2.float(), (2, 1, 1), 0)[0]
    got2 = from_cl(y2)
    assert got2.shape == ref2.shape == (C, 4, H, W)
    assert rel_l2(got2, ref2) < 3e-3


CONVS256 = [  # Cin, Cout, k, stride, pad, T, H, W, t_off, residual
    (64, 256, (1, 3, 3), (1, 1, 1), 1, 2, 20, 24, None, True),       # 960 positions: 3.75 tiles (ragged last tile), 9 taps
    (128, 256, (3, 1, 1), (1, 1, 1), 0, 5, 16, 16, None, False),     # causal clamp: frames 0 / 1 read frame 0 for kt < 2
    (64, 512, (1, 1, 1), (1, 1, 1), 0, 3, 16, 20, None, True),       # one K step: prologue only
    (192, 256, (1, 3, 3), (1, 2, 2), 0, 2, 33, 41, None, False),     # strided, 3 channel slices per tap, 27 K steps (odd)
    (64, 256, (3, 1, 1), (2, 1, 1), 0, 9, 12, 16, 2, False),         # temporal stride 2 with t_off
    (256, 256, (3, 3, 3), (1, 1, 1), 1, 3, 10, 12, None, True),      # all three tap axes, 108 K steps
]


@pytest.mark.parametrize("Cin,Cout,k,stride,pad,T,H,W,t_off,with_res", CONVS256)
def test_conv256_streamed_kernel_equals_128_kernel(pkg, gpu, Cin, Cout, k, stride, pad, T, H, W, t_off, with_res):
    """csrc/conv256s.hip (the streamed 256 x 256 tile, gathered A rows) against the 128 x 128 kernel on the same launch:
    bit-identical (same K order per output element), and against torch's conv3d; the zero halo stays untouched; the residual
    is also given IN PLACE (output aliases it), as the tokenizer's residual blocks do."""
    V = pkg.native_vae
    lib = pkg.native.load_library()
    x = rnd((Cin, T, H, W), seed=21)
    w = rnd((Cout, Cin) + k, 1.0 / (Cin * k[0] * k[1] * k[2]) ** 0.5, seed=22)
    b = rnd((Cout,), 0.1, seed=23)
    kw = {}
    if stride[1] == 2:
        Ho, Wo = (H - k[1]) // 2 + 1, (W - k[2]) // 2 + 1
        kw["out_dims"] = (T, Ho, Wo)
    else:
        Ho, Wo = H, W
    To = T if stride[0] == 1 else (T + (t_off if t_off is not None else 0) - k[0]) // stride[0] + 1
    if t_off is not None:
        kw["t_off"] = t_off
    res = rnd((Cout, To, Ho, Wo), seed=24) if with_res else None
    xc, wc, bc = to_cl(pkg, x, gpu), repack(w, gpu), b.to(gpu)
    outs = {}
    for tile in (0, 1):
        lib.drn_conv_force_tile(tile)
        try:
            r = to_cl(pkg, res, gpu) if with_res else None
            y = V.conv3d(xc, wc, bc, Cout, k, stride, pad, residual=r, **kw)
            assert lib.drn_conv_last_tile() == tile
            outs[tile] = y
            if with_res and tile == 1:
                y2 = V.conv3d(xc, wc, bc, Cout, k, stride, pad, residual=r, out=r, **kw)      # in place
                assert y2 is r and torch.equal(y2.t, y.t)
        finally:
            lib.drn_conv_force_tile(-1)
    assert (outs[0].T, outs[0].H, outs[0].W) == (To, Ho, Wo)
    assert torch.equal(outs[0].t, outs[1].t)                                                   # incl. the zero halo
    xf = x.float()[None]
    if stride[1] == 2:
        xf = F.pad(xf, (0, 1, 0, 1, 0, 0))[..., : 2 * (Ho - 1) + k[1], : 2 * (Wo - 1) + k[2]]
    if t_off is not None:
        xf = torch.cat([xf[:, :, :1]] * t_off + [xf], 2)
        ref = F.conv3d(xf, w.float(), b.float(), stride=stride)[0]
    else:
        ref = VO.causal_conv3d(xf, w.float(), b.float(), stride, pad)[0]
    ref = ref.to(BF)
    if with_res:
        ref = ref + res
    got = from_cl(outs[1])
    assert got.shape == ref.shape
    assert rel_l2(got, ref.float()) < 3e-3


def test_conv256_kernel_is_the_automatic_choice_for_big_convolutions(pkg, gpu):
    """>= 192 tiles of 256 positions x 256 channels -> the streamed kernel without forcing; small launches stay on 128 x 128."""
    V = pkg.native_vae
    lib = pkg.native.load_library()
    x = rnd((64, 4, 96, 128), seed=31)                  # 49 152 positions = 192 tiles at N = 256
    w, b = rnd((256, 64, 1, 3, 3), 0.04, seed=32), rnd((256,), 0.1, seed=33)
    xc, wc, bc = to_cl(pkg, x, gpu), repack(w, gpu), b.to(gpu)
    y = V.conv3d(xc, wc, bc, 256, (1, 3, 3), (1, 1, 1), 1)
    assert lib.drn_conv_last_tile() == 1
    lib.drn_conv_force_tile(0)
    try:
        y0 = V.conv3d(xc, wc, bc, 256, (1, 3, 3), (1, 1, 1), 1)
    finally:
        lib.drn_conv_force_tile(-1)
    assert torch.equal(y.t, y0.t)
    small = to_cl(pkg, rnd((64, 1, 16, 16), seed=34), gpu)
    V.conv3d(small, wc, bc, 256, (1, 3, 3), (1, 1, 1), 1)
    assert lib.drn_conv_last_tile() == 0


def test_dense_gemm_softmax_transpose(pkg, gpu):
    V = pkg.native_vae
    q, k, v = rnd((100, 128), seed=10).to(gpu), rnd((100, 128), seed=11).to(gpu), rnd((100, 128), seed=12).to(gpu)
    s = V.dense_gemm(q, k, out_f32=True, alpha=0.25)
    assert torch.allclose(s, (q.float() @ k.float().t()) * 0.25, atol=2e-3, rtol=1e-3)
    p = V.softmax_rows(s, 100, 128)
    ref_p = torch.softmax(s, -1)
    assert torch.allclose(p[:, :100].float(), ref_p, atol=4e-3) and (p[:, 100:] == 0).all()
    vt = V.transpose(v, 128)
    assert torch.equal(vt[:, :100], v.t()) and (vt[:, 100:] == 0).all()
    o = V.dense_gemm(p, vt)
    assert rel_l2(o, p.float()[:, :100] @ v.float()) < 3e-3


@pytest.mark.parametrize("P,C", [(2304, 512), (512, 128), (9216, 512)])
def test_scores_on_the_tile_gemm_equal_the_implicit_gemm_path(pkg, gpu, P, C):
    """Spatial attention of the mid block: raw fp32 scores from the 256 x 256 tile kernel (drn_gemm_bf16_f32out) + a softmax that
    applies the scale == scores scaled in the implicit-GEMM epilogue + plain softmax, bit for bit (same K order per element, the
    same fp32 product scale * score); ragged shapes keep the old kernel."""
    V = pkg.native_vae
    q, k = rnd((P, C), seed=40).to(gpu), rnd((P, C), seed=41).to(gpu)
    scale = C ** -0.5
    s_old = V.dense_gemm(q, k, out_f32=True, alpha=scale)
    s_new = V.scores_f32(q, k)
    assert torch.equal(s_new * scale, s_old)
    assert torch.equal(V.softmax_rows(s_new, P, P, scale), V.softmax_rows(s_old, P, P))
    qr, kr = q[:200].contiguous(), k[:300].contiguous()              # not multiples of 256: the implicit-GEMM kernel
    assert torch.equal(V.scores_f32(qr, kr), V.dense_gemm(qr, kr, out_f32=True))


@pytest.mark.parametrize("n,ldp", [(100, 128), (2304, 2304), (4100, 4160), (9216, 9216), (16384, 16384), (16388, 16448), (4098, 4160)])
def test_softmax_rows_register_and_three_pass_kernels(pkg, gpu, n, ldp):
    """drn_softmax_rows: rows of up to 4096 / 9216 / 16384 scores stay in registers (one read of the fp32 scores), longer or
    unaligned rows take the three-pass kernel; both against torch.softmax, padding columns zero."""
    V = pkg.native_vae
    g = torch.Generator().manual_seed(n)
    s = (torch.randn(37, n, generator=g) * 3.0).to(gpu)
    p = V.softmax_rows(s, n, ldp)
    ref = torch.softmax(s, -1)
    assert p.shape == (37, ldp) and (p[:, n:] == 0).all()
    assert torch.allclose(p[:, :n].float(), ref, atol=4e-3, rtol=8e-3)
    assert (p[:, :n].float().sum(-1) - 1).abs().max().item() < 2e-2


def test_pooled_activation_buffers_keep_a_zero_halo(pkg, gpu):
    """Inside encode / decode activations recycle buffers of their shape instead of a fresh torch.zeros each; two encodes of
    different clips must not leak into each other and must equal the un-pooled result bit for bit."""
    V = pkg.native_vae
    sw = pkg.synthetic_weights
    vae = pkg.CleanVAE.CleanVAE(state_dict=sw.synth_vae_state_dict(device=gpu), device=gpu)
    a = sw.synth_tensor("pool.a", (1, 3, 9, 32, 48), torch.float32, device=gpu).to(torch.bfloat16)
    b = sw.synth_tensor("pool.b", (1, 3, 9, 32, 48), torch.float32, device=gpu).to(torch.bfloat16)
    V.pool_clear()
    za, zb = vae.encode(a), vae.encode(b)
    ya = vae.decode(za)
    assert len(V._POOL) > 0 and all(t.dtype == torch.bfloat16 for free in V._POOL.values() for t in free)
    for free in V._POOL.values():               # every pooled buffer still has its zero halo
        for t in free:
            assert (t[:, 0] == 0).all() and (t[:, -1] == 0).all() and (t[:, :, 0] == 0).all() and (t[:, :, -1] == 0).all()
    za2 = vae.encode(a)                         # recycled buffers, same bits
    assert torch.equal(za, za2) and not torch.equal(za, zb)
    V.pool_clear()
    enc, dec = vae.model._encode, vae.model._decode       # the same work without the pool
    assert not V._POOL_ON[0]
    assert torch.equal(enc(a, None), za) and torch.equal(dec(za, None), ya) and len(V._POOL) == 0


@pytest.mark.parametrize("silu", [True, False])
def test_groupnorm_silu(pkg, gpu, silu):
    V = pkg.native_vae
    C, T, H, W = 128, 3, 9, 11
    x = rnd((C, T, H, W), 2.0, seed=13)
    g, b = 1 + 0.1 * rnd((C,), seed=14), rnd((C,), 0.2, seed=15)
    y = V.groupnorm_silu(to_cl(pkg, x, gpu), g.to(gpu), b.to(gpu), silu)
    ref = VO.group_norm_causal(x[None], g, b)
    if silu:
        ref = F.silu(ref)
    got = from_cl(y)
    assert rel_l2(got, ref[0].float()) < 4e-3
    assert ((got.float() - ref[0].float()).abs() <= 2.0 ** -6 * ref[0].float().abs() + 2e-2).all()
    assert (y.t[:, 0] == 0).all() and (y.t[:, :, -1] == 0).all()


def test_haar_patch_unpatch_match_oracle(pkg, gpu):
    V = pkg.native_vae
    sw = pkg.synthetic_weights
    orc = VO.VaeOracle({}, sw.COSMOS_CV8x8x8, BF)
    x = rnd((3, 9, 16, 24), seed=16)
    ref = orc.patch(x[None])[0]                                   # [192, 3, 4, 6]
    got = from_cl(V.haar_patch(x.to(gpu)))
    assert got.shape == ref.shape
    assert torch.equal(got, ref), f"{(got != ref).sum().item()} of {ref.numel()} differ"
    p = rnd((192, 3, 4, 6), seed=17)
    back = V.haar_unpatch(to_cl(pkg, p, gpu)).cpu()
    ref_b = orc.unpatch(p[None])[0]
    assert back.shape == ref_b.shape == (3, 9, 16, 24)
    assert torch.equal(back, ref_b), f"{(back != ref_b).sum().item()} of {ref_b.numel()} differ"


def test_resample_modes(pkg, gpu):
    V = pkg.native_vae
    x = rnd((64, 5, 6, 8), seed=18)
    xc = to_cl(pkg, x, gpu)
    x5 = x[None]
    assert torch.equal(from_cl(V.resample(xc, 0)), VO.avg_pool(F.pad(x5, (0, 1, 0, 1, 0, 0)), (1, 2, 2))[0])
    assert torch.equal(from_cl(V.resample(xc, 1)), VO.avg_pool(torch.cat([x5[:, :, :1], x5], 2), (2, 1, 1))[0])
    assert torch.equal(from_cl(V.resample(xc, 2)), x5.repeat_interleave(2, dim=2)[:, :, 1:][0])
    assert torch.equal(from_cl(V.resample(xc, 3)), x5.repeat_interleave(2, dim=3).repeat_interleave(2, dim=4)[0])
    one = to_cl(pkg, x[:, :1], gpu)
    assert torch.equal(from_cl(V.resample(one, 2)), x[:, :1])


def test_temporal_attention(pkg, gpu):
    V = pkg.native_vae
    T, P, C = 5, 37, 128
    q, k, v = rnd((T, P, C), seed=19), rnd((T, P, C), seed=20), rnd((T, P, C), seed=21)
    o = V.temporal_attention(q.to(gpu), k.to(gpu), v.to(gpu), C ** -0.5).cpu()
    qq, kk, vv = (t.float().permute(1, 0, 2)[:, None] for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qq, kk, vv, attn_mask=torch.tril(torch.ones(T, T, dtype=torch.bool)))
    assert rel_l2(o, ref[:, 0].permute(1, 0, 2)) < 5e-3


def _tok(pkg, gpu):
    sw = pkg.synthetic_weights
    sd = sw.synth_vae_state_dict(sw.COSMOS_CV8x8x8, BF)
    return sd, pkg.CleanVAE.CleanVAE(state_dict={k: v.to(gpu) for k, v in sd.items()}, device=gpu)


@pytest.mark.parametrize("T,H,W", [(9, 32, 32), (1, 32, 48), (17, 288, 512)])
def test_tokenizer_encode_decode_match_oracle(pkg, gpu, T, H, W):
    """(17, 288, 512) is the mid-size case: the real channel counts (128-512) at spatial sizes where every conv runs many
    128 x 128 tiles per level, temporal down / up-sampling over 3 latent frames, a 36 x 64 mid-block attention (2304 keys)."""
    torch.set_num_threads(max(1, min(32, len(__import__("os").sched_getaffinity(0)))))
    sw = pkg.synthetic_weights
    sd, vae = _tok(pkg, gpu)
    x = sw.synth_tensor("vae.x", (1, 3, T, H, W), torch.float32).to(BF)
    o16 = VO.VaeOracle(sd, sw.COSMOS_CV8x8x8, BF)
    o32 = VO.VaeOracle(sd, sw.COSMOS_CV8x8x8, torch.float32)
    o16.trace = {}
    vae.model.trace = {}
    with torch.no_grad():
        z16, z32 = o16.encode(x), o32.encode(x.float())
    z = vae.encode(x.to(gpu)).cpu()
    assert z.shape == z16.shape == (1, 16, vae.get_latent_num_frames(T), H // 8, W // 8)
    for key in ("patch", "conv_in", "down0.res0", "down0.res1", "down0.ds", "down1.res1", "down2.res1", "mid"):
        got = from_cl(vae.model.trace[key])
        print(key, rel_l2(got, o16.trace[key][0].float()))
    e_ref, e_hip = rel_l2(z16, z32), rel_l2(z, z32)
    print(f"encode: e_ref={e_ref:.3e} e_hip={e_hip:.3e} hip-vs-bf16-oracle={rel_l2(z, z16.float()):.3e}")
    assert e_hip <= 1.5 * e_ref + 1e-3
    with torch.no_grad():
        y16, y32 = o16.decode(z16), o32.decode(z16.float())
    y = vae.decode(z16.to(gpu)).cpu()
    assert y.shape == y16.shape == (1, 3, T, H, W)
    e_ref, e_hip = rel_l2(y16, y32), rel_l2(y, y32)
    print(f"decode: e_ref={e_ref:.3e} e_hip={e_hip:.3e} hip-vs-bf16-oracle={rel_l2(y, y16.float()):.3e}")
    assert e_hip <= 1.5 * e_ref + 1e-3


def test_tokenizer_full_size_clip_causality_and_determinism(pkg, gpu):
    """BASELINE configs 3-5 size (57 f x 576 x 1024; the CPU oracle would take hours here): size-independent properties of the
    causal tokenizer.  Latent frame t depends on pixel frames <= 8t only and pixel frame 8t-7..8t on latent frames <= t, per-frame
    GroupNorm / spatial attention, causal temporal attention - so a prefix of the clip must encode / decode to the SAME BITS as
    the prefix of the full result (the big launches run conv256s.hip, the prefix ones partly the 128 x 128 kernel: bit-identical
    kernels), T = 1 is the image path, and two runs agree bit for bit."""
    sw = pkg.synthetic_weights
    _, vae = _tok(pkg, gpu)
    T, H, W = 57, 576, 1024
    x = sw.synth_tensor("vae.full", (1, 3, T, H, W), torch.float32, device=gpu).to(BF)
    z = vae.encode(x)
    assert z.shape == (1, 16, 8, H // 8, W // 8) and torch.isfinite(z.float()).all()
    assert torch.equal(vae.encode(x), z)
    for frames in (1, 17):
        zp = vae.encode(x[:, :, :frames].contiguous())
        lf = vae.get_latent_num_frames(frames)
        same = (zp == z[:, :, :lf]).float().mean().item()
        print(f"encode prefix {frames} f: {same:.6f} of the latent identical to the full clip's first {lf} frame(s)")
        assert torch.equal(zp, z[:, :, :lf])
    y = vae.decode(z)
    assert y.shape == (1, 3, T, H, W) and torch.isfinite(y.float()).all()
    for lf in (1, 3):
        yp = vae.decode(z[:, :, :lf].contiguous())
        pf = vae.get_pixel_num_frames(lf)
        same = (yp == y[:, :, :pf]).float().mean().item()
        print(f"decode prefix {lf} latent frame(s): {same:.6f} of the pixels identical to the full clip's first {pf} frame(s)")
        assert torch.equal(yp, y[:, :, :pf])
    # the round trip of a random-init tokenizer is not an identity, but it is a bounded map of a bounded clip
    assert y.float().abs().max() < 1e4


def test_tokenizer_rejects_bad_inputs(pkg, gpu):
    _, vae = _tok(pkg, gpu)
    with pytest.raises(ValueError):
        vae.encode(torch.zeros(3, 9, 32, 32, dtype=BF, device=gpu))
    with pytest.raises(ValueError):
        vae.encode(torch.zeros(1, 3, 8, 32, 32, dtype=BF, device=gpu))
    assert vae.get_latent_num_frames(57) == 8 and vae.get_pixel_num_frames(8) == 57 and vae.get_latent_num_frames(1) == 1
    assert vae.latent_ch == 16 and vae.spatial_compression_factor == 8
