#!/usr/bin/env python3
"""Generate tests/golden/*.safetensors by RUNNING THE REFERENCE on CPU in this container.

    python tools/make_goldens.py [--full]        (--full adds the 7.2 B-parameter cfg-1 case)

The reference (/root/reference) never travels to the GPU box, so its outputs are
committed as small fixtures.  Weights and inputs are NOT stored: they come from
the build-owned hash generator (synthetic_weights.py), which produces identical
bits on any machine.  Each file's metadata records how it was produced,
including the F1 head-flatten patch applied in memory.

While generating, the script asserts that oracle/dit_oracle.py reproduces the
reference bit-for-bit (that is what pins the oracle).
"""
import argparse
import gc
import json
import os
import sys
import time

import torch
from safetensors.torch import save_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
from oracle import dit_oracle as O  # noqa: E402
from oracle.ref_import import load_reference  # noqa: E402

sw = pkg.synthetic_weights
GOLD = os.path.join(ROOT, "tests", "golden")
META_COMMON = {
    "generator": "tools/make_goldens.py",
    "reference": "eggsbenedicto/DiffusionRenderer-ComfyUI @ 2025-08-15 (/root/reference), imported on CPU",
    "f1_patch": "PytorchDotProductAttention.forward output reshaped (S,B,H,dh)->(S,B,H*dh) in memory",
    "weights": "synthetic_weights.synth_state_dict (hash generator; not stored)",
}


def net_config(ref, D, L, Hn, forward=False):
    cfg = ref.config.get_forward_renderer_config() if forward else ref.config.get_inverse_renderer_config()
    net = dict(cfg["net"])
    net.update(model_channels=D, num_blocks=L, num_heads=Hn)
    return cfg, net


def build_ref_model(ref, cfg, net, sd, dtype):
    cfg = dict(cfg)
    cfg["net"] = dict(net)
    cfg["scheduler"] = dict(cfg["scheduler"])
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)      # build directly in the target dtype (memory)
    try:
        model = ref.model.CleanDiffusionRendererModel(cfg)
    finally:
        torch.set_default_dtype(prev)
    model = model.to(dtype)
    missing, unexpected = model.load_state_dict({k: v.to(dtype) for k, v in sd.items()}, strict=False)
    # non-persistent RoPE range buffers are legitimately absent from the state dict
    assert not unexpected, unexpected
    assert all("dim_spatial_range" in m or "dim_temporal_range" in m for m in missing), missing
    return model.eval()


def inputs_for(tag, latent, cond_ch, dtype):
    F_, h, w = latent
    x = sw.synth_tensor(tag + ".x", (1, 16, F_, h, w), torch.float32, scale=2.0).to(dtype)
    cond = sw.synth_tensor(tag + ".cond", (1, cond_ch, F_, h, w), torch.float32, scale=1.0).to(dtype)
    return x, cond


def hook_blocks(model, store):
    handles = []
    for name, mod in model.net.named_modules():
        if mod.__class__.__name__ == "OfficialDITBuildingBlock":
            parts = name.split(".")          # blocks.block{i}.blocks.{j}
            key = f"{parts[1]}.{parts[3]}"
            handles.append(mod.register_forward_hook(lambda m, i, o, key=key: store.__setitem__(key, o.detach().clone())))
    return handles


def dit_case(ref, tag, D, L, Hn, latent, sigma, ctx_idx, dtypes=(torch.bfloat16, torch.float32), forward=False,
             keep_blocks=True):
    """One DiT forward through the reference; returns dict of tensors for the fixture."""
    cfg, net = net_config(ref, D, L, Hn, forward)
    out = {}
    cond_ch = net["additional_concat_ch"]
    for dtype in dtypes:
        dn = "bf16" if dtype == torch.bfloat16 else "fp32"
        sd = sw.synth_state_dict(net, torch.bfloat16)            # weights are bf16-representable in both modes
        model = build_ref_model(ref, cfg, net, sd, dtype)
        x, cond = inputs_for(tag, latent, cond_ch, torch.bfloat16)
        x, cond = x.to(dtype), cond.to(dtype)
        t = torch.tensor(sigma, dtype=torch.float32)
        ci = torch.full((1, 1), ctx_idx, dtype=torch.long)
        blocks = {}
        hs = hook_blocks(model, blocks)
        with torch.no_grad():
            kw = dict(x=x, timesteps=t, latent_condition=cond, context_index=ci)
            y = model.net(**kw)
        for h in hs:
            h.remove()
        # the oracle must reproduce the reference bit-for-bit in the same dtype
        orc = O.DitOracle(sd, net, dtype=dtype)
        orc.trace = {}
        with torch.no_grad():
            yo = orc.forward(x, t, cond, ci)
        assert torch.equal(y, yo), f"{tag}/{dn}: oracle != reference (max diff {(y.float()-yo.float()).abs().max()})"
        for k, v in blocks.items():
            assert torch.equal(v, orc.trace[k]), f"{tag}/{dn}: block {k} differs"
        out[f"out.{dn}"] = y.contiguous()
        if keep_blocks:
            for k, v in blocks.items():
                out[f"{k}.{dn}"] = v.squeeze(1).contiguous()
        del model, orc, blocks
        gc.collect()
        if dtype == torch.bfloat16:
            # fp32 evaluation with bf16-quantised host tables: the "exact" answer of the parity bound
            orc32 = O.DitOracle(sd, net, dtype=torch.float32, tables_dtype=torch.bfloat16)
            with torch.no_grad():
                out["out.fp32_tables_bf16"] = orc32.forward(x, t, cond, ci).contiguous()
            del orc32
            gc.collect()
    meta = dict(META_COMMON, case=tag, D=str(D), L=str(L), heads=str(Hn), latent=json.dumps(latent),
                sigma=repr(sigma), context_index=str(ctx_idx), forward=str(forward),
                inputs="synth_tensor(tag+'.x',(1,16,F,h,w),fp32,scale=2).to(bf16); synth_tensor(tag+'.cond',...,scale=1).to(bf16)")
    return out, meta


class StubVAE:
    """Duck-typed tokenizer (model_diffusion_renderer.py:146,156,168-171): 8x8x8 average pool / nearest upsample."""
    latent_ch = 16
    spatial_compression_factor = 8

    def get_latent_num_frames(self, n):
        return 1 if n == 1 else (n - 1) // 8 + 1

    def encode(self, x):
        B, C, T, H, W = x.shape
        F_ = self.get_latent_num_frames(T)
        xs = x.float()
        first = xs[:, :, :1]
        rest = xs[:, :, 1:]
        frames = [first.mean(2, keepdim=True)]
        if F_ > 1:
            frames.append(rest.reshape(B, C, F_ - 1, 8, H, W).mean(3))
        z = torch.cat(frames, 2).reshape(B, C, F_, H // 8, 8, W // 8, 8).mean((4, 6))
        reps = (16 + C - 1) // C
        z = z.repeat(1, reps, 1, 1, 1)[:, :16]
        scale = torch.linspace(0.5, 2.0, 16).view(1, 16, 1, 1, 1)
        return (z * scale).to(x.dtype)

    def decode(self, z):
        B, C, F_, h, w = z.shape
        T = 1 if F_ == 1 else (F_ - 1) * 8 + 1
        v = z[:, :3].float()
        v = v.repeat_interleave(8, 3).repeat_interleave(8, 4)
        if F_ > 1:
            v = torch.cat([v[:, :, :1], v[:, :, 1:].repeat_interleave(8, 2)], 2)
        assert v.shape[2] == T
        return v.to(z.dtype)


def sampler_case(ref, tag, D, L, Hn, T, HW, steps, guidance, seed, ctx_idx, normalize_normal):
    """Whole generate_video through the reference pipeline on CPU with the stub VAE."""
    cfg, net = net_config(ref, D, L, Hn)
    sd = sw.synth_state_dict(net, torch.bfloat16)
    model = build_ref_model(ref, cfg, net, sd, torch.bfloat16)
    p = ref.pipeline.CleanDiffusionRendererPipeline("/nonexistent", "x.pt", model_type=None, vae_instance=StubVAE(),
                                                    model_instance=model, guidance=guidance, num_steps=steps, seed=seed)
    p.device = torch.device("cpu")
    p.set_model_type("inverse")
    rgb = sw.synth_tensor(tag + ".rgb", (1, 3, T, HW[0], HW[1]), torch.float32, scale=1.0)
    ci = torch.full((1, 1), ctx_idx, dtype=torch.long)
    video = p.generate_video({"rgb": rgb, "video": rgb, "context_index": ci}, normalize_normal=normalize_normal, seed=seed)

    # reproduce the trajectory with the oracle (injecting the same x_T the reference drew)
    vae = StubVAE()
    F_ = vae.get_latent_num_frames(T)
    torch.manual_seed(seed)
    rgb_b = rgb.to(torch.bfloat16)
    cond = (vae.encode(rgb_b) * 0.5).contiguous()
    sig = O.edm_sigmas(steps)
    xT = torch.randn(size=(1, 16, F_, HW[0] // 8, HW[1] // 8), dtype=torch.bfloat16) * sig[0]
    orc = O.DitOracle(sd, net, dtype=torch.bfloat16)
    traj = []
    with torch.no_grad():
        x0 = O.sample_loop(orc.forward, xT, cond, ci.to(torch.bfloat16), steps, guidance, record=traj)
        vid = vae.decode(x0 / 0.5)
        u8 = O.postprocess(vid, normalize_normal).numpy()
    assert (u8 == video).all(), f"{tag}: oracle pipeline != reference pipeline"
    out = {"xT": xT.contiguous(), "latent_condition": cond, "x0": x0.contiguous(),
           "video_u8": torch.from_numpy(video).contiguous()}
    for i, s in enumerate(traj):
        out[f"traj.{i}"] = s.contiguous()
    meta = dict(META_COMMON, case=tag, D=str(D), L=str(L), heads=str(Hn), T=str(T), HW=json.dumps(HW), steps=str(steps),
                guidance=repr(guidance), seed=str(seed), context_index=str(ctx_idx), normalize_normal=str(normalize_normal),
                vae="StubVAE (8x8x8 mean-pool / nearest upsample; tools/make_goldens.py)",
                inputs="rgb = synth_tensor(tag+'.rgb',(1,3,T,H,W),fp32,scale=1)")
    return out, meta


def scheduler_and_post_case(ref):
    out = {}
    for n in (4, 15, 35):
        sch = ref.model.CleanEDMEulerScheduler()
        sch.set_timesteps(n)
        assert torch.equal(sch.sigmas, O.edm_sigmas(n))
        out[f"sigmas.{n}"] = sch.sigmas.clone()
    sch = ref.model.CleanEDMEulerScheduler()
    sch.set_timesteps(4)
    x = sw.synth_tensor("sched.x", (1, 16, 2, 8, 8), torch.float32, scale=40.0).to(torch.bfloat16)
    mo = sw.synth_tensor("sched.mo", (1, 16, 2, 8, 8), torch.float32, scale=1.0).to(torch.bfloat16)
    sch.current_step = 1
    t = sch.timesteps[1]
    out["scaled"] = sch.scale_model_input(x, t)
    out["stepped"] = sch.step(mo, t, x).prev_sample
    assert torch.equal(out["scaled"], O.edm_scale_input(x, t))
    assert torch.equal(out["stepped"], O.edm_step(mo, t, O.edm_sigmas(4)[2], x))
    # last step: sigma_next = 0 -> x = denoised
    sch.current_step = 3
    t = sch.timesteps[3]
    out["stepped_last"] = sch.step(mo, t, x).prev_sample
    assert torch.equal(out["stepped_last"], O.edm_step(mo, t, O.edm_sigmas(4)[4], x))
    meta = dict(META_COMMON, case="scheduler", inputs="x=synth_tensor('sched.x',(1,16,2,8,8),fp32,40).bf16; mo=synth_tensor('sched.mo',...,1).bf16; N=4, step index 1 and 3")
    return out, meta


def pretrained_vae_case(ref):
    """Chunk / normalise plumbing of pretrained_vae.py with a fake linear encoder/decoder (bit-exact)."""
    pv = ref.pretrained_vae

    class FakeEnc(torch.nn.Module):
        def forward(self, x):               # (b,3,t,h,w) -> (b,16,(t-1)//8+1,h/8,w/8)
            b, c, t, h, w = x.shape
            f = (t - 1) // 8 + 1
            idx = torch.arange(f) * 8
            z = x[:, :, idx].float().reshape(b, c, f, h // 8, 8, w // 8, 8).mean((4, 6))
            return z.repeat(1, 6, 1, 1, 1)[:, :16].to(x.dtype)

    class FakeDec(torch.nn.Module):
        def forward(self, z):
            b, c, f, h, w = z.shape
            t = (f - 1) * 8 + 1
            v = z[:, :3].repeat_interleave(8, 3).repeat_interleave(8, 4)
            idx = (torch.arange(t) + 7) // 8
            return v[:, :, idx]

    tok = pv.VideoJITTokenizer(name="fake", latent_ch=16, is_bf16=True, spatial_compression_factor=8,
                               temporal_compression_factor=8, pixel_chunk_duration=17, max_enc_batch_size=2,
                               max_dec_batch_size=1)
    tok.encoder, tok.decoder = FakeEnc(), FakeDec()
    mean = sw.synth_tensor("pv.mean", (16, 16), torch.float32, scale=0.1)
    std = 1.0 + sw.synth_tensor("pv.std", (16, 16), torch.float32, scale=0.2)
    lc = tok.latent_chunk_duration
    tok.register_buffer("latent_mean", mean[:, :lc].to(tok.dtype).reshape(1, 16, lc, 1, 1), persistent=False)
    tok.register_buffer("latent_std", std[:, :lc].to(tok.dtype).reshape(1, 16, lc, 1, 1), persistent=False)
    x = sw.synth_tensor("pv.x", (1, 3, 51, 16, 24), torch.float32, scale=1.0).to(torch.bfloat16)   # 3 chunks of 17
    z = tok.encode(x)
    y = tok.decode(z)
    out = {"x": x, "z": z.contiguous(), "y": y.contiguous()}
    meta = dict(META_COMMON, case="pretrained_vae chunking", chunk="17", latent_chunk=str(lc), enc_mb="2", dec_mb="1",
                frames=json.dumps({"latent": tok.get_latent_num_frames(51), "pixel": tok.get_pixel_num_frames(9)}),
                fake="FakeEnc: frame-subsample + 8x8 mean-pool, channels tiled to 16; FakeDec: nearest upsample")
    return out, meta


def manifest_case(ref):
    """Names and shapes of the reference model's full-size state dict (28 blocks, D = 4096), built on the meta device: what
    a real checkpoint holds, i.e. what CleanDiffusionRendererModel.load_state_dict(strict=True) must accept."""
    import json
    out = {}
    for kind in ("inverse", "forward"):
        cfg = ref.config.get_forward_renderer_config() if kind == "forward" else ref.config.get_inverse_renderer_config()
        with torch.device("meta"):
            m = ref.model.CleanDiffusionRendererModel(cfg)
        out[kind] = {k: list(v.shape) for k, v in m.state_dict().items()}
    path = os.path.join(GOLD, "state_dict_manifest.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote", path, {k: len(v) for k, v in out.items()})


COND_KEYS8 = ["basecolor", "normal", "metallic", "roughness", "depth", "env_ldr", "env_log", "env_nrm"]


def conditions_case(ref):
    """Condition assembly of both renderers through the REFERENCE's own `_get_conditions` /
    `prepare_diffusion_renderer_latent_conditions` (model_diffusion_renderer.py:158-209) with the stub tokenizer:
    forward 8 keys x (16 latent + mask), one key missing (zeros + zero mask), without the mask, and the inverse
    renderer's `image` -> `rgb` alias.  Inputs come from synth_tensor('cond8.<key>', ...); only outputs are stored."""
    out = {}
    T, H, W = 9, 32, 48

    def maps(keys):
        return {k: sw.synth_tensor("cond8." + k, (1, 3, T, H, W), torch.float32, scale=1.0).to(torch.bfloat16) for k in keys}

    def run(cfg, batch, mask):
        cfg = dict(cfg, net=dict(cfg["net"], model_channels=128, num_blocks=1, num_heads=1), scheduler=dict(cfg["scheduler"]),
                   append_condition_mask=mask)
        m = ref.model.CleanDiffusionRendererModel(cfg)
        m.vae = StubVAE()
        batch = dict(batch)
        cond, uncond = m._get_conditions(batch)
        lc = cond.to_dict()["latent_condition"]
        assert torch.equal(batch["latent_condition"], lc)
        assert not uncond.to_dict()["latent_condition"].any()
        return lc.contiguous(), m.input_data_key

    fwd = ref.config.get_forward_renderer_config(H, W, T)
    inv = ref.config.get_inverse_renderer_config(H, W, T)
    assert fwd["condition_keys"] == COND_KEYS8
    b8 = maps(COND_KEYS8)
    out["forward.all8.mask"], k1 = run(fwd, b8, True)
    out["forward.all8.nomask"], _ = run(fwd, b8, False)
    b7 = {k: v for k, v in b8.items() if k != "roughness"}
    out["forward.no_roughness.mask"], _ = run(fwd, b7, True)
    b5 = {k: v for k, v in b8.items() if not k.startswith("env_")}
    out["forward.no_env.mask"], k2 = run(fwd, b5, True)
    rgb = maps(["rgb"])
    inv_keys = inv["condition_keys"]
    out["inverse.rgb"], k3 = run(inv, {"rgb": rgb["rgb"], "video": rgb["rgb"]}, inv.get("append_condition_mask", True))
    # the model's own default key list (no `condition_keys` in the config): ["image", "rgb"] -> alias + the rgb key itself
    inv_default = {k: v for k, v in inv.items() if k != "condition_keys"}
    out["inverse.default_keys.mask"], _ = run(inv_default, {"rgb": rgb["rgb"], "video": rgb["rgb"]}, True)
    meta = dict(META_COMMON, case="condition assembly", T=str(T), HW=json.dumps([H, W]), forward_keys=json.dumps(COND_KEYS8),
                inverse_keys=json.dumps(inv_keys), inverse_mask=str(inv.get("append_condition_mask", True)),
                input_data_key=json.dumps({"forward.all8": k1, "forward.no_env": k2, "inverse.rgb": k3}),
                vae="StubVAE (tools/make_goldens.py)",
                inputs="synth_tensor('cond8.'+key,(1,3,T,H,W),fp32,scale=1).to(bf16) per key")
    return out, meta


def full_width_cases(ref):
    """The 7.2 B-parameter model (28 blocks, D = 4096, 32 heads) through the reference at BASELINE config 2's two token
    counts (latent (1,64,64) -> S = 1024 and (2,64,64) -> S = 2048) and SURVEY 8c G4: the 4-step config-1 pipeline run
    (1 frame 256 x 256, guidance 0) whose trajectory is re-traced by the oracle from the x_T the reference drew.
    The reference model is built once; its parameters ARE the synthetic state dict (load_state_dict(assign=True))."""
    cfg, net = net_config(ref, 4096, 28, 32)
    sd = sw.synth_state_dict(net, torch.bfloat16)
    cfgm = dict(cfg, net=dict(net), scheduler=dict(cfg["scheduler"]))
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        model = ref.model.CleanDiffusionRendererModel(cfgm)
    finally:
        torch.set_default_dtype(prev)
    model = model.to(torch.bfloat16)
    missing, unexpected = model.load_state_dict(sd, strict=False, assign=True)
    assert not unexpected and all("dim_spatial_range" in m or "dim_temporal_range" in m for m in missing), (missing, unexpected)
    model.eval()
    gc.collect()
    orc = O.DitOracle(sd, net, dtype=torch.bfloat16)
    files = {}

    fwd_cases = [("full28_s1024", (1, 64, 64), 1.5, 0), ("full28_s2048", (2, 64, 64), 12.0, 4)]
    fwd_inputs = {}
    for tag, latent, sigma, ci_ in fwd_cases:
        t0 = time.time()
        x, cond = inputs_for(tag, latent, 16, torch.bfloat16)
        t, ci = torch.tensor(sigma, dtype=torch.float32), torch.full((1, 1), ci_, dtype=torch.long)
        with torch.no_grad():
            y = model.net(x=x, timesteps=t, latent_condition=cond, context_index=ci)
            yo = orc.forward(x, t, cond, ci)
        assert torch.equal(y, yo), f"{tag}: oracle != reference"
        files[tag] = ({"out.bf16": y.contiguous()},
                      dict(META_COMMON, case=tag, D="4096", L="28", heads="32", latent=json.dumps(latent), sigma=repr(sigma),
                           context_index=str(ci_), forward="False",
                           inputs="synth_tensor(tag+'.x',(1,16,F,h,w),fp32,scale=2).to(bf16); synth_tensor(tag+'.cond',...,scale=1).to(bf16)"))
        fwd_inputs[tag] = (x, t, cond, ci)
        print(f"{tag}: reference + oracle bf16 forwards {time.time()-t0:.0f}s", flush=True)

    # G4: whole pipeline, 4 steps, guidance 0, normal pass (context 3, re-normalisation on), 1 frame 256 x 256
    t0 = time.time()
    tag, steps, seed = "g4", 4, 42
    p = ref.pipeline.CleanDiffusionRendererPipeline("/nonexistent", "x.pt", model_type=None, vae_instance=StubVAE(),
                                                    model_instance=model, guidance=0.0, num_steps=steps, seed=seed)
    p.device = torch.device("cpu")
    p.set_model_type("inverse")
    rgb = sw.synth_tensor(tag + ".rgb", (1, 3, 1, 256, 256), torch.float32, scale=1.0)
    ci = torch.full((1, 1), 3, dtype=torch.long)
    video = p.generate_video({"rgb": rgb, "video": rgb, "context_index": ci}, normalize_normal=True, seed=seed)
    vae = StubVAE()
    torch.manual_seed(seed)
    condl = (vae.encode(rgb.to(torch.bfloat16)) * 0.5).contiguous()
    sig = O.edm_sigmas(steps)
    xT = torch.randn(size=(1, 16, 1, 32, 32), dtype=torch.bfloat16) * sig[0]
    traj = []
    with torch.no_grad():
        x0 = O.sample_loop(orc.forward, xT, condl, ci.to(torch.bfloat16), steps, 0.0, record=traj)
        u8 = O.postprocess(vae.decode(x0 / 0.5), True).numpy()
    assert (u8 == video).all(), "g4: oracle pipeline != reference pipeline"
    g4 = {"xT": xT.contiguous(), "latent_condition": condl, "x0": x0.contiguous(), "video_u8": torch.from_numpy(video).contiguous()}
    for i, s in enumerate(traj):
        g4[f"traj.{i}"] = s.contiguous()
    print(f"g4: reference pipeline + oracle trajectory {time.time()-t0:.0f}s", flush=True)

    # exact answers: fp32 oracle with bf16-quantised host tables (parity bound)
    del model, p, orc
    gc.collect()
    orc32 = O.DitOracle(sd, net, dtype=torch.float32, tables_dtype=torch.bfloat16)
    for tag, _, _, _ in fwd_cases:
        t0 = time.time()
        x, t, cond, ci2 = fwd_inputs[tag]
        with torch.no_grad():
            files[tag][0]["out.fp32_tables_bf16"] = orc32.forward(x, t, cond, ci2).contiguous()
        print(f"{tag}: fp32 oracle {time.time()-t0:.0f}s", flush=True)
    t0 = time.time()
    traj32 = []
    with torch.no_grad():
        # the sampler's own roundings (bf16 latent between steps) kept; only the network evaluates in fp32
        x0_32 = O.sample_loop(lambda xs, t, c, i: orc32.forward(xs, t, c, i).to(torch.bfloat16), xT, condl,
                              ci.to(torch.bfloat16), steps, 0.0, record=traj32)
    g4["x0.fp32net"] = x0_32.contiguous()
    for i, s in enumerate(traj32):
        g4[f"traj32.{i}"] = s.contiguous()
    print(f"g4: fp32-network trajectory {time.time()-t0:.0f}s", flush=True)
    files["g4"] = (g4, dict(META_COMMON, case="g4 full-width 4-step pipeline", D="4096", L="28", heads="32", T="1",
                            HW=json.dumps([256, 256]), steps=str(steps), guidance="0.0", seed=str(seed), context_index="3",
                            normalize_normal="True", vae="StubVAE (tools/make_goldens.py)",
                            inputs="rgb = synth_tensor('g4.rgb',(1,3,1,256,256),fp32,scale=1)"))
    return files


def vae_config_case():
    """Hyper-parameters of the reference's VAE_config.json (the only in-repo specification of the tokenizer; read by no
    reference code) as a small JSON fixture; the 2 x 256 latent mean / std values as count + sums."""
    import json
    with open("/root/reference/VAE_config.json") as f:
        c = json.load(f)
    out = {k: v for k, v in c.items() if k not in ("latents_mean", "latents_std")}
    for k in ("latents_mean", "latents_std"):
        out[k + "_count"] = len(c[k])
        out[k + "_sum"] = float(sum(c[k]))
    path = os.path.join(GOLD, "vae_config_hparams.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path)


def save(name, tensors, meta):
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name)
    save_file({k: v.contiguous() for k, v in tensors.items()}, path, metadata={k: str(v) for k, v in meta.items()})
    print(f"wrote {path}: {os.path.getsize(path)/1e6:.2f} MB, {len(tensors)} tensors")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also run the 28-block 7.2B cfg-1 case (needs ~35 GB RAM)")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    ref = load_reference()
    todo = args.only.split(",") if args.only else None

    def want(n):
        return todo is None or n in todo

    if want("manifest"):
        manifest_case(ref)
    if want("sched"):
        save("scheduler.safetensors", *scheduler_and_post_case(ref))
    if want("pv"):
        save("pretrained_vae_chunking.safetensors", *pretrained_vae_case(ref))
    if want("tinyA"):
        save("dit_tinyA.safetensors", *dit_case(ref, "tinyA", 256, 1, 2, (2, 16, 16), 3.5, 3))
    if want("tinyB"):
        save("dit_tinyB.safetensors", *dit_case(ref, "tinyB", 512, 2, 4, (2, 16, 16), 80.0, 0))
    if want("tinyF"):
        save("dit_tinyF_forward.safetensors", *dit_case(ref, "tinyF", 256, 1, 2, (1, 16, 16), 0.7, 0, forward=True))
    if want("samplerA"):
        save("sampler_tinyA_g0.safetensors", *sampler_case(ref, "samplerA", 256, 2, 2, 9, (128, 128), 4, 0.0, 42, 3, True))
    if want("samplerB"):
        save("sampler_tinyB_g2.safetensors", *sampler_case(ref, "samplerB", 256, 1, 2, 1, (128, 128), 3, 2.0, 7, 1, False))
    if want("wide1"):
        t0 = time.time()
        t, m = dit_case(ref, "wide1", 4096, 1, 32, (1, 32, 32), 2.0, 3, dtypes=(torch.bfloat16,))
        save("dit_wide1.safetensors", t, m)
        print(f"wide1 took {time.time()-t0:.1f}s")
    if want("vaecfg"):
        vae_config_case()
    if want("cond"):
        save("conditions.safetensors", *conditions_case(ref))
    if args.full and want("fullwidth"):
        for tag, (t, m) in full_width_cases(ref).items():
            save({"g4": "sampler_full28_cfg1_g4.safetensors"}.get(tag, f"dit_{tag}.safetensors"), t, m)
    if args.full and want("full"):
        t0 = time.time()
        t, m = dit_case(ref, "full28", 4096, 28, 32, (1, 32, 32), 2.0, 3, dtypes=(torch.bfloat16,), keep_blocks=False)
        save("dit_full28_cfg1.safetensors", t, m)
        print(f"full28 took {time.time()-t0:.1f}s")


if __name__ == "__main__":
    main()
