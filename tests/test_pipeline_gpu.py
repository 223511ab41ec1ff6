"""Pipeline-level parity on the GPU: CleanDiffusionRendererPipeline.generate_video (HipDiT + fused sampler and
post-process kernels) against goldens captured from the reference's own pipeline on CPU (stub tokenizer), with the
reference's x_T injected (it draws noise with the device RNG, model_diffusion_renderer.py:222); then the whole node
with the HIP tokenizer."""
import pytest
import torch

from conftest import load_golden, rel_l2, tiny_net
from stub_vae import StubVAE

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _pipeline(pkg, gpu, net, vae, guidance, steps):
    sw = pkg.synthetic_weights
    cfg = pkg.diffusion_renderer_config.get_inverse_renderer_config()
    cfg["net"] = dict(net)
    model = pkg.model_diffusion_renderer.CleanDiffusionRendererModel(cfg, device=gpu)
    model.load_state_dict(sw.synth_state_dict(net, BF, device=gpu), strict=True)
    p = pkg.diffusion_renderer_pipeline.CleanDiffusionRendererPipeline(
        "/nonexistent", "x.pt", model_type=None, vae_instance=vae, model_instance=model, guidance=guidance, num_steps=steps)
    p.device = gpu
    p.set_model_type("inverse")
    return p, model


@pytest.mark.parametrize("fixture,tag", [("sampler_tinyA_g0.safetensors", "samplerA"), ("sampler_tinyB_g2.safetensors", "samplerB")])
def test_generate_video_matches_reference_pipeline(pkg, gpu, fixture, tag):
    gold, meta = load_golden(fixture)
    net = tiny_net(pkg, int(meta["D"]), int(meta["L"]), int(meta["heads"]))
    steps, g = int(meta["steps"]), float(meta["guidance"])
    import json
    T, (H, W) = int(meta["T"]), json.loads(meta["HW"])
    p, model = _pipeline(pkg, gpu, net, StubVAE(), g, steps)
    rgb = pkg.synthetic_weights.synth_tensor(tag + ".rgb", (1, 3, T, H, W), torch.float32, scale=1.0)
    ci = torch.full((1, 1), int(meta["context_index"]), dtype=torch.long)
    batch = {"rgb": rgb, "video": rgb, "context_index": ci}
    video = p.generate_video(batch, normalize_normal=meta["normalize_normal"] == "True", seed=int(meta["seed"]),
                             init_noise=gold["xT"])
    ref = gold["video_u8"].numpy()
    assert video.shape == ref.shape and video.dtype == ref.dtype
    # the exact answer both bf16 evaluations approximate: fp32 oracle (bf16 host tables), same x_T, same stub tokenizer
    from oracle import dit_oracle as O
    sd = pkg.synthetic_weights.synth_state_dict(net, BF)
    o32 = O.DitOracle(sd, net, dtype=torch.float32, tables_dtype=BF)
    with torch.no_grad():
        x0_32 = O.sample_loop(lambda *a: o32.forward(*a).to(BF), gold["xT"], gold["latent_condition"], ci.to(BF), steps, g)
        u8_32 = O.postprocess(StubVAE().decode(x0_32 / 0.5), meta["normalize_normal"] == "True").numpy()
    d_hip = abs(video.astype(int) - u8_32.astype(int))
    d_ref = abs(ref.astype(int) - u8_32.astype(int))
    print(f"{tag}: uint8 mean |diff| vs fp32 oracle: hip {d_hip.mean():.4f} (max {d_hip.max()}), reference-bf16 {d_ref.mean():.4f} "
          f"(max {d_ref.max()}); hip vs reference {abs(video.astype(int) - ref.astype(int)).mean():.4f}")
    assert d_hip.mean() <= 1.5 * d_ref.mean() + 0.25
    # latent-level check of the sampler against the reference trajectory
    x0 = model.generate_samples_from_batch(dict(p._move_to_device(batch)), guidance=g, seed=int(meta["seed"]),
                                           state_shape=list(gold["xT"].shape[1:]), num_steps=steps, init_noise=gold["xT"])
    e_hip, e_ref = rel_l2(x0.cpu(), x0_32.float()), rel_l2(gold["x0"], x0_32.float())
    print(f"{tag}: x0 rel-L2 vs fp32 oracle: hip {e_hip:.3e}, reference-bf16 {e_ref:.3e}")
    assert e_hip <= 1.5 * e_ref + 1e-3
    assert torch.equal(model.scheduler.sigmas, gold_sigmas(steps))


def test_full_width_4step_pipeline_matches_reference_g4(pkg, gpu, full28_dit):
    """SURVEY 8c G4 / BASELINE config 1 end to end: the 7.2 B-parameter model, 1 frame 256 x 256, 4 Euler steps, guidance 0,
    normal pass, through generate_video with the x_T the reference drew - against the reference pipeline's own uint8 video and
    against the trajectory re-traced by the oracle (bf16 = what the reference computes, fp32 network = the exact answer)."""
    gold, meta = load_golden("sampler_full28_cfg1_g4.safetensors")
    steps = int(meta["steps"])
    cfg = pkg.diffusion_renderer_config.get_inverse_renderer_config()
    cfg["net"] = dict(full28_dit.net)
    model = pkg.model_diffusion_renderer.CleanDiffusionRendererModel(cfg, device=gpu)
    model.net = full28_dit                                   # weights already repacked (conftest.full28_dit)
    p = pkg.diffusion_renderer_pipeline.CleanDiffusionRendererPipeline(
        "/nonexistent", "x.pt", model_type=None, vae_instance=StubVAE(), model_instance=model, guidance=0.0, num_steps=steps)
    p.device = gpu
    p.set_model_type("inverse")
    rgb = pkg.synthetic_weights.synth_tensor("g4.rgb", (1, 3, 1, 256, 256), torch.float32, scale=1.0)
    ci = torch.full((1, 1), int(meta["context_index"]), dtype=torch.long)
    batch = {"rgb": rgb, "video": rgb, "context_index": ci}
    video = p.generate_video(batch, normalize_normal=True, seed=int(meta["seed"]), init_noise=gold["xT"])
    ref = gold["video_u8"].numpy()
    assert video.shape == ref.shape and video.dtype == ref.dtype
    x0 = model.generate_samples_from_batch(dict(p._move_to_device(batch)), guidance=0.0, seed=int(meta["seed"]),
                                           state_shape=list(gold["xT"].shape[1:]), num_steps=steps, init_noise=gold["xT"]).cpu()
    exact = gold["x0.fp32net"].float()
    e_hip, e_ref = rel_l2(x0, exact), rel_l2(gold["x0"], exact)
    from oracle import dit_oracle as O
    u8_32 = O.postprocess(StubVAE().decode(gold["x0.fp32net"] / 0.5), True).numpy()
    d_hip = abs(video.astype(int) - u8_32.astype(int))
    d_ref = abs(ref.astype(int) - u8_32.astype(int))
    print(f"g4: x0 rel-L2 vs fp32-network trajectory: hip {e_hip:.3e}, reference-bf16 {e_ref:.3e}; uint8 mean |diff|: hip "
          f"{d_hip.mean():.4f} (max {d_hip.max()}), reference {d_ref.mean():.4f} (max {d_ref.max()}); hip vs reference "
          f"{abs(video.astype(int) - ref.astype(int)).mean():.4f}")
    assert e_hip <= 1.5 * e_ref + 1e-3
    assert d_hip.mean() <= 1.5 * d_ref.mean() + 0.25


def gold_sigmas(n):
    from oracle import dit_oracle as O
    return O.edm_sigmas(n)


def test_inverse_node_end_to_end_with_hip_tokenizer(pkg, gpu):
    """Cosmos1InverseRenderer.run_inverse_pass over 5 G-buffer passes, HIP tokenizer + HIP DiT (tiny width)."""
    sw = pkg.synthetic_weights
    net = tiny_net(pkg, 256, 1, 2)
    vae = pkg.CleanVAE.CleanVAE(state_dict=sw.synth_vae_state_dict(device=gpu), device=gpu)
    p, _ = _pipeline(pkg, gpu, net, vae, 0.0, 2)
    node = pkg.NODE_CLASS_MAPPINGS["Cosmos1InverseRenderer"]()
    image = sw.synth_tensor("node.img", (1, 9, 64, 64, 3), torch.float32).abs()
    outs = node.run_inverse_pass(p, image, guidance=0.0, seed=3)
    assert len(outs) == 5
    for o in outs:
        assert o.shape == (9, 64, 64, 3) and o.dtype == torch.float32
        assert 0.0 <= o.min() and o.max() <= 1.0
        assert torch.equal(o, torch.round(o * 255) / 255)          # quantised to k/255
    # same seed, same input -> same noise -> deterministic
    outs2 = node.run_inverse_pass(p, image, guidance=0.0, seed=3)
    assert all(torch.equal(a, b) for a, b in zip(outs, outs2))
    # the five passes stepped as one batch (default) reproduce the passes run one after the other (reference order)
    p.batch_passes = False
    outs_seq = node.run_inverse_pass(p, image, guidance=0.0, seed=3)
    p.batch_passes = True
    for name, a, b in zip(("basecolor", "metallic", "roughness", "normal", "depth"), outs, outs_seq):
        assert torch.equal(a, b), f"{name}: batched vs sequential differ by up to {((a - b).abs() * 255).max():.0f} uint8 levels"
    assert not torch.equal(outs[0], outs[3])                        # passes differ (context rows, normal blend)
    # 4-D batch with N > 1 is rejected like the reference (B must be 1)
    with pytest.raises((ValueError, RuntimeError)):
        node.run_inverse_pass(p, image[0], guidance=0.0, seed=3)


def test_forward_node_end_to_end(pkg, gpu):
    """Cosmos1ForwardRenderer: 5 G-buffers + env map -> 8 encoded conditions (136 ch) -> forward DiT (in_ch 153) -> RGB.
    The reference cannot run this pass as committed; built by the intent of get_forward_renderer_config."""
    sw = pkg.synthetic_weights
    cfgm = pkg.diffusion_renderer_config
    net = tiny_net(pkg, 256, 1, 2, forward=True)
    cfg = cfgm.get_forward_renderer_config()
    cfg["net"] = dict(net)
    cfg["model_type"] = "forward"
    model = pkg.model_diffusion_renderer.CleanDiffusionRendererModel(cfg, device=gpu)
    model.load_state_dict(sw.synth_state_dict(net, BF, device=gpu), strict=True)
    vae = pkg.CleanVAE.CleanVAE(state_dict=sw.synth_vae_state_dict(device=gpu), device=gpu)
    p = pkg.diffusion_renderer_pipeline.CleanDiffusionRendererPipeline(
        "/nonexistent", "x.pt", model_type=None, vae_instance=vae, model_instance={"forward": model}, guidance=0.0, num_steps=2)
    p.device = gpu
    node = pkg.NODE_CLASS_MAPPINGS["Cosmos1ForwardRenderer"]()
    g = {k: sw.synth_tensor("fw." + k, (1, 9, 32, 32, 3), torch.float32).abs() for k in ("depth", "normal", "roughness", "metallic", "base_color")}
    env = sw.synth_tensor("fw.env", (1, 32, 64, 3), torch.float32).abs() * 4.0
    (out,) = node.run_forward_pass(p, g["depth"], g["normal"], g["roughness"], g["metallic"], g["base_color"], env,
                                   guidance=0.0, seed=5, env_format="proj", env_brightness=1.0, env_flip_horizontal=False,
                                   env_rotation=180.0)
    assert out.shape == (1, 9, 32, 32, 3) and out.dtype == torch.float32 and 0.0 <= out.min() and out.max() <= 1.0
    assert model.condition_keys == cfgm.FORWARD_CONDITION_KEYS
    # an inverse-built net cannot serve the forward pass: clear error instead of the reference's TypeError / shape crash
    inv_net = tiny_net(pkg, 256, 1, 2)
    inv = pkg.model_diffusion_renderer.CleanDiffusionRendererModel(dict(cfgm.get_inverse_renderer_config(), net=inv_net), device=gpu)
    inv.load_state_dict(sw.synth_state_dict(inv_net, BF, device=gpu), strict=True)
    p2 = pkg.diffusion_renderer_pipeline.CleanDiffusionRendererPipeline("/n", "x.pt", model_type=None, vae_instance=vae,
                                                                        model_instance=inv, guidance=0.0, num_steps=1)
    p2.device = gpu
    with pytest.raises(ValueError, match="channel count"):
        node.run_forward_pass(p2, g["depth"], g["normal"], g["roughness"], g["metallic"], g["base_color"], env, env_format="ball")


def test_forward_pass_full_size_cfg5(pkg, gpu):
    """BASELINE config 5 at size: forward renderer, 57 f x 576 x 1024, the full 28-block in_ch-153 network (random-init), the
    HIP tokenizer (8 condition encodes + the decode), 2 Euler steps through the node.  No CPU oracle exists at this size and the
    reference cannot run the pass as committed: what is asserted is the boundary contract (shape, dtype, k/255
    levels in [0, 1]) and that the whole pass - env-map conditions, encodes, sampler with device RNG, decode, uint8 - is
    reproducible bit for bit from its seed."""
    sw = pkg.synthetic_weights
    cfgm = pkg.diffusion_renderer_config
    T, H, W = 57, 576, 1024
    cfg = cfgm.get_forward_renderer_config(H, W, T)
    cfg["model_type"] = "forward"
    net = dict(cfg["net"])
    model = pkg.model_diffusion_renderer.CleanDiffusionRendererModel(cfg, device=gpu)
    model.load_state_dict(sw.synth_state_dict(net, BF, device=gpu), strict=True)
    vae = pkg.CleanVAE.CleanVAE(state_dict=sw.synth_vae_state_dict(device=gpu), device=gpu)
    p = pkg.diffusion_renderer_pipeline.CleanDiffusionRendererPipeline(
        "/nonexistent", "x.pt", model_type=None, vae_instance=vae, model_instance={"forward": model}, guidance=0.0, num_steps=2,
        height=H, width=W, num_video_frames=T)
    p.device = gpu
    node = pkg.NODE_CLASS_MAPPINGS["Cosmos1ForwardRenderer"]()
    gen = torch.Generator().manual_seed(77)
    base = torch.rand((1, T, H, W, 3), generator=gen)
    g = {k: base.roll(i * 37, dims=3) for i, k in enumerate(("depth", "normal", "roughness", "metallic", "base_color"))}
    env = torch.rand((1, 64, 128, 3), generator=gen) * 4.0
    outs = []
    for _ in range(2):
        (out,) = node.run_forward_pass(p, g["depth"], g["normal"], g["roughness"], g["metallic"], g["base_color"], env,
                                       guidance=0.0, seed=11, env_format="proj", env_brightness=1.0, env_flip_horizontal=False,
                                       env_rotation=90.0)
        outs.append(out)
    out = outs[0]
    assert out.shape == (1, T, H, W, 3) and out.dtype == torch.float32
    assert torch.isfinite(out).all() and 0.0 <= out.min() and out.max() <= 1.0
    lv = out[0, ::8, ::16, ::16] * 255.0
    assert torch.equal(lv, lv.round())                                   # k / 255 levels (the uint8 post-process)
    assert out.std() > 0.01                                              # not a constant image
    assert torch.equal(outs[0], outs[1])


def test_pipeline_error_behaviour(pkg, gpu):
    """Errors at the boundary: no usable key, a clip batch > 1 (the reference cannot run it either), a pass-flag count
    that does not match the context indices."""
    net = tiny_net(pkg, 256, 1, 2)
    p, _ = _pipeline(pkg, gpu, net, StubVAE(), 0.0, 1)
    with pytest.raises(ValueError, match="No suitable input tensor"):
        p.generate_video({"context_index": torch.zeros((1, 1), dtype=torch.long)})
    rgb2 = pkg.synthetic_weights.synth_tensor("err.rgb", (2, 3, 1, 64, 64), torch.float32)
    with pytest.raises(ValueError, match="one clip per call"):
        p.generate_video({"rgb": rgb2, "video": rgb2, "context_index": torch.zeros((2, 1), dtype=torch.long)})
    rgb = rgb2[:1]
    idx = torch.tensor([[0], [3]], dtype=torch.long)
    with pytest.raises(ValueError, match="normalize_normal flags"):
        p.generate_video_passes({"rgb": rgb, "video": rgb, "context_index": idx}, normalize_normal=[False])
    outs = p.generate_video_passes({"rgb": rgb, "video": rgb, "context_index": idx}, normalize_normal=[False, True], seed=1)
    assert len(outs) == 2 and outs[0].shape == (1, 1, 64, 64, 3) and outs[0].dtype.name == "uint8"


class _CountingVAE(StubVAE):
    def __init__(self):
        self.encodes = 0

    def encode(self, x):
        self.encodes += 1
        return super().encode(x)


def test_clip_is_uploaded_and_encoded_once_across_passes(pkg, gpu):
    """The pipeline as the loader node builds it (device = torch.device("cuda"), NOT overridden): the inverse node hands the
    same clip tensor to five generate_video calls; the host->device copy and the tokenizer encode must happen once (N1)."""
    net = tiny_net(pkg, 256, 1, 2)
    sw = pkg.synthetic_weights
    cfg = pkg.diffusion_renderer_config.get_inverse_renderer_config()
    cfg["net"] = dict(net)
    model = pkg.model_diffusion_renderer.CleanDiffusionRendererModel(cfg, device=gpu)
    model.load_state_dict(sw.synth_state_dict(net, BF, device=gpu), strict=True)
    vae = _CountingVAE()
    p = pkg.diffusion_renderer_pipeline.CleanDiffusionRendererPipeline(
        "/nonexistent", "x.pt", model_type=None, vae_instance=vae, model_instance=model, guidance=0.0, num_steps=1)
    assert p.device == torch.device("cuda")                     # the reference's constructor value (:81), no index
    p.set_model_type("inverse")
    rgb = sw.synth_tensor("cache.rgb", (1, 3, 1, 64, 64), torch.float32)
    outs = []
    for ci in (0, 3, 4):
        outs.append(p.generate_video({"rgb": rgb, "video": rgb, "context_index": torch.full((1, 1), ci, dtype=torch.long)}, seed=1))
    assert vae.encodes == 1
    assert len(p._h2d_cache) <= 3 and len(model._enc_cache) == 1
    # a new clip replaces the cached one (nothing accumulates)
    rgb2 = rgb.clone()
    p.generate_video({"rgb": rgb2, "video": rgb2, "context_index": torch.zeros((1, 1), dtype=torch.long)}, seed=1)
    assert vae.encodes == 2 and len(p._h2d_cache) <= 3 and len(model._enc_cache) <= 2
    # in-place edits of the clip are seen (version counter)
    rgb2.mul_(0.5)
    p.generate_video({"rgb": rgb2, "video": rgb2, "context_index": torch.zeros((1, 1), dtype=torch.long)}, seed=1)
    assert vae.encodes == 3


def test_loader_node_load_pipeline_and_one_pass(pkg, gpu, tmp_path, monkeypatch):
    """LoadDiffusionRendererModel.load_pipeline as ComfyUI calls it (reference nodes.py:74-127): VAE directory (config.json +
    safetensors) under <models>/vae/Cosmos-1.0-Tokenizer-CV8x8x8/vae, checkpoint {"model": state_dict} through
    comfy.utils.load_torch_file, strict load_state_dict, pipeline(guidance 0, 15 steps, seed 42) - then one inverse pass.
    ComfyUI's modules are faked; the network is shrunk (same code path, tiny weights)."""
    import json
    import sys
    import types
    from safetensors.torch import save_file

    sw = pkg.synthetic_weights
    models = tmp_path / "models"
    vae_dir = models / "vae" / "Cosmos-1.0-Tokenizer-CV8x8x8" / "vae"
    vae_dir.mkdir(parents=True)
    (vae_dir / "config.json").write_text(json.dumps({**{k: (list(v) if isinstance(v, tuple) else v) for k, v in sw.COSMOS_CV8x8x8.items()},
                                                     "_class_name": "AutoencoderKLCosmos"}))
    save_file({k: v.contiguous() for k, v in sw.synth_vae_state_dict().items()}, str(vae_dir / "diffusion_pytorch_model.safetensors"))
    net = tiny_net(pkg, 256, 1, 2)
    sd = sw.synth_state_dict(net, BF)
    assert any(k.startswith("logvar.") for k in sd) and any(k.startswith("net.") for k in sd)     # the checkpoint contract (a23)
    ckpt_dir = models / "diffusion_models"
    ckpt_dir.mkdir()
    torch.save({"model": sd}, str(ckpt_dir / "tiny.pt"))

    fp = types.ModuleType("folder_paths")
    fp.models_dir = str(models)
    fp.get_filename_list = lambda kind: ["tiny.pt"] if kind == "diffusion_models" else []
    fp.get_full_path = lambda kind, name: str(models / kind / name)
    comfy = types.ModuleType("comfy")
    mm = types.ModuleType("comfy.model_management")
    mm.get_torch_device = lambda: gpu
    mm.soft_empty_cache = lambda: None
    cu = types.ModuleType("comfy.utils")
    loads = []

    def load_torch_file(path, safe_load=False):
        loads.append((path, safe_load))
        return torch.load(path, map_location="cpu", weights_only=True)
    cu.load_torch_file = load_torch_file
    comfy.model_management, comfy.utils = mm, cu
    for name, mod in (("folder_paths", fp), ("comfy", comfy), ("comfy.model_management", mm), ("comfy.utils", cu)):
        monkeypatch.setitem(sys.modules, name, mod)
    tiny_cfg = dict(pkg.diffusion_renderer_config.get_inverse_renderer_config(), net=dict(net))
    monkeypatch.setattr(pkg.nodes, "get_inverse_renderer_config", lambda *a, **k: dict(tiny_cfg))

    Loader = pkg.NODE_CLASS_MAPPINGS["LoadDiffusionRendererModel"]
    assert Loader.INPUT_TYPES()["required"]["model"][0] == ["tiny.pt"]
    (pipe,) = Loader().load_pipeline("tiny.pt")
    assert loads == [(str(ckpt_dir / "tiny.pt"), True)]
    assert (pipe.guidance, pipe.num_steps, pipe.seed, pipe.model_type) == (0.0, 15, 42, None)
    assert pipe.vae_instance.latent_ch == 16 and pipe.vae_instance.spatial_compression_factor == 8
    pipe.num_steps = 2
    image = sw.synth_tensor("ldr.img", (1, 9, 64, 64, 3), torch.float32).abs()
    outs = pkg.NODE_CLASS_MAPPINGS["Cosmos1InverseRenderer"]().run_inverse_pass(pipe, image, guidance=0.0, seed=42)
    assert len(outs) == 5 and all(o.shape == (9, 64, 64, 3) and 0.0 <= o.min() and o.max() <= 1.0 for o in outs)
    # a missing tokenizer directory / a checkpoint with a wrong tensor fail like the reference (FileNotFoundError / strict load)
    (vae_dir / "config.json").rename(vae_dir / "config.json.bak")
    with pytest.raises(ValueError, match="Failed to load VAE"):
        Loader().load_pipeline("tiny.pt")
    (vae_dir / "config.json.bak").rename(vae_dir / "config.json")
    bad = dict(sd)
    bad.pop("net.final_layer.linear.weight")
    torch.save({"model": bad}, str(ckpt_dir / "bad.pt"))
    with pytest.raises(RuntimeError, match="state_dict"):
        Loader().load_pipeline("bad.pt")
    import shutil
    shutil.rmtree(models / "vae")
    with pytest.raises(FileNotFoundError):
        Loader().load_pipeline("tiny.pt")
