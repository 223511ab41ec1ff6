"""Drop-in boundary surface: node mappings, INPUT/RETURN types, tensor conventions, errors.  CPU only."""
import pytest
import torch


def test_node_mappings_and_types(pkg):
    assert set(pkg.NODE_CLASS_MAPPINGS) == {"LoadDiffusionRendererModel", "Cosmos1InverseRenderer", "Cosmos1ForwardRenderer",
                                            "LoadHDRImage"}
    assert pkg.NODE_DISPLAY_NAME_MAPPINGS["Cosmos1InverseRenderer"] == "Cosmos1 Inverse Renderer"
    inv = pkg.NODE_CLASS_MAPPINGS["Cosmos1InverseRenderer"]
    it = inv.INPUT_TYPES()
    assert set(it["required"]) == {"pipeline", "image"} and it["required"]["pipeline"] == ("DIFFUSION_RENDERER_PIPELINE",)
    assert it["optional"]["guidance"][1] == {"default": 0.0, "min": 0.0, "max": 10.0, "step": 0.1}
    assert it["optional"]["seed"][1]["max"] == 0xffffffffffffffff
    assert inv.RETURN_TYPES == ("IMAGE",) * 5 and inv.RETURN_NAMES == ("base_color", "metallic", "roughness", "normal", "depth")
    assert inv.FUNCTION == "run_inverse_pass" and inv.CATEGORY == "Cosmos1"
    fwd = pkg.NODE_CLASS_MAPPINGS["Cosmos1ForwardRenderer"]
    ft = fwd.INPUT_TYPES()
    assert list(ft["required"]) == ["pipeline", "depth", "normal", "roughness", "metallic", "base_color", "env_map"]
    assert ft["optional"]["env_format"][0] == ["proj", "ball"] and fwd.FUNCTION == "run_forward_pass"
    ld = pkg.NODE_CLASS_MAPPINGS["LoadDiffusionRendererModel"]
    assert ld.RETURN_TYPES == ("DIFFUSION_RENDERER_PIPELINE",) and ld.FUNCTION == "load_pipeline"
    assert pkg.nodes.GBUFFER_INDEX_MAPPING == {"basecolor": 0, "metallic": 1, "roughness": 2, "normal": 3, "depth": 4}


def test_standardize_inputs(pkg):
    f = pkg.nodes.standardize_to_5d
    assert f(torch.zeros(4, 5, 3)).shape == (1, 1, 4, 5, 3)
    assert f(torch.zeros(2, 4, 5, 3)).shape == (2, 1, 4, 5, 3)
    assert f(torch.zeros(1, 9, 4, 5, 3)).shape == (1, 9, 4, 5, 3)
    assert f([torch.zeros(9, 4, 5, 3)]).shape == (1, 9, 4, 5, 3)
    with pytest.raises(ValueError):
        f(torch.zeros(3, 3))
    with pytest.raises(TypeError):
        f("not a tensor")


def test_pipeline_constructor_and_errors(pkg):
    P = pkg.diffusion_renderer_pipeline.CleanDiffusionRendererPipeline
    p = P("/ckpt", "m.pt")
    assert (p.model_type, p.guidance, p.num_steps, p.seed, p.dtype) == ("inverse", 2.0, 20, 42, torch.bfloat16)
    assert p.device == torch.device("cuda")
    p = P("/ckpt", "m.pt", model_type=None, guidance=0.0, num_steps=15)
    assert p.model_type is None
    p.set_model_type("Inverse")
    assert p.model_type == "inverse"
    p.device = torch.device("cpu")
    with pytest.raises(ValueError, match="No suitable input tensor"):
        p.generate_video({"context_index": torch.zeros(1, 1)})
    cfgm = pkg.diffusion_renderer_config
    with pytest.raises(ValueError):
        cfgm.get_config_from_tensor_shape("inverse", (1, 3, 64, 64))
    with pytest.raises(ValueError):
        cfgm.get_config_by_model_type("sideways")
    cfg = cfgm.get_config_from_tensor_shape("forward", (1, 3, 57, 704, 1280))
    assert cfg["net"]["additional_concat_ch"] == 136 and cfg["latent_shape"] == [16, 8, 88, 160]
    assert cfg["condition_keys"][-3:] == ["env_ldr", "env_log", "env_nrm"] and cfg["append_condition_mask"] is True
    cfgm.validate_config(cfg)


def test_configs_equal_reference_where_present(pkg):
    from oracle.ref_import import reference_available, load_reference
    if not reference_available():
        pytest.skip("reference not present")
    ref = load_reference().config
    mine = pkg.diffusion_renderer_config
    assert ref.get_inverse_renderer_config(576, 1024, 57) == mine.get_inverse_renderer_config(576, 1024, 57)
    assert ref.get_forward_renderer_config(704, 1280, 57) == mine.get_forward_renderer_config(704, 1280, 57)
    assert ref.get_network_config() == mine.get_network_config()
    assert ref.get_scheduler_config() == mine.get_scheduler_config()


def test_model_strict_state_dict(pkg):
    M = pkg.model_diffusion_renderer.CleanDiffusionRendererModel
    cfg = pkg.diffusion_renderer_config.get_inverse_renderer_config()
    cfg["net"].update(model_channels=256, num_blocks=1, num_heads=2)
    m = M(cfg, device="cpu")
    shapes = m.expected_state_dict_shapes()
    assert "net.blocks.block0.blocks.1.block.attn.to_k.0.weight" in shapes and "logvar.1.weight" in shapes
    with pytest.raises(RuntimeError, match="loading state_dict"):
        m.load_state_dict({"net.bogus": torch.zeros(1)}, strict=True)
    with pytest.raises(RuntimeError, match="weights not loaded"):
        m.generate_samples_from_batch({}, state_shape=(16, 1, 4, 4))


def test_state_dict_names_equal_reference_manifest(pkg):
    """the names / shapes load_state_dict(strict=True) expects are those of the reference model's own full-size
    state dict (tests/golden/state_dict_manifest.json, written by tools/make_goldens.py from the reference on the meta device)."""
    import json
    import os
    from conftest import GOLDEN
    man = json.load(open(os.path.join(GOLDEN, "state_dict_manifest.json")))
    M = pkg.model_diffusion_renderer.CleanDiffusionRendererModel
    cm = pkg.diffusion_renderer_config
    for kind, cfg in (("inverse", cm.get_inverse_renderer_config()), ("forward", cm.get_forward_renderer_config())):
        cfg["model_type"] = kind
        mine = {k: list(v) for k, v in M(cfg, device="cpu").expected_state_dict_shapes().items()}
        assert mine == man[kind], kind
    assert man["inverse"]["net.x_embedder.proj.1.weight"] == [4096, 132]
    assert man["forward"]["net.x_embedder.proj.1.weight"] == [4096, 612]
