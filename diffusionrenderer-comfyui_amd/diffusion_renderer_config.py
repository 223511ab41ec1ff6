"""Plain-dict configuration of the renderer models (boundary support).

Mirrors the names and values of the reference's diffusion_renderer_config.py
(:47-128 network, :106-118 scheduler, :121-128 vae, :131-188 inverse, :191-251 forward,
:277-305 from-shape, :308-348 validate) so configs hash/compare the same way at the
pipeline boundary.  Only the keys the hot path consumes matter to the HIP engine:
model_channels / num_blocks / num_heads / mlp_ratio / in,out_channels / patch sizes /
block_config / crossattn_emb_channels / additional_concat_ch / use_context_embedding.
"""
from typing import Any, Dict

_NETWORK = {
    "model_channels": 4096, "num_blocks": 28, "num_heads": 32, "head_dim": 128, "mlp_ratio": 4.0,
    "context_dim": 1024, "adaln_lora_dim": 256, "time_embed_dim": 4096, "max_time_embed_period": 10000,
    "in_channels": 16, "out_channels": 16, "patch_spatial": 2, "patch_temporal": 1,
    "max_img_h": 240, "max_img_w": 240, "max_frames": 128,
    "block_config": "FA-CA-MLP", "concat_padding_mask": True, "block_x_format": "THWBD",
    "pos_emb_cls": "rope3d", "pos_emb_learnable": False, "pos_emb_interpolation": "crop",
    "rope_h_extrapolation_ratio": 1.0, "rope_w_extrapolation_ratio": 1.0, "rope_t_extrapolation_ratio": 2.0,
    "affline_emb_norm": True, "use_adaln_lora": True,
    "extra_per_block_abs_pos_emb": True, "extra_per_block_abs_pos_emb_type": "sincos",
    "extra_h_extrapolation_ratio": 1.0, "extra_w_extrapolation_ratio": 1.0, "extra_t_extrapolation_ratio": 1.0,
    "crossattn_emb_channels": 1024,
}

_SCHEDULER = {
    "type": "EDMEulerScheduler", "sigma_max": 80.0, "sigma_min": 0.02, "sigma_data": 0.5,
    "num_train_timesteps": 1000, "beta_start": 0.00085, "beta_end": 0.012,
    "beta_schedule": "scaled_linear", "prediction_type": "v_prediction",
}

FORWARD_CONDITION_KEYS = ["basecolor", "normal", "metallic", "roughness", "depth", "env_ldr", "env_log", "env_nrm"]


def get_network_config() -> Dict[str, Any]:
    return dict(_NETWORK)


def get_scheduler_config() -> Dict[str, Any]:
    return dict(_SCHEDULER)


def get_vae_config(num_frames: int = 57) -> Dict[str, Any]:
    return {"pixel_chunk_duration": num_frames, "latent_channels": 16,
            "spatial_compression_ratio": 8, "temporal_compression_ratio": 8}


def _renderer_config(height, width, num_frames, condition_keys, drop_rate, append_mask, concat_ch, ctx_emb):
    return {
        "sigma_data": 0.5, "precision": "bfloat16", "input_data_key": "video",
        # the reference computes the config's latent T as num_frames // 8 + 1 (:145); the pipeline itself
        # uses (T-1)//8+1 for the state shape (diffusion_renderer_pipeline.py:275) - both kept
        "latent_shape": [16, num_frames // 8 + 1, height // 8, width // 8],
        "condition_keys": list(condition_keys), "condition_drop_rate": drop_rate, "append_condition_mask": append_mask,
        "net": {**get_network_config(), "additional_concat_ch": concat_ch, "use_context_embedding": ctx_emb,
                "crossattn_emb_channels": 1024},
        "scheduler": get_scheduler_config(), "vae": get_vae_config(num_frames),
        "guidance": 2.0, "num_steps": 20, "height": height, "width": width, "num_video_frames": num_frames,
    }


def get_inverse_renderer_config(height: int = 704, width: int = 1280, num_frames: int = 57) -> Dict[str, Any]:
    """RGB -> G-buffer.  16 latent condition channels, no condition mask, context embedding on."""
    return _renderer_config(height, width, num_frames, ["rgb"], 0.1, False, 16, True)


def get_forward_renderer_config(height: int = 704, width: int = 1280, num_frames: int = 57) -> Dict[str, Any]:
    """G-buffers + env-map -> RGB.  8 x (16 latent + 1 mask) = 136 condition channels, no context embedding."""
    return _renderer_config(height, width, num_frames, FORWARD_CONDITION_KEYS, 0.05, True, 17 * 8, False)


def get_config_by_model_type(model_type: str, height: int = 704, width: int = 1280, num_frames: int = 57):
    mt = model_type.lower()
    if mt == "inverse":
        return get_inverse_renderer_config(height, width, num_frames)
    if mt == "forward":
        return get_forward_renderer_config(height, width, num_frames)
    raise ValueError(f"Unknown model type: {model_type}. Must be 'inverse' or 'forward'")


def get_config_from_tensor_shape(model_type, tensor_shape):
    if len(tensor_shape) != 5:
        raise ValueError(f"Expected a 5D tensor shape, but got {len(tensor_shape)} dimensions.")
    _, _, T, H, W = tensor_shape
    if model_type == "inverse":
        return get_inverse_renderer_config(height=H, width=W, num_frames=T)
    if model_type == "forward":
        return get_forward_renderer_config(height=H, width=W, num_frames=T)
    raise ValueError(f"Unknown model type for config generation: {model_type}")


def validate_config(config: Dict[str, Any]) -> None:
    for key in ("sigma_data", "precision", "input_data_key", "latent_shape", "condition_keys", "net", "scheduler", "vae"):
        if key not in config:
            raise ValueError(f"Missing required config key: {key}")
    ls = config["latent_shape"]
    if not isinstance(ls, list) or len(ls) != 4:
        raise ValueError(f"Invalid latent_shape: {ls}. Expected [C, T, H, W] format.")
    for key in ("model_channels", "num_blocks", "num_heads", "in_channels", "out_channels"):
        if key not in config["net"]:
            raise ValueError(f"Missing required net config key: {key}")


PRESET_CONFIGS = {
    "inverse_1024x1024": get_inverse_renderer_config(1024, 1024, 1),
    "forward_1024x1024": get_forward_renderer_config(1024, 1024, 1),
    "inverse_704x1280_video": get_inverse_renderer_config(704, 1280, 57),
    "forward_704x1280_video": get_forward_renderer_config(704, 1280, 57),
}


def get_preset_config(preset_name: str) -> Dict[str, Any]:
    if preset_name not in PRESET_CONFIGS:
        raise ValueError(f"Unknown preset: {preset_name}. Available: {list(PRESET_CONFIGS.keys())}")
    return PRESET_CONFIGS[preset_name].copy()
