"""Drop-in boundary: CleanDiffusionRendererPipeline with the reference's constructor and generate_video
signatures (diffusion_renderer_pipeline.py:38-111, :242-321).  Shape inference -> config -> (cached) model ->
sample -> decode -> post-process -> uint8 (B,T,H,W,C) numpy array.

Kept from the reference: attribute knobs (`guidance`, `num_steps`, `seed`, `set_model_type`), the shape-key
search order, the md5 config cache, every tensor of the batch cast to (device, dtype), ValueError when no usable
key exists.  Only the pre-loaded-model path is contractual: the reference's dynamic loader calls a method that
does not exist; here it raises a clear error instead.
"""
import hashlib
import json
import os
from typing import Dict

import numpy as np
import torch

from . import native as N
from .diffusion_renderer_config import get_config_from_tensor_shape, validate_config
from .model_diffusion_renderer import CleanDiffusionRendererModel


class CleanDiffusionRendererPipeline:
    def __init__(self, checkpoint_dir: str, checkpoint_name: str, model_type: str = "inverse", vae_instance=None,
                 model_instance=None, guidance: float = 2.0, num_steps: int = 20, height: int = 1024, width: int = 1024,
                 num_video_frames: int = 1, seed: int = 42, dtype: torch.dtype = torch.bfloat16):
        self.checkpoint_dir = checkpoint_dir
        self.checkpoint_name = checkpoint_name
        self.model_type = model_type.lower() if model_type else None
        self.vae_instance = vae_instance
        self.pre_loaded_model_instance = model_instance
        self.guidance = guidance
        self.num_steps = num_steps
        self.default_height = height
        self.default_width = width
        self.default_num_video_frames = num_video_frames
        self.seed = seed
        self.device = torch.device("cuda")     # torch "cuda" == HIP on ROCm (reference :81)
        self.dtype = dtype
        self.config = None
        self.model = None
        self._config_cache = {}
        self._model_cache = {}

    def set_model_type(self, model_type: str):
        new = model_type.lower()
        if self.model_type != new:
            self.model_type = new
            self.config = None
            self.model = None
            self.__dict__.pop("_h2d_cache", None)     # a new renderer starts from a new clip

    @staticmethod
    def _resolved(device) -> torch.device:
        """torch.device("cuda") (what the reference's constructor stores, :81) names the CURRENT device; tensors report an
        indexed one ("cuda:0").  Compare devices only after resolving the default index."""
        d = torch.device(device)
        if d.type == "cuda" and d.index is None:
            return torch.device("cuda", torch.cuda.current_device())
        return d

    @staticmethod
    def _get_config_hash(config) -> str:
        return hashlib.md5(json.dumps(config, sort_keys=True, default=str).encode()).hexdigest()

    def _ensure_model_loaded(self, input_tensor_shape: tuple):
        new_config = get_config_from_tensor_shape(self.model_type, input_tensor_shape)
        new_config["model_type"] = self.model_type
        h = self._get_config_hash(new_config)
        if self.config is not None and self._get_config_hash(self.config) == h:
            return self.model
        if h in self._model_cache:
            self.config = new_config
            self.model = self._model_cache[h]
            return self.model
        self.config = new_config
        validate_config(self.config)
        inst = self.pre_loaded_model_instance
        if isinstance(inst, dict):
            # {"inverse": model, "forward": model}: the two renderers are different checkpoints (in_channels 33 vs 153); the
            # reference reuses the inverse-built net for both and cannot run the forward pass
            inst = inst.get(self.model_type)
        if inst is not None:
            self.model = self._configure_pre_loaded_model(inst, self.config)
        else:
            self.model = self._load_model_with_config()
        self._model_cache[h] = self.model
        return self.model

    def _configure_pre_loaded_model(self, model_instance, config):
        """Re-point a loaded model at a new shape/condition config without touching weights (reference :168-198).

        The weights' own network config (channel counts) stays the model's: the reference rewrites
        `model.config` wholesale, which silently mismatches in_channels for the forward pass."""
        net_cfg = model_instance.config.get("net") if isinstance(model_instance.config, dict) else None
        model_instance.config = dict(config)
        if net_cfg is not None and model_instance.net is not None:
            model_instance.config["net"] = net_cfg
        model_instance.condition_keys = config.get("condition_keys", ["image", "depth", "normal", "basecolor", "roughness", "metallic"])
        model_instance.condition_drop_rate = config.get("condition_drop_rate", 0.0)
        model_instance.append_condition_mask = config.get("append_condition_mask", True)
        model_instance.input_data_key = config.get("input_data_key", "video")
        if self.vae_instance:
            model_instance.vae = self.vae_instance
        return model_instance.to(self.device)

    def _move_to_device(self, data_batch):
        """Every tensor -> (device, dtype) as the reference does (:200-208); the copy of an unchanged source tensor is
        reused across calls (the inverse node hands the same clip to 5 consecutive passes)."""
        out = {}
        cache = self.__dict__.setdefault("_h2d_cache", {})
        dev = self._resolved(self.device)
        live = {}
        for k, v in data_batch.items():
            if not isinstance(v, torch.Tensor):
                out[k] = v
                continue
            key = (id(v), v._version, tuple(v.shape), v.dtype)
            hit = live.get(key) or cache.get(key)          # (rgb and video are usually the SAME tensor: one copy)
            if hit is None or hit[0] is not v or hit[1].device != dev:
                hit = (v, v.to(device=dev, dtype=self.dtype))
            live[key] = hit
            out[k] = hit[1]
        # only the tensors of the batch in hand stay referenced (the node builds a fresh clip tensor per call: an entry of an
        # earlier call can never hit again and would pin ~0.6 GB of host + HBM per cfg-3 clip)
        self._h2d_cache = live
        return out

    def _load_model_with_config(self):
        path = os.path.join(self.checkpoint_dir, self.checkpoint_name)
        if not os.path.isfile(path):
            raise FileNotFoundError(f"checkpoint not found: {path} (pass a pre-loaded model_instance, as the loader node does)")
        if path.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd = load_file(path)
        else:
            sd = torch.load(path, map_location="cpu", weights_only=True)
        if "model" in sd:
            sd = sd["model"]
        model = CleanDiffusionRendererModel(self.config, device=self.device)
        model.load_state_dict(sd, strict=True)
        if self.vae_instance:
            model.vae = self.vae_instance
        return model

    def generate_video(self, data_batch: Dict[str, torch.Tensor], normalize_normal: bool = False, seed: int = None,
                       init_noise: torch.Tensor = None) -> np.ndarray:
        sample = self._sample(data_batch, seed, init_noise)
        video = self.model.decode(sample)
        u8 = N.postprocess_u8(video.to(self.dtype).contiguous(), normalize_normal)   # fused :299-318
        return u8.cpu().numpy()

    def generate_video_passes(self, data_batch: Dict[str, torch.Tensor], normalize_normal, seed: int = None,
                              init_noise: torch.Tensor = None):
        """P G-buffer passes of ONE clip stepped together: `data_batch["context_index"]` is [P, 1], the
        clip tensors keep batch 1, `normalize_normal` is a sequence of P flags.  Returns the P uint8 arrays generate_video
        would return for the passes run one after the other (reference nodes.py:191-213): same seed -> same start noise,
        one encode instead of P, one batched denoiser call per step."""
        flags = list(normalize_normal)
        sample = self._sample(data_batch, seed, init_noise)
        if sample.shape[0] != len(flags):
            raise ValueError(f"{len(flags)} normalize_normal flags for {sample.shape[0]} passes")
        outs = []
        for p, flag in enumerate(flags):
            video = self.model.decode(sample[p:p + 1])
            outs.append(N.postprocess_u8(video.to(self.dtype).contiguous(), bool(flag)))
        return [o.cpu().numpy() for o in outs]

    def _sample(self, data_batch, seed, init_noise):
        effective_seed = seed if seed is not None else self.seed
        data_batch = self._move_to_device(data_batch)
        video_tensor = None
        for key in ("rgb", "image", "basecolor", "normal", "depth", "roughness", "metallic"):
            if key in data_batch:
                video_tensor = data_batch[key]
                break
        if video_tensor is None:
            raise ValueError("No suitable input tensor for shape inference found in data_batch. Looked for "
                             "['rgb', 'image', 'basecolor', 'normal', 'depth', 'roughness', 'metallic']")
        self._ensure_model_loaded(tuple(video_tensor.shape))
        C = self.config["latent_shape"][0]
        _, _, T, H, W = video_tensor.shape
        state_shape = [C, (T - 1) // 8 + 1, H // 8, W // 8]
        return self.model.generate_samples_from_batch(data_batch, guidance=self.guidance, state_shape=state_shape,
                                                      num_steps=self.num_steps, is_negative_prompt=False,
                                                      seed=effective_seed, init_noise=init_noise)
