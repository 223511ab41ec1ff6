"""EDM Euler sampler + condition plumbing around the HIP DiT and tokenizer.

Host-side mirror of the reference's model_diffusion_renderer.py: same class and method names, argument
meaning and error behaviour (CleanEDMEulerScheduler :16-82, CleanConditioner :84-96,
CleanDiffusionRendererModel :98-234), so the pipeline above it is unchanged.  Differences by design:
  * `net` is a HipDiT (hand-written gfx950 kernels) instead of an nn.Module; weights arrive through
    load_state_dict() with the reference's strict parameter names and are repacked once;
  * scheduler scalars are evaluated on the host with the reference's own fp32 torch expressions and the
    latent update runs in fused elementwise kernels (bit-exact with the unfused fp32 reference ops);
  * sigma values never leave the host, so the denoising loop has no device->host synchronisation.
"""
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from . import native as N
from .diffusion_renderer_config import get_inverse_renderer_config
from .dit_engine import HipDiT
from .synthetic_weights import dit_param_shapes


class CleanEDMEulerScheduler:
    def __init__(self, sigma_max=80.0, sigma_min=0.02, sigma_data=0.5, **kwargs):
        self.sigma_max = sigma_max
        self.sigma_min = sigma_min
        self.sigma_data = sigma_data
        self.sigmas, self.timesteps, self.current_step = None, None, 0

    def set_timesteps(self, num_steps, device=None):
        # fp32 logspace + trailing 0, kept on the HOST (reference :23-28 builds the same table on `device`)
        sigmas = torch.logspace(np.log10(self.sigma_max), np.log10(self.sigma_min), num_steps, dtype=torch.float32)
        self.sigmas = torch.cat([sigmas, torch.tensor([0.0], dtype=torch.float32)])
        self.timesteps = self.sigmas[:-1]
        self.current_step = 0

    def _sigma(self, timestep) -> Tensor:
        return torch.as_tensor(timestep, dtype=torch.float32).detach().to("cpu").reshape(())

    def scale_model_input(self, sample, timestep):
        """sample * 1/sqrt(sigma^2 + sigma_data^2), fp32 math, result in sample.dtype (reference :30-44)."""
        c_in = 1 / torch.sqrt(self._sigma(timestep) ** 2 + self.sigma_data ** 2)
        return N.edm_scale_input(sample.contiguous(), float(c_in))

    def step(self, model_output, timestep, sample):
        if self.sigmas is None or self.current_step >= len(self.sigmas) - 1:
            raise RuntimeError("Scheduler not initialized or timesteps exhausted")
        sigma = self._sigma(timestep)
        sigma_next = self.sigmas[self.current_step + 1]
        sd = self.sigma_data
        c_skip = sd ** 2 / (sigma ** 2 + sd ** 2)
        c_out = (sigma * sd) / torch.sqrt(sigma ** 2 + sd ** 2)
        dt = sigma_next - sigma
        prev = N.edm_step(model_output.contiguous(), sample.contiguous(), float(c_skip), float(c_out), float(sigma),
                          float(dt))
        self.current_step += 1

        class StepResult:
            def __init__(self, prev_sample):
                self.prev_sample = prev_sample

        return StepResult(prev)


class CleanCondition:
    def __init__(self, **kwargs):
        self.data = kwargs

    def to_dict(self):
        return self.data


class CleanConditioner:
    def get_condition_uncondition(self, data_batch: Dict) -> Tuple[CleanCondition, CleanCondition]:
        cond, uncond = {}, {}
        for key in ("latent_condition", "context_index"):
            if key in data_batch:
                cond[key] = data_batch[key]
                uncond[key] = torch.zeros_like(data_batch[key])
        return CleanCondition(**cond), CleanCondition(**uncond)


class CleanDiffusionRendererModel:
    """Sampler + conditioner + HipDiT.  Not an nn.Module: there are no torch parameters on this path."""

    def __init__(self, config: Dict[str, Any] = None, device=None, process_group=None):
        if config is None:
            config = get_inverse_renderer_config()
        self.config = config
        sched = dict(config.get("scheduler", {}))
        sched.pop("prediction_type", None)
        self.scheduler = CleanEDMEulerScheduler(**sched)
        self.conditioner = CleanConditioner()
        self.net: Optional[HipDiT] = None
        self.vae = None
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        self.dtype = torch.bfloat16
        self.process_group = process_group
        model_type = config.get("model_type", "inverse")
        default_keys = ["image", "rgb"] if model_type == "inverse" else ["depth", "normal", "basecolor", "roughness", "metallic"]
        self.condition_keys = config.get("condition_keys", default_keys)
        self.condition_drop_rate = config.get("condition_drop_rate", 0.0)
        self.append_condition_mask = config.get("append_condition_mask", True)
        self.input_data_key = config.get("input_data_key", "video")
        self.tokenizer = None
        self._enc_cache = {}      # (data_ptr, shape, version) -> (tensor ref, latent): the node encodes the SAME clip 5x

    # ---- nn.Module-shaped conveniences used by the loader node (nodes.py:103-114 in the reference)
    def eval(self):
        return self

    def to(self, *args, **kwargs):
        return self

    def to_empty(self, device=None):
        if device is not None:
            self.device = torch.device(device)
        return self

    def expected_state_dict_shapes(self) -> Dict[str, tuple]:
        shapes = dict(dit_param_shapes(self.config["net"]))
        shapes.update({"logvar.0.freqs": (128,), "logvar.0.phases": (128,), "logvar.1.weight": (1, 128)})
        return shapes

    def load_state_dict(self, state_dict: Dict[str, Tensor], strict: bool = True):
        """Reference checkpoint names in (net.* + the dead logvar.* placeholders, :114-117), repacked HipDiT out."""
        expect = self.expected_state_dict_shapes()
        missing = [k for k in expect if k not in state_dict]
        unexpected = [k for k in state_dict if k not in expect]
        bad = [k for k in expect if k in state_dict and tuple(state_dict[k].shape) != tuple(expect[k])]
        if strict and (missing or unexpected or bad):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:5]} unexpected {unexpected[:5]} "
                               f"shape mismatch {bad[:5]}")
        self.net = HipDiT(self.config["net"], state_dict, device=self.device, process_group=self.process_group)
        return missing, unexpected

    def _get_tensor_kwargs(self):
        return {"device": self.device, "dtype": self.dtype}

    # ---- tokenizer seam (reference :138-156)
    def encode(self, x: Tensor, **vae_kwargs) -> Tensor:
        if self.vae is None:
            raise RuntimeError("VAE not initialized in model.")
        if x.ndim != 5:
            raise ValueError(f"Model encode expects a 5D tensor (B,C,T,H,W), but got {x.ndim}D.")
        return self.vae.encode(x, **vae_kwargs) * self.scheduler.sigma_data

    def decode(self, x: Tensor) -> Tensor:
        if self.vae is None:
            raise RuntimeError("VAE not initialized in model.")
        if x.ndim != 5:
            raise ValueError(f"Model decode expects a 5D latent (B,C,T,H,W), but got {x.ndim}D.")
        return self.vae.decode(x / self.scheduler.sigma_data)

    def _encode_cached(self, x: Tensor, **vae_kwargs) -> Tensor:
        """encode(x) with a small identity cache: the inverse node runs 5 G-buffer passes over
        one clip and the reference re-encodes it every pass (model_diffusion_renderer.py:191); same tensor -> same latent."""
        key = (x.data_ptr(), tuple(x.shape), x._version, x.dtype, id(self.vae))
        hit = self._enc_cache.get(key)
        if hit is not None and hit[0] is x:
            return hit[1]
        latent = self.encode(x, **vae_kwargs).contiguous()
        while len(self._enc_cache) >= 8:                 # the forward renderer's 8 condition maps of ONE clip; oldest first
            self._enc_cache.pop(next(iter(self._enc_cache)))
        self._enc_cache[key] = (x, latent)
        return latent

    def prepare_diffusion_renderer_latent_conditions(self, data_batch: Dict[str, Tensor], condition_keys: list = None,
                                                     **kwargs) -> Tensor:
        """Per condition key: encode (or zeros) [+ ones/zeros mask], concat on C (reference :158-197)."""
        if self.vae is None:
            raise RuntimeError("VAE not initialized in model.")
        if condition_keys is None:
            condition_keys = self.condition_keys
        latent_shape = None
        for key in condition_keys:
            if key in data_batch:
                B, C, T, H, W = data_batch[key].shape
                f = self.vae.spatial_compression_factor
                latent_shape = (B, self.vae.latent_ch, self.vae.get_latent_num_frames(T), H // f, W // f)
                break
        if latent_shape is None:
            raise ValueError(f"Could not determine latent shape from keys {condition_keys}.")
        ref = data_batch[self.input_data_key]
        mask_shape = (latent_shape[0], 1, *latent_shape[2:])
        actual_keys = [k if k in data_batch else ("rgb" if "rgb" in data_batch and k == "image" else None)
                       for k in condition_keys]
        states = self._encode_conditions([data_batch[k] for k in actual_keys if k is not None])
        parts = []
        for actual in actual_keys:
            if actual is None:
                parts.append(torch.zeros(latent_shape, dtype=ref.dtype, device=ref.device))
                if self.append_condition_mask:
                    parts.append(torch.zeros(mask_shape, dtype=ref.dtype, device=ref.device))
            else:
                state = states.pop(0)
                parts.append(state)
                if self.append_condition_mask:
                    parts.append(torch.ones(mask_shape, dtype=state.dtype, device=state.device))
        return torch.cat(parts, dim=1)

    def _encode_conditions(self, tensors: list) -> list:
        """Tokenizer encodes of the condition maps.  One rank: one after the other (cached per tensor).  Several ranks: the
        maps are independent, so rank r encodes maps r, r + world, ... and the 2.4 MB latents are all-gathered - the forward
        renderer's 8 encodes ('other free parallelism') cost one encode per rank on an 8-GPU node."""
        from .parallel import allgather_stack, group_info
        rank, world = group_info(self.process_group) if self.process_group is not None else (0, 1)
        n = len(tensors)
        if world == 1 or n < 2:
            return [self._encode_cached(t) for t in tensors]
        per = -(-n // world)
        f = self.vae.spatial_compression_factor
        B, _, T, H, W = tensors[0].shape
        mine = torch.zeros((per, B, self.vae.latent_ch, self.vae.get_latent_num_frames(T), H // f, W // f),
                           dtype=self.dtype, device=self.device)          # zero slots pad ranks with fewer maps
        # (a tokenizer that owns a process group would cut every map into row bands - a collective; here each rank works alone)
        alone = {"bands": 1} if getattr(getattr(self.vae, "model", None), "world", 1) > 1 else {}
        for j, i in enumerate(range(rank, n, world)):
            mine[j] = self._encode_cached(tensors[i], **alone)
        allv = allgather_stack(mine, self.process_group)                  # [world, per, B, C, F, h, w]
        return [allv[i % world, i // world] for i in range(n)]

    def _get_conditions(self, data_batch: Dict, is_negative_prompt: bool = False):
        for key in ("rgb", "basecolor", "normal", "depth", "roughness", "metallic", "image"):
            if key in data_batch:
                self.input_data_key = key
                break
        with torch.no_grad():
            data_batch["latent_condition"] = self.prepare_diffusion_renderer_latent_conditions(data_batch, self.condition_keys)
        return self.conditioner.get_condition_uncondition(data_batch)

    def generate_samples_from_batch(self, data_batch: Dict, guidance: float = 0.0, seed: int = 1000,
                                    state_shape: Tuple = None, num_steps: int = 15, init_noise: Tensor = None,
                                    **kwargs) -> Tensor:
        """The denoising loop (reference :211-234).  `init_noise` (already scaled by sigma_0) replaces the
        device-side randn for parity runs; without it the reference's seeding convention is kept."""
        if self.net is None:
            raise RuntimeError("weights not loaded: call load_state_dict() first")
        with torch.no_grad():
            torch.manual_seed(seed)
            condition, uncondition = self._get_conditions(data_batch)
            self.scheduler.set_timesteps(num_steps)
            self.net.prepare_timesteps([float(t) for t in self.scheduler.timesteps])   # all AdaLN vectors, one batched pass
            if init_noise is not None:
                xt = init_noise.to(**self._get_tensor_kwargs()).contiguous()
            else:
                xt = torch.randn(size=(1, *state_shape), **self._get_tensor_kwargs()) * self.scheduler.sigmas[0]
            cond = condition.to_dict()
            unc = uncondition.to_dict()
            # the context indices are host scalars for the whole loop (no per-step .item() sync).  P > 1 indices over ONE
            # clip = P G-buffer passes stepped together: the reference runs them one after the other from
            # the same seed, i.e. from the same noise, so the batch starts from P copies of xt.
            cis = [0]
            if "context_index" in cond:
                cis = [int(v) for v in cond["context_index"].flatten().tolist()]
            P = len(cis)
            if xt.shape[0] == 1 and P > 1:
                xt = xt.expand(P, *xt.shape[1:]).contiguous()
            if xt.shape[0] != P:
                raise ValueError(f"{P} context indices for a latent batch of {xt.shape[0]}")
            lat_c = cond["latent_condition"]
            if lat_c.shape[0] != 1:
                # as in the reference, whose batch-1 noise cannot be concatenated with a batch-B condition (:222)
                raise ValueError("the renderer runs one clip per call (noise is drawn with batch 1)")
            if P > 1:
                lat_c = lat_c.expand(P, *lat_c.shape[1:])
            if guidance > 0:
                # cond and uncond halves as one batch of 2P clips (uncond = zero condition latent, index 0: reference :92-96)
                lat = torch.cat([lat_c, torch.zeros_like(lat_c)], 0)
                idx = cis + [0] * P
            else:
                lat, idx = lat_c, cis
            for t in self.scheduler.timesteps:
                xt_scaled = self.scheduler.scale_model_input(xt, timestep=t)
                if guidance > 0:
                    both = self.net(x=torch.cat([xt_scaled, xt_scaled], 0), timesteps=t, latent_condition=lat,
                                    context_index=idx)
                    net_output = N.cfg_combine(both[:P].contiguous(), both[P:].contiguous(), float(guidance))
                else:
                    net_output = self.net(x=xt_scaled, timesteps=t, latent_condition=lat, context_index=idx)
                xt = self.scheduler.step(net_output, t, xt).prev_sample
            return xt
