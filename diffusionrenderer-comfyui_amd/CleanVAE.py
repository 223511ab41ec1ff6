"""CleanVAE: the Cosmos-1.0 CV8x8x8 tokenizer behind the reference's wrapper API, on gfx950 kernels.

Mirrors CleanVAE.py:9-67 of the reference (constructor from a local model directory, encode/decode of 5-D
(B,C,T,H,W) tensors, get_latent_num_frames / get_pixel_num_frames, to(), reset_dtype(), latent_ch /
spatial_compression_factor / temporal_compression_factor attributes).  The reference delegates the arithmetic to
diffusers.AutoencoderKLCosmos; here HipCosmosTokenizer runs it through libdrn.so:

  * activations are channels-last bf16 with a 1-pixel zero halo, so spatial zero padding costs nothing;
  * every conv (1x3x3, 3x1x1 causal, 1x1x1, strided down-sampling convs) is one implicit-GEMM MFMA kernel with
    bias (+ residual) fused in the epilogue;
  * GroupNorm(1 group, per frame)+SiLU, the 2-level Haar patcher / unpatcher, the resampling helpers and the
    mid-block attentions are dedicated HBM-bound kernels.

Weight names are diffusers' AutoencoderKLCosmos state-dict names (see synthetic_weights.vae_param_shapes), so a real
`vae/diffusion_pytorch_model.safetensors` loads unchanged.  PARITY UNPINNED (diffusers is not available offline,
SURVEY.md F2): checked against oracle/vae_oracle.py, which is itself checked against torch.nn.functional.
"""
import json
import math
import os
from typing import Dict

import torch

from . import native as N
from . import native_vae as V
from .native_vae import CL
from .synthetic_weights import COSMOS_CV8x8x8, vae_param_shapes

BF = torch.bfloat16


def _ceil(x, m):
    return (x + m - 1) // m * m


class HipCosmosTokenizer:
    """The tokenizer on HIP kernels.  With a process group of `world` ranks every frame is cut into `world` bands of image rows
    (SURVEY.md 8f N3): causal temporal convs, 1x1x1 convs, Haar patching, resampling and the temporal attention are local to
    a band; the spatial 3x3 convs read one neighbour row per side (exchanged into the 1-pixel halo that already surrounds
    every activation), the per-frame GroupNorm sums its statistics over the ranks, and the mid block's spatial attention
    gathers K and V.  The latent / the decoded video are gathered at the end, so callers see the single-GPU interface."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], cfg: dict = None, device=None, process_group=None):
        self.cfg = dict(cfg or COSMOS_CV8x8x8)
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        self.pg = process_group
        self.rank, self.world = (0, 1)
        if process_group is not None:
            from .parallel import group_info
            self.rank, self.world = group_info(process_group)
        self._bands = 1                       # > 1 while a band-sharded encode / decode is running
        N.load_library()
        self.ps = self.cfg["patch_size"]
        if self.ps != 4 or self.cfg.get("patch_type", "haar") != "haar":
            raise ValueError("only the CV8x8x8 configuration (haar, patch 4) is built")
        self.n_sp = int(math.log2(self.cfg["spatial_compression_ratio"])) - int(math.log2(self.ps))
        self.n_tp = int(math.log2(self.cfg["temporal_compression_ratio"])) - int(math.log2(self.ps))
        expect = vae_param_shapes(self.cfg)
        missing = [k for k in expect if k not in state_dict]
        if missing:
            raise RuntimeError(f"tokenizer state dict is missing {len(missing)} tensors, e.g. {missing[:4]}")
        self.w, self.b, self.g = {}, {}, {}
        for name, shape in expect.items():
            t = state_dict[name].to(device=self.device, dtype=BF)
            if tuple(t.shape) != tuple(shape):
                raise RuntimeError(f"shape mismatch for {name}: {tuple(t.shape)} vs {shape}")
            if name.endswith(".weight") and t.dim() == 5:
                co, ci, kt, kh, kw = t.shape
                cip = _ceil(ci, 64)                                   # K padding (latent convs have 16 inputs)
                wt = torch.zeros((co, kt, kh, kw, cip), dtype=BF, device=self.device)
                wt[..., :ci] = t.permute(0, 2, 3, 4, 1)
                self.w[name[:-7]] = (wt.reshape(co, kt * kh * kw * cip).contiguous(), (kt, kh, kw), co, cip)
            elif name.endswith(".bias"):
                self.b[name[:-5]] = t.contiguous()
            else:
                self.g[name[:-7]] = t.contiguous()                    # GroupNorm gains (".norm.weight")
        self.trace = None

    def _rec(self, k, x):
        if self.trace is not None:
            self.trace[k] = x

    # ------------------------------------------------------------------ building blocks
    def conv(self, name, x: CL, stride=(1, 1, 1), pad=0, t_off=None, residual=None, out_halo=1, out_dims=None) -> CL:
        w, k, co, cip = self.w[name]
        assert x.C == cip, (name, x.C, cip)
        if self._bands > 1 and k[1] == 3:
            self._exchange_rows(x)                                    # neighbour rows into the halo before a spatial conv
        stored = _ceil(co, 64) if co % 64 else co                     # 16-channel latents live in 64-channel buffers
        return V.conv3d(x, w, self.b[name], co, k, stride, pad, t_off, residual=residual, out_halo=out_halo,
                        out_dims=out_dims, out_channels_stored=stored)

    def proj(self, name, x: CL, residual=None) -> CL:                 # CosmosConvProjection3d
        return self.conv(name + ".conv_t", self.conv(name + ".conv_s", x, pad=1), residual=residual)

    def norm(self, name, x: CL, silu: bool) -> CL:
        if self._bands == 1:
            return V.groupnorm_silu(x, self.g[name + ".norm"], self.b[name + ".norm"], silu)
        from .parallel import allgather_stack
        part = allgather_stack(V.groupnorm_stats(x), self.pg)                         # [world, T, 64, 2]
        part = part.permute(1, 0, 2, 3).reshape(x.T, self.world * 64, 2).contiguous()
        count = float(x.H * self.world) * float(x.W) * float(x.C)
        return V.groupnorm_apply(x, part, count, self.g[name + ".norm"], self.b[name + ".norm"], silu)

    def _exchange_rows(self, x: CL) -> None:
        """Top / bottom halo rows of this band <- last / first interior row of the neighbour bands (image border: stay zero).
        One small all-gather of the two edge rows of every rank (T x (W+2) x C each); the neighbours' rows are picked out."""
        from .parallel import allgather_stack
        assert x.halo == 1
        x.dirty_halo()                                                                # (its halo rows are about to hold data)
        edges = torch.stack([x.t[:, 1], x.t[:, x.H]], 0)                              # [2, T, W+2, C]
        allv = allgather_stack(edges, self.pg)                                         # [world, 2, T, W+2, C]
        if self.rank > 0:
            x.t[:, 0].copy_(allv[self.rank - 1, 1])
        if self.rank + 1 < self.world:
            x.t[:, x.H + 1].copy_(allv[self.rank + 1, 0])

    def resnet(self, name, x: CL) -> CL:
        res = self.conv(name + ".conv_shortcut", x) if (name + ".conv_shortcut") in self.w else x
        h = self.proj(name + ".conv1", self.norm(name + ".norm1", x, True))
        return self.proj(name + ".conv2", self.norm(name + ".norm2", h, True), residual=res)

    def attention(self, name, x: CL, temporal: bool) -> CL:
        T, H, W, C = x.T, x.H, x.W, x.C
        h = self.norm(name + ".norm", x, False)
        q, k, v = (self.conv(f"{name}.{p}", h, out_halo=0).t.view(T, H * W, C) for p in ("to_q", "to_k", "to_v"))
        scale = 1.0 / math.sqrt(C)
        if temporal:
            o = V.temporal_attention(q, k, v, scale)
        else:
            if self._bands > 1:                                       # queries of this band against the whole frame's keys
                from .parallel import allgather_stack
                kv = allgather_stack(torch.stack([k, v], 0), self.pg)                  # [world, 2, T, H*W, C]
                k = kv[:, 0].permute(1, 0, 2, 3).reshape(T, self.world * H * W, C)     # bands in rank order = row order
                v = kv[:, 1].permute(1, 0, 2, 3).reshape(T, self.world * H * W, C)
            P, Pk = H * W, k.shape[1]
            kp = _ceil(Pk, 64)
            o = torch.empty((T, P, C), dtype=BF, device=self.device)
            for f in range(T):                                        # one head of dim C per frame
                s = V.scores_f32(q[f], k[f].contiguous())            # raw fp32 scores; the softmax applies 1 / sqrt(C)
                p = V.softmax_rows(s, Pk, kp, scale)
                vt = V.transpose(v[f].contiguous(), kp)
                if kp % 64 == 0 and C % 128 == 0:
                    N.gemm(p, vt, out=o[f])                           # P . V on the DiT's tile GEMM (bf16 out, fp32 accumulate)
                else:
                    V.dense_gemm(p, vt, out=o[f])
        oc = CL(T, H, W, C, 0, self.device, tensor=o.view(T, H, W, C))
        return self.conv(name + ".to_out.0", oc, residual=x)

    def mid(self, name, x: CL) -> CL:
        x = self.resnet(name + ".resnets.0", x)
        x = self.attention(name + ".attentions.0", x, False)
        x = self.attention(name + ".temp_attentions.0", x, True)
        return self.resnet(name + ".resnets.1", x)

    def downsample(self, name, x: CL, sp: bool, tp: bool) -> CL:      # CosmosDownsample3d
        if not sp and not tp:
            return x
        if sp:    # conv1 (1,3,3)/s2 on the (0,1,0,1)-padded image + 2x2 mean
            x = self.conv(name + ".conv1", x, stride=(1, 2, 2), pad=0, residual=V.resample(x, 0),
                          out_dims=(x.T, x.H // 2, x.W // 2))
        if tp:    # conv2 (3,1,1)/s2 on [x0, x] (+1 causal pad frame) + 2-frame mean
            x = self.conv(name + ".conv2", x, stride=(2, 1, 1), t_off=2, residual=V.resample(x, 1))
        return self.conv(name + ".conv3", x)

    def upsample(self, name, x: CL, sp: bool, tp: bool) -> CL:        # CosmosUpsample3d
        if not sp and not tp:
            return x
        if tp:
            x = V.resample(x, 2)
            x = self.conv(name + ".conv1", x, residual=x)
        if sp:
            x = V.resample(x, 3)
            x = self.conv(name + ".conv2", x, pad=1, residual=x)
        return self.conv(name + ".conv3", x)

    # ------------------------------------------------------------------ encoder / decoder
    @torch.no_grad()
    def encode(self, video: torch.Tensor, bands: int = None) -> torch.Tensor:
        """[1,3,T,H,W] -> [1,16,F,H/8,W/8]  (AutoencoderKLCosmos.encode(x).latent_dist.sample(): identity posterior).
        bands: None = one band of rows per rank of the process group (a collective call), 1 = this rank alone."""
        B, C, T, H, W = video.shape
        if B != 1:
            return torch.cat([self.encode(video[i:i + 1], bands) for i in range(B)], 0)
        if (T - 1) % 8 != 0 or H % 8 != 0 or W % 8 != 0:
            raise ValueError(f"CV8x8x8 tokenizer needs T = 8k+1 frames and H, W multiples of 8, got {(T, H, W)}")
        with V.pooled_buffers():
            return self._encode(video, bands)

    def _encode(self, video, bands):
        B, C, T, H, W = video.shape
        cfg = self.cfg
        self._bands = self.world if (bands is None and self.world > 1 and (H // 8) % self.world == 0) else 1
        if self._bands > 1:                                           # this rank's band of image rows
            Hb = H // self._bands
            video = video[:, :, :, self.rank * Hb:(self.rank + 1) * Hb]
        h = V.haar_patch(video[0].to(device=self.device, dtype=BF).contiguous())
        self._rec("patch", h)
        h = self.proj("encoder.conv_in", h)
        self._rec("conv_in", h)
        enc = cfg["encoder_block_out_channels"]
        for i in range(len(enc) - 1):
            for j in range(cfg["num_layers"]):
                h = self.resnet(f"encoder.down_blocks.{i}.resnets.{j}", h)
                self._rec(f"down{i}.res{j}", h)
            if i < len(enc) - 2:
                h = self.downsample(f"encoder.down_blocks.{i}.downsamplers.0", h, i < self.n_sp, i < self.n_tp)
                self._rec(f"down{i}.ds", h)
        h = self.mid("encoder.mid_block", h)
        self._rec("mid", h)
        h = self.proj("encoder.conv_out", self.norm("encoder.norm_out", h, True))
        h = self.conv("quant_conv", h)
        z = V.cl_to_planar(h, cfg["latent_channels"])
        if self._bands > 1:
            z = self._gather_rows(z)
            self._bands = 1
        return z.unsqueeze(0)

    def _gather_rows(self, t: torch.Tensor) -> torch.Tensor:
        """[C, T, H_band, W] on every rank -> [C, T, H, W] (bands in rank order)."""
        from .parallel import allgather_stack
        allv = allgather_stack(t.contiguous(), self.pg)                               # [world, C, T, Hb, W]
        C_, T_, Hb, W_ = t.shape
        return allv.permute(1, 2, 0, 3, 4).reshape(C_, T_, self.world * Hb, W_).contiguous()

    @torch.no_grad()
    def decode(self, z: torch.Tensor, bands: int = None) -> torch.Tensor:
        """[1,16,F,h,w] -> [1,3,8(F-1)+1,8h,8w]  (AutoencoderKLCosmos.decode(z).sample).  bands: as in encode."""
        B = z.shape[0]
        if B != 1:
            return torch.cat([self.decode(z[i:i + 1], bands) for i in range(B)], 0)
        with V.pooled_buffers():
            return self._decode(z, bands)

    def _decode(self, z, bands):
        cfg = self.cfg
        hl = z.shape[3]
        self._bands = self.world if (bands is None and self.world > 1 and hl % self.world == 0) else 1
        if self._bands > 1:
            hb = hl // self._bands
            z = z[:, :, :, self.rank * hb:(self.rank + 1) * hb]
        h = V.planar_to_cl(z[0].to(device=self.device, dtype=BF).contiguous(), 64)
        h = self.conv("post_quant_conv", h)
        h = self.proj("decoder.conv_in", h)
        h = self.mid("decoder.mid_block", h)
        self._rec("dec.mid", h)
        dec = tuple(reversed(cfg["decode_block_out_channels"]))
        for i in range(len(dec) - 1):
            for j in range(cfg["num_layers"] + 1):
                h = self.resnet(f"decoder.up_blocks.{i}.resnets.{j}", h)
            if i < len(dec) - 2:
                tp = 0 < i < self.n_tp + 1
                sp = tp or (i < self.n_sp and self.n_sp > self.n_tp)
                h = self.upsample(f"decoder.up_blocks.{i}.upsamplers.0", h, sp, tp)
            self._rec(f"up{i}", h)
        h = self.proj("decoder.conv_out", self.norm("decoder.norm_out", h, True))
        self._rec("dec.conv_out", h)
        video = V.haar_unpatch(h, cfg["out_channels"])
        if self._bands > 1:
            video = self._gather_rows(video)
            self._bands = 1
        return video.unsqueeze(0)


class _Config(dict):
    __getattr__ = dict.__getitem__


class CleanVAE:
    """Drop-in for the reference's CleanVAE (CleanVAE.py:9-67)."""

    def __init__(self, model_path: str = None, state_dict: Dict[str, torch.Tensor] = None, config: dict = None, device=None,
                 process_group=None):
        cfg = dict(COSMOS_CV8x8x8)
        if model_path is not None:
            cpath = os.path.join(model_path, "config.json")
            if not os.path.isfile(cpath):
                raise ValueError(f"Failed to load VAE model from {model_path}")
            with open(cpath) as f:
                cfg.update({k: v for k, v in json.load(f).items() if k in cfg})
            if state_dict is None:
                from safetensors.torch import load_file
                files = sorted(f for f in os.listdir(model_path) if f.endswith(".safetensors"))
                if not files:
                    raise ValueError(f"Failed to load VAE model from {model_path}")
                state_dict = {}
                for f in files:
                    state_dict.update(load_file(os.path.join(model_path, f)))
        if config:
            cfg.update(config)
        if state_dict is None:
            raise ValueError("CleanVAE needs a model_path or a state_dict")
        self.config = _Config(cfg)
        self.model = HipCosmosTokenizer(state_dict, cfg, device=device, process_group=process_group)
        self.spatial_compression_factor = cfg["spatial_compression_ratio"]
        self.latent_ch = cfg["latent_channels"]
        self.temporal_compression_factor = 8

    def get_latent_num_frames(self, num_pixel_frames: int) -> int:
        if num_pixel_frames == 1:
            return 1
        return (num_pixel_frames - 1) // self.temporal_compression_factor + 1

    def get_pixel_num_frames(self, num_latent_frames: int) -> int:
        if num_latent_frames == 1:
            return 1
        return (num_latent_frames - 1) * self.temporal_compression_factor + 1

    @torch.no_grad()
    def encode(self, state_5d: torch.Tensor, bands: int = None) -> torch.Tensor:
        if state_5d.ndim != 5:
            raise ValueError(f"CleanVAE expects a 5D input (B, C, T, H, W), but got {state_5d.shape}")
        return self.model.encode(state_5d, bands)

    @torch.no_grad()
    def decode(self, latent_5d: torch.Tensor, bands: int = None) -> torch.Tensor:
        if latent_5d.ndim != 5:
            raise ValueError(f"CleanVAE expects a 5D latent (B, C, T, H, W), but got {latent_5d.shape}")
        return self.model.decode(latent_5d, bands)

    def to(self, device):
        return self          # weights were placed on the GPU at load; the HIP path has no CPU mode

    def reset_dtype(self, dtype: torch.dtype):
        if dtype != torch.bfloat16:
            raise ValueError("the HIP tokenizer computes in bf16 (nodes.py:91 asks for bf16)")
