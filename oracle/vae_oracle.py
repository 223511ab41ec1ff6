"""CPU oracle of the Cosmos-1.0 CV8x8x8 tokenizer (what CleanVAE.py:18,50-51,59-60 reaches through
diffusers.AutoencoderKLCosmos).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The arithmetic lives in the un-vendored third-party dependency `diffusers`
(class AutoencoderKLCosmos, file models/autoencoders/autoencoder_kl_cosmos.py, effective pin >= 0.34.0:
VAE_config.json:3 `_diffusers_version 0.34.0.dev0`; requirements.txt:2).  It is not installed in the build
container, the reference ships no tests, fixtures or weights for it, and there is no network, so this
restatement follows (a) every hyper-parameter of the reference's VAE_config.json, (b) the wrapper semantics of
CleanVAE.py:34-60 (deterministic identity posterior, no latent mean/std, 5-D in/out) and (c) the published
architecture of the NVIDIA Cosmos causal tokenizer as diffusers restates it (factorised causal convs,
per-frame 1-group GroupNorm, 2-level 3-D Haar patching, hybrid down/up-sampling, 1-head spatial +
causal temporal attention in the mid block).  What IS pinned: each building block below is checked
against torch.nn.functional on CPU and by invariants (IDWT(DWT(x)) == x, causality, T=1 == first frame)
in tests/test_vae_oracle.py; the HIP tokenizer is checked against this file.

Tensors are (B, C, T, H, W); weights use the diffusers state-dict names (synthetic_weights.vae_param_shapes).
All module outputs round to the weight dtype exactly where the bf16 nn.Module graph would.
"""
import math
from typing import Dict

import torch
import torch.nn.functional as F


def causal_conv3d(x, w, b, stride=(1, 1, 1), padding=0):
    """CosmosCausalConv3d: replicate frame 0 in front (k_t - 1 + 1 - stride_t) times, zero-pad H/W, plain conv3d."""
    kt = w.shape[2]
    tpad = (kt - 1) + (1 - stride[0])
    if tpad > 0:
        x = torch.cat([x[:, :, :1].repeat(1, 1, tpad, 1, 1), x], dim=2)
    if padding:
        x = F.pad(x, (padding, padding, padding, padding, 0, 0))
    return F.conv3d(x, w, b, stride=stride)


def group_norm_causal(x, w, b, eps=1e-6):
    """CosmosCausalGroupNorm(num_groups=1): GroupNorm over (C,H,W) of every frame separately."""
    B, C, T, H, W = x.shape
    y = F.group_norm(x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W), 1, w, b, eps)
    return y.reshape(B, T, C, H, W).permute(0, 2, 1, 3, 4)


def avg_pool(x, k):
    """F.avg_pool3d with fp32 accumulation and one rounding (torch CPU has no bf16 avg_pool3d kernel)."""
    return F.avg_pool3d(x.float(), k, k).to(x.dtype)


class VaeOracle:
    def __init__(self, sd: Dict[str, torch.Tensor], cfg: dict, dtype=torch.bfloat16):
        self.cfg, self.dtype = cfg, dtype
        self.sd = {k: v.to(dtype) for k, v in sd.items()}
        self.ps = cfg["patch_size"]
        self.n_sp = int(math.log2(cfg["spatial_compression_ratio"])) - int(math.log2(self.ps))
        self.n_tp = int(math.log2(cfg["temporal_compression_ratio"])) - int(math.log2(self.ps))
        self.trace = None

    def _rec(self, k, v):
        if self.trace is not None:
            self.trace[k] = v.detach().clone()

    # ---- parameterised pieces
    def conv(self, name, x, stride=(1, 1, 1), padding=0):
        return causal_conv3d(x, self.sd[name + ".weight"], self.sd[name + ".bias"], stride, padding)

    def proj(self, name, x):                     # CosmosConvProjection3d
        return self.conv(name + ".conv_t", self.conv(name + ".conv_s", x, padding=1))

    def norm(self, name, x):
        return group_norm_causal(x, self.sd[name + ".norm.weight"], self.sd[name + ".norm.bias"])

    def resnet(self, name, x):
        res = self.conv(name + ".conv_shortcut", x) if (name + ".conv_shortcut.weight") in self.sd else x
        h = self.proj(name + ".conv1", F.silu(self.norm(name + ".norm1", x)))
        h = self.proj(name + ".conv2", F.silu(self.norm(name + ".norm2", h)))
        return h + res

    def attention(self, name, x, temporal):
        B, C, T, H, W = x.shape
        h = self.norm(name + ".norm", x)
        q, k, v = (self.conv(f"{name}.{p}", h) for p in ("to_q", "to_k", "to_v"))
        if temporal:        # [B*H*W, 1 head, T, C], causal
            q, k, v = (t.permute(0, 3, 4, 2, 1).reshape(B * H * W, 1, T, C) for t in (q, k, v))
            mask = torch.tril(torch.ones(T, T, dtype=torch.bool))
            o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
            o = o.reshape(B, H, W, T, C).permute(0, 4, 3, 1, 2)
        else:               # [B*T, 1 head, H*W, C]
            q, k, v = (t.permute(0, 2, 3, 4, 1).reshape(B * T, 1, H * W, C) for t in (q, k, v))
            o = F.scaled_dot_product_attention(q, k, v)
            o = o.reshape(B, T, H, W, C).permute(0, 4, 1, 2, 3)
        return self.conv(name + ".to_out.0", o) + x

    def mid(self, name, x):
        x = self.resnet(name + ".resnets.0", x)
        x = self.attention(name + ".attentions.0", x, temporal=False)
        x = self.attention(name + ".temp_attentions.0", x, temporal=True)
        return self.resnet(name + ".resnets.1", x)

    def downsample(self, name, x, sp, tp):
        if not sp and not tp:
            return x
        if sp:
            x = F.pad(x, (0, 1, 0, 1, 0, 0))
            x = self.conv(name + ".conv1", x, stride=(1, 2, 2)) + avg_pool(x, (1, 2, 2))
        if tp:
            x = torch.cat([x[:, :, :1], x], dim=2)
            x = self.conv(name + ".conv2", x, stride=(2, 1, 1)) + avg_pool(x, (2, 1, 1))
        return self.conv(name + ".conv3", x)

    def upsample(self, name, x, sp, tp):
        if not sp and not tp:
            return x
        if tp:
            f = 2 if x.shape[2] > 1 else 1
            x = x.repeat_interleave(f, dim=2)[:, :, f - 1:]
            x = self.conv(name + ".conv1", x) + x
        if sp:
            x = x.repeat_interleave(2, dim=3).repeat_interleave(2, dim=4)
            x = self.conv(name + ".conv2", x, padding=1) + x
        return self.conv(name + ".conv3", x)

    # ---- Haar patching (CosmosPatchEmbed3d / CosmosUnpatcher3d, patch_method "haar")
    def _dwt(self, x):
        dt = x.dtype
        g = x.shape[1]
        wv = torch.tensor([0.7071067811865476, 0.7071067811865476])
        hl = wv.flip(0).reshape(1, 1, -1).repeat(g, 1, 1).to(dt)
        hh = (wv * ((-1) ** torch.arange(2))).reshape(1, 1, -1).repeat(g, 1, 1).to(dt)
        x = F.pad(x, (0, 1, 0, 1, 0, 1), mode="reflect").to(dt)
        xl = F.conv3d(x, hl.unsqueeze(3).unsqueeze(4), groups=g, stride=(2, 1, 1))
        xh = F.conv3d(x, hh.unsqueeze(3).unsqueeze(4), groups=g, stride=(2, 1, 1))
        outs = []
        for a in (xl, xh):
            al = F.conv3d(a, hl.unsqueeze(2).unsqueeze(4), groups=g, stride=(1, 2, 1))
            ah = F.conv3d(a, hh.unsqueeze(2).unsqueeze(4), groups=g, stride=(1, 2, 1))
            for bnd in (al, ah):
                outs.append(F.conv3d(bnd, hl.unsqueeze(2).unsqueeze(3), groups=g, stride=(1, 1, 2)))
                outs.append(F.conv3d(bnd, hh.unsqueeze(2).unsqueeze(3), groups=g, stride=(1, 1, 2)))
        return torch.cat(outs, dim=1) / 8 ** 0.5          # [lll llh lhl lhh hll hlh hhl hhh]

    def patch(self, x):
        xi, xv = x[:, :, :1], x[:, :, 1:]
        x = torch.cat([xi.repeat_interleave(self.ps, dim=2), xv], dim=2)
        for _ in range(int(math.log2(self.ps))):
            x = self._dwt(x)
        return x

    def _idwt(self, x):
        dt = x.dtype
        g = x.shape[1] // 8
        wv = torch.tensor([0.7071067811865476, 0.7071067811865476])
        hl = wv.flip(0).reshape(1, 1, -1).repeat(g, 1, 1).to(dt)
        hh = (wv * ((-1) ** torch.arange(2))).reshape(1, 1, -1).repeat(g, 1, 1).to(dt)
        lll, llh, lhl, lhh, hll, hlh, hhl, hhh = torch.chunk(x, 8, dim=1)
        ct = F.conv_transpose3d

        def w_(a, b):
            y = ct(a, hl.unsqueeze(2).unsqueeze(3), groups=g, stride=(1, 1, 2))
            return ct(b, hh.unsqueeze(2).unsqueeze(3), groups=g, stride=(1, 1, 2)) + y

        def h_(a, b):
            y = ct(a, hl.unsqueeze(2).unsqueeze(4), groups=g, stride=(1, 2, 1))
            return ct(b, hh.unsqueeze(2).unsqueeze(4), groups=g, stride=(1, 2, 1)) + y

        xl = h_(w_(lll, llh), w_(lhl, lhh))
        xh = h_(w_(hll, hlh), w_(hhl, hhh))
        y = ct(xl, hl.unsqueeze(3).unsqueeze(4), groups=g, stride=(2, 1, 1))
        y = ct(xh, hh.unsqueeze(3).unsqueeze(4), groups=g, stride=(2, 1, 1)) + y
        return y * 8 ** 0.5

    def unpatch(self, x):
        for _ in range(int(math.log2(self.ps))):
            x = self._idwt(x)
        return x[:, :, self.ps - 1:]

    # ---- encoder / decoder (CosmosEncoder3d / CosmosDecoder3d) and the KL wrapper's (post_)quant convs
    def encode(self, x):
        """AutoencoderKLCosmos.encode(x).latent_dist.sample() as CleanVAE.encode uses it (identity posterior)."""
        cfg = self.cfg
        x = x.to(self.dtype)
        h = self.patch(x)
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
        h = self.proj("encoder.conv_out", F.silu(self.norm("encoder.norm_out", h)))
        return self.conv("quant_conv", h)

    def decode(self, z):
        """AutoencoderKLCosmos.decode(z).sample."""
        cfg = self.cfg
        h = self.conv("post_quant_conv", z.to(self.dtype))
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
        h = self.proj("decoder.conv_out", F.silu(self.norm("decoder.norm_out", h)))
        self._rec("dec.conv_out", h)
        return self.unpatch(h)
