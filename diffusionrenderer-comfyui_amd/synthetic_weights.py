"""Deterministic, machine-independent synthetic weights for the renderer DiT.

There is no network for checkpoints, so benches and parity tests run on
random-init weights.  The generator is a counter-based integer hash
(splitmix64 of crc32(name)<<32 | element index) followed by exact float
conversions, so the same state dict is produced bit-for-bit on any host or
device (the golden fixtures under tests/golden/ hold only inputs/outputs).

Parameter names/shapes follow the reference's strict state-dict contract
(nodes.py:103-110; CleanGeneralDIT.py module tree; model_diffusion_renderer.py:114-117).
"""
import math
import zlib
from typing import Dict

import torch

_M64 = (1 << 64) - 1


def _s64(v: int) -> int:
    v &= _M64
    return v - (1 << 64) if v >= (1 << 63) else v


_GOLD = _s64(0x9E3779B97F4A7C15)
_C1 = _s64(0xBF58476D1CE4E5B9)
_C2 = _s64(0x94D049BB133111EB)


def _lsr(z: torch.Tensor, k: int) -> torch.Tensor:
    return (z >> k) & ((1 << (64 - k)) - 1)


def hash_uniform(name: str, numel: int, device="cpu", chunk: int = 1 << 24) -> torch.Tensor:
    """fp32 tensor of `numel` values in [-1, 1), a pure function of (name, index)."""
    seed = zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF
    out = torch.empty(numel, dtype=torch.float32, device=device)
    base = _s64(seed << 32)
    for start in range(0, numel, chunk):
        n = min(chunk, numel - start)
        z = torch.arange(start, start + n, dtype=torch.int64, device=device) + base
        z = z + _GOLD
        z = (z ^ _lsr(z, 30)) * _C1
        z = (z ^ _lsr(z, 27)) * _C2
        z = z ^ _lsr(z, 31)
        u24 = _lsr(z, 40).to(torch.float32)              # exact: < 2^24
        out[start:start + n] = u24 * (2.0 / 16777216.0) - 1.0   # exact in fp32
    return out


def _linear(name, out_f, in_f, dtype, device, gain=1.0):
    bound = gain * math.sqrt(3.0) / math.sqrt(in_f)     # unit-variance-preserving uniform
    w = hash_uniform(name, out_f * in_f, device) * bound
    return w.reshape(out_f, in_f).to(dtype)


def _norm_weight(name, dim, dtype, device):
    return (1.0 + 0.1 * hash_uniform(name, dim, device)).to(dtype)


def dit_param_shapes(net: dict) -> Dict[str, tuple]:
    """name -> shape for every tensor under `net.` (reference module tree)."""
    D = net["model_channels"]
    L = net["num_blocks"]
    H = net["num_heads"]
    dh = D // H
    r = net.get("adaln_lora_dim", 256)
    ctx = net["crossattn_emb_channels"]
    hid = int(D * net["mlp_ratio"])
    in_ch = net["in_channels"] + net.get("additional_concat_ch", 16) + (1 if net.get("concat_padding_mask", True) else 0)
    patch_dim = in_ch * net["patch_spatial"] ** 2 * net["patch_temporal"]
    out_dim = net["patch_spatial"] ** 2 * net["patch_temporal"] * net["out_channels"]
    s = {
        "net.x_embedder.proj.1.weight": (D, patch_dim),
        "net.t_embedder.1.linear_1.weight": (D, D),
        "net.t_embedder.1.linear_2.weight": (3 * D, D),
        "net.pos_embedder.seq": (max(512, dh),),
        "net.affline_norm.weight": (D,),
        "net.final_layer.linear.weight": (out_dim, D),
        "net.final_layer.adaLN_modulation.1.weight": (r, D),
        "net.final_layer.adaLN_modulation.2.weight": (2 * D, r),
    }
    if net.get("use_context_embedding", True):
        s["net.context_embedding.weight"] = (16, ctx)
    for i in range(L):
        p = f"net.blocks.block{i}.blocks."
        for j, kind in enumerate(net["block_config"].split("-")):
            q = f"{p}{j}."
            s[q + "adaLN_modulation.1.weight"] = (r, D)
            s[q + "adaLN_modulation.2.weight"] = (3 * D, r)
            kind = kind.strip().lower()
            if kind in ("fa", "ca"):
                kv_in = D if kind == "fa" else ctx
                a = q + "block.attn."
                s[a + "to_q.0.weight"] = (D, D)
                s[a + "to_q.1.weight"] = (dh,)
                s[a + "to_k.0.weight"] = (D, kv_in)
                s[a + "to_k.1.weight"] = (dh,)
                s[a + "to_v.0.weight"] = (D, kv_in)
                s[a + "to_out.0.weight"] = (D, D)
            else:
                s[q + "block.layer1.weight"] = (hid, D)
                s[q + "block.layer2.weight"] = (D, hid)
    return s


def synth_state_dict(net: dict, dtype=torch.bfloat16, device="cpu", with_logvar: bool = True) -> Dict[str, torch.Tensor]:
    """Random-init state dict with the reference's names; same bits everywhere."""
    sd = {}
    for name, shape in dit_param_shapes(net).items():
        if name.endswith("pos_embedder.seq"):
            sd[name] = torch.arange(shape[0], dtype=torch.float32, device=device).to(dtype)
        elif len(shape) == 1:
            sd[name] = _norm_weight(name, shape[0], dtype, device)
        elif name.endswith("context_embedding.weight"):
            sd[name] = (hash_uniform(name, shape[0] * shape[1], device) * math.sqrt(3.0)).reshape(shape).to(dtype)
        else:
            sd[name] = _linear(name, shape[0], shape[1], dtype, device)
    if with_logvar:
        sd["logvar.0.freqs"] = hash_uniform("logvar.0.freqs", 128, device).to(dtype)
        sd["logvar.0.phases"] = hash_uniform("logvar.0.phases", 128, device).to(dtype)
        sd["logvar.1.weight"] = _linear("logvar.1.weight", 1, 128, dtype, device)
    return sd


def synth_tensor(name: str, shape, dtype=torch.bfloat16, device="cpu", scale: float = 1.0) -> torch.Tensor:
    """Named synthetic activation/input tensor in [-scale, scale)."""
    n = 1
    for d in shape:
        n *= d
    return (hash_uniform(name, n, device) * scale).reshape(*shape).to(dtype)


# ------------------------------------------------------------------------------------------------ tokenizer
COSMOS_CV8x8x8 = {      # the reference's VAE_config.json:1-30 (AutoencoderKLCosmos hyper-parameters)
    "in_channels": 3, "out_channels": 3, "latent_channels": 16,
    "encoder_block_out_channels": (128, 256, 512, 512), "decode_block_out_channels": (256, 512, 512, 512),
    "num_layers": 2, "attention_resolutions": (32,), "resolution": 1024, "patch_size": 4, "patch_type": "haar",
    "spatial_compression_ratio": 8, "temporal_compression_ratio": 8, "scaling_factor": 1.0,
}


def vae_param_shapes(cfg: dict = None) -> Dict[str, tuple]:
    """name -> shape of an AutoencoderKLCosmos state dict (diffusers >= 0.34 naming; UNVERIFIED offline)."""
    cfg = cfg or COSMOS_CV8x8x8
    s: Dict[str, tuple] = {}

    def conv(name, cin, cout, k):
        s[name + ".weight"] = (cout, cin) + tuple(k)
        s[name + ".bias"] = (cout,)

    def proj(name, cin, cout):           # CosmosConvProjection3d: (1,3,3) then (3,1,1)
        conv(name + ".conv_s", cin, cout, (1, 3, 3))
        conv(name + ".conv_t", cout, cout, (3, 1, 1))

    def norm(name, c):
        s[name + ".norm.weight"] = (c,)
        s[name + ".norm.bias"] = (c,)

    def resnet(name, cin, cout):
        norm(name + ".norm1", cin)
        proj(name + ".conv1", cin, cout)
        norm(name + ".norm2", cout)
        proj(name + ".conv2", cout, cout)
        if cin != cout:
            conv(name + ".conv_shortcut", cin, cout, (1, 1, 1))

    def attn(name, c):
        norm(name + ".norm", c)
        for p in ("to_q", "to_k", "to_v", "to_out.0"):
            conv(f"{name}.{p}", c, c, (1, 1, 1))

    def mid(name, c):
        resnet(name + ".resnets.0", c, c)
        attn(name + ".attentions.0", c)
        attn(name + ".temp_attentions.0", c)
        resnet(name + ".resnets.1", c, c)

    ps = cfg["patch_size"]
    n_sp = int(math.log2(cfg["spatial_compression_ratio"])) - int(math.log2(ps))
    n_tp = int(math.log2(cfg["temporal_compression_ratio"])) - int(math.log2(ps))
    enc = tuple(cfg["encoder_block_out_channels"])
    proj("encoder.conv_in", cfg["in_channels"] * ps ** 3, enc[0])
    for i in range(len(enc) - 1):
        for j in range(cfg["num_layers"]):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", enc[i] if j == 0 else enc[i + 1], enc[i + 1])
        if i < len(enc) - 2:
            sp, tp = i < n_sp, i < n_tp
            d = f"encoder.down_blocks.{i}.downsamplers.0"
            if sp:
                conv(d + ".conv1", enc[i + 1], enc[i + 1], (1, 3, 3))
            if tp:
                conv(d + ".conv2", enc[i + 1], enc[i + 1], (3, 1, 1))
            if sp or tp:
                conv(d + ".conv3", enc[i + 1], enc[i + 1], (1, 1, 1))
    mid("encoder.mid_block", enc[-1])
    norm("encoder.norm_out", enc[-1])
    proj("encoder.conv_out", enc[-1], cfg["latent_channels"])
    conv("quant_conv", cfg["latent_channels"], cfg["latent_channels"], (1, 1, 1))
    conv("post_quant_conv", cfg["latent_channels"], cfg["latent_channels"], (1, 1, 1))
    dec = tuple(reversed(cfg["decode_block_out_channels"]))
    proj("decoder.conv_in", cfg["latent_channels"], dec[0])
    mid("decoder.mid_block", dec[0])
    for i in range(len(dec) - 1):
        for j in range(cfg["num_layers"] + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", dec[i] if j == 0 else dec[i + 1], dec[i + 1])
        if i < len(dec) - 2:
            tp = 0 < i < n_tp + 1
            sp = tp or (i < n_sp and n_sp > n_tp)
            u = f"decoder.up_blocks.{i}.upsamplers.0"
            if tp:
                conv(u + ".conv1", dec[i + 1], dec[i + 1], (3, 1, 1))
            if sp:
                conv(u + ".conv2", dec[i + 1], dec[i + 1], (1, 3, 3))
            if sp or tp:
                conv(u + ".conv3", dec[i + 1], dec[i + 1], (1, 1, 1))
    norm("decoder.norm_out", dec[-1])
    proj("decoder.conv_out", dec[-1], cfg["out_channels"] * ps ** 3)
    return s


def synth_vae_state_dict(cfg: dict = None, dtype=torch.bfloat16, device="cpu") -> Dict[str, torch.Tensor]:
    """Random-init tokenizer weights (variance-preserving uniform convs, small biases, norm gains near 1)."""
    sd = {}
    for name, shape in vae_param_shapes(cfg).items():
        n = 1
        for d in shape:
            n *= d
        key = "vae." + name
        if name.endswith(".norm.weight"):
            sd[name] = (1.0 + 0.1 * hash_uniform(key, n, device)).to(dtype)
        elif name.endswith(".bias"):
            sd[name] = (0.05 * hash_uniform(key, n, device)).to(dtype)
        else:
            fan_in = n // shape[0]
            sd[name] = (hash_uniform(key, n, device) * (math.sqrt(3.0) / math.sqrt(fan_in))).reshape(shape).to(dtype)
    return sd
