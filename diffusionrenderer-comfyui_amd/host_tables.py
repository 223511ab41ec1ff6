"""Host-built tables of the DiT: 3-D RoPE cos/sin and the timestep sinusoid.

The reference evaluates these with torch ops in the MODEL dtype (bf16 after
nodes.py:109 casts the whole module): theta**range, 1/x, outer(),
cos(), sin() each round to bf16, which moves the angles by up to 0.14 rad from
their fp32 values.  An "accurate" device-side RoPE would therefore break parity,
so the tables are built here, on the host, with the same torch ops in the same
dtype and handed to the kernels, which only apply them.  They are tiny
([tokens,128] and [1,D]) and cached per shape / per sigma.
"""
import math

import torch


def rope_cos_sin(T: int, H: int, W: int, head_dim: int, seq: torch.Tensor, dtype=torch.bfloat16):
    """cos/sin [(T H W), head_dim] as CleanRoPE3D.forward + apply_rotary_pos_emb_pure_torch produce them.

    CleanGeneralDIT.py:86-159 (angle table, lanes [t | h | w | t | h | w], theta_t = 20000, theta_h = theta_w = 10000;
    range buffers take their constructor values) and :77-78 (cos/sin of the dtype-rounded angles).
    """
    seq = seq.detach().to("cpu", dtype)
    dim_h = head_dim // 6 * 2
    dim_t = head_dim - 2 * dim_h
    rng_s = (torch.arange(0, dim_h, 2)[: dim_h // 2].float() / dim_h).to(dtype)
    rng_t = (torch.arange(0, dim_t, 2)[: dim_t // 2].float() / dim_t).to(dtype)
    f_h = 1.0 / (10000.0 ** rng_s)
    f_w = 1.0 / (10000.0 ** rng_s)
    f_t = 1.0 / (20000.0 ** rng_t)
    e_t = torch.outer(seq[:T], f_t)[:, None, None, :].expand(T, H, W, -1)
    e_h = torch.outer(seq[:H], f_h)[None, :, None, :].expand(T, H, W, -1)
    e_w = torch.outer(seq[:W], f_w)[None, None, :, :].expand(T, H, W, -1)
    ang = torch.cat([e_t, e_h, e_w, e_t, e_h, e_w], dim=-1).reshape(T * H * W, head_dim)
    return ang.cos().to(dtype).contiguous(), ang.sin().to(dtype).contiguous()


def timestep_sinusoid(sigma: float, channels: int, dtype=torch.bfloat16) -> torch.Tensor:
    """[1, channels] = [cos | sin](bf16(sigma) * exp(-ln(1e4) * i / half)), fp32 math, one rounding.

    CleanTimesteps.forward, CleanGeneralDIT.py:316-335; the raw sigma is cast to the model dtype first (:664, F10).
    """
    t = torch.tensor([sigma], dtype=torch.float32).to(dtype)
    half = channels // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32) / (half - 0.0)
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1).to(dtype)
