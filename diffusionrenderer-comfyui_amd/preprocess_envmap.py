"""Environment-map conditions of the forward (relighting) renderer: env_ldr / env_log / env_nrm.

One-shot preprocessing per clip, outside the denoising loop, so this is plain torch on
the device - no HIP kernels.  Mirrors the public functions the reference's forward node imports (nodes.py:25-31):
render_projection_from_panorama (:408-467), tonemap_image_direct (:469-526), latlong_vec (:320-338),
clear_environment_cache, get_cache_stats, with the same tone-mapping arithmetic (rgb2srgb :109-113, reinhard :115-117,
hdr_mapping :119-140), HDR clean-up (:263-286) and lat-long -> cube-map resampling (:161-206).

One deliberate difference: the reference projects the cube map with nvdiffrast (`dr.texture(..., boundary_mode='cube')`,
:448-449), a CUDA-only third-party library that is neither installed here nor portable to ROCm.  `cube_lookup` below is
a from-scratch bilinear cube-map fetch (major-axis face selection, the inverse of the reference's own `cube_to_dir`
face convention, per-face bilinear filtering with clamped edges).  PARITY UNPINNED for that lookup (no nvdiffrast to
compare with); it agrees with sampling the panorama directly up to the cube-map's own resampling error
(tests/test_envmap.py).
"""
import hashlib
import logging
import time
from typing import Dict, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

logger = logging.getLogger(__name__)


class EnvironmentMapCache:
    """LRU cache of processed environment maps (reference :23-66)."""

    def __init__(self, max_size: int = 10):
        self.cache, self.access_times, self.max_size = {}, {}, max_size

    @staticmethod
    def _key(env_hash, resolution, fmt, brightness, flip, rot):
        return f"{env_hash}_{resolution}_{fmt}_{brightness}_{flip}_{rot}"

    def get(self, env_hash, resolution, fmt, brightness, flip, rot):
        k = self._key(env_hash, resolution, fmt, brightness, flip, rot)
        if k in self.cache:
            self.access_times[k] = time.time()
            return self.cache[k]
        return None

    def put(self, env_hash, resolution, fmt, brightness, flip, rot, result):
        k = self._key(env_hash, resolution, fmt, brightness, flip, rot)
        if len(self.cache) >= self.max_size and k not in self.cache:
            oldest = min(self.access_times, key=self.access_times.get)
            del self.cache[oldest], self.access_times[oldest]
        self.cache[k] = result
        self.access_times[k] = time.time()

    def clear(self):
        self.cache.clear()
        self.access_times.clear()


_env_cache = EnvironmentMapCache()


def clear_environment_cache():
    _env_cache.clear()


def get_cache_stats() -> Dict[str, int]:
    return {"cache_size": len(_env_cache.cache), "max_size": _env_cache.max_size}


def compute_tensor_hash(t: torch.Tensor) -> str:
    flat = t.detach().reshape(-1)
    idx = torch.linspace(0, flat.numel() - 1, min(1000, flat.numel())).long()
    sample = flat[idx.to(flat.device)].float().cpu().numpy().tobytes()
    return hashlib.md5(sample + str(tuple(t.shape)).encode()).hexdigest()


# ---------------------------------------------------------------------------------------------- tone mapping
def rgb2srgb_official(rgb: torch.Tensor) -> torch.Tensor:
    return torch.where(rgb <= 0.0031308, 12.92 * rgb, 1.055 * torch.pow(torch.clamp(rgb, 1e-8, 1.0), 1.0 / 2.4) - 0.055)


def reinhard_official(x: torch.Tensor, max_point: float = 16.0) -> torch.Tensor:
    return x / (x + 1.0) * max_point


def hdr_mapping_official(env_hdr: torch.Tensor, log_scale: float = 10000.0) -> Dict[str, torch.Tensor]:
    env_ev0 = rgb2srgb_official(reinhard_official(env_hdr, max_point=16.0).clamp(0, 1))
    env_log = rgb2srgb_official(torch.log1p(env_hdr) / np.log1p(log_scale)).clamp(0, 1)
    return {"env_hdr": env_hdr, "env_ev0": env_ev0, "env_log": env_log}


# ---------------------------------------------------------------------------------------------- geometry
def safe_normalize(v: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    return v / (torch.norm(v, dim=-1, keepdim=True) + eps)


def cube_to_dir(s: int, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """Face s, face coordinates (x, y) in [-1, 1] -> direction (reference :142-155)."""
    one = torch.ones_like(x)
    return [torch.stack([one, -y, -x], -1), torch.stack([-one, -y, x], -1), torch.stack([x, one, y], -1),
            torch.stack([x, -one, -y], -1), torch.stack([x, -y, one], -1), torch.stack([-x, -y, -one], -1)][s]


def latlong_to_cubemap_official(latlong_map: torch.Tensor, res) -> torch.Tensor:
    """(H, W, C) equirectangular -> (6, res0, res1, C) cube map by bilinear grid_sample (reference :161-206)."""
    dev = latlong_map.device
    cube = torch.zeros(6, res[0], res[1], latlong_map.shape[-1], dtype=torch.float32, device=dev)
    gy, gx = torch.meshgrid(torch.linspace(-1.0 + 1.0 / res[0], 1.0 - 1.0 / res[0], res[0], device=dev),
                            torch.linspace(-1.0 + 1.0 / res[1], 1.0 - 1.0 / res[1], res[1], device=dev), indexing="ij")
    src = latlong_map.permute(2, 0, 1).unsqueeze(0).float()
    for s in range(6):
        v = safe_normalize(cube_to_dir(s, gx, gy))
        tu = torch.atan2(v[..., 0:1], -v[..., 2:3]) / (2 * np.pi) + 0.5
        tv = torch.acos(torch.clamp(v[..., 1:2], min=-1, max=1)) / np.pi
        grid = (torch.cat((tu, tv), dim=-1) * 2.0 - 1.0).unsqueeze(0)
        cube[s] = F.grid_sample(src, grid, mode="bilinear", padding_mode="border", align_corners=False)[0].permute(1, 2, 0)
    return cube


def cube_lookup(cubemap: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """Bilinear cube-map fetch: cubemap (6, R, R, C), dirs (..., 3) -> (..., C).  Inverse of cube_to_dir's convention."""
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    ax, ay, az = x.abs(), y.abs(), z.abs()
    is_x = (ax >= ay) & (ax >= az)
    is_y = (~is_x) & (ay >= az)
    ma = torch.where(is_x, ax, torch.where(is_y, ay, az)).clamp_min(1e-12)
    face = torch.where(is_x, torch.where(x > 0, 0, 1), torch.where(is_y, torch.where(y > 0, 2, 3), torch.where(z > 0, 4, 5)))
    # face coordinates (fx, fy) such that cube_to_dir(face, fx, fy) is parallel to dirs
    fx = torch.where(is_x, torch.where(x > 0, -z, z), torch.where(is_y, x, torch.where(z > 0, x, -x))) / ma
    fy = torch.where(is_x, -y, torch.where(is_y, torch.where(y > 0, z, -z), -y)) / ma
    R = cubemap.shape[1]
    out = torch.zeros(dirs.shape[:-1] + (cubemap.shape[-1],), dtype=cubemap.dtype, device=cubemap.device)
    grid = torch.stack([fx, fy], -1)
    for s in range(6):
        m = face == s
        if m.any():
            g = grid[m].view(1, 1, -1, 2)
            tex = cubemap[s].permute(2, 0, 1).unsqueeze(0)
            out[m] = F.grid_sample(tex, g, mode="bilinear", padding_mode="border", align_corners=False)[0, :, 0].t()
    return out


def latlong_vec(res: Tuple[int, int] = None, device="cuda", resolution: Tuple[int, int] = None) -> torch.Tensor:
    """(H, W, 3) direction of every lat-long pixel (reference :320-338); `resolution=` is the keyword the node uses."""
    H, W = res if res is not None else resolution
    gy, gx = torch.meshgrid(torch.linspace(0.0 + 1.0 / H, 1.0 - 1.0 / H, H, device=device),
                            torch.linspace(-1.0 + 1.0 / W, 1.0 - 1.0 / W, W, device=device), indexing="ij")
    sintheta, costheta = torch.sin(gy * np.pi), torch.cos(gy * np.pi)
    sinphi, cosphi = torch.sin(gx * np.pi), torch.cos(gx * np.pi)
    return torch.stack((sintheta * sinphi, costheta, -sintheta * cosphi), dim=-1)


# ---------------------------------------------------------------------------------------------- loading / clean-up
def process_comfyui_tensor(t: torch.Tensor) -> torch.Tensor:
    if t.ndim == 4:
        if t.shape[1] in (3, 4):
            t = t.permute(0, 2, 3, 1)
        t = t[0]
    if t.shape[-1] == 4:
        t = t[..., :3]
    elif t.shape[-1] == 1:
        t = t.repeat(1, 1, 3)
    return t


def load_hdr_file(path: str) -> torch.Tensor:
    import imageio.v3 as iio          # optional dependency of the node pack (requirements.txt:5)
    img = np.asarray(iio.imread(path))
    if img.dtype == np.uint8:
        img = img.astype(np.float32) / 255.0
    elif img.dtype == np.uint16:
        img = img.astype(np.float32) / 65535.0
    else:
        img = img.astype(np.float32)
    if img.ndim == 2:
        img = np.stack([img] * 3, axis=-1)
    elif img.shape[-1] == 4:
        img = img[..., :3]
    return torch.from_numpy(img)


def _load(env_input: Union[str, torch.Tensor]) -> torch.Tensor:
    if isinstance(env_input, str):
        return load_hdr_file(env_input)
    if isinstance(env_input, torch.Tensor):
        return process_comfyui_tensor(env_input)
    raise ValueError(f"Unsupported input type: {type(env_input)}")


def apply_hdr_preprocessing(latlong: torch.Tensor, env_brightness: float, env_flip: bool, env_rot: float, device) -> torch.Tensor:
    latlong = latlong.to(device).float().clone()
    if env_brightness != 1.0:
        latlong = latlong * env_brightness
    latlong = torch.nan_to_num(latlong, nan=0.0, posinf=65504.0, neginf=0.0).clamp(0.0, 65504.0)
    if env_flip:
        latlong = torch.flip(latlong, dims=[1])
    if env_rot != 0:
        latlong = torch.roll(latlong, shifts=int(latlong.shape[1] * env_rot / 360), dims=1)
    return latlong


def _frames(t: torch.Tensor, n: int) -> torch.Tensor:
    return t.unsqueeze(0).expand(n, -1, -1, -1) if n > 1 else t.unsqueeze(0)


def render_projection_from_panorama(env_input, resolution: Tuple[int, int], env_brightness: float = 1.0, env_flip: bool = True,
                                    env_rot: float = 180.0, device="cuda", num_frames: int = 1, use_cache: bool = True,
                                    **kwargs) -> Dict[str, torch.Tensor]:
    """Panorama -> 512^2 cube map -> lat-long projection -> tone maps: {'env_ldr','env_log'} as (T, H, W, 3)."""
    if use_cache:
        h = compute_tensor_hash(env_input) if isinstance(env_input, torch.Tensor) else hashlib.md5(str(env_input).encode()).hexdigest()
        hit = _env_cache.get(h, resolution, "proj", env_brightness, env_flip, env_rot)
        if hit is not None:
            return hit
    H, W = resolution
    latlong = apply_hdr_preprocessing(_load(env_input), env_brightness, env_flip, env_rot, device)
    cubemap = latlong_to_cubemap_official(latlong, [512, 512])
    vec = latlong_vec((H, W), device=device)                 # camera = identity, y-rotation 0 (reference :441-446)
    env_proj = torch.flip(cube_lookup(cubemap, -vec), dims=[0, 1])
    m = hdr_mapping_official(env_proj, log_scale=10000.0)
    result = {"env_ldr": _frames(m["env_ev0"], num_frames), "env_log": _frames(m["env_log"], num_frames)}
    if use_cache:
        _env_cache.put(h, resolution, "proj", env_brightness, env_flip, env_rot, result)
    return result


def tonemap_image_direct(env_input, resolution: Tuple[int, int], device="cuda", num_frames: int = 1, use_cache: bool = True,
                         **kwargs) -> Dict[str, torch.Tensor]:
    """A pre-rendered HDR image (chrome ball) -> resize -> tone maps."""
    if use_cache:
        h = compute_tensor_hash(env_input) if isinstance(env_input, torch.Tensor) else hashlib.md5(str(env_input).encode()).hexdigest()
        hit = _env_cache.get(h, resolution, "ball", 1.0, False, 0.0)
        if hit is not None:
            return hit
    H, W = resolution
    env = _load(env_input).to(device).float()
    if tuple(env.shape[:2]) != (H, W):
        env = F.interpolate(env.permute(2, 0, 1).unsqueeze(0), size=(H, W), mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
    m = hdr_mapping_official(env, log_scale=10000.0)
    result = {"env_ldr": _frames(m["env_ev0"], num_frames), "env_log": _frames(m["env_log"], num_frames)}
    if use_cache:
        _env_cache.put(h, resolution, "ball", 1.0, False, 0.0, result)
    return result


def envmap_conditions(env_map, resolution, num_frames, env_format="proj", env_brightness=1.0, env_flip=False, env_rot=0.0,
                      device="cuda") -> Dict[str, torch.Tensor]:
    """The three forward-renderer conditions as (1, 3, T|1, H, W) tensors in [-1, 1] / unit vectors (reference nodes.py:283-304)."""
    if env_format == "proj":
        d = render_projection_from_panorama(env_map, resolution, env_brightness, env_flip, env_rot, device, num_frames)
    elif env_format == "ball":
        d = tonemap_image_direct(env_map, resolution, device, num_frames)
    else:
        raise ValueError(f"unknown env_format {env_format!r}")
    env_ldr = d["env_ldr"].permute(3, 0, 1, 2).unsqueeze(0) * 2.0 - 1.0
    env_log = d["env_log"].permute(3, 0, 1, 2).unsqueeze(0) * 2.0 - 1.0
    env_nrm = latlong_vec(resolution, device=device).permute(2, 0, 1).unsqueeze(0).unsqueeze(2)
    return {"env_ldr": env_ldr, "env_log": env_log, "env_nrm": env_nrm}
