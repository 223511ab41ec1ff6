"""Tokenizer wrappers of the reference's pretrained_vae.py: latent mean/std normalisation, temporal chunking
(`pixel_chunk_duration`), encoder/decoder micro-batching and the joint image/video dispatch
(pretrained_vae.py:127-151, :348-470, :472-617).  The node pack imports these but never instantiates them
(nodes.py:19); they are kept because the hot path's north star names them, with the same class names, constructor
arguments, asserts and shape arithmetic.

The reference loads TorchScript `encoder.jit` / `decoder.jit` files (pretrained_vae.py:194-220, torch.load with
weights_only=False).  Those artefacts are not shipped and unpickling them executes code, so here the encoder and
decoder are plain callables handed in by the caller - normally the HIP tokenizer's (CleanVAE.HipCosmosTokenizer)
encode/decode - and mean/std come in as tensors.  Only shape / chunk / normalise logic lives in this file.
"""
from typing import Callable, Optional

import torch


class BasePretrainedImageVAE:
    """(enc(x) - mean) / std and dec(z * std + mean) with dtype handling (pretrained_vae.py:95-162)."""

    def __init__(self, name: str, latent_ch: int = 16, is_image: bool = True, is_bf16: bool = True):
        self.latent_ch = latent_ch
        self.name = name
        self.dtype = torch.bfloat16 if is_bf16 else torch.float32
        self.is_image = is_image
        self.encoder: Optional[Callable] = None
        self.decoder: Optional[Callable] = None
        self.latent_mean = None
        self.latent_std = None

    def register_buffer(self, name, tensor, persistent=False):
        setattr(self, name, tensor)

    def set_mean_std(self, latent_mean: torch.Tensor, latent_std: torch.Tensor) -> None:
        """Tensor form of register_mean_std (the reference reads `image_mean_std.pt`, :110-125)."""
        shape = [1, self.latent_ch, 1, 1] if self.is_image else [1, self.latent_ch, 1, 1, 1]
        self.latent_mean = latent_mean.to(self.dtype).reshape(*shape)
        self.latent_std = latent_std.to(self.dtype).reshape(*shape)

    @torch.no_grad()
    def encode(self, state: torch.Tensor) -> torch.Tensor:
        in_dtype = state.dtype
        mean = self.latent_mean.to(in_dtype)
        std = self.latent_std.to(in_dtype)
        enc = self.encoder(state.to(self.dtype))
        if isinstance(enc, tuple):
            assert isinstance(enc[0], torch.Tensor)
            enc = enc[0]
        elif not isinstance(enc, torch.Tensor):
            raise ValueError("Invalid type of encoded state")
        return (enc.to(in_dtype) - mean) / std

    @torch.no_grad()
    def decode(self, latent: torch.Tensor) -> torch.Tensor:
        in_dtype = latent.dtype
        latent = latent * self.latent_std.to(in_dtype) + self.latent_mean.to(in_dtype)
        return self.decoder(latent.to(self.dtype)).to(in_dtype)

    def reset_dtype(self, *args, **kwargs):
        del args, kwargs


class JITVAE(BasePretrainedImageVAE):
    """Name kept for drop-in imports; TorchScript loading is replaced by explicit callables (see module docstring)."""

    def load_encoder(self, vae_dir: str) -> None:
        raise RuntimeError("encoder.jit loading is not supported (unpickling executes code); assign `.encoder` a callable, "
                           "e.g. CleanVAE.HipCosmosTokenizer(...).encode")

    def load_decoder(self, vae_dir: str) -> None:
        raise RuntimeError("decoder.jit loading is not supported (unpickling executes code); assign `.decoder` a callable, "
                           "e.g. CleanVAE.HipCosmosTokenizer(...).decode")


class BasePretrainedVideoTokenizer:
    """Temporal chunking + micro-batching mixin (pretrained_vae.py:320-470).  Must precede the VAE base in the MRO."""

    def __init__(self, pixel_chunk_duration: int = 17, temporal_compress_factor: int = 8, max_enc_batch_size: int = 8,
                 max_dec_batch_size: int = 4):
        self._pixel_chunk_duration = pixel_chunk_duration
        self._temporal_compress_factor = temporal_compress_factor
        self.max_enc_batch_size = max_enc_batch_size
        self.max_dec_batch_size = max_dec_batch_size

    def set_mean_std(self, latent_mean: torch.Tensor, latent_std: torch.Tensor) -> None:
        """[latent_ch, n] tables cut to the latent chunk length (reference reads `mean_std.pt`, :348-365)."""
        lc = self.latent_chunk_duration
        mean = latent_mean.view(self.latent_ch, -1)[:, :lc]
        std = latent_std.view(self.latent_ch, -1)[:, :lc]
        self.latent_mean = mean.to(self.dtype).reshape(1, self.latent_ch, lc, 1, 1)
        self.latent_std = std.to(self.dtype).reshape(1, self.latent_ch, lc, 1, 1)

    def transform_encode_state_shape(self, state: torch.Tensor) -> torch.Tensor:
        B, C, T, H, W = state.shape
        t = self.pixel_chunk_duration
        assert T % t == 0, f"Temporal dimension {T} is not divisible by chunk_length {t}"
        return state.reshape(B, C, T // t, t, H, W).permute(0, 2, 1, 3, 4, 5).reshape(B * (T // t), C, t, H, W)

    def transform_decode_state_shape(self, latent: torch.Tensor) -> torch.Tensor:
        B, C, T, H, W = latent.shape
        t = self.latent_chunk_duration
        assert T % t == 0, f"Temporal dimension {T} is not divisible by chunk_length {t}"
        return latent.reshape(B, C, T // t, t, H, W).permute(0, 2, 1, 3, 4, 5).reshape(B * (T // t), C, t, H, W)

    @staticmethod
    def _merge_chunks(x: torch.Tensor, b: int) -> torch.Tensor:       # '(b n) c t h w -> b c (n t) h w'
        bn, c, t, h, w = x.shape
        n = bn // b
        return x.reshape(b, n, c, t, h, w).permute(0, 2, 1, 3, 4, 5).reshape(b, c, n * t, h, w)

    @torch.no_grad()
    def encode(self, state: torch.Tensor) -> torch.Tensor:
        origin_T = None
        if self._temporal_compress_factor == 1:
            _, _, origin_T, _, _ = state.shape
            state = state.permute(0, 2, 1, 3, 4).reshape(-1, state.shape[1], 1, *state.shape[3:])
        B = state.shape[0]
        state = self.transform_encode_state_shape(state)
        if state.shape[0] > self.max_enc_batch_size:
            parts = [super(BasePretrainedVideoTokenizer, self).encode(state[i:i + self.max_enc_batch_size])
                     for i in range(0, state.shape[0], self.max_enc_batch_size)]
            latent = torch.cat(parts, dim=0)
        else:
            latent = super().encode(state)
        latent = self._merge_chunks(latent, B)
        if origin_T is not None:
            latent = latent.reshape(-1, origin_T, latent.shape[1], *latent.shape[3:]).permute(0, 2, 1, 3, 4)
        return latent

    @torch.no_grad()
    def decode(self, latent: torch.Tensor) -> torch.Tensor:
        origin_T = None
        if self._temporal_compress_factor == 1:
            _, _, origin_T, _, _ = latent.shape
            latent = latent.permute(0, 2, 1, 3, 4).reshape(-1, latent.shape[1], 1, *latent.shape[3:])
        B = latent.shape[0]
        latent = self.transform_decode_state_shape(latent)
        if latent.shape[0] > self.max_dec_batch_size:
            parts = [super(BasePretrainedVideoTokenizer, self).decode(latent[i:i + self.max_dec_batch_size])
                     for i in range(0, latent.shape[0], self.max_dec_batch_size)]
            state = torch.cat(parts, dim=0)
        else:
            state = super().decode(latent)
        assert state.shape[2] == self.pixel_chunk_duration
        state = self._merge_chunks(state, B)
        if origin_T is not None:
            return state.reshape(-1, origin_T, state.shape[1], *state.shape[3:]).permute(0, 2, 1, 3, 4)
        return state

    @property
    def pixel_chunk_duration(self) -> int:
        return self._pixel_chunk_duration

    @property
    def latent_chunk_duration(self) -> int:
        assert (self.pixel_chunk_duration - 1) % self.temporal_compression_factor == 0, (
            f"Pixel chunk duration {self.pixel_chunk_duration} is not of the form k * "
            f"{self.temporal_compression_factor} + 1")
        return (self.pixel_chunk_duration - 1) // self.temporal_compression_factor + 1

    @property
    def temporal_compression_factor(self):
        return self._temporal_compress_factor

    def get_latent_num_frames(self, num_pixel_frames: int) -> int:
        if num_pixel_frames == 1:
            return 1
        assert num_pixel_frames % self.pixel_chunk_duration == 0, (
            f"Temporal dimension {num_pixel_frames} is not divisible by chunk_length {self.pixel_chunk_duration}")
        return num_pixel_frames // self.pixel_chunk_duration * self.latent_chunk_duration

    def get_pixel_num_frames(self, num_latent_frames: int) -> int:
        if num_latent_frames == 1:
            return 1
        assert num_latent_frames % self.latent_chunk_duration == 0, (
            f"Temporal dimension {num_latent_frames} is not divisible by chunk_length {self.latent_chunk_duration}")
        return num_latent_frames // self.latent_chunk_duration * self.pixel_chunk_duration


class VideoJITTokenizer(BasePretrainedVideoTokenizer, JITVAE):
    def __init__(self, name: str, latent_ch: int = 16, is_bf16: bool = True, spatial_compression_factor: int = 16,
                 temporal_compression_factor: int = 8, pixel_chunk_duration: int = 17, max_enc_batch_size: int = 8,
                 max_dec_batch_size: int = 4, spatial_resolution: str = "720"):
        BasePretrainedVideoTokenizer.__init__(self, pixel_chunk_duration, temporal_compression_factor, max_enc_batch_size,
                                              max_dec_batch_size)
        JITVAE.__init__(self, name, latent_ch, False, is_bf16)
        self._spatial_compression_factor = spatial_compression_factor
        self._spatial_resolution = spatial_resolution

    @property
    def spatial_compression_factor(self):
        return self._spatial_compression_factor

    @property
    def spatial_resolution(self) -> str:
        return self._spatial_resolution


class JointImageVideoTokenizer:
    """T == 1 -> image VAE, otherwise the video tokenizer (pretrained_vae.py:514-592)."""

    def __init__(self, image_vae, video_vae, name: str, latent_ch: int = 16, squeeze_for_image: bool = True):
        self.latent_ch = latent_ch
        self.name = name
        self.image_vae = image_vae
        self.video_vae = video_vae
        self.squeeze_for_image = squeeze_for_image

    def encode_image(self, state):
        return self.image_vae.encode(state)

    def decode_image(self, latent):
        return self.image_vae.decode(latent)

    @torch.no_grad()
    def encode(self, state: torch.Tensor) -> torch.Tensor:
        return self.encode_image(state) if state.shape[2] == 1 else self.video_vae.encode(state)

    @torch.no_grad()
    def decode(self, latent: torch.Tensor) -> torch.Tensor:
        return self.decode_image(latent) if latent.shape[2] == 1 else self.video_vae.decode(latent)

    def reset_dtype(self, *args, **kwargs):
        del args, kwargs
        self.video_vae.reset_dtype()

    def get_latent_num_frames(self, num_pixel_frames: int) -> int:
        return 1 if num_pixel_frames == 1 else self.video_vae.get_latent_num_frames(num_pixel_frames)

    def get_pixel_num_frames(self, num_latent_frames: int) -> int:
        return 1 if num_latent_frames == 1 else self.video_vae.get_pixel_num_frames(num_latent_frames)

    @property
    def spatial_compression_factor(self):
        return self.video_vae.spatial_compression_factor

    @property
    def temporal_compression_factor(self):
        return self.video_vae.temporal_compression_factor

    @property
    def spatial_resolution(self) -> str:
        return self.video_vae.spatial_resolution

    @property
    def pixel_chunk_duration(self) -> int:
        return self.video_vae.pixel_chunk_duration

    @property
    def latent_chunk_duration(self) -> int:
        return self.video_vae.latent_chunk_duration


class JointImageVideoSharedJITTokenizer(JointImageVideoTokenizer):
    def __init__(self, image_vae, video_vae, name: str, latent_ch: int = 16):
        super().__init__(image_vae, video_vae, name, latent_ch, squeeze_for_image=False)
        assert isinstance(image_vae, JITVAE)
        assert isinstance(video_vae, VideoJITTokenizer), f"video_vae should be an instance of VideoJITVAE, got {type(video_vae)}"

    def load_weights(self, vae_dir: str):
        raise RuntimeError("TorchScript tokenizer loading is not supported; assign encoder/decoder callables and "
                           "mean/std tensors (see module docstring)")
