"""pretrained_vae.py wrappers (chunking / normalisation / micro-batching) against a golden captured from the
reference's own classes driven by a fake linear encoder/decoder (tools/make_goldens.py: pretrained_vae_case).
Bit-exact; CPU only."""
import json

import pytest
import torch

from conftest import load_golden


class FakeEnc(torch.nn.Module):          # same fake as in tools/make_goldens.py
    def forward(self, x):
        b, c, t, h, w = x.shape
        f = (t - 1) // 8 + 1
        idx = torch.arange(f) * 8
        z = x[:, :, idx].float().reshape(b, c, f, h // 8, 8, w // 8, 8).mean((4, 6))
        return z.repeat(1, 6, 1, 1, 1)[:, :16].to(x.dtype)


class FakeDec(torch.nn.Module):
    def forward(self, z):
        b, c, f, h, w = z.shape
        t = (f - 1) * 8 + 1
        v = z[:, :3].repeat_interleave(8, 3).repeat_interleave(8, 4)
        idx = (torch.arange(t) + 7) // 8
        return v[:, :, idx]


def _tokenizer(pkg):
    pv = pkg.pretrained_vae if hasattr(pkg, "pretrained_vae") else None
    if pv is None:
        import importlib
        pv = importlib.import_module(pkg.__name__ + ".pretrained_vae")
    sw = pkg.synthetic_weights
    tok = pv.VideoJITTokenizer(name="fake", latent_ch=16, is_bf16=True, spatial_compression_factor=8,
                               temporal_compression_factor=8, pixel_chunk_duration=17, max_enc_batch_size=2,
                               max_dec_batch_size=1)
    tok.encoder, tok.decoder = FakeEnc(), FakeDec()
    mean = sw.synth_tensor("pv.mean", (16, 16), torch.float32, scale=0.1)
    std = 1.0 + sw.synth_tensor("pv.std", (16, 16), torch.float32, scale=0.2)
    tok.set_mean_std(mean, std)
    return pv, tok


def test_chunked_encode_decode_match_reference(pkg):
    gold, meta = load_golden("pretrained_vae_chunking.safetensors")
    pv, tok = _tokenizer(pkg)
    assert tok.latent_chunk_duration == int(meta["latent_chunk"]) == 3
    z = tok.encode(gold["x"])
    assert torch.equal(z, gold["z"])
    assert torch.equal(tok.decode(z), gold["y"])
    frames = json.loads(meta["frames"])
    assert tok.get_latent_num_frames(51) == frames["latent"] and tok.get_pixel_num_frames(9) == frames["pixel"]
    assert tok.get_latent_num_frames(1) == 1 and tok.get_pixel_num_frames(1) == 1


def test_chunk_asserts_and_joint_dispatch(pkg):
    pv, tok = _tokenizer(pkg)
    with pytest.raises(AssertionError):
        tok.encode(torch.zeros(1, 3, 18, 16, 16, dtype=torch.bfloat16))
    with pytest.raises(AssertionError):
        tok.get_latent_num_frames(20)
    img = pv.JITVAE("img", 16, True, True)
    img.encoder, img.decoder = (lambda x: x[:, :1].repeat(1, 16, 1, 1)), (lambda z: z[:, :3])
    img.set_mean_std(torch.zeros(16), torch.ones(16))
    joint = pv.JointImageVideoSharedJITTokenizer(img, tok, "joint")
    assert joint.pixel_chunk_duration == 17 and joint.latent_chunk_duration == 3
    assert joint.get_latent_num_frames(1) == 1 and joint.get_latent_num_frames(34) == 6
    with pytest.raises(RuntimeError):
        joint.load_weights("/nonexistent")
    with pytest.raises(RuntimeError):
        img.load_encoder("/nonexistent")
