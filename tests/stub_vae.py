"""Duck-typed tokenizer used to capture the sampler/pipeline goldens (same arithmetic as tools/make_goldens.StubVAE)."""
import torch


class StubVAE:
    latent_ch = 16
    spatial_compression_factor = 8

    def get_latent_num_frames(self, n):
        return 1 if n == 1 else (n - 1) // 8 + 1

    def encode(self, x):
        B, C, T, H, W = x.shape
        F_ = self.get_latent_num_frames(T)
        xs = x.float()
        frames = [xs[:, :, :1].mean(2, keepdim=True)]
        if F_ > 1:
            frames.append(xs[:, :, 1:].reshape(B, C, F_ - 1, 8, H, W).mean(3))
        z = torch.cat(frames, 2).reshape(B, C, F_, H // 8, 8, W // 8, 8).mean((4, 6))
        z = z.repeat(1, (16 + C - 1) // C, 1, 1, 1)[:, :16]
        scale = torch.linspace(0.5, 2.0, 16, device=x.device).view(1, 16, 1, 1, 1)
        return (z * scale).to(x.dtype)

    def decode(self, z):
        B, C, F_, h, w = z.shape
        v = z[:, :3].float().repeat_interleave(8, 3).repeat_interleave(8, 4)
        if F_ > 1:
            v = torch.cat([v[:, :, :1], v[:, :, 1:].repeat_interleave(8, 2)], 2)
        return v.to(z.dtype)
