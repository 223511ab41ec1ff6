"""ComfyUI node API of the renderer, kept verbatim at the surface (reference nodes.py:61-347):
same class names, INPUT_TYPES / RETURN_TYPES / RETURN_NAMES / FUNCTION / CATEGORY, tensor conventions
((B,T,H,W,C) in [0,1]  <->  (B,C,T,H,W) in [-1,1]) and error types.  ComfyUI modules (folder_paths, comfy.*)
are imported lazily inside the methods that need them so the package also loads in tests and benches.
"""
import os

import numpy as np
import torch

from .diffusion_renderer_config import get_inverse_renderer_config
from .diffusion_renderer_pipeline import CleanDiffusionRendererPipeline
from .model_diffusion_renderer import CleanDiffusionRendererModel

GBUFFER_INDEX_MAPPING = {"basecolor": 0, "metallic": 1, "roughness": 2, "normal": 3, "depth": 4}   # nodes.py:35-41
INVERSE_PASSES = ["basecolor", "metallic", "roughness", "normal", "depth"]


def _progress_bar(n):
    try:
        from comfy.utils import ProgressBar
        return ProgressBar(n)
    except ImportError:
        class _NoBar:
            def update(self, *_):
                pass
        return _NoBar()


def standardize_to_5d(image, name="image") -> torch.Tensor:
    """list / (H,W,C) / (B,H,W,C)->T=1 / (B,T,H,W,C) -> (B,T,H,W,C)   (reference nodes.py:156-177, :258-268)."""
    if isinstance(image, list):
        try:
            return torch.stack(image, dim=0)
        except Exception:
            return image[0].unsqueeze(0)
    if isinstance(image, torch.Tensor):
        if image.ndim == 3:
            return image.unsqueeze(0).unsqueeze(0)
        if image.ndim == 4:
            return image.unsqueeze(1)
        if image.ndim == 5:
            return image
        raise ValueError(f"Unsupported tensor dimension for '{name}': {image.ndim}. Expected 3D, 4D, or 5D.")
    raise TypeError(f"Unsupported input type for '{name}': {type(image)}. Expected torch.Tensor or list of Tensors.")


class LoadDiffusionRendererModel:
    @classmethod
    def INPUT_TYPES(s):
        import folder_paths
        return {"required": {"model": (folder_paths.get_filename_list("diffusion_models"),
                                       {"tooltip": "Models are loaded from 'ComfyUI/models/diffusion_models'"})}}

    RETURN_TYPES = ("DIFFUSION_RENDERER_PIPELINE",)
    FUNCTION = "load_pipeline"
    CATEGORY = "Cosmos1"

    def load_pipeline(self, model):
        import folder_paths
        import comfy.model_management as mm
        import comfy.utils
        from .CleanVAE import CleanVAE

        device = mm.get_torch_device()
        dtype = torch.bfloat16
        vae_dir = os.path.join(folder_paths.models_dir, "vae", "Cosmos-1.0-Tokenizer-CV8x8x8", "vae")
        if not os.path.isdir(vae_dir):
            raise FileNotFoundError(f"Image VAE subfolder not found at: {vae_dir}")
        vae_instance = CleanVAE(model_path=vae_dir)
        vae_instance.to(device)
        vae_instance.reset_dtype(dtype)

        checkpoint_path = folder_paths.get_full_path("diffusion_models", model)
        state_dict = comfy.utils.load_torch_file(checkpoint_path, safe_load=True)
        if "model" in state_dict:
            state_dict = state_dict["model"]
        model_instance = CleanDiffusionRendererModel(get_inverse_renderer_config(), device=device)
        model_instance.load_state_dict(state_dict, strict=True)     # repack to the HIP layouts, weights stay on GPU
        del state_dict
        mm.soft_empty_cache()
        pipeline = CleanDiffusionRendererPipeline(
            checkpoint_dir=os.path.dirname(checkpoint_path), checkpoint_name=os.path.basename(checkpoint_path),
            model_type=None, vae_instance=vae_instance, model_instance=model_instance,
            guidance=0.0, num_steps=15, seed=42)
        return (pipeline,)


class Cosmos1InverseRenderer:
    @classmethod
    def INPUT_TYPES(s):
        return {
            "required": {"pipeline": ("DIFFUSION_RENDERER_PIPELINE",), "image": ("IMAGE",)},
            "optional": {"guidance": ("FLOAT", {"default": 0.0, "min": 0.0, "max": 10.0, "step": 0.1}),
                         "seed": ("INT", {"default": 42, "min": 0, "max": 0xffffffffffffffff})},
        }

    RETURN_TYPES = ("IMAGE", "IMAGE", "IMAGE", "IMAGE", "IMAGE")
    RETURN_NAMES = ("base_color", "metallic", "roughness", "normal", "depth")
    FUNCTION = "run_inverse_pass"
    CATEGORY = "Cosmos1"

    def run_inverse_pass(self, pipeline, image, guidance=0.0, seed=42):
        pipeline.set_model_type("inverse")
        pipeline.guidance = guidance
        pipeline.seed = seed
        image_5d = standardize_to_5d(image)
        image_tensor = image_5d.permute(0, 4, 1, 2, 3) * 2.0 - 1.0
        outputs = {}
        pbar = _progress_bar(len(INVERSE_PASSES))

        def to_image(out):
            t = torch.from_numpy(out).float() / 255.0
            b, tt, h, w, c = t.shape
            return t.reshape(b * tt, h, w, c)

        if image_tensor.shape[0] == 1 and getattr(pipeline, "batch_passes", True) and hasattr(pipeline, "generate_video_passes"):
            # the five passes share the clip, the seed and every weight: step them as one batch
            data_batch = {
                "rgb": image_tensor,
                "video": image_tensor,
                "context_index": torch.tensor([[GBUFFER_INDEX_MAPPING[p]] for p in INVERSE_PASSES], dtype=torch.long),
            }
            outs = pipeline.generate_video_passes(data_batch=data_batch, seed=seed,
                                                  normalize_normal=[p == "normal" for p in INVERSE_PASSES])
            for gbuffer_pass, out in zip(INVERSE_PASSES, outs):
                outputs[gbuffer_pass] = to_image(out)
                pbar.update(1)
            return (outputs["basecolor"], outputs["metallic"], outputs["roughness"], outputs["normal"], outputs["depth"])
        for gbuffer_pass in INVERSE_PASSES:
            data_batch = {
                "rgb": image_tensor,
                "video": image_tensor,
                "context_index": torch.full((image_tensor.shape[0], 1), GBUFFER_INDEX_MAPPING[gbuffer_pass], dtype=torch.long),
            }
            out = pipeline.generate_video(data_batch=data_batch, normalize_normal=(gbuffer_pass == "normal"), seed=seed)
            outputs[gbuffer_pass] = to_image(out)
            pbar.update(1)
        return (outputs["basecolor"], outputs["metallic"], outputs["roughness"], outputs["normal"], outputs["depth"])


class Cosmos1ForwardRenderer:
    @classmethod
    def INPUT_TYPES(s):
        return {
            "required": {"pipeline": ("DIFFUSION_RENDERER_PIPELINE",), "depth": ("IMAGE",), "normal": ("IMAGE",),
                         "roughness": ("IMAGE",), "metallic": ("IMAGE",), "base_color": ("IMAGE",), "env_map": ("IMAGE",)},
            "optional": {"guidance": ("FLOAT", {"default": 0.0, "min": 0.0, "max": 2.0, "step": 0.1}),
                         "seed": ("INT", {"default": 42, "min": 0, "max": 0xffffffffffffffff}),
                         "env_format": (["proj", "ball"], {"default": "proj"}),
                         "env_brightness": ("FLOAT", {"default": 1.0, "min": 0.0, "max": 2.0, "step": 0.1}),
                         "env_flip_horizontal": ("BOOLEAN", {"default": False}),
                         "env_rotation": ("FLOAT", {"default": 180.0, "min": 0, "max": 360, "step": 1.0})},
        }

    RETURN_TYPES = ("IMAGE",)
    FUNCTION = "run_forward_pass"
    CATEGORY = "Cosmos1"

    def run_forward_pass(self, pipeline, depth, normal, roughness, metallic, base_color, env_map, guidance=0.0, seed=42,
                         env_format="proj", env_brightness=1.0, env_flip_horizontal=False, env_rotation=0.0):
        pipeline.set_model_type("forward")
        pipeline.guidance = guidance
        pipeline.seed = seed
        key_mapping = {"base_color": "basecolor", "depth": "depth", "normal": "normal", "roughness": "roughness",
                       "metallic": "metallic"}
        inputs = {"depth": depth, "normal": normal, "roughness": roughness, "metallic": metallic, "base_color": base_color}
        data_batch = {}
        for name, t in inputs.items():
            t5 = standardize_to_5d(t, name)
            data_batch[key_mapping[name]] = t5.permute(0, 4, 1, 2, 3) * 2.0 - 1.0
        B, _, T, H, W = data_batch["depth"].shape
        data_batch["video"] = data_batch["depth"]
        # env-map conditions (env_ldr / env_log / env_nrm): one-shot preprocessing outside the denoise loop (reference :283-304)
        from . import preprocess_envmap as pe
        env = pe.envmap_conditions(env_map, (H, W), T, env_format, env_brightness, env_flip_horizontal, env_rotation,
                                   device=getattr(pipeline, "device", "cuda"))
        data_batch["env_ldr"] = env["env_ldr"].expand(B, -1, -1, -1, -1)
        data_batch["env_log"] = env["env_log"].expand(B, -1, -1, -1, -1)
        data_batch["env_nrm"] = env["env_nrm"].expand(B, -1, T, -1, -1)
        out = pipeline.generate_video(data_batch=data_batch, seed=seed)
        return (torch.from_numpy(out).float() / 255.0,)


class LoadHDRImage:
    @classmethod
    def INPUT_TYPES(s):
        return {"required": {"path": ("STRING", {"tooltip": "Path to HDR image (.hdr, .exr)"})}}

    RETURN_TYPES = ("IMAGE",)
    FUNCTION = "load_hdr"
    CATEGORY = "Cosmos1"

    def load_hdr(self, path):
        import imageio
        img = imageio.imread(path, format="HDR-FI")
        if img.ndim == 2:
            img = np.stack([img] * 3, axis=-1)
        elif img.ndim == 3 and img.shape[2] == 1:
            img = np.repeat(img, 3, axis=2)
        return (torch.from_numpy(img).float().unsqueeze(0),)


NODE_CLASS_MAPPINGS = {
    "LoadDiffusionRendererModel": LoadDiffusionRendererModel,
    "Cosmos1InverseRenderer": Cosmos1InverseRenderer,
    "Cosmos1ForwardRenderer": Cosmos1ForwardRenderer,
    "LoadHDRImage": LoadHDRImage,
}

NODE_DISPLAY_NAME_MAPPINGS = {
    "LoadDiffusionRendererModel": "Load Diffusion Renderer Model",
    "Cosmos1InverseRenderer": "Cosmos1 Inverse Renderer",
    "Cosmos1ForwardRenderer": "Cosmos1 Forward Renderer",
    "LoadHDRImage": "Load HDR Image",
}
