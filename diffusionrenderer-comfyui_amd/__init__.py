"""MI355X-native drop-in for the denoising hot path of DiffusionRenderer-ComfyUI.

Loaded by ComfyUI as a custom-node folder (exports NODE_CLASS_MAPPINGS like the reference's
__init__.py:1-3) or, in tests/bench, through __graft_entry__.load_package().
"""
from . import synthetic_weights, diffusion_renderer_config, host_tables, native, native_vae  # noqa: F401
from . import dit_engine, parallel, CleanVAE, pretrained_vae  # noqa: F401

try:  # boundary modules (nodes need nothing from ComfyUI at import time)
    from . import model_diffusion_renderer, diffusion_renderer_pipeline, nodes  # noqa: F401
    from .nodes import NODE_CLASS_MAPPINGS, NODE_DISPLAY_NAME_MAPPINGS  # noqa: F401
except ImportError:  # pragma: no cover - during bring-up only
    NODE_CLASS_MAPPINGS, NODE_DISPLAY_NAME_MAPPINGS = {}, {}

__all__ = ["NODE_CLASS_MAPPINGS", "NODE_DISPLAY_NAME_MAPPINGS"]
