"""Import shim for the read-only reference at /root/reference (test infra only).

Used by tools/make_goldens.py (fixture generation, in this container) and by the
`not gpu` test that checks oracle == reference when the reference is present.
Never used on the GPU box (the reference does not travel) and never by the
product package.

Adaptations (none edits a reference file):
  1. synthetic parent package so the reference's relative imports resolve without
     running its __init__.py (which needs ComfyUI);
  2. F1 patch: PytorchDotProductAttention.forward returns (S,B,H,dh) but
     Attention.cal_attn feeds it to Linear(H*dh, D) (CleanGeneralDIT.py:199-203,
     :299-304) -> wrap it to flatten heads, the "s b n c -> s b (n c)" the
     Transformer-Engine op it imitates performs.
"""
import importlib
import os
import sys
import types

REFERENCE_DIR = "/root/reference"
_PKG = "drn_refpkg"


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_DIR, "CleanGeneralDIT.py"))


def load_reference():
    """Returns a namespace with the reference modules (dit, model, pipeline, config, pretrained_vae)."""
    if not reference_available():
        raise RuntimeError("reference not present at /root/reference")
    if _PKG not in sys.modules:
        pkg = types.ModuleType(_PKG)
        pkg.__path__ = [REFERENCE_DIR]
        sys.modules[_PKG] = pkg
    dont_write = sys.dont_write_bytecode
    sys.dont_write_bytecode = True  # /root/reference is read-only
    try:
        dit = importlib.import_module(_PKG + ".CleanGeneralDIT")
        config = importlib.import_module(_PKG + ".diffusion_renderer_config")
        model = importlib.import_module(_PKG + ".model_diffusion_renderer")
        pipeline = importlib.import_module(_PKG + ".diffusion_renderer_pipeline")
        pretrained_vae = importlib.import_module(_PKG + ".pretrained_vae")
    finally:
        sys.dont_write_bytecode = dont_write

    if not getattr(dit.PytorchDotProductAttention, "_drn_f1_patched", False):
        orig = dit.PytorchDotProductAttention.forward

        def flat_forward(self, q, k, v, **kw):
            out = orig(self, q, k, v, **kw)
            return out.reshape(out.shape[0], out.shape[1], -1)

        dit.PytorchDotProductAttention.forward = flat_forward
        dit.PytorchDotProductAttention._drn_f1_patched = True

    return types.SimpleNamespace(dit=dit, config=config, model=model, pipeline=pipeline,
                                 pretrained_vae=pretrained_vae)
