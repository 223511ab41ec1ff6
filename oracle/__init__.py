"""CPU oracle for the DiffusionRenderer denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package
(`diffusionrenderer-comfyui_amd/`) imports from here; only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do, and
there only as the checker / the timed CPU baseline.

Contents
  dit_oracle.py   torch-CPU restatement of CleanGeneralDIT.py + the EDM sampler
                  of model_diffusion_renderer.py + the pipeline post-process.
                  PINNED: bit-exact against the imported reference (see
                  tools/make_goldens.py, tests/golden/*.safetensors and
                  tests/test_oracle_golden.py).
  vae_oracle.py   torch-CPU restatement of the Cosmos CV8x8x8 tokenizer that the
                  reference reaches through diffusers.AutoencoderKLCosmos.
                  PARITY UNPINNED: diffusers is not installed here, the reference
                  holds no tests / fixtures for it.
  ref_import.py   in-container import shim for /root/reference (fixture
                  generation and the oracle==reference check only).
"""
