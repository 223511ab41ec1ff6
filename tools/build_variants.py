#!/usr/bin/env python3
"""Build A/B variants of libdrn.so for tools/kbench.py --lib (interleaved timing in one process, CDNA guide rule 24).

    python tools/build_variants.py NAME=file.hip:-DFLAG=1,-DOTHER=2 [NAME2=...]  [NAME3=@git:REV:file.hip]

Each variant recompiles ONE source with extra flags (or takes that source from a git revision) and links it with the
objects of the regular build into build/variants/libdrn_NAME.so (git-ignored; travels to the GPU box with the snapshot)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

OUT = os.path.join(ROOT, "build", "variants")
PER_FILE = {"attention.hip": ["-fno-honor-nans", "-fno-slp-vectorize"], "attention16.hip": ["-fno-honor-nans", "-fno-slp-vectorize"]}


def main():
    G.build()
    os.makedirs(OUT, exist_ok=True)
    for spec in sys.argv[1:]:
        name, rest = spec.split("=", 1)
        flags = []
        if rest.startswith("@git:"):
            _, rev, fname = rest.split(":", 2)
            src = os.path.join(OUT, f"{name}_{fname}")
            with open(src, "wb") as f:
                f.write(subprocess.check_output(["git", "show", f"{rev}:diffusionrenderer-comfyui_amd/csrc/{fname}"], cwd=ROOT))
            per = ["-fno-honor-nans"] if fname == "attention.hip" else []
        else:
            fname, _, fl = rest.partition(":")
            src = os.path.join(G.CSRC, fname)
            flags = [f for f in fl.split(",") if f]
            per = PER_FILE.get(fname, [])
        obj = os.path.join(OUT, f"{name}.o")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj,
                               "-I", os.path.join(ROOT, "include"), "-I", G.CSRC, "-Wno-unused-result"] + per + flags)
        others = [os.path.join(G.CSRC, f[:-4] + ".o") for f in sorted(os.listdir(G.CSRC)) if f.endswith(".hip") and f != fname]
        lib = os.path.join(OUT, f"libdrn_{name}.so")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + others)
        print("built", lib)


if __name__ == "__main__":
    main()
