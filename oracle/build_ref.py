"""Build oracle/_ref/detops_ref_C*.so — the reference's own CPU kernels (ROIAlign forward, NMS)
compiled in place from /root/reference (read-only).  TEST INFRASTRUCTURE ONLY.

Usage:  python oracle/build_ref.py            (no-op when /root/reference is absent, e.g. on the
                                               GPU box, where the prebuilt .so travels with the
                                               snapshot)
`-O2` mirrors the flags the reference's setup.py gets from distutils.
"""
import glob
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_CSRC = os.environ.get("DETOPS_REFERENCE_CSRC", "/root/reference/maskrcnn_benchmark/csrc")
OUT = os.path.join(HERE, "_ref")
NAME = "detops_ref_C"


def built_path():
    hits = sorted(glob.glob(os.path.join(OUT, NAME + "*.so")))
    return hits[0] if hits else None


def build(force=False, verbose=False):
    if not os.path.isdir(REF_CSRC):
        return built_path()
    src = os.path.join(HERE, "ref_wrap.cpp")
    so = built_path()
    if so and not force and os.path.getmtime(so) >= os.path.getmtime(src):
        return so
    os.makedirs(OUT, exist_ok=True)
    from torch.utils.cpp_extension import load

    load(
        name=NAME,
        sources=[src],
        extra_include_paths=[REF_CSRC],
        extra_cflags=["-O2", "-w"],
        build_directory=OUT,
        verbose=verbose,
        is_python_module=False,
    )
    # keep only the shared object (objects / ninja files are scratch)
    for f in os.listdir(OUT):
        if not f.endswith(".so"):
            p = os.path.join(OUT, f)
            if os.path.isfile(p):
                os.remove(p)
    return built_path()


def load_ref():
    """Import the prebuilt module (returns None if it was never built)."""
    so = built_path()
    if so is None:
        return None
    import importlib.util

    import torch  # noqa: F401  (libtorch must be loaded first)

    spec = importlib.util.spec_from_file_location(NAME, so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose=True)
    print("reference oracle:", p)
