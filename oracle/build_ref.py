"""Build the reference's one native file for this path into oracle/_ref/ (test infrastructure only).

The reference's only compiled component on the path is
maua/audiovisual/audioreactive/selfsupervised/features/efficient_quantile/efficient_quantile.cpp
(a libtorch/pybind11 extension).  It is compiled FROM WHERE IT LIES under
/root/reference — never copied — with torch's own extension builder (hipcc is
not involved; it is host C++), outputs going to oracle/_ref/ (git-ignored,
travels to the GPU box with the snapshot like our own .so files).

Used by tests/golden/make_golden.py (to let the reference's Python import its
extension) and by tests/test_oracle_golden.py (oracle/quantile.c vs the real thing,
when the built module is present).
"""
import glob
import importlib.util
import os
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
OUT = HERE / "_ref"
REF_SRC = ("/root/reference/maua/audiovisual/audioreactive/selfsupervised/features/"
           "efficient_quantile/efficient_quantile.cpp")
NAME = "efficient_quantile_ref"


def _find_built():
    hits = sorted(glob.glob(str(OUT / f"{NAME}*.so")))
    return hits[0] if hits else None


def build_reference_quantile(verbose=False):
    """Compile (if the reference is mounted and no build exists yet). Returns the .so path or None."""
    so = _find_built()
    if so:
        return so
    if not os.path.exists(REF_SRC):
        return None
    OUT.mkdir(exist_ok=True)
    from torch.utils import cpp_extension
    cpp_extension.load(name=NAME, sources=[REF_SRC], build_directory=str(OUT), verbose=verbose,
                       extra_cflags=["-O2"], is_python_module=True)
    return _find_built()


def load_reference_quantile():
    """Import the built module (building it first when possible). Returns the module or None."""
    so = build_reference_quantile()
    if so is None:
        return None
    import torch  # noqa: F401  (libtorch symbols must be loaded first)
    spec = importlib.util.spec_from_file_location(NAME, so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules[NAME] = mod
    return mod


if __name__ == "__main__":
    print(build_reference_quantile(verbose=True))
