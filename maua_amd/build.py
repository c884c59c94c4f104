"""Build libmaua_hip.so (gfx950) in-tree with hipcc.  ``python -m maua_amd.build``.

One object per .hip file (compiled in parallel, rebuilt only when the source or a header changed),
linked into maua_amd/csrc/libmaua_hip.so.  The .so is git-ignored but travels to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
INCLUDE = Path(__file__).resolve().parent.parent / "include"
LIB = CSRC / "libmaua_hip.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
         f"-I{INCLUDE}"]


def _stale(target, deps):
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(verbose=False, force=False):
    srcs = sorted(CSRC.glob("*.hip"))
    hdrs = sorted(CSRC.glob("*.h")) + sorted(INCLUDE.glob("*.h"))
    objs = []

    def compile_one(src):
        obj = src.with_suffix(".o")
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC, *FLAGS, "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src.name}:\n{r.stdout}\n{r.stderr}")
            if verbose and r.stderr.strip():
                print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
