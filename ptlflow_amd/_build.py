"""In-tree build of the native libraries (gfx950 only).

    python -m ptlflow_amd._build            # build what is stale
    python -m ptlflow_amd._build --force

* ``ptlflow_amd/libpfk.so``      hipcc, the C-ABI kernels of include/pfk.h, no torch dependency.
* ``ptlflow_amd/_pfk_torch.so``  g++ against the torch headers, registers ``torch.ops.pfk.*`` and
  forwards raw pointers to libpfk.so.  Compiled directly (no hipify pass, no JIT cache under
  ~/.cache) so the .so travels with the tree.

hipcc cross-compiles without a GPU; nothing here needs one.
"""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
INCLUDE = ROOT / "include"
OBJ = PKG / "csrc" / "_obj"
ROCM = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))
HIPCC = str(ROCM / "bin" / "hipcc")

LIBPFK = PKG / "libpfk.so"
LIBTORCH_EXT = PKG / "_pfk_torch.so"

# (source, extra flags)
HIP_SOURCES = [
    ("pfk_gemm.hip", []),
    ("pfk_gemm_bf.hip", []),
    ("pfk_corr.hip", ["-ffp-contract=off"]),  # index-exact coordinate arithmetic
    ("pfk_misc.hip", ["-ffp-contract=off"]),
    ("pfk_altcorr.hip", []),
    ("pfk_encoder.hip", []),
    ("pfk_wgrad.hip", []),
    ("pfk_corr_bf16.hip", []),
    ("pfk_bwd.hip", ["-ffp-contract=off"]),   # same coordinate arithmetic as pfk_corr.hip (pfk_lookup.h)
]
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
             f"-I{INCLUDE}", f"-I{CSRC}", "-Wall", "-Wno-unused-function"]


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def _run(cmd) -> None:
    print("[pfk build]", " ".join(map(str, cmd)), flush=True)
    subprocess.run(list(map(str, cmd)), check=True)


def build_libpfk(force: bool = False) -> Path:
    OBJ.mkdir(exist_ok=True)
    headers = [INCLUDE / "pfk.h", CSRC / "pfk_common.h", CSRC / "pfk_gemm.h", CSRC / "pfk_lookup.h"]
    objs = []
    for src, extra in HIP_SOURCES:
        s = CSRC / src
        o = OBJ / (s.stem + ".o")
        if force or _stale(o, [s, *headers]):
            _run([HIPCC, *HIP_FLAGS, *extra, "-c", s, "-o", o])
        objs.append(o)
    if force or _stale(LIBPFK, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIBPFK])
    return LIBPFK


def build_torch_ext(force: bool = False) -> Path:
    import torch
    from torch.utils import cpp_extension as ce

    src = CSRC / "pfk_torch.cpp"
    if not (force or _stale(LIBTORCH_EXT, [src, INCLUDE / "pfk.h", LIBPFK])):
        return LIBTORCH_EXT
    tlib = Path(torch.__file__).parent / "lib"
    inc = [f"-I{p}" for p in ce.include_paths()] + [f"-I{INCLUDE}", f"-I{ROCM / 'include'}"]
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1",
           "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={abi}", "-DTORCH_EXTENSION_NAME=_pfk_torch",
           *inc, str(src), "-o", str(LIBTORCH_EXT),
           f"-L{tlib}", "-lc10", "-ltorch_cpu", "-ltorch", "-lc10_hip", "-ltorch_hip",
           f"-L{PKG}", "-lpfk", f"-L{ROCM / 'lib'}", "-lamdhip64",
           f"-Wl,-rpath,{tlib}", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{ROCM / 'lib'}"]
    _run(cmd)
    return LIBTORCH_EXT


def build_all(force: bool = False) -> None:
    build_libpfk(force)
    build_torch_ext(force)


if __name__ == "__main__":
    build_all("--force" in sys.argv)
