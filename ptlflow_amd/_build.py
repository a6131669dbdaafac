"""In-tree build of the native libraries (gfx950 only).

    python -m ptlflow_amd._build            # build what is stale
    python -m ptlflow_amd._build --force

* ``ptlflow_amd/libpfk.so``      hipcc, the C-ABI kernels of include/pfk.h, no torch dependency.
* ``ptlflow_amd/_pfk_torch.so``  g++ against the torch headers, registers ``torch.ops.pfk.*`` and
  forwards raw pointers to libpfk.so.  Compiled directly (no hipify pass, no JIT cache under
  ~/.cache) so the .so travels with the tree.

hipcc cross-compiles without a GPU; nothing here needs one.

Staleness is decided by CONTENT, not by mtime: `source_hash()` — sha256 over every file of csrc/ (sources and headers),
include/pfk.h and the compiler flags — is compiled into both libraries (`pfk_source_hash()`, `torch.ops.pfk.source_hash()`).
A library whose stamp differs from the tree's hash is rebuilt here and REFUSED by `ptlflow_amd.load_native()`: the `.so`
files travel with the working tree (git-ignored, not gpurun-ignored), and a `git checkout` or a fresh push resets mtimes.
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
    ("pfk_gemm_b16.hip", []),                 # K8b: bf16 activation storage, LDS-DMA for both operands
    ("pfk_corr.hip", ["-ffp-contract=off"]),  # index-exact coordinate arithmetic
    ("pfk_misc.hip", ["-ffp-contract=off"]),
    ("pfk_altcorr.hip", []),
    ("pfk_encoder.hip", []),
    ("pfk_wgrad.hip", []),
    ("pfk_corr_bf16.hip", []),
    ("pfk_bwd.hip", ["-ffp-contract=off"]),   # same coordinate arithmetic as pfk_corr.hip (pfk_lookup.h)
    ("pfk_stamp.hip", []),                    # pfk_source_hash(): compiled with -DPFK_SOURCE_HASH=<tree hash>
]
# PFK_BENCH_VARIANTS=1 (tuning scripts only: scripts/conv_bench.py, scripts/conv_b16_bench.py) compiles the timing ablations and the
# experimental tile / schedule variants of the implicit-GEMM kernels in; the shipped library does not contain them.  The flag is part
# of the source stamp, so a variants build is only accepted by a process that also runs with PFK_BENCH_VARIANTS=1.
BENCH_VARIANTS = os.environ.get("PFK_BENCH_VARIANTS") == "1"
if BENCH_VARIANTS:
    HIP_SOURCES = [(s, f + ["-DPFK_BENCH_VARIANTS"]) if s.startswith("pfk_gemm") else (s, f) for s, f in HIP_SOURCES]
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
             f"-I{INCLUDE}", f"-I{CSRC}", "-Wall", "-Wno-unused-function"]


def _sha(paths, extra: str = "") -> str:
    import hashlib
    h = hashlib.sha256(extra.encode())
    for p in sorted(map(Path, paths), key=lambda q: q.name):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()[:16]


def source_files():
    return sorted([p for p in CSRC.iterdir() if p.suffix in (".hip", ".h", ".cpp")] + [INCLUDE / "pfk.h"])


def _flag_signature(flags) -> str:
    """The flags that change code generation — without the -I paths, which name wherever this tree happens to lie (the GPU box
    runs a copy under a scratch path; the stamp must be the same there)."""
    return " ".join(f for f in flags if not f.startswith("-I"))


def source_hash() -> str:
    """Stamp of the tree: every kernel / binding source, the public header, the flags."""
    return _sha(source_files(), _flag_signature(HIP_FLAGS) + repr(HIP_SOURCES))


def _stale(target: Path, key: str) -> bool:
    """True unless `target` exists and was built for `key` (a content hash kept next to it)."""
    stamp = target.with_name(target.name + ".stamp")
    return not (target.exists() and stamp.exists() and stamp.read_text() == key)


def _mark(target: Path, key: str) -> None:
    target.with_name(target.name + ".stamp").write_text(key)


def embedded_hash(lib: Path, marker: bytes = b"PFK_SOURCE_HASH="):
    """The stamp compiled into a built libpfk.so, or None.  Read from the FILE (the `PFK_SOURCE_HASH=<stamp>` record of
    pfk_stamp.hip), never through dlopen: glibc caches handles by path, so a library relinked by this process would keep
    answering with the stamp of the mapping made before the relink."""
    if not lib.exists():
        return None
    data = lib.read_bytes()
    i = data.find(marker)
    while i >= 0:
        j = data.find(b"\0", i)
        val = data[i + len(marker): j if j >= 0 else i + len(marker) + 64]
        if val and all(32 < c < 127 for c in val):
            return val.decode()
        i = data.find(marker, i + 1)
    return None


def _run(cmd) -> None:
    print("[pfk build]", " ".join(map(str, cmd)), flush=True)
    subprocess.run(list(map(str, cmd)), check=True)


def build_libpfk(force: bool = False) -> Path:
    OBJ.mkdir(exist_ok=True)
    headers = [INCLUDE / "pfk.h", CSRC / "pfk_common.h", CSRC / "pfk_gemm.h", CSRC / "pfk_lookup.h"]
    tree = source_hash()
    objs, jobs = [], []
    for src, extra in HIP_SOURCES:
        s = CSRC / src
        o = OBJ / (s.stem + ".o")
        define = [f'-DPFK_SOURCE_HASH="{tree}"'] if src == "pfk_stamp.hip" else []      # the TU that defines pfk_source_hash()
        key = _sha([s, *headers], _flag_signature(HIP_FLAGS + extra + define))
        if force or _stale(o, key):
            jobs.append(([HIPCC, *HIP_FLAGS, *extra, *define, "-c", s, "-o", o], o, key))
        objs.append(o)
    rebuilt = bool(jobs)
    if jobs:       # independent translation units: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
            for (cmd, o, key), _ in zip(jobs, pool.map(lambda j: _run(j[0]), jobs)):
                _mark(o, key)
    if force or rebuilt or embedded_hash(LIBPFK) != tree:
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIBPFK])
    assert embedded_hash(LIBPFK) == tree, (embedded_hash(LIBPFK), tree)
    return LIBPFK


def build_torch_ext(force: bool = False) -> Path:
    import torch
    from torch.utils import cpp_extension as ce

    src = CSRC / "pfk_torch.cpp"
    tree = source_hash()
    key = tree + ":" + torch.__version__
    if not (force or _stale(LIBTORCH_EXT, key)):
        return LIBTORCH_EXT
    tlib = Path(torch.__file__).parent / "lib"
    inc = [f"-I{p}" for p in ce.include_paths()] + [f"-I{INCLUDE}", f"-I{ROCM / 'include'}"]
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1",
           "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={abi}", "-DTORCH_EXTENSION_NAME=_pfk_torch", f'-DPFK_SOURCE_HASH="{tree}"',
           *inc, str(src), "-o", str(LIBTORCH_EXT),
           f"-L{tlib}", "-lc10", "-ltorch_cpu", "-ltorch", "-lc10_hip", "-ltorch_hip",
           f"-L{PKG}", "-lpfk", f"-L{ROCM / 'lib'}", "-lamdhip64",
           f"-Wl,-rpath,{tlib}", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{ROCM / 'lib'}"]
    _run(cmd)
    _mark(LIBTORCH_EXT, key)
    return LIBTORCH_EXT


def up_to_date() -> bool:
    """Both libraries carry the tree's stamp (read from the FILES): nothing to compile, whatever csrc/_obj holds."""
    tree = source_hash()
    if embedded_hash(LIBPFK) != tree or not LIBTORCH_EXT.exists():
        return False
    return embedded_hash(LIBTORCH_EXT, b"PFK_EXT_SOURCE_HASH=") == tree


def build_all(force: bool = False) -> None:
    # a tree that ships correctly stamped libraries but no csrc/_obj (a wheel, a read-only checkout) needs no compiler and no write
    if not force and up_to_date():
        return
    # several ranks (torchrun, bench.py --gpus N) that find a stale tree must not relink the same .so side by side: one exclusive
    # lock file next to the libraries; whoever gets it second finds the libraries up to date
    import fcntl
    with open(PKG / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or not up_to_date():
                build_libpfk(force)
                build_torch_ext(force)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


if __name__ == "__main__":
    build_all("--force" in sys.argv)
