"""The C-ABI library loads without a GPU, exports every symbol include/pfk.h declares, and validates its
arguments before touching the device (status codes, never exceptions)."""
import ctypes
import os
import re

import pytest

import ptlflow_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not ptlflow_amd.LIBPFK_PATH.exists():
        from ptlflow_amd import _build
        _build.build_all()
    import torch  # noqa: F401  (load torch's HIP runtime first, as the product does)
    return ctypes.CDLL(str(ptlflow_amd.LIBPFK_PATH))


def declared():
    src = open(os.path.join(ROOT, "include", "pfk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pfk_[a-z0-9_]+)\s*\(", src)))


def test_exports_every_declared_symbol(lib):
    names = declared()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), f"libpfk.so does not export {n}"


def test_abi_version_and_status_strings(lib):
    assert lib.pfk_abi_version() == 7
    lib.pfk_status_string.restype = ctypes.c_char_p
    assert lib.pfk_status_string(0) == b"ok"
    assert b"alignment" in lib.pfk_status_string(-2)


def test_argument_validation_without_gpu(lib):
    assert lib.pfk_conv2d_f32(None, None) == -1
    assert lib.pfk_corr_lookup_f32(None, None) == -1
    assert lib.pfk_corr_volume_f32(None, 0, None, 0, None, 1, 1, 1, 1, ctypes.c_float(1.0), None) == -1
    assert lib.pfk_corr_pool2x2_f32(None, None, ctypes.c_int64(1), 2, 2, None) == -1
    assert lib.pfk_conv_ktot(None) == -1


def test_product_path_has_no_fallback():
    """ops must raise when the native library is absent / no GPU: no silent eager path."""
    import torch
    from ptlflow_amd.corr import CorrBlock
    with pytest.raises(RuntimeError):
        CorrBlock(torch.randn(1, 32, 8, 8), torch.randn(1, 32, 8, 8))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "ptlflow_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f"{f} imports the oracle"


def test_product_never_touches_the_staged_reference():
    """oracle/_ref (the reference staged for the GPU box) is test infrastructure like the rest of oracle/."""
    pkg = os.path.join(ROOT, "ptlflow_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle/_ref" not in txt and "stage_ref" not in txt and "ref_loader" not in txt, f"{f} refers to the staged reference"


def test_libraries_carry_the_tree_stamp():
    """Both libraries are stamped with the content hash of csrc/ + include/pfk.h (ptlflow_amd/_build.py); load_native()
    refuses any other stamp, so a stale `.so` travelling with the tree cannot be loaded silently."""
    import torch
    import ptlflow_amd
    from ptlflow_amd import _build
    want = _build.source_hash()
    assert _build.embedded_hash(ptlflow_amd.LIBPFK_PATH) == want
    ptlflow_amd.load_native()
    assert torch.ops.pfk.source_hash() == f"{want}:{want}"
    # a different tree hash must be refused
    real = _build.source_hash
    _build.source_hash = lambda: "0" * 16
    try:
        with pytest.raises(ptlflow_amd.NativeLibraryStale):
            ptlflow_amd._check_stamp()
    finally:
        _build.source_hash = real


def test_stamp_does_not_depend_on_where_the_tree_lies(tmp_path):
    """The GPU box runs a COPY of this tree under a scratch path: the stamp must be the same there (round 4: absolute -I
    paths in the hashed flags made every load on the box fail as stale)."""
    import shutil
    import subprocess
    import sys
    shutil.copytree(os.path.join(ROOT, "ptlflow_amd"), tmp_path / "ptlflow_amd", ignore=shutil.ignore_patterns("_obj", "__pycache__"))
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "include")
    run = subprocess.run([sys.executable, "-c", "import ptlflow_amd, torch; ptlflow_amd.load_native(); print(torch.ops.pfk.source_hash())"],
                         cwd=tmp_path, env={k: v for k, v in os.environ.items() if k != "PYTHONPATH"}, capture_output=True, text=True)
    assert run.returncode == 0, run.stderr[-1500:]
    from ptlflow_amd import _build
    assert run.stdout.strip().endswith(f"{_build.source_hash()}:{_build.source_hash()}")


def test_debug_knobs_are_inert_without_opt_in():
    """pfk_debug_set_* flip process-global kernel selection: PFK_ERR_DISABLED (-5) unless the process has PFK_DEBUG_KNOBS=1."""
    import subprocess
    import sys
    import ptlflow_amd
    code = ("import ctypes,sys; lib=ctypes.CDLL(sys.argv[1]); "
            "print(lib.pfk_debug_set_tile(3), lib.pfk_debug_set_lookup_pix(8), lib.pfk_debug_set_altcorr(1), lib.pfk_debug_set_wgrad(2))")
    env = {k: v for k, v in os.environ.items() if k != "PFK_DEBUG_KNOBS"}
    off = subprocess.run([sys.executable, "-c", code, str(ptlflow_amd.LIBPFK_PATH)], env=env, capture_output=True, text=True)
    assert off.stdout.split() == ["-5"] * 4, off.stdout + off.stderr
    on = subprocess.run([sys.executable, "-c", code, str(ptlflow_amd.LIBPFK_PATH)], env=dict(env, PFK_DEBUG_KNOBS="1"),
                        capture_output=True, text=True)
    assert on.stdout.split() == ["0"] * 4, on.stdout + on.stderr


def test_fastdiv_multipliers_are_exact():
    """The persistent convolution kernel turns a tile index / pixel index into coordinates with host-made multipliers
    (pfk_gemm.h `fastdiv_u32`): exact for every n < 2^31 — edges, powers of two and random pairs."""
    import ctypes
    import random
    import ptlflow_amd
    lib = ctypes.CDLL(str(ptlflow_amd.LIBPFK_PATH))
    f = lib.pfk_debug_fastdiv
    f.restype = ctypes.c_uint
    f.argtypes = [ctypes.c_uint, ctypes.c_uint]
    rng = random.Random(3)
    ds = list(range(1, 300)) + [2 ** k for k in range(1, 31)] + [2 ** k + e for k in range(3, 30) for e in (-1, 1)] + \
         [rng.randint(1, 2 ** 31 - 1) for _ in range(300)] + [7040, 56320, 7332, 2852, 1242, 1248, 156, 47, 62, 46, 110, 115]
    for d in ds:
        ns = [0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, 2 ** 31 - 1, 2 ** 31 - d, 2 ** 30] + [rng.randint(0, 2 ** 31 - 1) for _ in range(40)]
        for n in ns:
            if 0 <= n < 2 ** 31:
                assert f(n, d) == n // d, (n, d)
