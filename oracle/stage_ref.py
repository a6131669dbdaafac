"""TEST INFRASTRUCTURE — stages the reference's hot-path Python files so they can travel to the GPU box.

/root/reference exists only in the build container; the MI355X box gets a snapshot of THIS tree (untracked files
included).  `stage()` — called by `__graft_entry__.build()` wherever /root/reference is present — packs exactly the
files `oracle/ref_loader.py` imports (the raft / gma / sea_raft / ccmr / ms_raft_plus model packages, `base_model`,
the five `ptlflow/utils` modules they pull in, and the three sibling families seam B1 also engages in: rapidflow, rpknet, skflow) into ONE archive, `oracle/_ref/ptlflow_ref.zip`, next to a
manifest of per-file sha256 sums.  `oracle/_ref/` is git-ignored (no reference source ever enters the history) but
not gpurun-ignored, exactly like the built `.so` files.  On a machine without /root/reference,
`ref_loader.reference_root()` unpacks the archive into a per-content temporary directory and imports the reference's
own, unmodified files from there — so the `reference`-marked GPU tests and bench.py's `dropin` / `cpu_baseline` legs run
`ptlflow.models.raft.raft.RAFT` itself on the MI355X, not a stand-in.

Only tests/, `__graft_entry__`, and bench.py's checker / baseline legs may touch this (tests/test_capi.py forbids
`oracle` anywhere under ptlflow_amd/).
"""
from __future__ import annotations

import hashlib
import json
import os
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
STAGE_DIR = os.path.join(HERE, "_ref")
ARCHIVE = os.path.join(STAGE_DIR, "ptlflow_ref.zip")
MANIFEST = os.path.join(STAGE_DIR, "manifest.json")
SOURCE_ROOT = "/root/reference"

# packages whose *.py files are staged whole (they are small and import each other), and single utility modules
PACKAGES = ["ptlflow/models/raft", "ptlflow/models/gma", "ptlflow/models/sea_raft", "ptlflow/models/ccmr",
            "ptlflow/models/ms_raft_plus", "ptlflow/models/base_model"]
# sibling families whose model modules expose `get_corr_block` with RAFT's own CorrBlock (tests/test_patch_zoo.py): seam B1
# engages inside them, tests/test_gpu_reference_siblings.py runs the real classes on the MI355X.  Staged with their
# sub-packages (local_timm/).
PACKAGES_RECURSIVE = ["ptlflow/models/rapidflow", "ptlflow/models/rpknet", "ptlflow/models/skflow",
                      "ptlflow/models/lcv",      # lcv: RAFT's encoders / update block / loop around a learnable cost volume (B3-B5)
                      "ptlflow/models/llaflow"]  # llaflow: RAFT's / GMA's update block and encoders around its own cost volume (B3-B5)
MODULES = ["ptlflow/utils/correlation.py", "ptlflow/utils/external/raft.py", "ptlflow/utils/flow_metrics.py",
           "ptlflow/utils/registry.py", "ptlflow/utils/utils.py", "ptlflow/utils/timer.py", "LICENSE",
           # the reference's own entry points (ref_loader.load_scripts): `ptlflow.get_model` (ptlflow/__init__.py:65-125) and
           # `model_benchmark.estimate_inference_time` (model_benchmark.py:422-466) run on the accelerated model in
           # tests/test_gpu_reference_scripts.py and bench.py's `model_benchmark_protocol` leg
           "ptlflow/__init__.py", "ptlflow/utils/lightning/ptlflow_cli.py", "model_benchmark.py"]


def _file_list(root: str):
    files = []
    for pkg in PACKAGES:
        d = os.path.join(root, pkg)
        files += [f"{pkg}/{n}" for n in sorted(os.listdir(d)) if n.endswith(".py")]
    for pkg in PACKAGES_RECURSIVE:
        for d, _dirs, names in sorted(os.walk(os.path.join(root, pkg))):
            rel = os.path.relpath(d, root).replace(os.sep, "/")
            files += [f"{rel}/{n}" for n in sorted(names) if n.endswith(".py")]
    files += [m for m in MODULES if os.path.isfile(os.path.join(root, m))]
    return files


def stage(source_root: str = SOURCE_ROOT, force: bool = False) -> str | None:
    """Pack the reference files into oracle/_ref/ptlflow_ref.zip.  Returns the archive path, or None when the
    reference tree is absent (the GPU box: the archive made in the build container is used as it is)."""
    if not os.path.isdir(os.path.join(source_root, "ptlflow", "models", "raft")):
        return ARCHIVE if os.path.isfile(ARCHIVE) else None
    files = _file_list(source_root)
    sums = {}
    for rel in files:
        with open(os.path.join(source_root, rel), "rb") as fh:
            sums[rel] = hashlib.sha256(fh.read()).hexdigest()
    if not force and os.path.isfile(ARCHIVE) and os.path.isfile(MANIFEST):
        try:
            if json.load(open(MANIFEST)).get("sha256") == sums:
                return ARCHIVE
        except (OSError, ValueError):
            pass
    os.makedirs(STAGE_DIR, exist_ok=True)
    tmp = ARCHIVE + ".tmp"
    with zipfile.ZipFile(tmp, "w", zipfile.ZIP_DEFLATED) as z:
        for rel in files:
            info = zipfile.ZipInfo(rel, date_time=(2020, 1, 1, 0, 0, 0))      # reproducible archive bytes
            info.compress_type = zipfile.ZIP_DEFLATED
            with open(os.path.join(source_root, rel), "rb") as fh:
                z.writestr(info, fh.read())
    os.replace(tmp, ARCHIVE)
    with open(MANIFEST, "w") as fh:
        json.dump({"source": "hmorimitsu/ptlflow (Apache-2.0), staged unmodified by oracle/stage_ref.py",
                   "sha256": sums}, fh, indent=1, sort_keys=True)
    return ARCHIVE


def _verified(root: str, sums: dict) -> bool:
    """Every manifest file exists under `root` with the manifest's hash (and the directory is ours, not group/world writable)."""
    try:
        st = os.stat(root)
        if hasattr(os, "getuid") and (st.st_uid != os.getuid() or (st.st_mode & 0o022)):
            return False
        for rel, want in sums.items():
            with open(os.path.join(root, rel), "rb") as fh:
                if hashlib.sha256(fh.read()).hexdigest() != want:
                    return False
        return True
    except OSError:
        return False


def unpack() -> str | None:
    """Extract the staged archive into a directory keyed by the archive's content; returns its root (the directory
    that plays /root/reference), or None when nothing was staged.  Every file is checked against the manifest — also when an
    existing directory is REUSED (the path is predictable: another user of a shared box could have put files there), and the
    directory must be owned by this user with mode 0700.  Concurrent unpackers (torchrun ranks, xdist workers) each extract
    into their own temporary directory and publish it with one atomic rename; the loser of the race verifies the winner's."""
    if not (os.path.isfile(ARCHIVE) and os.path.isfile(MANIFEST)):
        return None
    import shutil
    import tempfile
    sums = json.load(open(MANIFEST))["sha256"]
    key = hashlib.sha256(json.dumps(sums, sort_keys=True).encode()).hexdigest()[:16]
    uid = os.getuid() if hasattr(os, "getuid") else 0
    root = os.path.join(tempfile.gettempdir(), f"pfk_staged_reference_{uid}_{key}")
    if os.path.isdir(root):
        if _verified(root, sums):
            return root
        raise RuntimeError(f"{root} exists but is not a verbatim, privately owned copy of the staged reference; remove it")
    tmp = tempfile.mkdtemp(prefix=f"pfk_staged_reference_{key}_")          # mode 0700, unique per process
    try:
        with zipfile.ZipFile(ARCHIVE) as z:
            for rel, want in sums.items():
                data = z.read(rel)
                if hashlib.sha256(data).hexdigest() != want:
                    raise RuntimeError(f"staged reference file {rel} does not match its manifest hash")
                dst = os.path.join(tmp, rel)
                os.makedirs(os.path.dirname(dst), exist_ok=True)
                with open(dst, "wb") as fh:
                    fh.write(data)
        try:
            os.rename(tmp, root)                 # atomic publish; fails if another process got there first
            tmp = None
        except OSError:
            if not _verified(root, sums):
                raise RuntimeError(f"{root} appeared while unpacking and does not verify against the manifest")
    finally:
        if tmp is not None:
            shutil.rmtree(tmp, ignore_errors=True)
    return root


if __name__ == "__main__":
    print(stage(force=True))
