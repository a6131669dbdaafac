"""oracle/stage_ref.py: the archive that carries the reference's hot-path files to the GPU box holds exactly the files
/root/reference has (byte for byte), and ref_loader can run the reference's RAFT out of it."""
import hashlib
import json
import os
import subprocess
import sys
import zipfile

import pytest

from oracle import stage_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.reference
@pytest.mark.skipif(not os.path.isdir("/root/reference/ptlflow"), reason="needs the reference tree to compare the archive with")
def test_archive_matches_the_reference_tree():
    arc = stage_ref.stage()
    sums = json.load(open(stage_ref.MANIFEST))["sha256"]
    with zipfile.ZipFile(arc) as z:
        names = sorted(z.namelist())
        assert names == sorted(sums)
        for n in names:
            data = z.read(n)
            assert hashlib.sha256(data).hexdigest() == sums[n]
            assert data == open(os.path.join("/root/reference", n), "rb").read(), n
    for must in ("ptlflow/models/raft/raft.py", "ptlflow/models/raft/corr.py", "ptlflow/models/raft/update.py",
                 "ptlflow/models/gma/gma.py", "ptlflow/models/sea_raft/sea_raft.py", "ptlflow/models/ccmr/ccmr.py",
                 "ptlflow/models/ms_raft_plus/ms_raft_plus.py", "ptlflow/models/base_model/base_model.py",
                 "ptlflow/utils/correlation.py"):
        assert must in sums


@pytest.mark.skipif(not os.path.isfile(stage_ref.ARCHIVE), reason="nothing staged (run __graft_entry__.build() where /root/reference exists)")
def test_reference_runs_from_the_staged_archive():
    """What the GPU box does: no /root/reference, the classes come out of the archive (forced here by the loader's flag)."""
    code = ("import torch; from oracle import ref_loader as R; assert R.REFERENCE_KIND == 'staged', R.REFERENCE_KIND; "
            "m = R.build_raft(iters=2); import sys; f = sys.modules[type(m).__module__].__file__; "
            "assert f.startswith(R.REFERENCE_ROOT) and '/root/reference' not in f, f; "
            "o = m({'images': torch.rand(1, 2, 3, 128, 192)}); assert tuple(o['flows'].shape) == (1, 1, 2, 128, 192); print('ok')")
    run = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, PFK_REFERENCE_FORCE_STAGED="1"),
                         capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and run.stdout.strip().endswith("ok"), run.stderr[-1500:]
