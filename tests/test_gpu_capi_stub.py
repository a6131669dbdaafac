"""INTEGRATION.md §2 shows the ctypes stub a ptlflow maintainer would add over include/pfk.h.  This test executes THAT code
block verbatim (cut out of the document) on the GPU and checks the lookup it performs against the oracle, bit for bit — so the
documented binding cannot drift from the ABI."""
import os
import re
import types

import pytest
import torch

from oracle import raft_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_source() -> str:
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc.split("## 2.", 1)[1]
    m = re.search(r"```python\n(# ptlflow/utils/external/pfk\.py.*?)```", sec, flags=re.S)
    assert m, "INTEGRATION.md section 2 lost its stub"
    return m.group(1)


def test_integration_md_ctypes_stub_runs_verbatim(gpu):
    import ptlflow_amd
    src = _stub_source()
    assert 'ctypes.CDLL("libpfk.so")' in src
    # the only edit: where the library lives (a maintainer would have it on the loader path)
    src = src.replace('ctypes.CDLL("libpfk.so")', f'ctypes.CDLL({str(ptlflow_amd.LIBPFK_PATH)!r})')
    mod = types.ModuleType("pfk_stub")
    exec(compile(src, "INTEGRATION.md#2", "exec"), mod.__dict__)
    g = torch.Generator().manual_seed(0)
    B, D, h, w = 2, 64, 18, 30
    f1, f2 = torch.randn(B, D, h, w, generator=g), torch.randn(B, D, h, w, generator=g)
    pyr = O.correlation_pyramid(f1, f2, 4)
    coords = O.coords_grid(B, h, w) + torch.rand(B, 2, h, w, generator=g) * 14 - 7
    want = O.lookup(pyr, coords, 4)
    levels = [p.reshape(B * h * w, p.shape[-2], p.shape[-1]).contiguous().to(gpu) for p in pyr]
    out = torch.zeros(B * h * w, 324, device=gpu)
    mod.corr_lookup(levels, coords.to(gpu).contiguous(), 4, out)
    torch.cuda.synchronize()
    got = out.view(B, h, w, 324).permute(0, 3, 1, 2).cpu()
    assert torch.equal(got, want)
    # error convention: a status code turned into RuntimeError by the stub
    with pytest.raises(RuntimeError):
        mod.corr_lookup(levels, coords.to(gpu).contiguous(), 9, out)
