"""Importable as ``alt_cuda_corr`` (the reference's optional extension name,
ptlflow/utils/external/alt_cuda_corr/setup.py:8) when this repo is on sys.path: same `forward` / `backward`
entry points, served by the gfx950 kernel in libpfk.so."""
from ptlflow_amd.altcorr import backward, forward  # noqa: F401
