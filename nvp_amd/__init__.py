"""nvp_amd - MI355X (gfx950) implementation of NVP's per-coordinate encoding path.

Module surface mirrors the reference's importable names:
    nvp_amd.modules.NVP, nvp_amd.tinycudann.Encoding, nvp_amd.sparsegrid.SparseGrid,
    nvp_amd.modulation.{Sine, Siren, SirenNet, Modulator, SirenWrapper}
`nvp_amd.compat.install()` puts them on sys.modules under the reference's own top-level
names (modules, tinycudann, sparsegrid, modulation, loss_functions) so the reference's
train_video.py / eval.py import them unmodified.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
