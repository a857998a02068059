"""Expose nvp_amd's modules under the reference's own top-level import names so that
`experiment_scripts/train_video.py` / `eval.py` (which do `import modules`, and whose
`modules.py` does `import tinycudann, modulation` / `from sparsegrid import SparseGrid`)
pick up the MI355X implementation unmodified.  Call `install()` before those imports
(e.g. from a sitecustomize.py or a two-line launcher); see INTEGRATION.md section A."""
import sys


def install() -> None:
    from . import modulation, modules, sparsegrid, tinycudann
    sys.modules["modules"] = modules
    sys.modules["modulation"] = modulation
    sys.modules["sparsegrid"] = sparsegrid
    sys.modules["tinycudann"] = tinycudann
