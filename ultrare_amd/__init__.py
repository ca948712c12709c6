"""ultrare_amd -- MI355X-native engine for UltraRE's SISA hot path.

Python surface (same names and arguments as the reference's modules):
    ultrare_amd.config   InsParam, Instance (runFull / runGroup)
    ultrare_amd.read     readRating, RatingData, loadData, readSparseMat
    ultrare_amd.group    Group
    ultrare_amd.method.scratch / sisa / utils   Scratch, Sisa, MF, baseTest, ot_cluster
Engine (HBM layout, jobs, evaluation):  ultrare_amd.engine
C ABI:  include/ultrare_hip.h  ->  ultrare_amd/libultrare_hip.so  (python -m ultrare_amd.build)
"""
__version__ = '0.1.0'

# (see rng.limit_torch_threads: torch's CPU thread pool must respect the container's CPU quota)
from . import rng as _rng  # noqa: E402

_rng.limit_torch_threads()
