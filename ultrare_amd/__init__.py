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

# Nothing process-global changes at import.  torch's intra-op pool is capped for the DURATION of a request only
# (rng.torch_threads(): Scratch.train, Sisa.learn / unlearn, Instance.run*) and restored afterwards; URE_TORCH_THREADS=0
# leaves torch alone, =n sets n (conventions).
