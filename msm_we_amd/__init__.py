"""msm_we_amd: MI355X-native haMSM construction hot path (stratified k-means assign/train,
flux-matrix accumulation) behind the msm_we ``modelWE`` / ``StratifiedClusters`` API.

The compute lives in ``csrc/`` (hand-written HIP for gfx950) behind the C ABI declared in
``include/hamsm.h``; this package is the ctypes binding plus the host-side mirror of the
reference interface.  There is no CPU fallback: without the built library the ops raise.
"""
__version__ = "0.1.0"
