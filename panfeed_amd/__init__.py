"""panfeed_amd: panfeed's k-mer extraction + presence/absence pattern hashing hot path on MI355X.

    from panfeed_amd.panfeed import cluster_cutter, pattern_hasher   # the reference's two callables
    from panfeed_amd.engine import Engine                            # batched driver

The compute lives in libpanfeed_hip.so (hand-written HIP for gfx950, C ABI in include/panfeed_hip.h).
"""
__version__ = "0.1.0"
