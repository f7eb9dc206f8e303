"""dartray_amd -- MI355X-native implementation of DartRay's per-pixel-sample hot path
(SamplerRenderer.render -> PathIntegrator.Li -> BVHAccel.intersect / Triangle.intersect).

Layout: csrc/ holds the HIP kernels and the C ABI (include/dartray_hip.h); core.py mirrors
the reference's plugin interface above that ABI; scenes.py builds BASELINE.json's synthetic
scenes; dist.py shards image tiles over ranks and reduces the film over RCCL.
"""
from . import _abi  # noqa: F401
from .core import *  # noqa: F401,F403
