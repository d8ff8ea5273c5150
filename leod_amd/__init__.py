"""leod_amd -- MI355X-native implementation of the LEOD hot path (RVT recurrent backbone + YOLOX
head/SimOTA/losses + pseudo-label NMS) behind the reference's Python operator API.

Compute lives in ``libleod_hip.so`` (hand-written gfx950 HIP kernels, C ABI in include/leod_hip.h);
this package is the host-side mirror of the reference interface.  No CPU / eager fallback exists.
"""
__version__ = '0.1.0'
