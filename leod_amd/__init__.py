"""leod_amd -- MI355X-native implementation of the LEOD hot path (RVT recurrent backbone + YOLOX
head/SimOTA/losses + pseudo-label NMS) behind the reference's Python operator API.

Compute lives in ``libleod_hip.so`` (hand-written gfx950 HIP kernels, C ABI in include/leod_hip.h);
this package is the host-side mirror of the reference interface.  No CPU / eager fallback exists.
"""
import os as _os

# Kernel arguments in device memory (the HIP runtime's HIP_FORCE_DEV_KERNARG): the training step is ~570 short launches whose
# start-to-start distance on the launch stream is bounded by the command processor fetching each launch's argument block from host
# memory -- with the blocks in device memory the RVT-S step is 0.8 ms (5 %) shorter (profiles/r03_zz_dev_kernarg_ab.txt).  Read by the
# runtime when it initialises, i.e. at the first HIP call of the process: set it before that (importing this package first is enough);
# an explicit value in the environment wins.
_os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')

__version__ = '0.1.0'
