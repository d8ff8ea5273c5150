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
# Opt out: LEOD_DEV_KERNARG=0 in the environment, or -- from code, before anything touches HIP -- ``leod_amd.configure(dev_kernarg=False)``
# (an embedding application that must not have its process environment edited by a library import sets LEOD_DEV_KERNARG=0 and calls
# ``configure`` as it sees fit).
_DEV_KERNARG_SET_HERE = False
if _os.environ.get('LEOD_DEV_KERNARG', '1') != '0' and 'HIP_FORCE_DEV_KERNARG' not in _os.environ:
    _os.environ['HIP_FORCE_DEV_KERNARG'] = '1'
    _DEV_KERNARG_SET_HERE = True


def configure(dev_kernarg=None, plan=None, precision=None):
    """Process-level settings of the library as an API (instead of environment variables):

    dev_kernarg  True / False: kernel-argument blocks in device memory (HIP_FORCE_DEV_KERNARG).  The HIP runtime reads the variable when
                 it initialises, so this only has an effect BEFORE the first HIP call of the process; afterwards it raises.
                 ``False`` also undoes the default this package's import installed.
    plan         True / False: launch plans for ``Module.training_step`` of modules created from now on (LEOD_PLAN).
    precision    'f32' | 'bf16' | '16f': ``ops.set_precision`` (overrides nothing that a later ``Module.setup`` derives from its config).
    Returns the settings in force."""
    global _DEV_KERNARG_SET_HERE
    if dev_kernarg is not None:
        import torch as _th
        if _th.cuda.is_initialized():
            raise RuntimeError('leod_amd.configure(dev_kernarg=...) must run before the first HIP call of the process '
                               '(the runtime reads HIP_FORCE_DEV_KERNARG when it initialises)')
        if dev_kernarg:
            _os.environ['HIP_FORCE_DEV_KERNARG'] = '1'
        else:
            _os.environ.pop('HIP_FORCE_DEV_KERNARG', None)
        _DEV_KERNARG_SET_HERE = False
    if plan is not None:
        _os.environ['LEOD_PLAN'] = '1' if plan else '0'
    if precision is not None:
        from . import ops as _ops
        _ops.set_precision(precision)
    return {'dev_kernarg': _os.environ.get('HIP_FORCE_DEV_KERNARG') == '1', 'plan': _os.environ.get('LEOD_PLAN', '1') == '1'}


__version__ = '0.1.0'
