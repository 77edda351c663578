"""Autograd operators over the C ABI of libsg2im_hip.so (include/sg2im_hip.h).

PyTorch-ROCm is used here as plumbing only: device allocation (torch.empty), the autograd tape and the current HIP stream.
Every forward / backward is one or a few hand-written gfx950 kernel launches through ctypes; there is no eager / CPU fallback --
a CPU tensor raises.

One namespace, six modules (the single 1 741-line ops.py of rounds 1-3, split by operator family in round 4):
  _core    ctypes call helpers, stream handle, workspace, parameter-gradient sinks (GradOut), path switches, layout hints,
           flat-buffer updates (Adam, fill, fold), profiler front end
  conv     Conv2d / ConvTranspose2d / sub-pixel up-conv / Linear (implicit GEMM, Winograd, head, skinny kernels)
  nn       activations, residual add, dropout mask, Instance / BatchNorm, pooling, padding
  graph    CSR build, row gather, bit-exact triple pool, embeddings, concat
  layout   masks_to_layout (dense / deferred / test mode / factored), per-image-weight convs, bilinear crops, VectorPool
  losses   scalar losses, weighted sum, cross-entropy
Everything is re-exported here, so ``ops.conv2d``, ``ops.GradOut``, ``ops._call`` ... keep working.  The path switches
(``ops.WINOGRAD``, ``ops.FACTORED_LAYOUT``, ``ops.UPCONV``, ``ops.HEADCONV``, ``ops.WINOGRAD24``, ``ops.COND_FOLD``) are WRITABLE through this
namespace: assigning ``ops.WINOGRAD = False`` updates the value the operator modules read (``_core.WINOGRAD``).
"""
import sys
import types

from . import _core, graph, losses, layout, conv, nn

_MODULES = (_core, graph, losses, layout, conv, nn)
_FLAGS = ('HEADCONV', 'WINOGRAD', 'WINOGRAD24', 'FACTORED_LAYOUT', 'UPCONV', 'COND_FOLD')

for _m in _MODULES:
    for _k, _v in vars(_m).items():
        if not _k.startswith('__') and not isinstance(_v, types.ModuleType):
            globals()[_k] = _v
del _m, _k, _v


class _OpsNamespace(types.ModuleType):
    def __setattr__(self, name, value):
        if name in _FLAGS:
            setattr(_core, name, value)                   # what the operator modules read
        elif name == '_call':
            for m in _MODULES:                            # tools/step_shapes.py wraps the ABI call: every module must see it
                if '_call' in vars(m):
                    setattr(m, '_call', value)
        super().__setattr__(name, value)


sys.modules[__name__].__class__ = _OpsNamespace
