"""scene_generation_amd -- MI355X-native G+D training path with the operator surface of ashual/scene_generation.

Module-for-module mirror of the reference package (graph, layers, layout, bilinear, generators, discriminators,
losses, model, trainer, utils, args); ``install_as('scene_generation')`` aliases it under the reference's
package name so existing ``from scene_generation.model import Model`` imports resolve to this implementation.
The compute path lives in csrc/libsg2im_hip.so (hand-written gfx950 HIP behind include/sg2im_hip.h); there is no
CPU fallback.
"""
import importlib
import sys

__version__ = '0.1.0'
_SUBMODULES = ('args', 'utils', 'layers', 'graph', 'layout', 'bilinear', 'generators', 'discriminators', 'losses',
               'model', 'trainer', 'optim', 'parallel', 'synthetic', 'ops')


def install_as(name='scene_generation'):
    """Register this package (and its submodules) in sys.modules under ``name`` -- the drop-in switch."""
    pkg = sys.modules[__name__]
    sys.modules[name] = pkg
    for sub in _SUBMODULES:
        sys.modules['%s.%s' % (name, sub)] = importlib.import_module('%s.%s' % (__name__, sub))
    return pkg
