"""scene_generation_amd -- MI355X-native G+D training path with the operator surface of ashual/scene_generation.

Module-for-module mirror of the reference package (graph, layers, layout, bilinear, generators, discriminators,
losses, model, trainer, utils, args); ``install_as('scene_generation')`` aliases it under the reference's
package name so existing ``from scene_generation.model import Model`` imports resolve to this implementation.
The compute path lives in csrc/libsg2im_hip.so (hand-written gfx950 HIP behind include/sg2im_hip.h); there is no
CPU fallback.
"""
import importlib
import os
import sys

# dmabuf IPC: the host driver of the MI355X boxes supports no legacy IPC handles, and ROCr reads this variable when the runtime
# initialises -- so the default must be in the environment before the first torch.cuda call (parallel.first_contact)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

__version__ = '0.1.0'
_SUBMODULES = ('args', 'utils', 'layers', 'graph', 'layout', 'bilinear', 'generators', 'discriminators', 'losses',
               'model', 'trainer', 'optim', 'parallel', 'synthetic', 'ops', 'pipeline')


def _find_host_package(name):
    """directory of a DIFFERENT package called ``name`` on sys.path / the working directory (the checkout whose train.py is
    being run): everything this package does not provide -- data/, metrics, vis -- keeps coming from there"""
    import os
    own = os.path.realpath(os.path.dirname(os.path.abspath(__file__)))
    for base in [os.getcwd()] + list(sys.path):
        d = os.path.join(base or '.', name)
        if os.path.isfile(os.path.join(d, '__init__.py')) and os.path.realpath(d) != own:
            return os.path.abspath(d)
    return None


def install_as(name='scene_generation', host_package_dir=None):
    """The drop-in switch: register this package in sys.modules under ``name`` so that
    ``from scene_generation.trainer import Trainer`` etc. resolve to the MI355X implementation.

    Only the hot-path modules are replaced (args, utils, layers, graph, layout, bilinear, generators, discriminators,
    losses, model, trainer).  Everything else the reference's scripts import from the package -- ``scene_generation.data.*``,
    ``scene_generation.metrics``, ``scene_generation.vis`` (train.py:10-12) -- is out of this package's scope and keeps
    resolving to the host checkout: its package directory (found on sys.path / the working directory, or given as
    ``host_package_dir``) is appended to the package search path."""
    pkg = sys.modules[__name__]
    host = host_package_dir if host_package_dir is not None else _find_host_package(name)
    if host and host not in pkg.__path__:
        pkg.__path__.append(host)
    for key in [k for k in sys.modules if k == name or k.startswith(name + '.')]:
        del sys.modules[key]                      # a previously imported reference package must not shadow the overlay
    sys.modules[name] = pkg
    for sub in _SUBMODULES:
        sys.modules['%s.%s' % (name, sub)] = importlib.import_module('%s.%s' % (__name__, sub))
    return pkg


def set_legacy_align_corners(on=True):
    """Switch every bilinear operator (masks_to_layout, boxes_to_layout, crop_bbox_batch, test-mode compositing) to the
    ``align_corners=True`` geometry of the PyTorch 1.0 the reference was written for (requirements.txt:8; SURVEY section 0
    item 4) -- use it when loading the authors' released checkpoints.  Default off: the reference as executed by
    torch >= 1.3.  Also settable with SG_LEGACY_ALIGN_CORNERS=1."""
    from . import _hip
    _hip.lib().sg_set_legacy_align_corners(1 if on else 0)
