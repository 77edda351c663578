"""ctypes access to oracle/libsg_index_oracle.so (plain-C restatement of the pool / layout / crop arithmetic).
TEST INFRASTRUCTURE ONLY -- see the header of sg_index_oracle.c."""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libsg_index_oracle.so')
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(_SO):
            subprocess.check_call(['make', '-s', '-C', _HERE])
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f(t):
    return np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)


def _i(t):
    return np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.int64)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def pool_triples(new_t, edges, O, H, Dout, avg):
    nt, ed = _f(new_t), _i(edges)
    out = np.empty((O, H), dtype=np.float32)
    lib().ora_pool_triples(_p(nt), _p(ed), ed.shape[0], O, H, Dout, int(avg), _p(out))
    return torch.from_numpy(out)


def masks_to_layout(vecs, boxes, masks, obj_to_img, H, W=None, pooling='sum'):
    W = H if W is None else W
    v, b, m, o2i = _f(vecs), _f(boxes), _f(masks.float()), _i(obj_to_img)
    O, D = v.shape
    N = int(o2i.max()) + 1
    out = np.empty((N, D, H, W), dtype=np.float32)
    lib().ora_masks_to_layout(_p(v), _p(b), _p(m), _p(o2i), O, D, m.shape[1], N, H, W, int(pooling == 'avg'), _p(out))
    return torch.from_numpy(out)


def masks_to_layout_test(vecs, boxes, masks, obj_to_img, H, W=None, pooling='sum'):
    """test-mode compositing (layout.py:87-92,157-169)"""
    W = H if W is None else W
    v, b, m, o2i = _f(vecs), _f(boxes), _f(masks.float()), _i(obj_to_img)
    O, D = v.shape
    N = int(o2i.max()) + 1
    out = np.empty((N, D, H, W), dtype=np.float32)
    lib().ora_masks_to_layout_test(_p(v), _p(b), _p(m), _p(o2i), O, D, m.shape[1], N, H, W, int(pooling == 'avg'), _p(out))
    return torch.from_numpy(out)


def crop_bbox_batch(feats, boxes, idx, HH, WW=None):
    WW = HH if WW is None else WW
    f, b, i = _f(feats), _f(boxes), _i(idx)
    N, C, H, W = f.shape
    out = np.empty((b.shape[0], C, HH, WW), dtype=np.float32)
    lib().ora_crop_bbox(_p(f), _p(b), _p(i), C, H, W, b.shape[0], HH, WW, _p(out))
    return torch.from_numpy(out)
