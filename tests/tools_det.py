"""closed-form test tensors shared by tools/make_golden.py (det) and the tests"""
import numpy as np

from scene_generation_amd.synthetic import _hash_uniform


def det(shape, salt, scale=1.0, shift=0.0):
    n = int(np.prod(shape))
    return _hash_uniform(n, salt).view(*shape) * 2 * scale + shift
