"""What does a small host -> device copy cost when it sits BETWEEN kernels of one stream?  (round 6, host path)

The step uploads ~11 small tables per iteration (VectorPool plan, segment offsets, factored-layout lists: utils.to_device_async =
pinned staging + hipMemcpyAsync on the launch stream).  Three forms of the same sequence, host far ahead of the GPU, timed with
events: kernels only; a pinned hipMemcpyAsync between every pair of kernels; the sg_stage_copy kernel (reads the pinned buffer
through its device mapping) in the same place."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scene_generation_amd import ops

torch.cuda.set_device(0)
x = torch.randn(16 << 20, device='cuda')            # 64 MB: ~30 us per pass
R = 200
for nbytes in (4096, 65536):
    pin = [torch.zeros(nbytes, dtype=torch.uint8, pin_memory=True) for _ in range(R)]
    dst = [torch.empty(nbytes, dtype=torch.uint8, device='cuda') for _ in range(R)]

    def run(kind):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(R):
            x.mul_(1.0000001)
            if kind == 'memcpy':
                dst[i].copy_(pin[i], non_blocking=True)
            elif kind == 'kernel':
                ops.stage_copy(dst[i], pin[i], nbytes)
        x.mul_(1.0000001)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / R

    for kind in ('none', 'memcpy', 'kernel', 'none', 'memcpy', 'kernel'):
        run(kind)
    res = {k: min(run(k) for _ in range(3)) for k in ('none', 'memcpy', 'kernel')}
    print('%6d B: kernel pair only %.1f us; + hipMemcpyAsync(pinned) %.1f us (+%.1f); + sg_stage_copy %.1f us (+%.1f)' % (
        nbytes, res['none'], res['memcpy'], res['memcpy'] - res['none'], res['kernel'], res['kernel'] - res['none']))
