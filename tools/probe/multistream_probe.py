"""Does running the three PatchGAN scales on three HIP streams pay?  Forward + backward of the five conv layers of each scale
(image-D shapes of BASELINE config 2), one stream vs one stream per scale.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scene_generation_amd import ops

dev = 'cuda'
torch.manual_seed(0)


def make_scale(H):
    specs = [(207, 64, 2), (64, 128, 2), (128, 256, 2), (256, 512, 1)]
    ws = [(torch.randn(co, ci, 4, 4, device=dev) * 0.05).requires_grad_(True) for ci, co, _ in specs]
    bs = [torch.zeros(co, device=dev, requires_grad=True) for _, co, _ in specs]
    x = torch.randn(32, 207, H, H, device=dev, requires_grad=True)
    return specs, ws, bs, x


def run_scale(sc):
    specs, ws, bs, x = sc
    h = x
    for (ci, co, st), w, b in zip(specs, ws, bs):
        h = ops.conv2d(h, w, b, stride=st, pad=2, act=ops.ACT_LEAKY, slope=0.2)
    loss = h.sum()
    loss.backward()


scales = [make_scale(H) for H in (128, 64, 32)]
streams = [torch.cuda.Stream() for _ in scales]


def single():
    for sc in scales:
        run_scale(sc)


def multi():
    main = torch.cuda.current_stream()
    for sc, s in zip(scales, streams):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            run_scale(sc)
    for s in streams:
        main.wait_stream(s)


def timeit(fn, n=10, reps=3):
    fn(); fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e3


for name, fn in (('one stream', single), ('stream per scale', multi), ('one stream', single), ('stream per scale', multi)):
    print('%-18s %.3f ms' % (name, timeit(fn)), flush=True)
for i, sc in enumerate(scales):
    print('scale %d alone %.3f ms' % (i, timeit(lambda: run_scale(sc))), flush=True)
