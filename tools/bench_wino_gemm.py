"""Micro-benchmark of the batched dense GEMM stage of the Winograd convs (sg_batched_gemm_nt, HIP events, GPU box) at the
shapes a form F(m x m, 3 x 3) would give the 1024-channel 8x8 trunk of the generator (generators.py:62-91, batch 32):

    F(2x2,3x3): 16 x [1024 x 1024] x [1024 x 512]   (what the step runs today, 54 launches)
    F(4x4,3x3): 36 x [1024 x 1024] x [1024 x 128]   (2x2 tiles of 4x4 outputs per plane: 1.78x fewer MACs)
    weight gradient of F(4x4,3x3): 36 x [1024 x 128] x [128 x 1024]  (K = 128 tiles only)

and checks each result against torch.bmm.  Round-5 study for VERDICT r4 item 3: profiles/r05_f43_study.md."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scene_generation_amd import _hip

DEV = 'cuda'
PEAK = 157.3


def timeit(fn, n=20, reps=3):
    fn(); fn(); torch.cuda.synchronize()
    best = float('inf')
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


def run(name, nb, M, cols, K, tiles=(0, 1, 2, 3)):
    L = _hip.lib()
    g = torch.Generator(device='cpu').manual_seed(nb * 7 + cols)
    a = (torch.randn(nb, M, K, generator=g) * 0.05).to(DEV)
    b = torch.randn(nb * cols, K, generator=g).to(DEV)
    c = torch.empty(M, nb * cols, device=DEV)
    want = torch.bmm(a.double(), b.view(nb, cols, K).double().transpose(1, 2))          # [nb][M][cols]
    st = torch.cuda.current_stream().cuda_stream
    flops = 2.0 * nb * M * cols * K
    for tile in tiles:
        bm, bn = (128, 128) if tile == 0 else ((64, 128) if tile == 1 else (64, 64))
        tag = ' k16' if tile == 3 else ''
        if M % bm or cols % bn:
            continue
        fn = lambda: _hip.check(L.sg_batched_gemm_nt(a.data_ptr(), b.data_ptr(), c.data_ptr(), nb, M, cols, K, tile, st), 'bgemm')
        fn()
        torch.cuda.synchronize()
        got = c.view(M, nb, cols).permute(1, 0, 2).double()
        err = float((got - want).abs().max() / want.abs().max())
        ms = timeit(fn)
        print('%-34s tile %dx%-3d%s  %4d workgroups  %7.1f us  %6.1f TFLOP/s (%.2f of peak)  rel.err %.1e' % (
            name, bm, bn, tag, (M // bm) * (cols // bn) * nb, 1e3 * ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / PEAK, err))


if __name__ == '__main__':
    torch.cuda.set_device(0)
    run('F(2x2,3x3) fwd/dgrad 16x1024x512', 16, 1024, 512, 1024)
    run('F(4x4,3x3) fwd/dgrad 36x1024x128', 36, 1024, 128, 1024)
    run('F(4x4,3x3) wgrad 36x1024x1024,K128', 36, 1024, 1024, 128)
    run('F(2x2,3x3) wgrad 16x1024x1024,K512', 16, 1024, 1024, 512)
