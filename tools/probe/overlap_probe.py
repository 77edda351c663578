"""Does an HBM-bound kernel hide under an MFMA-bound GEMM when it runs on a second stream?  (round 6 probe, GPU box)

The trunk of the generator alternates ~90 us batched Winograd GEMMs (matrix pipe bound, HBM nearly idle) with ~30 us filter /
weight-gradient-output transforms (HBM bound at ~6 TB/s, matrix pipe idle); the filter transform of conv i+1 depends on nothing
conv i computes.  Round 3's stream-per-scale experiment overlapped MFMA kernels with MFMA kernels and gained nothing; this probe
pairs COMPLEMENTARY kernels, with and without stream priorities:

  serial        gemm_i ; mem_i                      on one stream
  fork/join     gemm_i on the main stream || mem_i on a side stream, joined before gemm_(i+1)   (equal priority)
  prio          the same with the main stream at high priority (side-stream workgroups only fill free slots)

gemm = sg_batched_gemm_nt at the F(4x4,3x3) trunk shape (36 x [1024 x 1024] x [1024 x 128], 64x64 tiles);
mem  = the library's F(4x4,3x3) filter transform through sg_conv2d_wino_fwd is not callable alone, so an elementwise pass over a
       buffer of the same traffic (151 MB written + 38 MB read => here: 95 MB read + 95 MB written) stands in for it.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scene_generation_amd import _hip


def main():
    torch.cuda.set_device(0)
    L = _hip.lib()
    nb, M, cols, K = 36, 1024, 128, 1024
    A = (torch.randn(nb, M, K) * 0.05).cuda()
    B = torch.randn(nb * cols, K).cuda()
    C = torch.empty(M, nb * cols, device='cuda')
    buf = torch.randn(95 * 1024 * 1024 // 4, device='cuda')
    out = torch.empty_like(buf)
    n = 20

    def gemm(stream):
        _hip.check(L.sg_batched_gemm_nt(A.data_ptr(), B.data_ptr(), C.data_ptr(), nb, M, cols, K, 3, stream.cuda_stream), 'bgemm')

    def mem():
        torch.mul(buf, 1.0001, out=out)

    def timed(fn, stream):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            fn()
            e1.record(stream)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return 1e3 * best / n

    cur = torch.cuda.current_stream()

    def only_gemm():
        for _ in range(n):
            gemm(cur)

    def only_mem():
        for _ in range(n):
            mem()

    def serial():
        for _ in range(n):
            gemm(cur)
            mem()
    print('gemm alone   %7.1f us' % timed(only_gemm, cur))
    print('mem alone    %7.1f us   (%.0f GB/s)' % ((timed(only_mem, cur),) + (2 * buf.numel() * 4 / (timed(only_mem, cur) * 1e-6) / 1e9,)))
    print('serial       %7.1f us per (gemm + mem)' % timed(serial, cur))
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else (0, -1)
    for label, pm, ps in (('fork/join equal priority', 0, 0), ('fork/join main HIGH, side low', -1, 0)):
        main_s, side = torch.cuda.Stream(priority=pm), torch.cuda.Stream(priority=ps)

        def overlapped():
            with torch.cuda.stream(main_s):
                for _ in range(n):
                    side.wait_stream(main_s)
                    gemm(main_s)
                    with torch.cuda.stream(side):
                        mem()
                    main_s.wait_stream(side)
        main_s.wait_stream(cur)
        print('%-30s %7.1f us per (gemm || mem)' % (label, timed(overlapped, main_s)))
    # the same inside a captured graph (what the trunk segments are replayed from)
    main_s, side = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.stream(main_s):
        with torch.cuda.graph(g, stream=main_s, capture_error_mode='thread_local'):
            for _ in range(n):
                side.wait_stream(main_s)
                gemm(main_s)
                with torch.cuda.stream(side):
                    mem()
                main_s.wait_stream(side)
    print('%-30s %7.1f us per (gemm || mem)' % ('hipGraph replay of fork/join', timed(lambda: g.replay(), torch.cuda.current_stream())))
    gs = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(gs, capture_error_mode='thread_local'):
        s = torch.cuda.current_stream()
        for _ in range(n):
            gemm(s)
            mem()
    print('%-30s %7.1f us per (gemm + mem)' % ('hipGraph replay of serial', timed(lambda: gs.replay(), torch.cuda.current_stream())))


if __name__ == '__main__':
    main()
