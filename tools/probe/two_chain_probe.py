"""Two INDEPENDENT chains on two streams: does the data-gradient chain of the trunk overlap with the weight-gradient side work?
(round 6 probe, GPU box)

chain A (main): n batched GEMMs at the F(4x4,3x3) trunk shape (the data gradients of consecutive convs; each depends on the previous)
chain B (side): n x (batched GEMM + a 190 MB elementwise pass + a small kernel)  (weight gradient, its output transform, a bias sum)
serial = A and B interleaved on ONE stream (what the step does today); concurrent = A on one stream, B on another, joined once at the
end.  If the hardware interleaves workgroups of the two queues, concurrent < serial by the tails and boundaries of both chains.
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
    C1 = torch.empty(M, nb * cols, device='cuda')
    C2 = torch.empty(M, nb * cols, device='cuda')
    buf = torch.randn(95 * 1024 * 1024 // 4, device='cuda')
    out = torch.empty_like(buf)
    small = torch.zeros(4096, device='cuda')
    n = 18

    def gemm(stream, C):
        _hip.check(L.sg_batched_gemm_nt(A.data_ptr(), B.data_ptr(), C.data_ptr(), nb, M, cols, K, 3, stream.cuda_stream), 'bgemm')

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        cur = torch.cuda.current_stream()
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
            fn()
            e1.record(cur)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return 1e3 * best

    cur = torch.cuda.current_stream()

    def chain_a(s):
        for _ in range(n):
            gemm(s, C1)

    def chain_b(s):
        with torch.cuda.stream(s):
            for _ in range(n):
                gemm(s, C2)
                torch.mul(buf, 1.0001, out=out)
                small.add_(1.0)

    def serial():
        for _ in range(n):
            gemm(cur, C1)
            gemm(cur, C2)
            torch.mul(buf, 1.0001, out=out)
            small.add_(1.0)

    ta = timed(lambda: chain_a(cur))
    tb = timed(lambda: chain_b(cur))
    ts = timed(serial)
    print('chain A alone %8.1f us, chain B alone %8.1f us, serial (interleaved, one stream) %8.1f us' % (ta, tb, ts))
    for label, pa, pb in (('equal priority', 0, 0), ('A high, B low', -1, 0)):
        sa, sb = torch.cuda.Stream(priority=pa), torch.cuda.Stream(priority=pb)

        def concurrent():
            sa.wait_stream(cur)
            sb.wait_stream(cur)
            with torch.cuda.stream(sa):
                chain_a(sa)
            chain_b(sb)
            cur.wait_stream(sa)
            cur.wait_stream(sb)
        tc = timed(concurrent)
        print('two streams, %-16s %8.1f us  (%.1f %% of serial)' % (label, tc, 100 * tc / ts))


main()
