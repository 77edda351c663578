import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scene_generation_amd import _hip
L = _hip.lib()
L.sg_probe_bgemm.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
for (M, K, P, nb) in [(1024, 1024, 512, 16), (1024, 1024, 800, 16), (1024, 512, 1024, 16)]:
    A = torch.randn(nb, M, K, device='cuda'); B = torch.randn(nb * P, K, device='cuda'); C = torch.empty(M, nb * P, device='cuda')
    for tile in (0, 1):
        f = lambda: L.sg_probe_bgemm(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, K, P, nb, tile, torch.cuda.current_stream().cuda_stream)
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        ref = torch.einsum('bmk,bpk->mbp', A, B.view(nb, P, K)).reshape(M, nb * P)
        err = (C - ref).abs().max().item()
        print('M %d K %d P %d nb %d tile %d: %.3f ms %.1f TF  err %.2e' % (M, K, P, nb, tile, ms, 2.0 * M * K * P * nb / ms / 1e9, err))
