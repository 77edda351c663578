"""Stage-by-stage check of the Winograd F(4x4,3x3) conv (csrc/igemm.hip: w43_*) against float64 torch on the CPU: filter transform
U, input transform V, forward result, gradient transform Ytp, the data-gradient GEMM result G (read out of the workspace), gx,
gw.  GPU box only:  python tools/probe/debug_w43.py [N C H]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.nn.functional as F
from scene_generation_amd import ops, _hip
from scene_generation_amd.ops import _core

N, C, H = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (16, 128, 8)
M, W = C, H
dev = 'cuda'
AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, .5, -2, 0], [0, 1, 1, .25, 4, 0], [0, 1, -1, .125, -8, 1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [1 / 3, 1 / 3, 1 / 3], [-1 / 3, 1 / 3, -1 / 3], [-16 / 15, -8 / 15, -4 / 15], [1 / 15, -2 / 15, 4 / 15], [0, 0, 1]], dtype=torch.float64)
BT = torch.tensor([[1, -1.5, -2, 1.5, 1, 0], [0, -1, .5, 2.5, 1, 0], [0, 1, -2.5, .5, 1, 0], [0, -2, -1, 2, 1, 0], [0, .5, -1, -.5, 1, 0],
                   [0, 1, -1.5, -2, 1.5, 1]], dtype=torch.float64)
g = torch.Generator().manual_seed(0)
x = torch.randn(N, C, H, W, generator=g)
w = torch.randn(M, C, 3, 3, generator=g) * 0.05
gy = torch.randn(N, M, H, W, generator=g)
TH = TW = H // 4
P = N * TH * TW


def rel(a, b):
    return float((a.double().cpu() - b).abs().max() / b.abs().max())


# references
xp = F.pad(x.double(), (1, 1, 1, 1), mode='reflect')
patches = torch.stack([xp[:, :, 4 * ti:4 * ti + 6, 4 * tj:4 * tj + 6] for ti in range(TH) for tj in range(TW)], 1)    # N, T, C, 6, 6
V_ref = torch.einsum('ik,ntckl,jl->ijntc', BT, patches, BT).reshape(36, P, C)
U_ref = torch.einsum('ik,mckl,jl->ijmc', G, w.double(), G).reshape(36, M, C)
gt = torch.stack([gy.double()[:, :, 4 * ti:4 * ti + 4, 4 * tj:4 * tj + 4] for ti in range(TH) for tj in range(TW)], 1)  # N, T, M, 4, 4
Y_ref = torch.einsum('ki,ntmkl,lj->ijntm', AT, gt, AT).reshape(36, P, M)
G_ref = torch.einsum('xpk,xkc->pxc', Y_ref, U_ref)                         # [P][36][C]
xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
yr = F.conv2d(F.pad(xr, (1, 1, 1, 1), mode='reflect'), wr)
yr.backward(gy.double())

d = _core._conv_desc(N, C, 0, H, W, M, 3, 1, 1, True, 1, H, W, 0, 0)
L = _hip.lib()
print('ut_floats', L.sg_conv2d_wino_ut_floats(d._ref), 'v_floats', L.sg_conv2d_wino_v_floats(d._ref), 'ytp_floats',
      L.sg_conv2d_wino_ytp_floats(d._ref), 'expected', 36 * M * C, 36 * P * C, 36 * P * M)
wsb = L.sg_conv2d_wino_ws_bytes(d._ref)
ws = torch.zeros(wsb // 4 + 16, device=dev)
xg, wg, gyg = x.to(dev), w.to(dev), gy.to(dev)
U = torch.empty(36 * M * C, device=dev)
V = torch.empty(36 * P * C, device=dev)
Ytp = torch.empty(36 * P * M, device=dev)
y = torch.empty(N, M, H, W, device=dev)
st = torch.cuda.current_stream().cuda_stream
_hip.check(L.sg_conv2d_wino_fwd(d._ref, xg.data_ptr(), wg.data_ptr(), None, y.data_ptr(), 0, 0.0, U.data_ptr(), V.data_ptr(),
                                ws.data_ptr(), wsb, st), 'fwd')
torch.cuda.synchronize()
print('U   rel err', rel(U.view(36, M, C), U_ref))
print('V   rel err', rel(V.view(36, P, C), V_ref))
print('y   rel err', rel(y, yr.detach()))
gx = torch.empty(N, C, H, W, device=dev)
_hip.check(L.sg_conv2d_wino_dgrad(d._ref, gyg.data_ptr(), wg.data_ptr(), gx.data_ptr(), U.data_ptr(), Ytp.data_ptr(), ws.data_ptr(),
                                  wsb, st), 'dgrad')
torch.cuda.synchronize()
print('Ytp rel err', rel(Ytp.view(36, P, M), Y_ref))
goff = 36 * M * C + 36 * P * max(M, C)
Gd = ws[goff:goff + 36 * P * C].view(P, 36, C)
e = (Gd.double().cpu() - G_ref).abs()
print('G   rel err', float(e.max() / G_ref.abs().max()), 'worst (p, xi, c)', np.unravel_index(int(e.argmax()), e.shape))
print('    per xi max err / max:', [round(float(e[:, i].max() / G_ref[:, i].abs().max()), 6) for i in range(36)])
eg = (gx.double().cpu() - xr.grad).abs()
print('gx  rel err', float(eg.max() / xr.grad.abs().max()), 'worst (n, c, h, w)', np.unravel_index(int(eg.argmax()), eg.shape))
print('    per-pixel max err map (max over n, c):')
print(np.array2string(eg.amax(dim=(0, 1)).numpy() / float(xr.grad.abs().max()), precision=5, suppress_small=True, max_line_width=200))
gw = torch.empty(M, C, 3, 3, device=dev)
_hip.check(L.sg_conv2d_wino_wgrad(d._ref, gyg.data_ptr(), xg.data_ptr(), gw.data_ptr(), V.data_ptr(), Ytp.data_ptr(), ws.data_ptr(),
                                  wsb, st), 'wgrad')
torch.cuda.synchronize()
print('gw  rel err (saved operands)', rel(gw, wr.grad))
_hip.check(L.sg_conv2d_wino_wgrad(d._ref, gyg.data_ptr(), xg.data_ptr(), gw.data_ptr(), None, None, ws.data_ptr(), wsb, st), 'wgrad')
torch.cuda.synchronize()
print('gw  rel err (rebuilt operands)', rel(gw, wr.grad))
_hip.check(L.sg_conv2d_wino_dgrad(d._ref, gyg.data_ptr(), wg.data_ptr(), gx.data_ptr(), None, None, ws.data_ptr(), wsb, st), 'dgrad')
torch.cuda.synchronize()
print('gx  rel err (U rebuilt in dgrad)', rel(gx, xr.grad))
