import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scene_generation_amd import ops, _hip
torch.manual_seed(0)
dev = 'cuda'
bad = 0
for (N, Cin, Cout, H, KS, stride, pad, transposed) in itertools.product((3, 8), (8, 16, 32, 64), (8, 16, 64), (8, 16, 32), (3, 4), (1, 2), (1, 2), (False, True)):
    if pad >= KS:
        continue
    x = torch.randn(N, Cin, H, H, device=dev)
    res = []
    for opt in (2, 1):
        _hip.set_option('fixedtap', opt)
        xg = x.clone().requires_grad_()
        if transposed:
            if stride != 2:
                continue
            w = torch.randn(Cin, Cout, KS, KS, device=dev, generator=None) * 0.1
            torch.manual_seed(1); w = torch.randn(Cin, Cout, KS, KS, device=dev) * 0.1
            wg = w.clone().requires_grad_()
            try:
                y = ops.conv_transpose2d(xg, wg, None, stride=2, pad=pad, out_pad=1 if KS == 3 else 0)
            except Exception as e:
                res = None; break
        else:
            torch.manual_seed(1); w = torch.randn(Cout, Cin, KS, KS, device=dev) * 0.1
            wg = w.clone().requires_grad_()
            y = ops.conv2d(xg, wg, None, stride=stride, pad=pad)
        torch.manual_seed(2); gy = torch.randn_like(y)
        y.backward(gy)
        res.append((y.detach().clone(), xg.grad.clone(), wg.grad.clone()))
    if not res or len(res) < 2:
        continue
    for name, a, b in zip(('y', 'gx', 'gw'), res[0], res[1]):
        if not torch.equal(a, b):
            bad += 1
            print('DIFF', (N, Cin, Cout, H, KS, stride, pad, transposed), name, float((a - b).abs().max()), float(a.abs().max()))
_hip.set_option('fixedtap', 1)
print('done, differing tensors:', bad)
