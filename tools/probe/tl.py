import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scene_generation_amd import ops
which = sys.argv[1]
if which == 'res':
    x = torch.randn(32, 1024, 8, 8, device='cuda', requires_grad=True); w = (torch.randn(1024, 1024, 3, 3, device='cuda') * 0.05).requires_grad_()
    f = lambda: ops.conv2d(x, w, None, stride=1, pad=1, reflect=True)
elif which == 'first':
    x = torch.randn(32, 204, 128, 128, device='cuda', requires_grad=True); w = (torch.randn(64, 204, 7, 7, device='cuda') * 0.05).requires_grad_()
    f = lambda: ops.conv2d(x, w, None, stride=1, pad=3, reflect=True)
print('== fwd'); y = f(); torch.cuda.synchronize()
gy = torch.randn_like(y)
print('== dgrad+wgrad'); y.backward(gy); torch.cuda.synchronize()
