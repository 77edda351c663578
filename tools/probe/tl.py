import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scene_generation_amd import ops
x = torch.randn(32, 1024, 8, 8, device='cuda'); w = torch.randn(1024, 1024, 3, 3, device='cuda') * 0.05
for i in range(3):
    y = ops.conv2d(x, w, None, stride=1, pad=1, reflect=True)
torch.cuda.synchronize()
