#!/bin/bash
# round-4 GPU session L: cell-gather output fold of the adjoint Winograd data gradient, 16-deep k-tiles for short-K dense GEMMs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
B="python bench.py --steps 12 --warmup 4 --no_legs --no_secondary --cpu_baseline off --pmc off"
python - > $O/r4l_fold_identity.txt 2>&1 <<'P'
import torch
from scene_generation_amd import ops, _hip
for (N, C, H) in ((32, 1024, 8), (8, 1024, 16), (4, 128, 8), (2, 256, 16)):
    x = torch.randn(N, C, H, H, device='cuda', requires_grad=True)
    w = (torch.randn(C, C, 3, 3, device='cuda') * 0.02).requires_grad_()
    b = torch.zeros(C, device='cuda')
    outs = []
    for v in (0, 1):
        _hip.set_option('wino_fold_cells', v)
        y = ops.conv2d(x, w, b, stride=1, pad=1, reflect=True)
        g, = torch.autograd.grad(y, x, torch.ones_like(y) * 0.5 + y.detach() * 0.1)
        outs.append(g)
    print((N, C, H), 'bit-identical:', torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max()))
P
cat $O/r4l_fold_identity.txt | grep -v amdgpu
( SG_WINO_FOLD_CELLS=0 timeout 600 $B ) > $O/r4l_fold0.json 2> $O/r4l_fold0.err
( SG_WINO_FOLD_CELLS=1 timeout 600 $B ) > $O/r4l_fold1.json 2> $O/r4l_fold1.err
( SG_WINO_SHORTK=512 timeout 600 $B ) > $O/r4l_shortk512.json 2> $O/r4l_shortk512.err
( SG_WINO_SHORTK=256 timeout 600 $B ) > $O/r4l_shortk256.json 2> $O/r4l_shortk256.err
( SG_WINO_FOLD_CELLS=0 python tools/run_leg.py c4 10; SG_WINO_FOLD_CELLS=1 python tools/run_leg.py c4 10 ) 2>&1 | grep images
python - <<'P'
import json
for n in ('fold0','fold1','shortk512','shortk256'):
    try:
        d=json.loads([l for l in open('gpurun_out/r4l_%s.json'%n) if l.startswith('{')][-1])
        t=d['kernels']['top']
        print(n, round(d['value'],1), round(d['ms_per_step'],3), t.get('wino_transforms'), t.get('wino24_bgemm_t128'), t.get('wino_bgemm_t128'))
    except Exception as e: print(n,'failed',e); print(open('gpurun_out/r4l_%s.err'%n).read()[-1500:])
P
