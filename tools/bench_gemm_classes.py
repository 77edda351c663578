"""Per-class micro-benchmark of the MFMA GEMM kernels at the layer shapes of BASELINE configs[1] (128x128, batch 32: the shapes
bench.py times), GPU box.  VERDICT r5 item 1: the mid-size gather / weight-gradient GEMMs are iterated HERE, one layer and one
pass at a time, instead of on the whole step.

Every case runs ONE layer pass (forward, data gradient or weight gradient of a conv / transposed conv of the step, or one
batched Winograd GEMM) ``--iters`` times with the library's own HIP-event profiler on (an event pair around every launch, on the
launch stream) and prints, per profiler kind the pass touched: launches, us per launch, TFLOP/s on the flops the launch issues,
fraction of the 157.3 TFLOP/s f32 MFMA peak.  The kinds are the names of bench.py's ``kernels.top`` table (igemm_kn0_k4_t64 = forward
gather, 4x4 kernel, 64x64 tile; kn1 = transposed gather; nk = weight gradient; wino43_bgemm_t64; ...).

  python tools/bench_gemm_classes.py                         # every case, default launch plans
  python tools/bench_gemm_classes.py --only D1,D2 --pass dgrad
  python tools/bench_gemm_classes.py --opt tile=0 --opt splits=2      # A/B of a library option (sg_set_option) in this process
  python tools/bench_gemm_classes.py --sweep tile=-1,0,1,3           # the same cases once per value
  python tools/bench_gemm_classes.py --pmc gpurun_out/pmc            # re-runs itself under rocprofv3 with the SQ instruction-mix
                                                                     # counters (two passes) and prints tools/pmc_db_summary tables
"""
import argparse
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PEAK = 157.3

# name, kind, N, Cin, H, W, Cout, KS, stride, pad, reflect, upsample   (kind: conv | convT)
# Layer shapes of one configs[1] step (tools/step_shapes.py): PatchGAN image discriminator at both scales (the fake pass has
# N = 32, the batched real + wrong-texture pass N = 64), generator down / up path, object-side convs at the mean object count
LAYERS = [
    ('D1s0', 'conv', 32, 64, 65, 65, 128, 4, 2, 2, False, 1),       # discriminators.py:215-228, scale 0
    ('D2s0', 'conv', 32, 128, 33, 33, 256, 4, 2, 2, False, 1),
    ('D3s0', 'conv', 32, 256, 17, 17, 512, 4, 1, 2, False, 1),      # Winograd F(2x2,4x4)
    ('D1s0x2', 'conv', 64, 64, 65, 65, 128, 4, 2, 2, False, 1),     # real + wrong-texture as one 2N batch
    ('D2s0x2', 'conv', 64, 128, 33, 33, 256, 4, 2, 2, False, 1),
    ('D1s1', 'conv', 32, 64, 33, 33, 128, 4, 2, 2, False, 1),       # scale 1 (AvgPool(3,2,1) of the input)
    ('D2s1', 'conv', 32, 128, 17, 17, 256, 4, 2, 2, False, 1),
    ('D3s1', 'conv', 32, 256, 9, 9, 512, 4, 1, 2, False, 1),
    ('Gdn1', 'conv', 32, 64, 128, 128, 128, 3, 2, 1, False, 1),     # generators.py:68-71 down path
    ('Gdn2', 'conv', 32, 128, 64, 64, 256, 3, 2, 1, False, 1),
    ('Gdn3', 'conv', 32, 256, 32, 32, 512, 3, 2, 1, False, 1),
    ('Gdn4', 'conv', 32, 512, 16, 16, 1024, 3, 2, 1, False, 1),
    ('Gup1', 'convT', 32, 1024, 8, 8, 512, 3, 2, 1, False, 1),      # generators.py:84-87 up path (ConvTranspose2d k3 s2 p1 op1)
    ('Gup2', 'convT', 32, 512, 16, 16, 256, 3, 2, 1, False, 1),
    ('Gup3', 'convT', 32, 256, 32, 32, 128, 3, 2, 1, False, 1),
    ('Gup4', 'convT', 32, 128, 64, 64, 64, 3, 2, 1, False, 1),
    ('Gres', 'conv', 32, 1024, 8, 8, 1024, 3, 1, 1, True, 1),       # layers.py:251-270 ResnetBlock conv: Winograd F(4x4,3x3)
    ('Mask', 'conv', 204, 192, 16, 16, 192, 3, 1, 1, False, 2),     # generators.py:20-21 mask_net Interpolate(x2)+Conv3x3 (sub-pixel form)
    ('ObjD', 'conv', 204, 128, 8, 8, 256, 3, 1, 1, False, 1),       # object-side 3x3 conv at the mean object count
]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='', help='comma-separated layer names (default: all)')
    ap.add_argument('--pass', dest='passes', default='fwd,dgrad,wgrad')
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--opt', action='append', default=[], help='library option name=value (repeatable)')
    ap.add_argument('--sweep', default='', help='name=v1,v2,...: run the selected cases once per value of one option')
    ap.add_argument('--pmc', default='', help='output directory: re-run under rocprofv3 --pmc (SQ instruction mix)')
    ap.add_argument('--bgemm', action='store_true', help='also the raw batched Winograd GEMMs (sg_batched_gemm_nt) at the trunk shapes')
    return ap.parse_args()


def run_cases(a, tag=''):
    import torch
    from scene_generation_amd import ops
    dev = 'cuda'
    only = set(filter(None, a.only.split(',')))
    passes = a.passes.split(',')
    rows = []
    for name, kind, N, Cin, H, W, Cout, KS, st, pad, refl, ups in LAYERS:
        if only and name not in only:
            continue
        g = torch.Generator(device='cpu').manual_seed(len(name) * 131 + N)
        x = torch.randn(N, Cin, H, W, generator=g).to(dev).requires_grad_(True)
        if kind == 'conv':
            w = (torch.randn(Cout, Cin, KS, KS, generator=g) * 0.05).to(dev).requires_grad_(True)
            b = torch.zeros(Cout, device=dev)
            fwd = lambda xx, ww: ops.conv2d(xx, ww, b, stride=st, pad=pad, reflect=refl, upsample=ups)
        else:
            w = (torch.randn(Cin, Cout, KS, KS, generator=g) * 0.05).to(dev).requires_grad_(True)
            b = torch.zeros(Cout, device=dev)
            fwd = lambda xx, ww: ops.conv_transpose2d(xx, ww, b, stride=st, pad=pad, out_pad=1)
        y = fwd(x, w)
        gy = torch.randn(y.shape, generator=g).to(dev)
        yx = fwd(x, w.detach())                   # graph with a data gradient only
        yw = fwd(x.detach(), w)                   # graph with a weight gradient only
        fns = {'fwd': lambda: fwd(x.detach(), w.detach()),
               'dgrad': lambda: torch.autograd.grad(yx, x, gy, retain_graph=True),
               'wgrad': lambda: torch.autograd.grad(yw, w, gy, retain_graph=True)}
        for p in passes:
            fn = fns[p]
            with torch.no_grad() if p == 'fwd' else torch.enable_grad():
                fn(); fn()
                torch.cuda.synchronize()
                ops.prof_reset()
                ops.prof_enable(True)
                for _ in range(a.iters):
                    fn()
                torch.cuda.synchronize()
                ops.prof_enable(False)
            prof = {k: v for k, v in ops.prof_read().items() if v['launches']}
            tot = sum(v['ms'] for v in prof.values()) / a.iters
            for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms']):
                us = 1e3 * v['ms'] / v['launches']
                tf = v['flops'] / (v['ms'] * 1e-3) / 1e12 if v['flops'] and v['ms'] else 0.0
                gb = v['bytes'] / (v['ms'] * 1e-3) / 1e9 if v['bytes'] and v['ms'] else 0.0
                rows.append((tag, name, p, k, v['launches'] / a.iters, us, tf, gb, 1e3 * tot))
        del x, w, y, yx, yw, gy
    return rows


def run_bgemm(a, tag=''):
    import torch
    from scene_generation_amd import _hip
    L = _hip.lib()
    rows = []
    st = torch.cuda.current_stream().cuda_stream
    for name, nb, M, cols, K in (('bgemm43_fwd 36x[1024x1024]x[1024x128]', 36, 1024, 128, 1024),
                                 ('bgemm43_bal 32x[1024x1024]x[1024x128]', 32, 1024, 128, 1024),
                                 ('bgemm43_2x  72x[1024x1024]x[1024x128]', 72, 1024, 128, 1024),
                                 ('bgemm43_wg  36x[1024x128]x[128x1024]', 36, 1024, 1024, 128)):
        g = torch.Generator(device='cpu').manual_seed(nb)
        A = (torch.randn(nb, M, K, generator=g) * 0.05).cuda()
        B = torch.randn(nb * cols, K, generator=g).cuda()
        C = torch.empty(M, nb * cols, device='cuda')
        for tile in (2, 3):
            fn = lambda: _hip.check(L.sg_batched_gemm_nt(A.data_ptr(), B.data_ptr(), C.data_ptr(), nb, M, cols, K, tile, st), 'bgemm')
            fn(); fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(3):
                e0.record()
                for _ in range(a.iters):
                    fn()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / a.iters)
            fl = 2.0 * nb * M * cols * K
            rows.append((tag, name, 'tile%d' % tile, '%d workgroups' % ((M // 64) * (cols // 64) * nb), 1, 1e3 * best,
                         fl / (best * 1e-3) / 1e12, 0.0, 1e3 * best))
    return rows


def show(rows):
    print('| opt | layer | pass | kernel kind | launches | us/launch | TFLOP/s | frac of f32 MFMA peak | GB/s | pass total us |')
    print('|---|---|---|---|---|---|---|---|---|---|')
    for tag, name, p, k, n, us, tf, gb, tot in rows:
        print('| %s | %s | %s | %s | %.1f | %.1f | %s | %s | %s | %.1f |' % (
            tag, name, p, k, n, us, '%.1f' % tf if tf else '', '%.2f' % (tf / PEAK) if tf else '', '%.0f' % gb if gb else '', tot))
    sys.stdout.flush()


def main():
    a = parse()
    if a.pmc:
        # two counter passes (8 SQ slots each), kernel trace only -- the combination gpurun accepts; summaries by pmc_db_summary
        os.makedirs(a.pmc, exist_ok=True)
        here = os.path.abspath(__file__)
        base = [sys.executable, here, '--iters', '2', '--pass', a.passes] + (['--only', a.only] if a.only else [])
        for o in a.opt:
            base += ['--opt', o]
        # pass 0: instruction mix (tools/pmc_raw_summary.py); pass 1: where the wave cycles go + MFMA pipe busy
        # (tools/pmc_sq_db_summary.py expects exactly this set)
        groups = ['SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT',
                  'SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES '
                  'SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU']
        for i, gsel in enumerate(groups):
            d = os.path.join(os.path.abspath(a.pmc), 'pass%d' % i)
            cmd = ['rocprofv3', '--pmc'] + gsel.split() + ['--kernel-trace', '-d', d, '-o', 'pmc', '--'] + base
            print('+', ' '.join(cmd), flush=True)
            subprocess.call(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'))
        return
    import torch
    from scene_generation_amd import _hip
    torch.cuda.set_device(0)
    for o in a.opt:
        k, v = o.split('=')
        _hip.set_option(k, int(v))
    rows = []
    if a.sweep:
        k, vals = a.sweep.split('=')
        for v in vals.split(','):
            _hip.set_option(k, int(v))
            rows += run_cases(a, '%s=%s' % (k, v))
    else:
        rows += run_cases(a, ','.join(a.opt))
    if a.bgemm:
        rows += run_bgemm(a, ','.join(a.opt))
    show(rows)


if __name__ == '__main__':
    main()
