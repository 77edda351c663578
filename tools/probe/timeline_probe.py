"""Where does a workgroup's life go?  Per-workgroup s_memtime stamps of one GEMM launch (round 6, GPU box).

Needs the -DSG_TIMELINE build of the library (tools/probe/build_timeline.sh -> csrc/libsg2im_hip_tl.so); this script loads it through
SG_LIB_PATH, hands the translation unit of the kernel a stamp buffer, runs ONE layer pass of tools/bench_gemm_classes.py's table and
prints, for the launch with the most workgroups of that pass:

  * the phases of a workgroup (wave 0): init (index arithmetic of the loaders), first tile (first global loads -> LDS), main loop,
    epilogue (stores accepted) -- median / p10 / p90 in shader cycles, split by the k extent of the workgroup (parity classes of a
    stride-2 transposed gather have 1x / 2x / 4x the taps),
  * the dispatch timeline: when workgroups start (deciles of the start time over the launch), how many are alive on average, and
    the same per CU (hardware id),
  * launch span vs the sum of main-loop time.

  python tools/probe/timeline_probe.py --layer Gup4 --pass fwd [--unit igemm_kn1] [--opt name=value]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('SG_LIB_PATH', os.path.join(ROOT, 'scene_generation_amd', 'csrc', 'libsg2im_hip_tl.so'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layer', default='Gup4')
    ap.add_argument('--pass', dest='p', default='fwd')
    ap.add_argument('--unit', default='', help='igemm | igemm_kn0 | igemm_kn1 | igemm_nk (default: all four get the buffer)')
    ap.add_argument('--opt', action='append', default=[])
    ap.add_argument('--cap', type=int, default=1 << 16)
    a = ap.parse_args()
    import numpy as np
    import torch
    from scene_generation_amd import _hip, ops
    import bench_gemm_classes as B
    torch.cuda.set_device(0)
    for o in a.opt:
        k, v = o.split('=')
        _hip.set_option(k, int(v))
    L = _hip.lib()
    buf = torch.zeros(a.cap * 8, dtype=torch.int64, device='cuda')
    units = [a.unit] if a.unit else ['igemm', 'igemm_kn0', 'igemm_kn1', 'igemm_nk']

    def set_buf(on):
        for u in units:
            fn = getattr(L, 'sg_debug_timeline_set_' + u)
            fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint]
            assert fn(buf.data_ptr() if on else None, a.cap if on else 0) == 0

    spec = [l for l in B.LAYERS if l[0] == a.layer][0]
    name, kind, N, Cin, H, W, Cout, KS, st, pad, refl, ups = spec
    g = torch.Generator(device='cpu').manual_seed(1)
    x = torch.randn(N, Cin, H, W, generator=g).cuda().requires_grad_(True)
    if kind == 'conv':
        w = (torch.randn(Cout, Cin, KS, KS, generator=g) * 0.05).cuda().requires_grad_(True)
        b = torch.zeros(Cout, device='cuda')
        fwd = lambda xx, ww: ops.conv2d(xx, ww, b, stride=st, pad=pad, reflect=refl, upsample=ups)
    else:
        w = (torch.randn(Cin, Cout, KS, KS, generator=g) * 0.05).cuda().requires_grad_(True)
        b = torch.zeros(Cout, device='cuda')
        fwd = lambda xx, ww: ops.conv_transpose2d(xx, ww, b, stride=st, pad=pad, out_pad=1)
    y = fwd(x, w)
    gy = torch.randn_like(y)
    yx, yw = fwd(x, w.detach()), fwd(x.detach(), w)
    fns = {'fwd': lambda: fwd(x.detach(), w.detach()), 'dgrad': lambda: torch.autograd.grad(yx, x, gy, retain_graph=True),
           'wgrad': lambda: torch.autograd.grad(yw, w, gy, retain_graph=True)}
    fn = fns[a.p]
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    set_buf(True)
    buf.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    set_buf(False)
    rec = buf.cpu().numpy().astype(np.uint64).reshape(-1, 8)
    live = rec[:, 4] != 0
    rec = rec[live]
    print('%s %s: %d workgroups stamped (the LAST launch that wrote each slot; several launches of one pass overwrite each other), '
          'pass wall %.1f us' % (a.layer, a.p, len(rec), 1e3 * e0.elapsed_time(e1)))
    if not len(rec):
        return
    t = rec[:, :5].astype(np.int64)
    kext = (rec[:, 6] >> np.uint64(32)).astype(np.int64)
    hw = rec[:, 5]
    # s_memtime counters of different XCDs have different bases (seen: spans of 1e12 ticks): times are made relative to the first
    # workgroup entry of the SAME XCD -- the eight XCDs start a launch within a microsecond of each other
    # (HW_REG_XCC_ID read back 0 on every XCD in this build: the groups are found as clusters of the entry stamps instead -- bases
    #  differ by >= 1e9 ticks, a launch lasts < 1e6)
    order = np.argsort(t[:, 0])
    gaps = np.diff(t[order, 0])
    grp = np.zeros(len(order), dtype=np.int64)
    grp[order[1:]] = np.cumsum(gaps > 50_000_000)
    grp[order[0]] = 0
    for gq in set(grp.tolist()):
        sel = grp == gq
        t[sel] -= t[sel, 0].min()
    span = t[:, 4].max()
    # HW_ID (gfx9): wave [3:0], simd [5:4], pipe [7:6], cu [11:8], sh [12], se [15:13]; XCC id in the high word
    cu = ((hw >> np.uint64(8)) & np.uint64(0xF)).astype(np.int64)
    sh = ((hw >> np.uint64(12)) & np.uint64(0x1)).astype(np.int64)
    se = ((hw >> np.uint64(13)) & np.uint64(0x7)).astype(np.int64)
    cuid = ((grp * 8 + se) * 2 + sh) * 16 + cu
    ph = {'init': t[:, 1] - t[:, 0], 'first tile': t[:, 2] - t[:, 1], 'main loop': t[:, 3] - t[:, 2], 'epilogue': t[:, 4] - t[:, 3],
          'life': t[:, 4] - t[:, 0]}
    print('launch span %d ticks (first entry -> last exit).  s_memtime ticks; 100 MHz reference clock if span*10ns ~ wall, else shader '
          'cycles' % span)
    print('| k extent | workgroups | phase | p10 | median | p90 | mean |')
    print('|---|---|---|---|---|---|---|')
    for ke in sorted(set(kext.tolist())):
        sel = kext == ke
        for nm, v in ph.items():
            v = v[sel]
            print('| %d | %d | %s | %d | %d | %d | %.0f |' % (ke, sel.sum(), nm, np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.mean()))
    alive = ph['life'].sum() / max(span, 1)
    inloop = ph['main loop'].sum() / max(span, 1)
    ncu = len(set(cuid.tolist()))
    print('average workgroups alive: %.1f (%.2f per CU over %d CUs seen); in their main loop: %.1f (%.2f per CU)'
          % (alive, alive / ncu, ncu, inloop, inloop / ncu))
    print('start time deciles (fraction of the span): ' + ' '.join('%.2f' % (np.percentile(t[:, 0], q) / span) for q in range(0, 101, 10)))
    print('end   time deciles (fraction of the span): ' + ' '.join('%.2f' % (np.percentile(t[:, 4], q) / span) for q in range(0, 101, 10)))
    per_cu = np.bincount(cuid, minlength=cuid.max() + 1)
    per_cu = per_cu[per_cu > 0]
    print('workgroups per CU: min %d median %d max %d' % (per_cu.min(), np.median(per_cu), per_cu.max()))
    # busy time per CU: union of lifetimes is costly; use last exit - first entry per CU
    cu_span = [t[cuid == c, 4].max() - t[cuid == c, 0].min() for c in set(cuid.tolist())]
    print('per-CU (first entry -> last exit) / launch span: min %.2f median %.2f max %.2f' % (
        min(cu_span) / span, float(np.median(cu_span)) / span, max(cu_span) / span))


if __name__ == '__main__':
    main()
