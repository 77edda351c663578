"""How far may the launching thread run ahead of the GPU?  (round 6, host path)

A long kernel occupies the stream, then M tiny launches are issued behind it and every call is timed on the host: the index at
which a call first blocks is the lead the runtime grants (in launches), whatever resource bounds it (AQL ring, signal pool,
kernel-argument pool, command batch).  Run once per candidate environment setting (the parent re-executes itself):

    python tools/probe/queue_depth_probe.py            # all settings
    python tools/probe/queue_depth_probe.py child      # one measurement in the current environment
"""
import os
import subprocess
import sys
import time


def child():
    import torch
    torch.cuda.set_device(0)
    a = torch.randn(12288, 12288, device='cuda')
    b = torch.randn(12288, 12288, device='cuda')
    x = torch.zeros(256, device='cuda')
    for _ in range(3):
        (a @ b)
        x.add_(1.0)
    torch.cuda.synchronize()
    M = 6000
    for rep in range(2):
        t = [0.0] * (M + 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            c = a @ b                      # ~6 x 30 ms of GPU work in front of the small launches
        t[0] = time.perf_counter()
        for i in range(M):
            x.add_(1.0)
            t[i + 1] = time.perf_counter()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        stalls = [(i, 1e3 * (t[i + 1] - t[i])) for i in range(M) if t[i + 1] - t[i] > 0.5e-3]
        marks = [100, 500, 1000, 2000, 4000, 6000]
        print('rep %d: front issued in %.2f ms; %d small launches issued in %.1f ms (GPU done at %.1f ms); cumulative ms at %s = %s'
              % (rep, 1e3 * (t[0] - t0), M, 1e3 * (t[M] - t[0]), 1e3 * t_all, marks,
                 ['%.1f' % (1e3 * (t[m] - t[0])) for m in marks]))
        print('        first stalls (index, ms): %s; median call %.1f us' %
              (stalls[:6], 1e6 * sorted(t[i + 1] - t[i] for i in range(M))[M // 2]))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child()
        sys.exit(0)
    settings = [{}, {'ROC_SIGNAL_POOL_SIZE': '4096'}, {'ROC_AQL_QUEUE_SIZE': '65536'}, {'DEBUG_CLR_MAX_BATCH_SIZE': '4096'},
                {'HIP_FORCE_DEV_KERNARG': '0'}, {'ROC_SIGNAL_POOL_SIZE': '4096', 'ROC_AQL_QUEUE_SIZE': '65536'},
                {'AMD_DIRECT_DISPATCH': '0'}]
    for s in settings:
        env = dict(os.environ)
        env.update(s)
        print('==== %s' % (s or 'default'), flush=True)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, timeout=300)
        print(r.stdout.decode(errors='replace'), flush=True)
