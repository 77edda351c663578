#!/usr/bin/env python
"""bench.py -- images/sec of one full G+D training step (train.py:190-215 semantics) on MI355X.

  python bench.py --gpus 1 --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1]): COCO-Stuff-shaped synthetic scene graphs, 128x128, <=8 objects/image (+ the
__image__ node), batch 32 PER GPU (weak scaling; configs[2] = 256 over 8 GPUs), reference default widths (183 M-param
generator, 2-scale PatchGAN, object / mask discriminators), fp32, VGG loss off (needs pretrained weights).
A step = Model.forward + train_generator + the three discriminator steps incl. four Adam updates and, for N>1, the
RCCL gradient all-reduces.  Inputs are resident in HBM before the timed region.  One JSON line on rank 0.
"""
import argparse
import json
import os
import random
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32
HBM_PEAK_GBS = 8000.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=6)
    p.add_argument('--warmup', type=int, default=4)
    p.add_argument('--batch_per_gpu', type=int, default=32)
    p.add_argument('--image_size', type=int, default=128)
    p.add_argument('--cpu_baseline', default='auto', choices=['auto', 'off'])
    p.add_argument('--cpu_images', type=int, default=32, help='images per CPU-baseline step (SURVEY 8d: config 2 at N = 32)')
    p.add_argument('--cpu_steps', type=int, default=3)
    p.add_argument('--no_legs', action='store_true', help='skip the secondary config legs (c4: 256x256, c5: dense graphs)')
    p.add_argument('--no_prof', action='store_true')
    p.add_argument('--pmc', default='auto', choices=['auto', 'off'],
                   help="roofline.traffic measured live: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short child run "
                        "of this script, restricted to the dominant kernel + the Adam kernel used for calibration ('off': the "
                        "committed table under profiles/)")
    p.add_argument('--no_secondary', action='store_true', help='skip the secondary passes (default-flags step with the VGG '
                   'loss on; the step with every fast path off)')
    p.add_argument('--vgg', type=float, default=0.0, help='--vgg_features_weight of the HEADLINE pass (SURVEY 8d: 0; the '
                   'reference default 10 is reported as secondary.default_flags_vgg_on)')
    p.add_argument('--no_share_d_forward', action='store_true',
                   help='re-run the mask/image discriminator forwards in the D steps like the reference does')
    return p.parse_args()


def physical_cores():
    try:
        seen = set()
        phys = core = None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                phys = line.split(':')[1].strip()
            elif line.startswith('core id'):
                core = line.split(':')[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or os.cpu_count()
    except OSError:
        return os.cpu_count()


def cpu_throttle_count():
    """``nr_throttled`` of the cgroup this process runs in: how many 100 ms scheduler periods ended with the process frozen because
    it had spent its CPU quota (None without cgroup v2 accounting).  A throttle inside a timed region starves the GPU."""
    try:
        for line in open('/sys/fs/cgroup/cpu.stat'):
            if line.startswith('nr_throttled'):
                return int(line.split()[1])
    except (OSError, ValueError):
        pass
    return None


class ClockSampler(object):
    """sclk / mclk / socket power of one GPU sampled on a background thread while the timed region runs (two
    boxes of the pool differ by ~5 % in kernel time at equal host speed; the line now says at what clocks and power it was
    measured).  Source: the amdgpu hwmon files of the device (freq1_input = sclk, freq2_input = mclk, power1_average / power1_input
    in microwatts) -- plain file reads, ~20 us each, nothing is launched on the GPU; ``rocm-smi --json`` as a (slow, ~100 ms per
    sample) fallback when sysfs is not readable.  Reported clocks are the SMU's view, not an in-kernel cycle count."""

    def __init__(self, index=0, period_s=0.02):
        import glob
        import threading
        self.period, self.samples, self._stop, self._th = period_s, [], threading.Event(), None
        self.src, self._files = None, {}
        cards = sorted(glob.glob('/sys/class/drm/card[0-9]*/device/hwmon/hwmon*'))
        cards = [c for c in cards if os.path.isfile(os.path.join(c, 'freq1_input'))]
        if cards:
            h = cards[min(index, len(cards) - 1)]
            # the HIP device index is not the DRM card index when the box exposes more cards than the process may use: match the
            # PCI address (domain:bus:device) of the HIP device against the card's sysfs path
            try:
                pr = torch.cuda.get_device_properties(index)
                addr = '%04x:%02x:%02x.' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
                hit = [c for c in cards if addr in os.path.realpath(c)]
                if hit:
                    h = hit[0]
            except Exception:
                pass
            for key, names in (('sclk_mhz', ('freq1_input',)), ('mclk_mhz', ('freq2_input',)),
                               ('power_w', ('power1_average', 'power1_input'))):
                for n in names:
                    if os.path.isfile(os.path.join(h, n)):
                        self._files[key] = os.path.join(h, n)
                        break
            self.src = 'sysfs:' + h
        elif os.path.exists('/opt/rocm/bin/rocm-smi'):
            self.src, self.period = 'rocm-smi', 0.1
        self._index = index

    def _read(self):
        t = time.perf_counter()
        if self._files:
            row = {'t': t}
            for key, path in self._files.items():
                try:
                    v = float(open(path).read().strip())
                    row[key] = v / 1e6                       # Hz -> MHz, microwatt -> W
                except (OSError, ValueError):
                    pass
            return row
        import re as _re
        import subprocess
        try:
            txt = subprocess.run(['/opt/rocm/bin/rocm-smi', '-d', str(self._index), '--showclocks', '--showpower', '--json'],
                                 capture_output=True, text=True, timeout=10).stdout
        except Exception:
            return {'t': t}
        row = {'t': t}
        for key, rx in (('sclk_mhz', r'"sclk clock speed:?": "\((\d+)Mhz\)"'), ('mclk_mhz', r'"mclk clock speed:?": "\((\d+)Mhz\)"'),
                        ('power_w', r'Power \(W\)": "([0-9.]+)"')):
            m = _re.search(rx, txt)
            if m:
                row[key] = float(m.group(1))
        return row

    def _loop(self):
        while not self._stop.is_set():
            self.samples.append(self._read())
            self._stop.wait(self.period)

    def start(self):
        if self.src is None:
            return self
        import threading
        self.samples, self._stop = [], threading.Event()
        self._th = threading.Thread(target=self._loop, daemon=True)
        self._th.start()
        return self

    def stop(self, t0=None, t1=None):
        """-> summary of the samples taken inside [t0, t1] (perf_counter times; default: all of them)"""
        if self._th is not None:
            self._stop.set()
            self._th.join(timeout=5)
            self._th = None
        rows = [r for r in self.samples if (t0 is None or r['t'] >= t0) and (t1 is None or r['t'] <= t1)]
        out = {'source': self.src, 'samples': len(rows), 'period_ms': 1e3 * self.period}
        for key in ('sclk_mhz', 'mclk_mhz', 'power_w'):
            v = sorted(r[key] for r in rows if key in r)
            if v:
                out[key] = {'min': round(v[0], 1), 'median': round(v[len(v) // 2], 1), 'max': round(v[-1], 1)}
        return out


def rccl_info(backend):
    """what transport the multi-GPU line was measured on: torch's view of the RCCL version, the version line RCCL itself
    printed under NCCL_DEBUG=VERSION (NCCL_DEBUG_FILE of this process), and the environment that selects the transport"""
    import glob
    import socket
    info = {'backend': backend, 'hip': getattr(torch.version, 'hip', None), 'torch': torch.__version__}
    try:
        info['torch_cuda_nccl_version'] = list(torch.cuda.nccl.version())
    except Exception as e:
        info['torch_cuda_nccl_version'] = 'unavailable: %r' % (e,)
    pat = os.environ.get('NCCL_DEBUG_FILE', '')
    line = None
    if pat:
        path = pat.replace('%h', socket.gethostname()).replace('%p', str(os.getpid()))
        for f in [path] + sorted(glob.glob(pat.replace('%h', '*').replace('%p', '*'))):
            try:
                for ln in open(f, errors='replace'):
                    if 'version' in ln.lower() and ('nccl' in ln.lower() or 'rccl' in ln.lower()):
                        line = ln.strip()
                        break
            except OSError:
                continue
            if line:
                break
    info['version_line'] = line
    info['env'] = {k: os.environ.get(k) for k in ('NCCL_DEBUG', 'HSA_ENABLE_IPC_MODE_LEGACY', 'NCCL_P2P_DISABLE', 'NCCL_SOCKET_IFNAME',
                                                  'RCCL_MSCCL_ENABLE', 'HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES')
                   if os.environ.get(k) is not None}
    return info


def split_blocks(n_steps, want=5):
    """step counts of the back-to-back timing blocks the K timed steps are cut into (>= 5 blocks when K allows)"""
    nb = max(1, min(want, n_steps))
    base, extra = divmod(n_steps, nb)
    return [base + (1 if i < extra else 0) for i in range(nb)]


def cpu_baseline(image_size, n_images, n_steps):
    """The oracle (oracle/sg_oracle.py, kind "port") timed on the host cores on a bounded sample of the same workload
    (SURVEY 8d): the full G+D step at the same widths and flags as the headline pass on ``n_images`` images, 1 warm-up +
    ``n_steps`` timed steps."""
    from oracle import sg_oracle as O
    from scene_generation_amd.args import parser
    from scene_generation_amd.synthetic import make_batch, make_vocab
    args = parser.parse_args(['--image_size', '%d,%d' % (image_size, image_size), '--batch_size', str(n_images),
                              '--vgg_features_weight', '0', '--output_dir', '/tmp/o'])
    from scene_generation_amd.utils import cpu_quota
    # as many threads as the process may actually run: the cgroup's CPU quota when there is one (16 CPUs on the GPU boxes of this
    # project, on a 128-core host).  torch's default -- one thread per physical core -- oversubscribes the quota 8x; the kernel
    # then freezes the process for most of every 100 ms period and the "128-core" figure of rounds 1-5 was really a throttled one
    quota = cpu_quota()
    threads0 = torch.get_num_threads()
    threads = max(1, min(threads0, int(quota))) if quota else threads0
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    tr = O.Trainer(args, make_vocab())
    batches = [make_batch(N=n_images, min_objs=3, max_objs=8, size=image_size, seed=i) for i in range(2)]
    random.seed(0)
    try:
        tr.step(batches[0], use_gt=True)
        t0 = time.perf_counter()
        for i in range(n_steps):
            tr.step(batches[(i + 1) % 2], use_gt=random.randint(0, 1) != 0)
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(threads0)
    return {'value': n_images * n_steps / dt, 'unit': 'images/s', 'cores': threads, 'kind': 'port',
            'physical_cores': physical_cores(), 'logical_cpus': os.cpu_count(), 'cgroup_cpu_quota': quota,
            'sample': 'full G+D step (same widths and flags as the headline pass, fp32, torch-CPU oracle), %d images of the '
                      '%dx%d workload per step, 1 warm-up + %d timed steps, %.1f s' % (n_images, image_size, image_size,
                                                                                        n_steps, dt)}


PMC_KERNELS = {
    # profiler kind -> regex of the kernel symbol (the two dense Winograd instantiations: stores at the top / interleaved)
    # (rocprofv3's --kernel-include-regex is not PCRE: no \d, no (?:...) -- character classes only)
    'wino_bgemm_t128': r'igemm_kernel.*TileCfg<128, 128, 2, 2, [12][, 0-9]*>.*EpRowMajorPlain',
    # the three 64x64-tile instantiations of the F(4x4,3x3) GEMMs (forward, data gradient, weight gradient)
    # (16- or 32-deep k-tiles, with or without the chunked channel sum: TileCfg<64, 64, 2, NSUB, 2, KFOLD>)
    'wino43_bgemm_t64': r'igemm_kernel.*TileCfg<64, 64, 2, [12], 2[, 0-9]*>.*EpRowMajorPlain',
}


def measure_traffic(kind, timeout_s=150):
    """HBM-side bytes per launch of the kernel behind profiler kind ``kind``, measured NOW: two child runs of this script under
    ``rocprofv3 --pmc <counter> --kernel-trace`` (FETCH_SIZE and WRITE_SIZE need separate passes), counters restricted to that
    kernel and adam_kernel.  Corrections exactly as /opt/skills/guides/MI355X_MICROARCH.md prescribes (tools/pmc_db_summary.py):
    the counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (x2); WRITE_SIZE is
    calibrated on the largest adam_kernel launch, whose traffic is known exactly (12 B written per parameter).
    Returns (dict | None, note)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    rx = PMC_KERNELS.get(kind)
    rp = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if rx is None or not os.path.exists(rp):
        return None, 'no live measurement: %s' % ('kind %s has no kernel pattern' % kind if rx is None else 'rocprofv3 not found')
    per = {}
    t_start = time.perf_counter()
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='sg_pmc_', dir='/tmp')
        cmd = [rp, '--pmc', ctr, '--kernel-trace', '--kernel-include-regex', '%s|adam_kernel' % rx, '-d', d, '-o', 'pmc', '--',
               sys.executable, os.path.abspath(__file__), '--steps', '2', '--warmup', '3', '--no_legs', '--no_secondary',
               '--no_prof', '--cpu_baseline', 'off', '--pmc', 'off']
        env = dict(os.environ, TMPDIR='/tmp', SG_GRAPHS='0', SG_STREAM_GROUPS='')      # eager launches: every dispatch is visible to the counters
        try:
            # own process group: on a timeout the whole tree (rocprofv3 AND the python it started) is killed, nothing is left
            # running on the GPU next to the legs that follow
            pr = subprocess.Popen(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                                  start_new_session=True)
            try:
                _, err = pr.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(pr.pid, signal.SIGKILL)
                pr.communicate()
                raise
            rc = pr.returncode
        except Exception as e:
            shutil.rmtree(d, ignore_errors=True)
            return None, 'no live measurement: rocprofv3 %s pass failed: %r' % (ctr, e)
        dbs = glob.glob(os.path.join(d, '**', '*.db'), recursive=True)
        if rc != 0 or not dbs:
            shutil.rmtree(d, ignore_errors=True)
            return None, 'no live measurement: rocprofv3 %s pass rc=%d: %s' % (ctr, rc, (err or '')[-300:])
        rows = sqlite3.connect(dbs[0]).execute('select kernel_name, grid_size, value from counters_collection where '
                                              'counter_name = ?', (ctr,)).fetchall()
        shutil.rmtree(d, ignore_errors=True)
        agg = {}
        for name, grid, val in rows:
            key = ('adam' if 'adam_kernel' in name else 'gemm', grid)
            a = agg.setdefault(key, [0, 0.0])
            a[0] += 1
            a[1] += val * 1024.0
        per[ctr] = agg
    try:
        f, w = per['FETCH_SIZE'], per['WRITE_SIZE']
        k_adam = max((k for k in w if k[0] == 'adam'), key=lambda k: k[1])          # the generator's flat buffer
        wcal = 12.0 * k_adam[1] / (w[k_adam][1] / w[k_adam][0])
        fcal = 16.0 * k_adam[1] / (2.0 * f[k_adam][1] / f[k_adam][0])
        k_gemm = max((k for k in f if k[0] == 'gemm'), key=lambda k: f[k][0])        # most frequent grid = the F(2x2,3x3) GEMMs
        fetch = 2.0 * f[k_gemm][1] / f[k_gemm][0]
        write = wcal * w[k_gemm][1] / max(w[k_gemm][0], 1)
        return ({'bytes_per_launch': fetch + write, 'fetch_bytes_per_launch': fetch, 'write_bytes_per_launch': write,
                 'launches_sampled': f[k_gemm][0], 'grid': k_gemm[1], 'write_calibration': wcal,
                 'fetch_check_on_adam': fcal, 'seconds': round(time.perf_counter() - t_start, 1)},
                'MEASURED IN THIS RUN: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two child passes of bench.py --steps 2, eager '
                'launches), FETCH x2 (gfx950), WRITE calibrated on adam_kernel (x%.3f; the x2 FETCH rule reproduces its 16 B / '
                'parameter to x%.3f)' % (wcal, fcal))
    except Exception as e:
        return None, 'no live measurement: could not reduce the counter tables: %r' % (e,)


# what the profiler kinds are, as template instantiations (include/sg2im_hip.h, csrc/igemm.hip)
KERNEL_INSTANTIATIONS = {
    'wino43_bgemm_t64': 'igemm_kernel<TileCfg<64,64,2,2,2>, A, B, EpRowMajorPlain> with (A, B) = (LoadKContig<64>, LoadKContig<64>) forward, '
                        '(LoadKContig<64>, LoadXContigS<64>) data gradient, (LoadXContigS<64>, LoadXContigS<64>) weight gradient: the 36 '
                        'batched dense GEMMs of a Winograd F(4x4,3x3) conv (the 1024-channel ResnetBlock convs of the generator '
                        'trunk), batch-major tile order, 32-deep software-pipelined k-tiles, buffer loads with the k advance in '
                        'the SGPR offset',

    'wino_bgemm_t128': 'igemm_kernel<TileCfg<128,128,2,2,2>, LoadKContig<128,true,false>, LoadKContig<128,true,false>, EpRowMajorPlain>: '
                       'the 16 batched dense GEMMs of a Winograd F(2x2,3x3) conv (ResnetBlock / VGG19 convs), batch-major '
                       'tile order, 32-deep k-tiles, buffer loads with the k advance in the SGPR offset, software-pipelined '
                       'fragment reads, LDS stores of the next tile interleaved with the MFMAs of phase 0',
    'wino24_bgemm_t128': 'the same instantiation as wino_bgemm_t128: the 25 (x k-chunks) batched dense GEMMs of a Winograd F(2x2,4x4) '
                         'conv (stride-1 4x4 convs of the PatchGANs, K = 256 / 512 channels)',
    'wino_bgemm_t64': 'igemm_kernel<TileCfg<64,64,2,1>, LoadKContig<64,true,false>, LoadKContig<64,true,false>, EpRowMajor>: '
                      'Winograd GEMMs of the 192-channel mask_net convs',
}


def main():
    a = parse()
    if os.environ.get('SG_BENCH_STACKS'):          # debugging aid: dump every thread's stack to stderr every N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ['SG_BENCH_STACKS']), repeat=True)
    # the host driver only supports dmabuf IPC: RCCL / cross-process tensor sharing fails without it (set before HIP starts)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit('launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node %d bench.py '
                             '--gpus %d ...' % (a.gpus, a.gpus))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback in the product path)'
    if os.environ.get('SG_SHARE_GPU') == '1':        # debugging aid: every rank on GPU 0 (needs SG_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = 'cuda:%d' % local
    backend = None
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('SG_DIST_BACKEND', 'nccl')
        # the first real multi-GPU run describes itself: RCCL prints its version line at communicator
        # creation when NCCL_DEBUG >= VERSION; it goes to a per-process file (never into the one-JSON-line stdout) and is read back
        # into ``rccl`` below
        if backend == 'nccl' and 'NCCL_DEBUG' not in os.environ:
            os.environ['NCCL_DEBUG'] = 'VERSION'
            os.environ.setdefault('NCCL_DEBUG_FILE', '/tmp/sg_rccl_debug.%h.%p.log')
        import datetime
        # a rank that never arrives must fail the run, not hang it: 10 minutes covers the slowest first import on a fresh box
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=10))
        from scene_generation_amd.parallel import first_contact
        first_contact(dev)        # checked first collective: a wrong IPC mode / dead link fails HERE with a readable message

    from scene_generation_amd import ops, graphs
    from scene_generation_amd.args import parser
    from scene_generation_amd.synthetic import make_batch, make_vocab
    from scene_generation_amd.pipeline import DeviceBatchPrefetcher
    from scene_generation_amd.trainer import Trainer
    from scene_generation_amd.utils import cpu_quota

    S, B = a.image_size, a.batch_per_gpu
    vocab = make_vocab()

    def make_trainer(vgg_weight, size=None, batch=None, max_objs=8):
        size, batch = size or S, batch or B
        args = parser.parse_args(['--image_size', '%d,%d' % (size, size), '--batch_size', str(batch * world),
                                  '--vgg_features_weight', str(vgg_weight), '--output_dir', '/tmp/o'])
        torch.manual_seed(1234)                   # same initial weights on every rank (also broadcast in Trainer)
        tr = Trainer(args, vocab, device=dev, distributed=world > 1)
        tr.model.layout_objects_hint = max_objs + 1
        tr.share_d_forward = not a.no_share_d_forward
        tr.dense_layout_outputs = False           # nobody reads the three dense layouts here (TensorBoard-only outputs)
        return tr

    tr = make_trainer(a.vgg)
    # two collated host batches per rank (different data per rank: weak scaling), staged to HBM through the collate ->
    # device adapter (pinned double-buffered H2D, host lists for the VectorPool / factored layout planning)
    host_batches = [make_batch(N=B, min_objs=3, max_objs=8, size=S, seed=1000 * rank + i) for i in range(2)]
    staged = list(DeviceBatchPrefetcher(host_batches, dev))
    random.seed(0)                                # the use_gt coin (train.py:195): drawn on rank 0, broadcast (Trainer)
    torch.manual_seed(100 + rank)

    def one_step(trainer, i, batches=None):
        db = (batches or staged)[i % 2]
        trainer.model.objs_host, trainer.model.obj_to_img_host = db.objs_host, db.obj_to_img_host
        trainer.step(db.batch, use_gt=trainer.draw_use_gt())      # train.py:195

    issue = [0.0]

    blocks_ms = [None]

    def timed(trainer, n_steps, first, batches=None, blocks=None):
        """K steps between barrier + synchronize on both sides (the contract); ``blocks``: step counts of back-to-back blocks whose
        boundaries are HIP events recorded on the launch stream (no host synchronisation inside the region) -> blocks_ms[0]"""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        evs = []
        if blocks:
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(blocks) + 1)]
            bounds, acc = set(), 0
            for nb in blocks:
                acc += nb
                bounds.add(acc)
        t0 = time.perf_counter()
        if evs:
            evs[0].record()
        ie = 1
        for i in range(n_steps):
            one_step(trainer, first + i, batches)
            if evs and (i + 1) in bounds:
                evs[ie].record()
                ie += 1
        issue[0] = time.perf_counter() - t0          # host done issuing; the GPU may still be working
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        d = time.perf_counter() - t0
        timed.t0, timed.t1 = t0, t0 + d
        if evs:
            bm = [evs[j].elapsed_time(evs[j + 1]) for j in range(len(blocks))]
            if world > 1:
                tb = torch.tensor(bm, device=dev, dtype=torch.float64)
                dist.all_reduce(tb, op=dist.ReduceOp.MAX)
                bm = [float(x) for x in tb.tolist()]
            blocks_ms[0] = bm
        if world > 1:
            t = torch.tensor([d], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d = float(t.item())
        return d

    # untimed warm-up; with fewer than 3 steps the hipGraph capture of the static sub-networks (graphs.py: two eager steps,
    # capture on the third) would land inside the timed region
    for i in range(a.warmup):
        one_step(tr, i)
    # headline pass: exactly K steps, no per-launch instrumentation
    calls0, replays0 = ops.CALLS[0], graphs.REPLAYS[0]
    blocks = split_blocks(a.steps)
    sampler = ClockSampler(local).start()
    thr0 = cpu_throttle_count()
    dt = timed(tr, a.steps, a.warmup, blocks=blocks)
    thr1 = cpu_throttle_count()
    clocks = sampler.stop(timed.t0, timed.t1)
    host_issue = issue[0]
    # value = the MEDIAN block: the K steps run back to back exactly as the contract says, cut into >= 5
    # blocks by HIP events on the launch stream; the whole-region figure (host clock around barrier + synchronize) stays in
    # ``whole_region``.  Block rates are per-step rates, so blocks of unequal length compare directly.
    per_step = sorted(ms / nb for ms, nb in zip(blocks_ms[0], blocks))
    med_ms = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    repeat = {'blocks': len(blocks), 'steps_per_block': blocks,
              'ms_per_step_blocks': [round(ms / nb, 4) for ms, nb in zip(blocks_ms[0], blocks)],
              'ms_per_step_median': med_ms, 'ms_per_step_min': per_step[0], 'ms_per_step_max': per_step[-1],
              'spread': (per_step[-1] - per_step[0]) / med_ms}
    calls, replays = (ops.CALLS[0] - calls0) / a.steps, (graphs.REPLAYS[0] - replays0) / a.steps
    # host cost of ISSUING one step, measured with an idle GPU in front of it: once the step is GPU-bound the number above
    # mostly measures back-pressure of the full launch queue, not host work
    iso = []
    quick = os.environ.get('SG_BENCH_QUICK') == '1'      # test harnesses: one isolated / one observed step instead of three
    for i in range(1 if quick else 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        one_step(tr, a.warmup + a.steps + i)
        iso.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    # roofline pass: the SAME K steps again with a HIP event pair around every kernel launch on the launch stream
    dt_prof = None
    if not a.no_prof:
        # ... and with the side streams off (streams.py, group 'front'): a launch that shares the chip with another stream's
        # kernels is stretched by what they take, and the pass is there to price every kernel on its own.  The headline pass
        # above runs WITH them (`side_streams` in the JSON line); rocprofv3's durations of that pass are the stretched ones.
        from scene_generation_amd import streams
        groups_on = sorted(streams.GROUPS)
        streams.GROUPS.clear()
        try:
            ops.prof_reset()
            ops.prof_enable(True)
            dt_prof = timed(tr, a.steps, a.warmup + a.steps + 3)
            ops.prof_enable(False)
        finally:
            streams.GROUPS.update(groups_on)
    # sanity: the step really trained (finite losses)
    total = dict(tr.generator_losses.items())['total_loss']
    assert total == total and abs(total) < 1e6, 'non-finite generator loss %r' % total

    # DP observability: isolated all-reduce time per bucket and how long the step was exposed to the collectives
    comm = None
    if world > 1:
        comm = {'world': world}
        for r in tr.reducers:
            r.profile = True
        n_obs = 1 if quick else 3
        d_obs = timed(tr, n_obs, a.warmup + 2 * a.steps + 3)
        names = {id(tr.optimizer): 'G', id(tr.optimizer_d_img): 'D_img', id(tr.optimizer_d_obj): 'D_obj',
                 id(tr.optimizer_d_mask): 'D_mask'}
        iso_total, exposed_total = 0.0, 0.0
        for r in tr.reducers:
            exp_ms, n = r.exposed_ms()
            r.profile = False
            buckets = r.time_buckets()
            iso_ms = sum(ms for _, ms in buckets)
            iso_total += iso_ms
            exposed_total += exp_ms / n_obs
            comm[names.get(id(r.optimizer), '?')] = {
                'buckets': len(buckets), 'bytes': sum(b for b, _ in buckets), 'overlap_mode': bool(r.overlap),
                'isolated_allreduce_ms': round(iso_ms, 3),
                'per_bucket_ms': [round(ms, 3) for _, ms in buckets],
                'algbw_GBps': round(sum(b for b, _ in buckets) / (iso_ms * 1e-3) / 1e9, 1) if iso_ms > 0 else None,
                'exposed_ms_per_step': round(exp_ms / n_obs, 3)}
        comm['exposed_ms_per_step'] = round(exposed_total, 3)
        comm['isolated_ms_per_step'] = round(iso_total, 3)
        comm['overlap_fraction'] = round(1.0 - exposed_total / iso_total, 3) if iso_total > 0 else None
        comm['ms_per_step_observed'] = 1e3 * d_obs / n_obs

    out = {
        'metric': 'images/sec G+D step, 128x128 <=8-obj scene graphs', 'value': B * world / (med_ms * 1e-3),
        'unit': 'images/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': med_ms,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[1]: COCO-Stuff-shaped %dx%d, <=8 objects/img (+__image__), batch %d '
                               'per GPU, full G+D train step (G fwd/bwd + 3 D steps + 4 Adam%s), reference default '
                               'widths, VGG loss %s' % (S, S, B, ' + RCCL grad all-reduce' if world > 1 else '',
                                                        'on (weight %g)' % a.vgg if a.vgg > 0 else 'off (SURVEY 8d)'),
                   'global_batch': B * world, 'image_size': S, 'parallelism': 'dp%d' % world,
                   'share_d_forward': not a.no_share_d_forward, 'vgg_features_weight': a.vgg,
                   'dense_layout_outputs': False, 'hip_graphs': graphs.ENABLED},
        # wall time the host needed to ISSUE the K steps (no sync): close to ms_per_step => launch-bound
        'host_issue_ms_per_step': 1e3 * host_issue / a.steps,
        # the same for a single step issued into an idle GPU (no queue back-pressure): the host-side cost of a step
        'host_issue_isolated_ms_per_step': 1e3 * sum(iso) / len(iso),
        # C-ABI calls (each launches 1..4 kernels) and hipGraph launches the host issues per step; the kernels inside a
        # replayed graph are dispatched by the GPU front-end without host involvement
        'host_calls_per_step': calls, 'graph_replays_per_step': replays,
        # streams.py groups active in the headline pass ('front': embeddings / graph convolutions / box_net / mask_net on a side
        # stream beside the image path); the per-launch figures under ``roofline`` / ``kernels`` come from a one-stream pass
        'side_streams': sorted(__import__('scene_generation_amd.streams', fromlist=['GROUPS']).GROUPS),
        # ``value`` / ``ms_per_step`` are the median of these back-to-back blocks of the K timed steps; ``whole_region`` is the
        # host clock around barrier + synchronize on both sides of the same K steps (what rounds 1-5 reported as ``value``)
        'repeat_spread': repeat['spread'], 'repeat': repeat,
        'whole_region': {'value': B * world * a.steps / dt, 'ms_per_step': 1e3 * dt / a.steps, 'seconds': dt},
        'clocks': clocks,
        # host side of the measurement: the cgroup's CPU quota (CPUs per period; None = unlimited) and how many scheduler periods
        # ended with the process frozen by it while the timed steps ran (0 = the host was never taken away from the launch thread)
        'host': {'cgroup_cpu_quota': cpu_quota(), 'cpu_throttled_periods_in_timed_region':
                 (thr1 - thr0) if thr0 is not None and thr1 is not None else None,
                 'logical_cpus': os.cpu_count(), 'torch_threads': torch.get_num_threads()},
    }
    if world > 1:
        out['rccl_ranks'] = dist.get_world_size()
        out['dist_backend'] = backend
        out['allreduce'] = comm
        out['rccl'] = rccl_info(backend)
    if rank == 0 and not a.no_prof:
        prof = ops.prof_read()
        mm = {k: v for k, v in prof.items() if v['launches'] > 0 and v['flops'] > 0 and k not in ('linear', 'head_conv')}
        all_ms = sum(v['ms'] for v in prof.values())
        out['launches_per_step'] = sum(v['launches'] for v in prof.values()) / a.steps
        if mm:
            name, v = max(mm.items(), key=lambda kv: kv[1]['ms'])
            ach = v['flops'] / (v['ms'] * 1e-3) / 1e12 if v['ms'] > 0 else 0.0
            traffic, tsrc, tdetail = None, None, None
            if a.pmc == 'auto' and world == 1:
                tdetail, tsrc = measure_traffic(name)
                if tdetail:
                    traffic = tdetail['bytes_per_launch']
            if traffic is None:
                live_note = tsrc
                for tname in ('r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json'):
                    tpath = os.path.join(ROOT, 'profiles', tname)
                    if os.path.isfile(tpath):
                        tab = json.load(open(tpath)).get(name)
                        if tab:
                            traffic = tab.get('bytes_per_launch')
                            tsrc = 'STATIC (not measured in this run%s): profiles/%s -- %s' % (
                                '; ' + live_note if live_note else '', tname, tab.get('source'))
                            break
            # algorithmic bytes of one launch of the dominant kernel (operands read once + result written once): the 16
            # batched GEMMs of a ResnetBlock conv, C = 64 << 4 channels both sides, P = B (S / 32)^2 tiles of 2x2 outputs
            alg = None
            if name == 'wino_bgemm_t128':
                Cc, P = 64 << 4, B * (S // 32) ** 2
                alg = 4.0 * 16 * (Cc * Cc + P * Cc + Cc * P)
            elif name == 'wino43_bgemm_t64':         # 36 batches, P = B (S / 64)^2 tiles of 4x4 outputs (forward / data gradient)
                Cc, P = 64 << 4, B * (S // 64) ** 2
                alg = 4.0 * 36 * (Cc * Cc + P * Cc + Cc * P)
            # what the launches of this kernel would cost in the DIRECT form of the convs they serve (SURVEY 8d counts the step in
            # direct-form MACs): a ResnetBlock conv pass is 2 * 9 * C^2 * H W * N flops whichever Winograd form runs it
            direct = None
            if name in ('wino_bgemm_t128', 'wino43_bgemm_t64'):
                Cc = 64 << 4
                direct = 2.0 * 9 * Cc * Cc * (S // 16) ** 2 * B
            out['roofline'] = {'bound': 'mfma', 'achieved': ach, 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                               'direct_form_flops_per_launch': direct,
                               'direct_form_equivalent_tflops': (direct * v['launches'] / (v['ms'] * 1e-3) / 1e12) if direct else None,
                               'flops_note': '``achieved`` / ``frac`` count the multiplies the matrix pipe really issues (F(4x4,3x3): 36 per '
                                             '16 outputs and channel pair, 1.78x fewer than F(2x2,3x3), 4x fewer than the direct 3x3 form); '
                                             '``direct_form_equivalent_tflops`` prices the same launches at the direct-form flops of the '
                                             'convs they compute (the unit of SURVEY 8d) -- it exceeds the peak by the MACs Winograd removes',
                               'frac': ach / F32_MFMA_PEAK_TFLOPS, 'traffic': traffic, 'traffic_source': tsrc,
                               'traffic_detail': tdetail, 'algorithmic_bytes_per_launch': alg,
                               'traffic_over_algorithmic': (traffic / alg) if (traffic and alg) else None,
                               'kernel': name, 'instantiation': KERNEL_INSTANTIATIONS.get(name, name),
                               'flops_counted': 'MACs of the non-padding tiles the launch computes (useful work)',
                               'launches': v['launches'], 'avg_us': 1e3 * v['ms'] / v['launches'],
                               'flops_per_launch': v['flops'] / v['launches'],
                               'share_of_step': v['ms'] / (1e3 * dt_prof),
                               'measured_in': 'second pass of the same %d steps with a HIP event pair around every launch, eager and on '
                                              'ONE stream (%.1f ms/step; the headline pass runs without the events, with hipGraph replays '
                                              'and with the object front on its side stream)' % (a.steps, 1e3 * dt_prof / a.steps)}
            igms = sum(x['ms'] for x in mm.values())
            igfl = sum(x['flops'] for x in mm.values())
            lin = prof.get('linear', {'flops': 0.0})
            step_flops = (igfl + lin['flops']) / a.steps
            out['kernels'] = {
                # every MFMA GEMM of the step together, and the step as a whole: FLOPs the matrix pipe really issued (Winograd
                # and sub-pixel forms count their own, reduced, MACs) over the time of the GEMMs / of the whole step
                'all_mfma_gemms': {'ms_per_step': igms / a.steps, 'tflops': igfl / (igms * 1e-3) / 1e12 if igms else 0.0,
                                   'frac': igfl / (igms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS if igms else 0.0},
                'step_mfma_frac': step_flops / (dt / a.steps) / 1e12 / F32_MFMA_PEAK_TFLOPS,
                # 237 GFLOP per image in the direct form of every conv / GEMM of the step at 128x128 -- the step runs above
                # the fp32 MFMA peak on that scale because the Winograd / sub-pixel / factored forms remove MACs
                'step_direct_form_tflops': (237e9 * B / (dt / a.steps) / 1e12) if S == 128 else None,
                'step_mfma_tflop': step_flops / 1e12,
                'timed_kernels_ms_per_step': all_ms / a.steps,
                'top': {k: {'ms_per_step': round(x['ms'] / a.steps, 3), 'launches_per_step': x['launches'] / a.steps,
                            'tflops': round(x['flops'] / (x['ms'] * 1e-3) / 1e12, 2) if x['ms'] and x['flops'] else None,
                            'gbs': round(x['bytes'] / (x['ms'] * 1e-3) / 1e9, 1) if x['ms'] and x['bytes'] else None}
                        for k, x in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])[:14] if x['launches']}}
        # HBM-bound kernel classes: ALGORITHMIC bytes (what the op must read + write once) over the measured time
        hbm = {}
        for k in ('instnorm', 'instnorm_bwd', 'batchnorm', 'adam', 'wino_transforms', 'head_conv', 'crop', 'layout_fwd',
                  'layout_bwd'):
            x = prof.get(k)
            if x and x['launches'] and x['bytes'] and x['ms']:
                gbs = x['bytes'] / (x['ms'] * 1e-3) / 1e9
                hbm[k] = {'launches_per_step': x['launches'] / a.steps, 'ms_per_step': round(x['ms'] / a.steps, 3),
                          'avg_us': round(1e3 * x['ms'] / x['launches'], 2),
                          'alg_bytes_per_launch': round(x['bytes'] / x['launches']), 'GBps': round(gbs, 1),
                          'frac': round(gbs / HBM_PEAK_GBS, 3)}
        out['roofline_hbm'] = {'peak_GBps': HBM_PEAK_GBS, 'note': 'algorithmic bytes / HIP-event time per launch; a kernel that '
                               're-reads its input through L2 shows less than its real traffic rate', 'kernels': hbm}
    if world == 1 and not a.no_secondary:
        sec = {}
        n2 = max(2, min(a.steps, 6))
        # (1) the reference's DEFAULT flags: VGG19 perceptual loss on (args.py:73), random-init VGG weights
        if a.vgg == 0:
            tr2 = make_trainer(10.0)
            for i in range(4):                    # 2 eager steps + the hipGraph capture + 1 replay before timing
                one_step(tr2, i)
            d2 = timed(tr2, n2, 4)
            sec['default_flags_vgg_on'] = {'images_per_s': B * n2 / d2, 'ms_per_step': 1e3 * d2 / n2, 'steps': n2,
                                           'note': '--vgg_features_weight 10 (args.py:73), He-normal VGG19 weights'}
            del tr2
        # (1b) the boundary as the reference's loop has it (train.py:190-193): every step is handed a HOST batch.  The K host
        # batches go through pipeline.DeviceBatchPrefetcher INSIDE the timed region (validation, host summaries, packing into a
        # page-locked slot on the staging thread, one copy kernel per batch on the launch stream) -- the PCIe-inclusive rate;
        # never the headline ``value``
        nh, nwarm = max(4, a.steps), 4
        it = iter(DeviceBatchPrefetcher([host_batches[i % 2] for i in range(nwarm + nh)], dev))

        def host_step():
            db = next(it)
            tr.model.objs_host, tr.model.obj_to_img_host = db.objs_host, db.obj_to_img_host
            tr.step(db.batch, use_gt=tr.draw_use_gt())
        for _ in range(nwarm):                # fills the pipeline: the page-locked staging slots exist
            host_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nh):
            host_step()
        torch.cuda.synchronize()
        dh = time.perf_counter() - t0
        hb_bytes = sum(t.numel() * t.element_size() for t in host_batches[0])
        sec['host_buffers'] = {'images_per_s': B * nh / dh, 'ms_per_step': 1e3 * dh / nh, 'steps': nh,
                               'host_bytes_per_batch': hb_bytes,
                               'note': 'headline configuration fed from pageable HOST batches through DeviceBatchPrefetcher '
                                       'inside the timed loop (collate-contract validation + single-threaded packing into re-used '
                                       'page-locked slots on a staging thread + one sg_stage_copy kernel per batch on the launch '
                                       'stream), after %d untimed steps of the same iterator' % nwarm}
        # (2) Trainer.step's own defaults: fast paths on AND the three dense (N,204,H,W) layouts of Model.forward written
        tr.dense_layout_outputs = True
        one_step(tr, 0)
        d4 = timed(tr, n2, 1)
        tr.dense_layout_outputs = False
        sec['fast_paths_on_dense_layouts'] = {'images_per_s': B * n2 / d4, 'ms_per_step': 1e3 * d4 / n2, 'steps': n2,
                                              'note': 'headline configuration + dense_layout_outputs=True (3 x 428 MB written)'}
        # the masks_to_layout kernel on its own: algorithmic bytes = 4 N D H W written + inputs
        ops.prof_reset()
        ops.prof_enable(True)
        for i in range(2):
            one_step(tr, i)
        tr.dense_layout_outputs = True
        for i in range(2):
            one_step(tr, i)
        tr.dense_layout_outputs = False
        torch.cuda.synchronize()
        ops.prof_enable(False)
        lf = ops.prof_read().get('layout_fwd')
        if lf and lf['launches']:
            per = sorted([(lf['bytes'] / lf['launches'])])[0]
            gbs = lf['bytes'] / (lf['ms'] * 1e-3) / 1e9
            sec['masks_to_layout_fwd'] = {'launches': lf['launches'], 'avg_us': round(1e3 * lf['ms'] / lf['launches'], 2),
                                          'alg_bytes_per_launch_mean': round(per), 'GBps': round(gbs, 1),
                                          'frac_of_hbm_peak': round(gbs / HBM_PEAK_GBS, 3),
                                          'note': 'all launches of the kind in 4 eager steps (dense (N,204,H,W) layouts of the '
                                                  'last two steps + the plane builders of the factored convs)'}
        # (3) every fast path off: direct convs instead of Winograd, dense 204-channel layout convs (channel-sparse),
        #     discriminator forwards re-run in the D steps like the reference, dense layouts materialised
        saved = (ops.WINOGRAD, ops.FACTORED_LAYOUT, ops.UPCONV, ops.HEADCONV)
        try:
            ops.WINOGRAD = ops.FACTORED_LAYOUT = ops.UPCONV = ops.HEADCONV = False
            tr3 = make_trainer(a.vgg)
            tr3.share_d_forward = False
            tr3.dense_layout_outputs = True
            for i in range(4):
                one_step(tr3, i)
            d3 = timed(tr3, n2, 4)
            sec['fast_paths_off'] = {'images_per_s': B * n2 / d3, 'ms_per_step': 1e3 * d3 / n2, 'steps': n2,
                                     'note': 'SG_WINOGRAD=0 SG_FACTORED_LAYOUT=0 SG_UPCONV=0 SG_HEADCONV=0 '
                                             '--no_share_d_forward, dense layouts written'}
            del tr3
        finally:
            ops.WINOGRAD, ops.FACTORED_LAYOUT, ops.UPCONV, ops.HEADCONV = saved
        out['secondary'] = sec
    if world == 1 and not a.no_legs:
        # the other BASELINE configs as secondary legs (parity-tested shapes; not the headline): c4 = configs[3] per-GPU shape
        # (256x256, <= 16 objects, 8 images per GPU), c5 = configs[4] (32 objects / 96 triples per image, 128x128, N = 32:
        # the GraphTripleConv scatter stress) with the graph kernels' own rates
        from scene_generation_amd.synthetic import make_config_batch, CONFIGS
        legs = {}
        del tr
        torch.cuda.empty_cache()
        for name in ('c5', 'c4'):
            cfg = CONFIGS[name]
            trl = make_trainer(a.vgg, size=cfg['size'], batch=cfg['N'], max_objs=cfg['max_objs'])
            hb = [make_config_batch(name, seed=2000 + i) for i in range(2)]
            st = list(DeviceBatchPrefetcher(hb, dev))
            for i in range(4):
                one_step(trl, i, st)
            nl = max(2, min(a.steps, 5))
            dl = timed(trl, nl, 4, st)
            leg = {'images_per_s': cfg['N'] * nl / dl, 'ms_per_step': 1e3 * dl / nl, 'steps': nl,
                   'shape': '%dx%d, %d images, %d..%d objects/img, O=%d T=%d' % (
                       cfg['size'], cfg['size'], cfg['N'], cfg['min_objs'], cfg['max_objs'], hb[0].objs.numel(),
                       hb[0].triples.size(0))}
            if not a.no_prof:
                ops.prof_reset()
                ops.prof_enable(True)
                dp = timed(trl, 2, 4 + nl, st)
                ops.prof_enable(False)
                pr = ops.prof_read()
                for k in ('linear', 'segsum'):
                    x = pr.get(k)
                    if x and x['launches'] and x['ms']:
                        leg[k] = {'ms_per_step': round(x['ms'] / 2, 3), 'launches_per_step': x['launches'] / 2,
                                  'tflops': round(x['flops'] / (x['ms'] * 1e-3) / 1e12, 2) if x['flops'] else None,
                                  'GBps': round(x['bytes'] / (x['ms'] * 1e-3) / 1e9, 1) if x['bytes'] else None}
                mmL = {k: v for k, v in pr.items() if v['launches'] and v['flops'] and k not in ('linear', 'head_conv')}
                tms = sum(v['ms'] for v in mmL.values())
                leg['all_mfma_gemms_tflops'] = round(sum(v['flops'] for v in mmL.values()) / (tms * 1e-3) / 1e12, 1) if tms else None
            legs[name] = leg
            del trl, st
            torch.cuda.empty_cache()
        out['legs'] = legs
    if rank == 0:
        if world == 1 and a.cpu_baseline == 'auto':
            try:
                out['cpu_baseline'] = cpu_baseline(S, a.cpu_images, a.cpu_steps)
            except Exception as e:           # the baseline is a report, never a reason to lose the bench line
                out['cpu_baseline'] = {'value': None, 'unit': 'images/s', 'cores': torch.get_num_threads(),
                                       'kind': 'port', 'sample': 'failed: %r' % (e,)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
