"""Side HIP streams for independent sub-networks.

The three PatchGAN scales of the image discriminator (discriminators.py:192-202 of the reference) -- and the three
discriminators themselves -- share no data after their inputs: on one stream their kernels queue behind each other, and the
small-scale layers (64x64 and 32x32 inputs: a few hundred workgroups, 20-50 us per launch) leave most of the 256 CUs idle
while they run.  Issued on separate streams they fill the gaps of the full-resolution scale (measured on MI355X,
tools/probe/multistream_probe.py: forward + backward of the three scales 6.14 -> 5.35 ms).

    with fork(x.device, 'imgD') as f:
        with f.branch(0): ...          # branch 0 stays on the current stream
        with f.branch(1): ...          # side stream, ordered after everything issued before the fork
    # leaving the fork makes the current stream wait for every branch

Rules that keep this safe:
  * every tensor a branch reads was produced before the fork (or inside the branch) and is named in ``branch(i, reads=...)``;
    branch outputs are consumed after the join and named in ``produced(...)`` -- both are ``record_stream``-ed on the other
    stream, so the caching allocator never recycles a block a kernel of the other stream may still be using,
  * autograd runs the backward of an operator on the stream of its forward and synchronises across streams itself -- but the
    weight-gradient kernels write straight into the optimiser's flat gradient buffer (ops.GradOut), which autograd does not
    see: whoever reads that buffer (optimiser step, gradient all-reduce) calls ``join_all()`` first,
  * the scratch buffers of ops.workspace() are per stream; results are bit-identical to the single-stream order (no atomics,
    every buffer has one writer).
Always off (everything stays on the current stream) under hipGraph capture and on CPU tensors.

OPT-IN (SG_MULTISTREAM=1).  Measured on MI355X inside the full step: 778.2 images/s on one stream vs 770.9 with a stream per
scale -- a kernel trace shows 2.2 ms/step of kernels overlapping, but the co-running kernels stretch by the same amount (the
full-resolution scale already fills the chip; only launch gaps and tails are there to win) and the fork / join edges add
waits.  The isolated three-scale chain does gain (6.14 -> 5.35 ms), which is why the switch stays; the GPU suite passes with
it on, and tests/test_gpu_parity.py checks that both settings give bit-identical results.
"""
import contextlib
import os

import torch

ENABLED = os.environ.get('SG_MULTISTREAM', '0') == '1'
# Groups that fork even when ENABLED is off.  'front' (round 6, default ON): the object front of Model.forward -- embeddings,
# graph convolutions, box_net, mask_net: ~250 launches of a few microseconds each, forward + backward -- shares NO data with the
# image path of the training branch (crops -> AppearanceEncoder -> layouts from the GROUND-TRUTH boxes and masks -> generator:
# model.py:98-124 of the reference); on a side stream its latency-bound launches run under the generator's GEMMs instead of in
# front of / behind them.  SG_STREAM_GROUPS='' switches it off, SG_STREAM_GROUPS=front,imgD adds groups by name.
# 'adam' (round 6, default ON): inside Trainer.step the generator's Adam step on a side stream under the discriminator sub-steps
# (trainer._step_or_defer) -- 0.6 ms of HBM streaming.  Next to the front ALONE it lost 1.0 % (30.30 -> 30.61 ms/step: the sub-steps'
# GEMMs, all on one stream, lost more to the contention than the step hid); with the mask / object / image discriminator work
# spread over their own streams it gains 1.3 % (28.94 -> 28.56, three same-box pairs; configs[3] shape +0.9 %, configs[4] level;
# profiles/r06_ab_adam_stream.txt).
# 'mstep' (round 6, default ON; needs 'front'): the mask discriminator's work -- its two forwards and data gradients inside the
# generator step, its own sub-step (backward through the shared forwards, Adam) -- continues the front's stream: O 16x16 masks,
# small launches fed by masks_pred only (trainer.train_generator / train_mask_discriminator).  +3.1 % on top of 'front'.
# 'imgD' (default ON since the end of round 6): a stream per PatchGAN scale of the image discriminator (the docstring above: a loss
# of 1 % in round 3; with this round's kernels and the front / mask work already beside them it is +1.0 %, three same-box pairs,
# profiles/r06_ab_imgd_stream.txt).  'maskD' (a stream per scale of the mask discriminator): +0.2 %, inside the noise, stays off.
# 'objD' (default ON): the object discriminator's branch of the generator step and, inside Trainer.step, its whole sub-step on a
# stream of their own: -0.2 % next to the front alone, +0.4 % (three pairs) once the mask / image discriminators run beside it
# (profiles/r06_ab_objd_stream.txt).
# What the last three stand on: HIP streams share ROCclr's hardware queues (GPU_MAX_HW_QUEUES, default 4).  With the default the
# seven streams of a step fold onto four queues and the step runs 28.5 ms; with 5 or more queues and ALL groups on it runs 42-45 ms
# (every stream its own queue: the mid-size kernels of 'imgD' / 'objD' / 'adam' then really share the chip), with 3 queues 29.5,
# with 2 30.2.  'front' + 'mstep' alone (small launches) are 29.5 ms whatever the queue count (profiles/r06_ab_hw_queues.txt).  So
# when the environment asks for another queue count, only those two are on by default.
_HWQ = os.environ.get('GPU_MAX_HW_QUEUES', '').strip()
_DEFAULT_GROUPS = 'front,mstep,imgD,objD,adam' if _HWQ in ('', '4') else 'front,mstep'
GROUPS = set(g for g in os.environ.get('SG_STREAM_GROUPS', _DEFAULT_GROUPS).split(',') if g)
_POOL = {}            # (device index, group, branch) -> torch.cuda.Stream
_LIVE = {}            # device index -> {side stream that has been handed out: its group}


def group_on(group):
    return ENABLED or group in GROUPS


def _usable(device, group=None):
    return (group_on(group) and device is not None and device.type == 'cuda' and torch.cuda.is_available()
            and not torch.cuda.is_current_stream_capturing())


def side_stream(device, group, i):
    key = (device.index if device.index is not None else torch.cuda.current_device(), group, i)
    s = _POOL.get(key)
    if s is None:
        s = _POOL[key] = torch.cuda.Stream(device=device)
        _LIVE.setdefault(key[0], {})[s] = group
    return s


class fork(object):
    def __init__(self, device, group, enabled=True):
        self.device, self.group = device, group
        self.on = bool(enabled) and _usable(device, group)
        self.used = []

    def __enter__(self):
        if self.on:
            self.main = torch.cuda.current_stream(self.device)
        return self

    def branch(self, i, reads=()):
        """``reads``: tensors allocated on the main stream that kernels of this branch read (its inputs; autograd keeps them
        for the branch's backward, which runs on the same side stream).  They are recorded on the side stream so that the
        caching allocator does not hand their block to another main-stream kernel while a side-stream kernel may still be
        reading it (the block is reused only after the side-stream work queued at free time has finished -- ADVICE r3)."""
        if not self.on or i == 0:
            return contextlib.nullcontext()
        s = side_stream(self.device, self.group, i)
        s.wait_stream(self.main)
        self.used.append(s)
        for t in _tensors(reads):
            t.record_stream(s)
        self._side = True
        return torch.cuda.stream(s)

    def produced(self, outputs):
        """tensors a side branch allocated and the code behind the join consumes (and frees) on the main stream"""
        if self.on and torch.cuda.current_stream(self.device) != self.main:
            for t in _tensors(outputs):
                t.record_stream(self.main)

    def join(self):
        """the fork's stream waits for every branch issued so far (what leaving the ``with`` block does; idempotent)"""
        if self.on:
            for s in self.used:
                self.main.wait_stream(s)
            self.used = []

    def __exit__(self, *exc):
        self.join()
        return False


def _tensors(x):
    if isinstance(x, torch.Tensor):
        if x.is_cuda and x.untyped_storage().size() > 0:
            yield x
    elif isinstance(x, (list, tuple)):
        for y in x:
            for t in _tensors(y):
                yield t


def join_all(device=None, exclude=()):
    """Make the current stream wait for every side stream of the device (cheap: one event per stream).  Called before anything
    reads memory that side-stream kernels write behind autograd's back: the flat gradient buffers.  ``exclude``: groups whose
    kernels never write what the caller is about to read (a discriminator's optimiser does not wait for the generator's front)."""
    if not _LIVE or not torch.cuda.is_available():
        return
    idx = torch.cuda.current_device() if device is None or device.index is None else device.index
    live = _LIVE.get(idx)
    if not live or torch.cuda.is_current_stream_capturing():
        return
    cur = torch.cuda.current_stream(idx)
    for s, group in live.items():
        if s != cur and group not in exclude:
            cur.wait_stream(s)

