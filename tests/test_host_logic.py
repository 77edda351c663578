"""CPU-only checks of the host side: C-ABI surface, flag surface, VectorPool planning, lazy LossManager,
flat-buffer bookkeeping, batch sharding.  No compute entry point of the HIP library is called here."""
import ctypes
import json
import os
import random
import re
import sys

import numpy as np
import pytest
import torch

from oracle import sg_oracle as O
from scene_generation_amd import _hip
from scene_generation_amd.args import parser
from scene_generation_amd.synthetic import make_batch, make_config_batch, shard_batch, make_vocab, fill_deterministic
from scene_generation_amd.utils import plan_pool_query, LossManager, int_tuple, bool_flag, str_tuple

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    src = open(os.path.join(ROOT, 'include', 'sg2im_hip.h')).read()
    declared = set(re.findall(r'\b(sg_[A-Za-z0-9_]+)\s*\(', re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)))
    assert declared == set(_hip.PROTOS), declared ^ set(_hip.PROTOS)
    assert len(declared) >= 50
    lib = _hip.lib()                       # loads on a CPU-only host too (no kernel is launched)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.sg_version() >= 100
    assert isinstance(lib.sg_last_error_string(), bytes)


def test_argument_errors_are_reported_not_thrown():
    lib = _hip.lib()
    rc = lib.sg_linear_fwd(None, None, None, None, 4, 4, 4, 0, 0.0, None)
    assert rc < 0 and b'sg_linear_fwd' in lib.sg_last_error_string()
    d = _hip.sgConvDesc(1, 3, 0, 8, 8, 4, 5, 1, 0, 0, 1, 4, 4, 0, 0)      # kernel size 5 unsupported
    rc = lib.sg_conv2d_fwd(ctypes.byref(d), ctypes.c_void_p(8), None, ctypes.c_void_p(8), None, ctypes.c_void_p(8), 0, 0.0, ctypes.c_void_p(8), 1 << 20, None)
    assert rc < 0 and b'kernel size' in lib.sg_last_error_string()


def test_tensor_size_limit_of_the_buffer_load_masking():
    """The gather loaders mask with the range check of raw buffer loads: byte offsets are 32 bits against num_records = 2^31
    bytes, so an operand may hold at most 2^29 elements (up to 2^31 used to pass validation and would have read
    zeros beyond 2 GiB).  Checked before any launch, so it runs without a device."""
    lib = _hip.lib()
    p8 = ctypes.c_void_p(8)

    def fwd(N, C, H):
        d = _hip.sgConvDesc(N, C, 0, H, H, 64, 1, 1, 0, 0, 1, H, H, 0, 0)      # 1x1 conv: no padded grid
        return lib.sg_conv2d_fwd(ctypes.byref(d), p8, None, p8, None, p8, 0, 0.0, None, 1 << 20, None)
    # 128 x 64 x 256 x 256 = 2^29 elements: the largest legal input (fails LATER, on the null workspace, not on the size)
    assert fwd(128, 64, 256) < 0 and b'elements' not in lib.sg_last_error_string()
    # one more image: rejected for its size
    assert fwd(129, 64, 256) < 0 and b'elements' in lib.sg_last_error_string()
    # dense layers: rows x features
    assert lib.sg_linear_fwd(p8, p8, None, p8, 1 << 20, 1 << 10, 8, 0, 0.0, None) < 0
    assert b'exceeds' in lib.sg_last_error_string()


def test_late_gradient_contribution_is_refused_once_the_bucket_is_reduced():
    """GradReducer(overlap=True) treats a parameter as final on its first delivery; a second CONTRIBUTION (ops.GradOut mode 1)
    after the bucket went to the all-reduce must raise instead of letting the ranks diverge.  A repeated REPORT
    of the same delivery stays a no-op."""
    from scene_generation_amd.optim import FlatParams
    from scene_generation_amd.parallel import GradReducer
    fp = FlatParams([torch.nn.Parameter(torch.zeros(8)), torch.nn.Parameter(torch.zeros(8))])
    red = GradReducer(fp, bucket_bytes=16)
    red.world, red.overlap = 2, True              # pretend: two ranks (no process group is touched below)
    launched = []
    red._launch = lambda b: (launched.append(b), red._launched.__setitem__(b, True), setattr(red, '_next', b + 1))
    red.begin_step()
    red.late_contribution(1)                       # nothing launched yet: fine
    red.param_ready(1)                             # reverse parameter order: bucket 0 holds parameter 1
    assert launched == [0]
    red.param_ready(1)                             # repeated report: no-op
    with pytest.raises(RuntimeError, match='another gradient contribution'):
        red.late_contribution(1)
    red.late_contribution(0)                       # its bucket is still local


def test_cpu_tensors_fail_loudly():
    from scene_generation_amd import ops
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.linear(torch.ones(2, 3), torch.ones(4, 3))
    from scene_generation_amd.layers import build_mlp
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        build_mlp([3, 4])(torch.ones(2, 3))


def test_flag_surface_matches_reference(golden):
    ref = json.loads(str(golden('args_defaults')['json']))
    mine = vars(parser.parse_args([]))
    assert set(ref) <= set(mine) and set(mine) - set(ref) <= {'vgg_weights'}      # one documented extension flag
    assert mine['vgg_weights'] is None
    for k, v in ref.items():
        if k == 'output_dir':
            continue
        got = mine[k]
        got = list(got) if isinstance(got, tuple) else got
        assert got == v, (k, got, v)
    assert int_tuple('3,4') == (3, 4) and str_tuple('a,b') == ('a', 'b') and bool_flag('1') is True
    with pytest.raises(ValueError):
        bool_flag('yes')


def test_vector_pool_plan_replays_reference_loop():
    """plan_pool_query + (gather, scatter) == the sequential loop of utils.py:67-90, incl. RNG consumption."""
    for pool_size in (1, 2, 5):
        random.seed(3)
        ref = O.VectorPool(pool_size)
        st = random.getstate()
        g = torch.Generator().manual_seed(pool_size)
        batches = [(torch.randint(0, 4, (9,), generator=g), torch.randn(9, 3, generator=g)) for _ in range(8)]
        want = [ref.query(o, v) for o, v in batches]
        end_state = random.getstate()
        random.setstate(st)
        pool = torch.zeros(4, pool_size, 3)
        fill = {}
        for (objs, vec), w in zip(batches, want):
            cls = objs.tolist()
            kind, idx, slot = plan_pool_query(cls, fill, pool_size)
            out = torch.stack([pool[c, j] if k else vec[j] for c, k, j in zip(cls, kind, idx)])
            for i, (c, sl) in enumerate(zip(cls, slot)):
                if sl >= 0:
                    pool[c, sl] = vec[i]
            assert torch.equal(out, w)
        assert random.getstate() == end_state
        for c in range(4):
            for j, v in enumerate(ref.vectors.get(c, [])):
                assert torch.equal(pool[c, j], v)


def test_loss_manager_is_lazy_but_equivalent():
    L, R = LossManager(), O.LossManager()
    for i, w in enumerate([1.0, 0.5, 10]):
        t = torch.tensor(float(i + 1), requires_grad=True)
        L.add_loss(t * 2, 'l%d' % i, w)
        R.add_loss(t * 2, 'l%d' % i, w)
    assert dict(L.items()) == dict(R.items())
    assert float(L.total_loss) == float(R.total_loss)


def test_flat_params_rehoming_and_inplace_grad_accumulation():
    from scene_generation_amd.optim import FlatParams
    m = fill_deterministic(torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.ReLU(), torch.nn.Linear(4, 2)))
    before = [p.detach().clone() for p in m.parameters()]
    fp = FlatParams(m.parameters())
    assert fp.numel >= sum(p.numel() for p in m.parameters())
    assert all(o % FlatParams.ALIGN == 0 for o in fp.offsets)      # aligned slices => vector weight loads
    for p, b in zip(m.parameters(), before):
        assert torch.equal(p, b)
        assert p.data_ptr() >= fp.flat.data_ptr() and p.data_ptr() < fp.flat.data_ptr() + fp.numel * 4
    m(torch.ones(3, 5)).sum().backward()
    m(torch.ones(3, 5)).sum().backward()            # second backward accumulates in place
    for i, p in enumerate(m.parameters()):
        assert p.grad.data_ptr() == fp.grad_view(i).data_ptr()
    ref = fill_deterministic(torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.ReLU(), torch.nn.Linear(4, 2)))
    (ref(torch.ones(3, 5)).sum() * 2).backward()
    flat_ref = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    assert torch.allclose(fp.packed(fp.grad), flat_ref)


def test_synthetic_batches_follow_the_collate_contract():
    for name, (Omax, Tmax) in {'c1': (20, 32), 'c2': (288, 512), 'c5': (1056, 3072)}.items():
        b = make_config_batch(name)
        N = b.imgs.size(0)
        assert b.objs.numel() <= Omax and b.triples.size(0) <= Tmax
        assert bool((b.obj_to_img[1:] >= b.obj_to_img[:-1]).all()) and int(b.obj_to_img.max()) == N - 1
        assert b.masks.dtype == torch.int64 and b.triples.dtype == torch.int64 and b.attributes.shape[1] == 35
        # every triple stays inside its image (block-diagonal graph, coco.py:527-529)
        assert torch.equal(b.obj_to_img[b.triples[:, 0]], b.obj_to_img[b.triples[:, 2]])
        last = torch.cat([b.obj_to_img[1:] != b.obj_to_img[:-1], torch.tensor([True])])
        assert bool((b.objs[last] == 0).all()) and bool((b.boxes[last] == torch.tensor([0., 0, 1, 1])).all())
    b = make_config_batch('c2')
    parts = [shard_batch(b, r, 4) for r in range(4)]
    assert sum(p.objs.numel() for p in parts) == b.objs.numel()
    assert torch.equal(torch.cat([p.imgs for p in parts]), b.imgs)
    for p in parts:
        assert int(p.obj_to_img.min()) == 0 and int(p.triples[:, [0, 2]].max()) < p.objs.numel()


def test_active_layout_channels():
    """host index plumbing for the channel-sparse first conv: classes of the image's objects + the dense block."""
    from scene_generation_amd.utils import active_layout_channels
    objs = [3, 7, 3, 171, 0, 5]
    o2i = [0, 0, 0, 1, 1, 2]
    cl, cc = active_layout_channels(objs, o2i, 3, 172, 4)
    assert cl.dtype == np.int32 and cc.dtype == np.int32 and cl.shape == (3, 2 + 4)
    assert cc.tolist() == [6, 6, 5]
    assert cl[0].tolist() == [3, 7, 172, 173, 174, 175]
    assert cl[1].tolist() == [0, 171, 172, 173, 174, 175]
    assert cl[2, :5].tolist() == [5, 172, 173, 174, 175]
    with pytest.raises(ValueError):
        active_layout_channels([172], [0], 1, 172, 4)


def test_factored_layout_lists_host_logic():
    """index plumbing of the factored layout convs: per-image plane lists, with the channels of a concatenated second
    source sitting at channel ids J.. and at list positions cnt[n].. (ops.FactoredLayout.lists)"""
    from scene_generation_amd.ops import FactoredLayout
    Z = torch.zeros(3, 4, 2, 2)                      # N=3 images, J=4 planes
    f = FactoredLayout(Z, torch.tensor([1, 2, 3, 4, 5, 6, 7]), torch.zeros(7, 2), 10,
                       torch.tensor([0, 0, 1, 1, 1, 1, 2]), torch.tensor([0, 1, 0, 1, 2, 3, 0]), [2, 4, 1])
    cl, cc, ep, L = f.lists(0)
    assert L == 4 and cc.tolist() == [2, 4, 1]
    assert cl[0, :2].tolist() == [0, 1] and cl[1].tolist() == [0, 1, 2, 3] and cl[2, :1].tolist() == [0]
    cl, cc, ep, L = f.lists(3)
    assert L == 7 and cc.tolist() == [5, 7, 4]
    assert cl[0, :5].tolist() == [0, 1, 4, 5, 6] and cl[2, :4].tolist() == [0, 4, 5, 6]
    assert ep.tolist() == [[2, 3, 4], [4, 5, 6], [1, 2, 3]]
    assert f.lists(3) is f.lists(3)                  # cached per layout
    g = f.detached()
    assert g.lists(3) is f.lists(3) and not g.repr.requires_grad


def test_factored_layout_stacked_is_the_batch_concatenation():
    """ops.FactoredLayout.stacked (the image discriminator's real + wrong-texture passes as one 2N batch): same planes and
    objects twice, images N.. carry the second layout's appearance vectors; everything detached."""
    from scene_generation_amd.ops import FactoredLayout
    Z = torch.arange(3 * 4 * 2 * 2, dtype=torch.float32).view(3, 4, 2, 2)
    objs, o2i, pidx = torch.tensor([1, 2, 3, 4, 5, 6, 7]), torch.tensor([0, 0, 1, 1, 1, 1, 2]), torch.tensor([0, 1, 0, 1, 2, 3, 0])
    ra, rb = torch.rand(7, 2, requires_grad=True), torch.rand(7, 2)
    a = FactoredLayout(Z, objs, ra, 10, o2i, pidx, [2, 4, 1])
    b = FactoredLayout(Z, objs, rb, 10, o2i, pidx, [2, 4, 1], a.seg)
    f = FactoredLayout.stacked(a, b)
    assert f.Z.shape == (6, 4, 2, 2) and torch.equal(f.Z[:3], Z) and torch.equal(f.Z[3:], Z)
    assert f.objs.tolist() == objs.tolist() * 2 and f.counts_host == [2, 4, 1, 2, 4, 1]
    assert f.img_idx.tolist() == o2i.tolist() + (o2i + 3).tolist() and f.plane_idx.tolist() == pidx.tolist() * 2
    assert f.seg.tolist() == [0, 2, 6, 7, 9, 13, 14]
    assert torch.equal(f.repr[:7], ra.detach()) and torch.equal(f.repr[7:], rb) and not f.repr.requires_grad
    cl, cc, ep, L = f.lists(3)
    cl1, cc1, ep1, L1 = a.lists(3)
    assert L == L1 and cc.tolist() == cc1.tolist() * 2 and torch.equal(cl[:3], cl1) and torch.equal(cl[3:], cl1)
    with pytest.raises(AssertionError):
        FactoredLayout.stacked(a, FactoredLayout(Z.clone(), objs, rb, 10, o2i, pidx, [2, 4, 1]))      # other planes: not a twin


def test_weighted_sum_and_lazy_loss_manager():
    from scene_generation_amd.utils import weighted_sum
    a, b, c = torch.tensor(1.5, requires_grad=True), torch.tensor(-2.0, requires_grad=True), torch.tensor(0.25)
    s = weighted_sum([a, b, c], [2.0, 0.5, 4.0])
    assert abs(float(s) - (3.0 - 1.0 + 1.0)) < 1e-6
    s.backward()
    assert float(a.grad) == 2.0 and float(b.grad) == 0.5
    L = LossManager()
    L.add_loss(torch.tensor(2.0), 'x', 3.0)
    L.add_loss(torch.tensor(1.0), 'logged_only', 5.0, use_loss=False)
    L.add_loss(torch.tensor(4.0), 'y')
    assert abs(float(L.total_loss) - 10.0) < 1e-6
    assert dict(L.items()) == {'x': 6.0, 'logged_only': 5.0, 'y': 4.0}
    L.add_loss(torch.tensor(1.0), 'z', 2.0)          # adding after a read invalidates the cached sum
    assert abs(float(L.total_loss) - 12.0) < 1e-6


def test_checkpoint_round_trip_and_torch_adam_compatibility(tmp_path):
    """SURVEY 8f rank 4 (host logic only): Trainer.save_checkpoint / restore_checkpoint round-trip every network and all
    four optimisers (trainer.py:136-203 schema), and the saved optimiser state is what torch.optim.Adam -- the reference's
    optimiser -- loads (the oracle Trainer restores it)."""
    from scene_generation_amd.trainer import Trainer
    argv = ['--image_size', '32,32', '--batch_size', '3', '--vgg_features_weight', '0', '--output_dir', str(tmp_path),
            '--n_downsample_global', '2', '--gconv_hidden_dim', '64', '--gconv_num_layers', '3', '--mask_size', '8',
            '--ndf', '8', '--ndf_mask', '8', '--crop_size', '16', '--d_obj_arch', 'C4-8-2,C4-16-2', '--pool_size', '2']
    args = parser.parse_args(argv)
    vocab = make_vocab(12, 4, 35)
    tr = Trainer(args, vocab, device='cpu')
    nets = lambda t: [t.model, t.netD, t.obj_discriminator, t.mask_discriminator]
    opts = lambda t: [t.optimizer, t.optimizer_d_img, t.optimizer_d_obj, t.optimizer_d_mask]
    for m in nets(tr):
        fill_deterministic(m)
    g = torch.Generator().manual_seed(0)
    for o in opts(tr):                                   # pretend three Adam steps happened
        o.exp_avg.copy_(torch.randn(o.exp_avg.shape, generator=g))
        o.exp_avg_sq.copy_(torch.rand(o.exp_avg_sq.shape, generator=g))
        o.steps = [3] * len(o.steps)
    path = tr.save_checkpoint({}, 7, args, 2)
    sd = torch.load(path, weights_only=False)
    assert sd['counters'] == {'t': 7, 'epoch': 2}
    tr2 = Trainer(args, vocab, device='cpu')
    tr2.restore_checkpoint(sd)
    for a, b in zip(nets(tr), nets(tr2)):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa.keys()) == list(sb.keys())
        assert all(torch.equal(sa[k], sb[k]) for k in sa)
    for a, b in zip(opts(tr), opts(tr2)):
        assert a.steps == b.steps
        assert torch.equal(a.fp.packed(a.exp_avg), b.fp.packed(b.exp_avg))
        assert torch.equal(a.fp.packed(a.exp_avg_sq), b.fp.packed(b.exp_avg_sq))
    # the same file restores the torch-Adam based oracle (= the reference's optimiser format)
    ref = O.Trainer(args, vocab)
    ref.model.load_state_dict(sd['model_state'])
    ref.optimizer.load_state_dict(sd['optim_state'])
    p0 = next(iter(ref.model.parameters()))
    st = ref.optimizer.state[p0]
    assert float(st['step']) == 3.0
    assert torch.equal(st['exp_avg'], tr.optimizer.exp_avg[:p0.numel()].view(p0.shape))


def test_collate_adapter_contract_and_host_summaries():
    """pipeline.py: the collate contract of data/coco.py:517-534 is checked on the host, the host lists the step needs are
    derived without touching the device, and on a CPU device the prefetcher is a pass-through iterator."""
    import pytest as _pt
    from scene_generation_amd.pipeline import DeviceBatchPrefetcher, validate_collated, segment_offsets
    from scene_generation_amd.synthetic import make_batch, Batch
    hb = make_batch(N=5, min_objs=2, max_objs=6, size=16, mask_size=8, seed=3)
    o2i = validate_collated(hb)
    assert o2i == hb.obj_to_img.tolist()
    seg = segment_offsets(o2i, 5)
    assert seg[0] == 0 and seg[-1] == hb.objs.numel() and all(b > a for a, b in zip(seg, seg[1:]))
    for n in range(5):
        assert all(i == n for i in o2i[seg[n]:seg[n + 1]])
    out = list(DeviceBatchPrefetcher([hb, hb], 'cpu'))
    assert len(out) == 2
    db = out[0]
    assert db.num_images == 5 and db.objs_host == hb.objs.tolist() and db.obj_to_img_host == o2i
    assert db.seg_offsets_host == seg and all(torch.equal(a, b) for a, b in zip(db.batch, hb))
    # violations of the contract fail on the host, before anything is launched
    bad = hb._replace(obj_to_img=hb.obj_to_img.flip(0))
    with _pt.raises(ValueError):
        validate_collated(bad)
    hole = hb.obj_to_img.clone()
    hole[hole == 2] = 3                                   # image 2 owns no object
    with _pt.raises(ValueError):
        validate_collated(hb._replace(obj_to_img=hole))
    tri = hb.triples.clone()
    tri[0, 2] = hb.objs.numel()                            # object id outside the batch
    with _pt.raises(ValueError):
        validate_collated(hb._replace(triples=tri))
    with _pt.raises(ValueError):
        validate_collated(hb._replace(boxes=hb.boxes[:-1]))


def test_side_stream_fork_is_a_no_op_without_a_gpu():
    """streams.fork on CPU tensors / with the switch off: every branch runs inline on the caller's (only) stream, join_all
    does nothing -- the discriminators call it unconditionally."""
    import torch
    from scene_generation_amd import streams
    order = []
    with streams.fork(torch.device('cpu'), 'imgD') as f:
        for i in range(3):
            with f.branch(i):
                order.append(i)
    assert order == [0, 1, 2] and not f.on and f.used == []
    streams.join_all(torch.device('cpu'))
    saved = streams.ENABLED
    try:
        streams.ENABLED = True                     # opting in does not make a CPU device usable
        assert not streams.fork(torch.device('cpu'), 'imgD').on
    finally:
        streams.ENABLED = saved


def test_stream_groups_host_logic_and_lazy_zero_is_device_only():
    """Round 6: side-stream GROUPS (streams.py) and FusedAdam(lazy_zero).  On a CPU-only host every fork is off whatever the groups
    say (branch / produced / join / join_all with an exclude list are no-ops), the default groups are the ones the docs name, and
    lazy_zero silently stays off for host tensors (zero_grad() keeps filling, p.grad stays attached)."""
    import torch.nn as nn
    from scene_generation_amd import ops, streams
    from scene_generation_amd.optim import FusedAdam
    if 'SG_STREAM_GROUPS' not in os.environ:
        want = {'front', 'mstep', 'imgD', 'objD', 'adam'} if os.environ.get('GPU_MAX_HW_QUEUES', '').strip() in ('', '4') else {'front', 'mstep'}
        assert streams.GROUPS == want
    assert streams.group_on('front') == ('front' in streams.GROUPS or streams.ENABLED)
    assert streams.group_on('no-such-group') == streams.ENABLED
    cpu = torch.device('cpu')
    with streams.fork(cpu, 'front') as fk:
        assert not fk.on
        with fk.branch(1, reads=(torch.zeros(2),)):
            y = torch.ones(3) * 2
            fk.produced((y,))
        fk.join()
    assert float(y.sum()) == 6.0
    streams.join_all(cpu, exclude=('front',))
    lin = nn.Linear(3, 2)
    opt = FusedAdam(lin.parameters(), lr=1e-2, lazy_zero=True)
    assert opt.lazy_zero is False and opt.join_exclude == ()
    saved = ops.fill_
    ops.fill_ = lambda t, v: t.fill_(v)
    try:
        lin(torch.ones(4, 3)).sum().backward()
        opt.zero_grad()
        assert all(p.grad is not None and float(p.grad.abs().sum()) == 0.0 for p in lin.parameters())
        opt.finalize_grads()                       # idempotent no-op outside lazy mode
    finally:
        ops.fill_ = saved


def test_winograd_f24_constants_are_exact_and_adjoint():
    """The F(2x2,4x4) matrices hard-coded in csrc/igemm.hip (w24_bt / w24_g / w24_at / w24_a / w24_gt; derivation:
    tools/winograd_f24.py): y = A^T[(G g G^T) . (B^T d B)]A is the 4x4 correlation of a 5x5 patch exactly (float64), and the
    weight-gradient form G^T[(A dy A^T) . (B^T d B)]G is its adjoint."""
    import numpy as np
    AT = np.array([[1, 1, 1, 1, 0], [0, 1, -1, -2, 1]], dtype=np.float64)
    G = np.array([[-1 / 2, 0, 0, 0], [1 / 6, 1 / 6, 1 / 6, 1 / 6], [1 / 2, -1 / 2, 1 / 2, -1 / 2], [-1 / 6, 1 / 3, -2 / 3, 4 / 3],
                  [0, 0, 0, 1]], dtype=np.float64)
    BT = np.array([[-2, -1, 2, 1, 0], [0, 2, 3, 1, 0], [0, -2, 1, 1, 0], [0, -1, 0, 1, 0], [0, -2, -1, 2, 1]], dtype=np.float64)
    # the device helpers, transcribed: any edit of the kernel constants has to show up here
    def w24_bt(d): return np.array([-2 * d[0] - d[1] + 2 * d[2] + d[3], 2 * d[1] + 3 * d[2] + d[3], -2 * d[1] + d[2] + d[3],
                                    d[3] - d[1], -2 * d[1] - d[2] + 2 * d[3] + d[4]])
    def w24_g(g): return np.array([-0.5 * g[0], (g[0] + g[1] + g[2] + g[3]) / 6, (g[0] - g[1] + g[2] - g[3]) * 0.5,
                                   (-g[0] + 2 * g[1] - 4 * g[2] + 8 * g[3]) / 6, g[3]])
    def w24_at(m): return np.array([m[0] + m[1] + m[2] + m[3], m[1] - m[2] - 2 * m[3] + m[4]])
    def w24_a(y): return np.array([y[0], y[0] + y[1], y[0] - y[1], y[0] - 2 * y[1], y[1]])
    def w24_gt(t): return np.array([-0.5 * t[0] + (t[1] - t[3]) / 6 + 0.5 * t[2], (t[1] + 2 * t[3]) / 6 - 0.5 * t[2],
                                    (t[1] - 4 * t[3]) / 6 + 0.5 * t[2], (t[1] + 8 * t[3]) / 6 - 0.5 * t[2] + t[4]])
    rng = np.random.RandomState(3)
    e5, e4, e2 = np.eye(5), np.eye(4), np.eye(2)
    assert np.allclose(np.stack([w24_bt(e) for e in e5], 1), BT) and np.allclose(np.stack([w24_g(e) for e in e4], 1), G)
    assert np.allclose(np.stack([w24_at(e) for e in e5], 1), AT)
    assert np.allclose(np.stack([w24_a(e) for e in e2], 1), AT.T) and np.allclose(np.stack([w24_gt(e) for e in e5], 1), G.T)
    g, d, dy = rng.randn(4, 4), rng.randn(5, 5), rng.randn(2, 2)
    y = AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T
    ref = np.array([[(g * d[i:i + 4, j:j + 4]).sum() for j in range(2)] for i in range(2)])
    assert np.abs(y - ref).max() < 1e-12
    gw = G.T @ ((AT.T @ dy @ AT) * (BT @ d @ BT.T)) @ G               # d<dy, y>/dg
    gref = np.array([[sum(dy[i, j] * d[i + a, j + b] for i in range(2) for j in range(2)) for b in range(4)] for a in range(4)])
    assert np.abs(gw - gref).max() < 1e-12
    # data gradient = the forward form on dy with the rotated filter and padding 3 - pad (here: full correlation, pad 3)
    dyp = np.zeros((8, 8)); dyp[3:5, 3:5] = dy
    gd = np.array([[(g[::-1, ::-1] * dyp[i:i + 4, j:j + 4]).sum() for j in range(5)] for i in range(5)])
    gdref = np.zeros((5, 5))
    for i in range(2):
        for j in range(2):
            gdref[i:i + 4, j:j + 4] += dy[i, j] * g
    assert np.abs(gd - gdref).max() < 1e-12


def test_deepcopy_of_global_generator_rebinds_graphed_segments():
    """a deep-copied GlobalGenerator (EMA copy / snapshot) must run ITS OWN layers in the graphed tail segments and
    list ITS OWN parameters there, not the original's."""
    import copy
    from scene_generation_amd.generators import define_G
    g = define_G(6, 3, 4, n_downsample_global=2, n_blocks_global=3)
    g2 = copy.deepcopy(g)
    own = {id(p) for p in g2.parameters()}
    orig = {id(p) for p in g.parameters()}
    assert len(g2._tail) == len(g._tail) and own.isdisjoint(orig)
    for seg, seg0 in zip(g2._tail, g._tail):
        assert seg is not seg0 and seg.fn.seq is g2.model and seg0.fn.seq is g.model
        assert (seg.fn.a, seg.fn.b) == (seg0.fn.a, seg0.fn.b)
        assert seg.params and all(id(p) in own for p in seg.params)
        assert all(m2 is not m for m2, m in zip(seg.modules, seg0.modules))
        assert all(any(m2 is x for x in g2.model) for m2 in seg.modules)


def test_build_cnn_and_resnet_block_cover_every_reference_variant():
    """Surface the reference builds (layers.py:181-189,243-262) that raised NotImplementedError before round 4: 'P<k>' pooling
    of either kind and the 'zero' / 'replicate' ResnetBlock paddings -- module structure and state_dict keys (the numerics run
    on the GPU: tests/test_gpu_parity.py)."""
    import torch.nn as nn
    from scene_generation_amd.layers import (build_cnn, ResnetBlock, get_norm_layer, MaxPool2d, AvgPool2d, ReplicationPad2d,
                                            ReflectionPad2d)
    m, c = build_cnn('I5,C3-8,P3,C3-16,P2', pooling='avg')
    assert c == 16 and [type(x) for x in m if isinstance(x, (MaxPool2d, AvgPool2d))] == [AvgPool2d, AvgPool2d]
    assert [x.kernel_size for x in m if isinstance(x, AvgPool2d)] == [3, 2]
    m, _ = build_cnn('C3-8,P4', pooling='max')
    assert isinstance(m[-1], MaxPool2d) and m[-1].kernel_size == 4
    with pytest.raises(ValueError):
        build_cnn('C3-8,P2', pooling='median')
    with pytest.raises(ValueError):
        build_cnn('C3-8,X2')
    norm = get_norm_layer('instance')
    keys = {}
    for pt in ('reflect', 'replicate', 'zero'):
        b = ResnetBlock(8, pt, norm)
        keys[pt] = sorted(b.state_dict())
    assert keys['reflect'] == keys['replicate'] == ['conv_block.1.bias', 'conv_block.1.weight', 'conv_block.5.bias',
                                                     'conv_block.5.weight']
    assert keys['zero'] == ['conv_block.0.bias', 'conv_block.0.weight', 'conv_block.3.bias', 'conv_block.3.weight']
    assert isinstance(ResnetBlock(8, 'replicate', norm).conv_block[0], ReplicationPad2d)
    assert isinstance(ResnetBlock(8, 'reflect', norm).conv_block[0], ReflectionPad2d)
    assert ResnetBlock(8, 'zero', norm).conv_block[0].padding == (1, 1)
    with pytest.raises(NotImplementedError):
        ResnetBlock(8, 'circular', norm)


def test_library_options_table_and_no_stray_getenv():
    """VERDICT r3 (library hygiene): every tuning switch of libsg2im_hip.so lives in ONE table that is initialised from the
    environment when the library is loaded and changed through sg_set_option afterwards -- no getenv() anywhere else in the
    kernels' translation units."""
    import glob
    import subprocess
    from scene_generation_amd import _hip
    opts = _hip.options()
    assert opts['linear_skinny'] == (2048, 2048) and opts['tile'] == (-1, -1) and len(opts) == _hip.lib().sg_num_options()
    _hip.set_option('t128_min', 500)
    assert _hip.get_option('t128_min') == 500
    _hip.set_option('t128_min', opts['t128_min'][1])
    with pytest.raises(RuntimeError):
        _hip.set_option('no_such_switch', 1)
    csrc = os.path.join(ROOT, 'scene_generation_amd', 'csrc')
    for path in glob.glob(os.path.join(csrc, '*.hip')) + glob.glob(os.path.join(csrc, '*.h')):
        text = open(path).read()
        n = text.count('getenv(')
        assert n == (2 if path.endswith('runtime.hip') else 0), '%s: %d getenv() calls' % (path, n)
    # a fresh process picks the environment up at load time
    out = subprocess.run([sys.executable, '-c', 'from scene_generation_amd import _hip; print(_hip.get_option("bn_blocks"))'],
                         env=dict(os.environ, SG_BN_BLOCKS='123'), cwd=ROOT, capture_output=True, text=True)
    assert out.stdout.strip() == '123', out.stderr


def test_library_kernels_fit_their_register_budget():
    """Static check of the built gfx950 code objects (tools/isa_report.py; no GPU): no product kernel spills registers to
    scratch -- three instantiations of the register-resident InstanceNorm for 256x256 planes excepted -- and the GEMM tile
    classes keep the resident-wave counts the launch plans are built around."""
    import shutil
    from tools import isa_report
    if not (os.path.isfile(isa_report.DEFAULT_LIB) and shutil.which('objcopy')
            and os.path.isfile(os.path.join(isa_report.LLVM, 'clang-offload-bundler'))):
        pytest.skip('needs the built library and the ROCm LLVM tools')
    ks = isa_report.kernels()
    assert len(ks) > 300
    allowed = ('instnorm_fwd_reg_kernel<1024, 64>', 'instnorm_bwd_reg_kernel<1024, 32>', 'instnorm_bwd_reg_kernel<1024, 64>')
    spills = [(k['name'], k['scratch']) for k in ks if k['scratch'] and not k['name'].startswith(allowed)]
    assert not spills, spills
    dense = [k for k in ks if k['name'].startswith('igemm_kernel<T128x128')]
    assert dense and all(k['waves_per_simd'] >= 2 and k['lds'] <= 80 * 1024 for k in dense)      # two workgroups per CU
    t64 = [k for k in ks if k['name'].startswith('igemm_kernel<T64x64')]
    assert t64 and all(k['waves_per_simd'] >= 3 for k in t64)
    assert sum(1 for k in t64 if k['waves_per_simd'] >= 5) >= 0.8 * len(t64)       # the fixed-tap / K-contiguous forms: 6
    head = [k for k in ks if k['name'].startswith('smallm_fwd_kernel<7, 3>')]
    assert head and head[0]['vgpr'] <= 128                                          # 512 threads: two workgroups per CU


def test_threaded_prefetcher_order_and_error_propagation():
    """the background staging thread of DeviceBatchPrefetcher (the default on a GPU; forced here on CPU tensors): batches come out
    in source order with their host summaries, the iterator ends cleanly, and a batch that violates the collate contract raises
    in the CONSUMER, at its position in the stream."""
    from scene_generation_amd.pipeline import DeviceBatchPrefetcher
    from scene_generation_amd.synthetic import make_batch, Batch
    hbs = [make_batch(N=3, min_objs=2, max_objs=4, size=16, mask_size=8, num_objs=12, num_preds=4, seed=s) for s in range(5)]
    out = list(DeviceBatchPrefetcher(hbs, 'cpu', threaded=True, depth=2))
    assert len(out) == 5
    for db, hb in zip(out, hbs):
        assert torch.equal(db.batch.imgs, hb.imgs) and db.objs_host == hb.objs.tolist() and db.num_images == 3
        assert db.seg_offsets_host[-1] == hb.objs.numel()
    bad = Batch(*hbs[1])._replace(obj_to_img=hbs[1].obj_to_img.flip(0))
    it = DeviceBatchPrefetcher([hbs[0], bad, hbs[2]], 'cpu', threaded=True)
    assert torch.equal(next(it).batch.objs, hbs[0].objs)
    with pytest.raises(ValueError):
        next(it)
    with pytest.raises(StopIteration):
        next(it)


def test_abandoned_prefetcher_releases_its_worker_and_source():
    """the staging thread reads depth + 1 batches ahead; a consumer that stops early must not leave it parked in a
    blocking put forever.  Dropping the iterator (or leaving its ``with`` block, or close()) ends the thread within a fraction of
    a second; an exception raised by the source while the queue is FULL still reaches the consumer."""
    import gc
    import threading
    import time
    from scene_generation_amd.pipeline import DeviceBatchPrefetcher
    from scene_generation_amd.synthetic import make_batch
    hbs = [make_batch(N=2, min_objs=2, max_objs=3, size=16, mask_size=4, seed=s) for s in range(8)]

    def workers():
        return [t for t in threading.enumerate() if t.name == 'sg-prefetch' and t.is_alive()]

    def wait_gone():
        for _ in range(40):
            if not workers():
                return True
            time.sleep(0.05)
        return False

    assert wait_gone()
    p = DeviceBatchPrefetcher(hbs, 'cpu', threaded=True, depth=1)
    next(p)
    time.sleep(0.2)                       # the worker is now blocked on the full queue, two batches ahead
    assert workers()
    del p
    gc.collect()
    assert wait_gone(), 'an abandoned prefetcher kept its staging thread'
    with DeviceBatchPrefetcher(hbs, 'cpu', threaded=True, depth=1) as q:
        next(q)
        assert workers()
    assert wait_gone()

    def bad_source():
        yield hbs[0]
        yield hbs[1]
        raise RuntimeError('loader died')
    it = DeviceBatchPrefetcher(bad_source(), 'cpu', threaded=True, depth=1)
    time.sleep(0.3)                       # queue full (batch 0), the worker holds batch 1 and then hits the exception
    next(it)
    next(it)
    with pytest.raises(RuntimeError, match='loader died'):
        next(it)
    assert wait_gone()


def test_deferred_grad_scale_never_outlives_its_step():
    """FusedAdam.grad_scale (the 1 / world a data-parallel reduce hands over instead of scaling 765 MB) is reset by
    zero_grad() and by a step() that raises -- a later step without a reduce must see 1.0 -- and ``scaled_grad`` shows a hook the
    gradient the step will apply."""
    from scene_generation_amd import ops
    from scene_generation_amd.optim import FusedAdam
    m = fill_deterministic(torch.nn.Linear(4, 3))
    opt = FusedAdam(m.parameters(), lr=1e-3)
    m(torch.ones(2, 4)).sum().backward()
    opt.grad_scale = 0.125
    g = opt.fp.grad.clone()
    assert torch.allclose(opt.scaled_grad(), g * 0.125) and torch.allclose(opt.scaled_grad(0), opt.fp.grad_view(0) * 0.125)
    fill = ops.fill_
    ops.fill_ = lambda t, v: t.fill_(v)              # (the device fill of zero_grad(): host bookkeeping is what is tested here)
    try:
        opt.zero_grad()
    finally:
        ops.fill_ = fill
    assert opt.grad_scale == 1.0
    m(torch.ones(2, 4)).sum().backward()
    opt.grad_scale = 0.5

    def boom():
        raise RuntimeError('hook failed')
    opt.pre_step_hooks.append(boom)
    with pytest.raises(RuntimeError, match='hook failed'):
        opt.step()
    assert opt.grad_scale == 1.0


def test_fused_conv_instnorm_is_not_taken_for_affine_or_tracking_norms():
    """FusedSequential fuses ReflectionPad2d + Conv2d + InstanceNorm2d only for the plain (affine=False, no running
    statistics) norm; any other InstanceNorm2d must reach the module's own forward (which rejects it) instead of being
    computed as plain InstanceNorm.  Checked on the fusion predicate's host side: with an affine norm the fusable query is
    never consulted."""
    from scene_generation_amd import layers, ops
    calls = []
    orig = ops.conv_instnorm_fusable
    ops.conv_instnorm_fusable = lambda *a, **k: calls.append(1) or False
    try:
        for kw, consulted in ((dict(), True), (dict(affine=True), False), (dict(track_running_stats=True), False)):
            del calls[:]
            seq = layers.FusedSequential(layers.ReflectionPad2d(1), layers.Conv2d(4, 4, 3), layers.InstanceNorm2d(4, **kw))
            with pytest.raises(Exception):          # CPU tensors: whatever runs next fails loudly (no CPU compute path)
                seq(torch.zeros(1, 4, 8, 8))
            assert bool(calls) == consulted, kw
    finally:
        ops.conv_instnorm_fusable = orig


def test_winograd_f43_matrices_in_the_kernel_source_satisfy_the_identities():
    """The F(4x4,3x3) transform matrices as WRITTEN in csrc/igemm.hip (w43_bt / w43_g / w43_at), parsed from the source: forward
    y = A^T [(G g G^T) (.) (B^T d B)] A equals the 3x3 correlation of a 6x6 patch, and the data / weight gradient forms the kernels
    use -- B [U (.) (A gy A^T)] B^T and G^T [(A gy A^T) (.) (B^T d B)] G -- are its exact adjoints (float64).  Guards the constants
    against an edit that the GPU tolerances might absorb."""
    import re
    import numpy as np
    src = open(os.path.join(ROOT, 'scene_generation_amd', 'csrc', 'igemm.hip')).read()

    def matrix(fn, rows, cols):
        body = re.search(r'%s\(int i, int j\) \{\s*constexpr float m\[%d\]\[%d\] = (\{.*?\});' % (fn, rows, cols), src, re.S).group(1)
        vals = [eval(v.replace('f', ''), {'__builtins__': {}}) for v in re.findall(r'-?\d+\.?\d*f(?:\s*/\s*\d+\.?\d*f)?', body)]
        assert len(vals) == rows * cols, (fn, len(vals))
        return np.array(vals, dtype=np.float64).reshape(rows, cols)

    BT, G, AT = matrix('w43_bt', 6, 6), matrix('w43_g', 6, 3), matrix('w43_at', 4, 6)
    rng = np.random.RandomState(0)
    d, g, gy = rng.randn(6, 6), rng.randn(3, 3), rng.randn(4, 4)
    U, V = G @ g @ G.T, BT @ d @ BT.T
    y = AT @ (U * V) @ AT.T
    ref = np.array([[(g * d[i:i + 3, j:j + 3]).sum() for j in range(4)] for i in range(4)])
    assert np.abs(y - ref).max() < 1e-12
    Yt = AT.T @ gy @ AT
    gd = BT.T @ (U * Yt) @ BT                       # d/dd of sum(y * gy)
    gd_ref = np.zeros((6, 6))
    for i in range(4):
        for j in range(4):
            gd_ref[i:i + 3, j:j + 3] += gy[i, j] * g
    assert np.abs(gd - gd_ref).max() < 1e-12
    gw = G.T @ (Yt * V) @ G                         # d/dg of sum(y * gy)
    gw_ref = np.array([[(gy * d[a:a + 4, b:b + 4]).sum() for b in range(3)] for a in range(3)])
    assert np.abs(gw - gw_ref).max() < 1e-12


def test_cpu_quota_parsing_and_thread_cap(tmp_path, monkeypatch):
    """utils.cpu_quota reads the cgroup v2 quota in CPUs (None when unlimited / absent); respect_cpu_quota lowers torch's intra-op
    thread count to it once, never raises it, and SG_KEEP_TORCH_THREADS=1 opts out (round 6: a 128-thread OpenMP burst inside a
    16-CPU quota froze the launching thread for 20-60 ms of every 100 ms)."""
    import builtins
    import torch
    from scene_generation_amd import utils
    real_open = builtins.open
    content = {'v': '1600000 100000\n'}

    def fake_open(path, *a, **k):
        if path == '/sys/fs/cgroup/cpu.max':
            import io
            return io.StringIO(content['v'])
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, 'open', fake_open)
    assert utils.cpu_quota() == 16.0
    content['v'] = 'max 100000\n'
    assert utils.cpu_quota() is None
    content['v'] = '200000 100000\n'
    n0 = torch.get_num_threads()
    try:
        monkeypatch.setattr(utils, '_QUOTA_APPLIED', [False])
        monkeypatch.setenv('SG_KEEP_TORCH_THREADS', '1')
        assert utils.respect_cpu_quota(verbose=False) == n0
        monkeypatch.setenv('SG_KEEP_TORCH_THREADS', '0')
        got = utils.respect_cpu_quota(verbose=False)
        assert got == min(n0, 2) and torch.get_num_threads() == got
        content['v'] = '100000 100000\n'
        assert utils.respect_cpu_quota(verbose=False) == got          # once per process
    finally:
        torch.set_num_threads(n0)
