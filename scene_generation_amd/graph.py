"""Scene-graph convolution on MI355X (surface of /root/reference/scene_generation/graph.py).

GraphTripleConv.forward = 7 HIP launches instead of ~25 ATen kernels:
  gather+concat rows -> f32-MFMA GEMM(+bias+ReLU) x2 -> deterministic segmented pool (+avg, + new_p split)
  -> f32-MFMA GEMM(+bias+ReLU) x2,
with the destination-major CSR of the triples built ONCE per forward of the stack (Model / GraphTripleConvNet hand it down).
The pool walks the CSR in (pass, t) order, i.e. exactly the order CPU scatter_add applies the updates at graph.py:98-101,
so ``pooled`` is bit-identical to the reference given identical net1 outputs.
SG_GCONV_FUSED=1 switches to TWO fused launches per layer (csrc/gconv.hip: [row gather -> GEMM -> ReLU -> GEMM -> ReLU] with
the hidden block resident in LDS, [segmented pool -> GEMM -> ReLU -> GEMM -> ReLU], new_p a column view of new_t) --
bit-identical results and gradients, but measured slower on MI355X at the benchmark sizes (see ops.GCONV_FUSED), hence opt-in.
"""
import torch.nn as nn

from . import ops
from .layers import build_mlp, Linear


def _init_weights(module):
    if isinstance(module, Linear):
        nn.init.kaiming_normal_(module.weight)        # graph.py:27-30


class GraphTripleConv(nn.Module):
    """A single layer of scene graph convolution (graph.py:33-122)."""

    def __init__(self, input_dim, attributes_dim=0, output_dim=None, hidden_dim=512, pooling='avg',
                 mlp_normalization='none'):
        super().__init__()
        if output_dim is None:
            output_dim = input_dim
        self.input_dim, self.output_dim, self.hidden_dim = input_dim, output_dim, hidden_dim
        assert pooling in ['sum', 'avg'], 'Invalid pooling "%s"' % pooling
        self.pooling = pooling
        self.net1 = build_mlp([3 * input_dim + 2 * attributes_dim, hidden_dim, 2 * hidden_dim + output_dim],
                              batch_norm=mlp_normalization)
        self.net1.apply(_init_weights)
        self.net2 = build_mlp([hidden_dim, hidden_dim, output_dim], batch_norm=mlp_normalization)
        self.net2.apply(_init_weights)

    def _fusable(self, obj_vecs, pred_vecs):
        plain = all([type(m).__name__ for m in net] == ['Linear', 'ReLU', 'Linear', 'ReLU'] for net in (self.net1, self.net2))
        return (plain and obj_vecs.is_cuda and pred_vecs.dim() == 2 and pred_vecs.stride(-1) == 1
                and ops.gconv_fused_supported(obj_vecs.size(1), pred_vecs.size(1), self.hidden_dim, self.output_dim))

    def forward(self, obj_vecs, pred_vecs, edges, csr=None):
        """obj_vecs (O, D), pred_vecs (T, D), edges (T, 2) int64 -> new_obj_vecs (O, Dout), new_pred_vecs (T, Dout).
        ``csr``: the (offsets, entries) pair of ops.build_csr for these edges (GraphTripleConvNet builds it once)."""
        O = obj_vecs.size(0)
        H, Dout = self.hidden_dim, self.output_dim
        edges = edges if edges.is_contiguous() else edges.contiguous()
        off, ent = csr if csr is not None else ops.build_csr(edges, O)
        if edges.size(0) > 0 and self._fusable(obj_vecs, pred_vecs):
            l1, l2, l3, l4 = self.net1[0], self.net1[2], self.net2[0], self.net2[2]
            return ops.FusedTripleConvFn.apply(obj_vecs, pred_vecs, edges, off, ent, l1.weight, l1.bias, l2.weight, l2.bias,
                                               l3.weight, l3.bias, l4.weight, l4.bias, self.pooling == 'avg')
        pred_vecs = pred_vecs if pred_vecs.is_contiguous() else pred_vecs.contiguous()
        cur_t = ops.GatherConcatFn.apply(obj_vecs, pred_vecs, edges, off, ent)
        new_t = self.net1(cur_t)
        pooled, new_p = ops.TriplePoolFn.apply(new_t, edges, off, ent, O, H, Dout, self.pooling == 'avg')
        return self.net2(pooled), new_p


class GraphTripleConvNet(nn.Module):
    """A sequence of scene graph convolution layers (graph.py:125-147)."""

    def __init__(self, input_dim, num_layers=5, hidden_dim=512, pooling='avg', mlp_normalization='none'):
        super().__init__()
        self.num_layers = num_layers
        self.gconvs = nn.ModuleList([
            GraphTripleConv(input_dim=input_dim, hidden_dim=hidden_dim, pooling=pooling,
                            mlp_normalization=mlp_normalization) for _ in range(num_layers)])

    def forward(self, obj_vecs, pred_vecs, edges, csr=None):
        edges = edges if edges.is_contiguous() else edges.contiguous()
        if csr is None and obj_vecs.is_cuda:
            csr = ops.build_csr(edges, obj_vecs.size(0))        # one CSR for the whole stack
        for gconv in self.gconvs:
            obj_vecs, pred_vecs = gconv(obj_vecs, pred_vecs, edges, csr=csr)
        return obj_vecs, pred_vecs
