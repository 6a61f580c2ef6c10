"""PC-DARTS search network on the HIP primitives.

Drop-in for reference basic_vqa/pcdarts/model_search.py:11-250 (and darts_vqa's copy, which adds
``save_arch_parameters`` / ``load_arch_parameters``, darts_vqa/pcdarts/model_search.py:204-217): ``MixedOp``, ``Cell`` and
``Network`` keep the reference's constructor arguments, attribute names (``_ops``, ``preprocess0/1``, ``stem``, ``cells``,
``alphas_*``, ``betas_*``), module creation order (=> identical ``state_dict`` keys and, for the same seed, identical initial
parameters and architecture parameters), ``new()``, ``_loss``, ``arch_parameters()`` and ``genotype()``.

How it runs is this package's own:
  * activations are NHWC; the network takes the reference's NCHW image and returns its ``[B, 256*7*7]`` feature in NCHW
    flatten order (one kernel does the 7x7 average pool and the re-ordering);
  * a MixedOp computes its 7 live branches on the first C/4 channels and hands them to ONE kernel that forms the
    alpha-weighted sum, appends the bypass channels (2x2 max-pooled on reduction edges), applies the channel shuffle and
    accumulates beta * result into the node's running sum -- the reference's sum / cat / shuffle / beta-scale / node-sum
    chain never materialises; the 'none' branch is skipped (it adds zero and gets a zero gradient);
  * the DAG bookkeeping (which edge feeds which node, where a node's edges sit in the flat alpha / beta tables) lives in
    two small helpers, ``_node_slices`` and ``_edge_inputs``, shared by construction, forward and ``genotype()``.
"""
import weakref

import numpy as np
import torch
import torch.nn as nn

from ... import darts_ops as D
from .. import config as _config
from .genotypes import PRIMITIVES, Genotype
from .operations import OPS, FactorizedReduce, Identity, ReLUConvBN, Zero  # noqa: F401

_ARCH_NAMES = ('alphas_normal', 'alphas_reduce', 'betas_normal', 'betas_reduce')


def _node_slices(steps):
    """[(first, last+1)] positions of each intermediate node's incoming edges in the flat edge tables: node i has 2 + i."""
    out, first = [], 0
    for i in range(steps):
        out.append((first, first + 2 + i))
        first += 2 + i
    return out


def _edge_inputs(steps):
    """Input state index of every edge, in table order: 0, 1 | 0, 1, 2 | ..."""
    return [j for i in range(steps) for j in range(2 + i)]


def channel_shuffle(x, groups):
    """The reference's channel shuffle on an NHWC tensor (tests only; MixedOp's combine kernel fuses it)."""
    B, H, W, C = x.shape
    return x.view(B, H, W, groups, C // groups).transpose(3, 4).reshape(B, H, W, C)


class MixedOp(nn.Module):
    """All primitives on a quarter of the channels (partial-channel connection, k = 4), the rest bypasses."""

    def __init__(self, C, stride):
        super().__init__()
        self.k, self.stride = 4, stride
        self._ops = nn.ModuleList()
        self.mp = nn.MaxPool2d(2, 2)                               # holder for the reference's attribute; the kernel pools
        part = C // self.k
        for name in PRIMITIVES:
            op = OPS[name](part, stride, False)
            self._ops.append(nn.Sequential(op, nn.BatchNorm2d(part, affine=False)) if 'pool' in name else op)
        self._live = [i for i, op in enumerate(self._ops) if not isinstance(op, Zero)]
        self._live_index = {}                                      # device -> int64 tensor of self._live

    def branches(self, x):
        """(outputs of the live primitives on x[..., :C/4], their positions in PRIMITIVES)."""
        part = D.SliceChannelsFn.apply(x, 0, x.shape[3] // self.k)
        outs = []
        for i in self._live:
            op = self._ops[i]
            if isinstance(op, Identity):
                outs.append(part)
            elif isinstance(op, nn.Sequential):
                outs.append(D.run_seq(op, part))
            else:
                outs.append(op(part))
        return outs, self._live

    def _fused_plan(self):
        """(BatchNorm holder per slot, weight tensors in the fused kernels' order) when this MixedOp is the PRIMITIVES table
        the fused edge kernels implement (csrc/mixed_edge.hip), else None."""
        plan = self.__dict__.get('_plan')
        if plan is None:
            plan = False
            if list(PRIMITIVES) == ['none', 'max_pool_3x3', 'avg_pool_3x3', 'skip_connect', 'sep_conv_3x3', 'sep_conv_5x5',
                                    'dil_conv_3x3', 'dil_conv_5x5']:
                mx, av, skip, s3, s5, d3, d5 = (self._ops[i] for i in range(1, 8))
                convs = [s3.op[1], s5.op[1], d3.op[1], d5.op[1], s3.op[5], s5.op[5]]
                points = [s3.op[2], s5.op[2], d3.op[2], d5.op[2], s3.op[6], s5.op[6]]
                bns = [mx[1], av[1], skip.bn if self.stride == 2 else None, s3.op[3], s5.op[3], d3.op[3], d5.op[3], s3.op[7], s5.op[7]]
                if all(not b.affine and b.momentum is not None for b in bns if b is not None):
                    plan = (bns, convs + points + ([skip.conv_1, skip.conv_2] if self.stride == 2 else []))
            self.__dict__['_plan'] = plan
        return plan or None

    def forward(self, x, weights, state=None, beta=None, selected=False, slot=None, x1d=None):
        """``state (+)= beta * shuffle(cat(sum_k w_k op_k(x[:C/4]), bypass(x[C/4:])))``; with the defaults this is the
        reference's ``MixedOp.forward(x, weights)``.  ``selected``: ``weights`` already holds the live primitives only."""
        if selected:
            wsel = weights
        elif self._live == list(range(self._live[0], self._live[-1] + 1)):
            wsel = weights[self._live[0]:self._live[-1] + 1]        # a view: no gather / scatter kernels around the edge
        else:
            idx = self._live_index.get(weights.device)
            if idx is None:
                idx = self._live_index[weights.device] = torch.as_tensor(self._live, device=weights.device)
            wsel = weights[idx]
        if beta is None:
            beta = torch.ones(1, dtype=torch.float32, device=x.device)
        plan = self._fused_plan() if D.fused_edge_active() else None
        if plan is not None and D.lib.query('mixed_edge_supported', *x.shape, self.stride):
            bns, mods = plan
            modes = {b.training or not b.track_running_stats for b in bns if b is not None}
            if len(modes) == 1:                                   # one kernel, one mode: every BatchNorm of the edge agrees
                return D.MixedEdgeFn.apply(state, x, wsel, beta.reshape(1), self.stride, bns, modes.pop(), slot, x1d, *[m.weight for m in mods])
        outs, _ = self.branches(x)
        return D.MixedCombineFn.apply(state, x, wsel, beta.reshape(1), self.stride == 2, *outs)


class Cell(nn.Module):

    def __init__(self, steps, multiplier, C_prev_prev, C_prev, C, reduction, reduction_prev):
        super().__init__()
        self.reduction, self._steps, self._multiplier = reduction, steps, multiplier
        first = FactorizedReduce(C_prev_prev, C, affine=False) if reduction_prev else ReLUConvBN(C_prev_prev, C, 1, 1, 0, affine=False)
        self.preprocess0 = first
        self.preprocess1 = ReLUConvBN(C_prev, C, 1, 1, 0, affine=False)
        self._ops = nn.ModuleList()
        self._bns = nn.ModuleList()                                # empty in the reference too; kept for the key layout
        for src in _edge_inputs(steps):
            self._ops.append(MixedOp(C, 2 if reduction and src < 2 else 1))

    def forward(self, s0, s1, weights, weights2):
        states = [self.preprocess0(s0), self.preprocess1(s1)]
        live = self._ops[0]._live
        parts = None
        if live == list(range(live[0], live[-1] + 1)) and all(op._live == live for op in self._ops):
            parts = D.EdgeWeightsFn.apply(weights, weights2, live[0], live[-1] + 1)     # every edge's operands, one autograd node
        # a state is read by up to four edges and by the concatenation: one alias per reader (D.FanoutFn), the edges' input gradients
        # meet in one buffer inside their kernels instead of in ATen adds
        fan = D.fused_edge_active() and parts is not None and torch.is_grad_enabled()
        nstates = self._steps + 2
        readers = [sum(1 for first, last in _node_slices(self._steps) if last - first > i) + (1 if i >= nstates - self._multiplier else 0)
                   for i in range(nstates)]
        taps, slots, dense = {}, {}, {}
        edge_readers = [sum(1 for first, last in _node_slices(self._steps) if last - first > i) for i in range(nstates)]

        def x1_of(i):
            """dense x[..., :C/4] of state i when at least two fused edges read it (one strided pass instead of one per reader)"""
            if not (D.fused_edge_active() and parts is not None) or edge_readers[i] < 2:
                return None
            if i not in dense:
                with torch.no_grad():
                    dense[i] = D.dense_x1(states[i])
            return dense[i]

        def tap(i):
            if not fan or readers[i] < 2:
                return states[i], None
            if i not in taps:
                slots[i] = D.GradSlot()
                taps[i] = list(D.FanoutFn.apply(states[i], readers[i], slots[i]))
            return taps[i].pop(), slots[i]

        for first, last in _node_slices(self._steps):
            node = None
            for e in range(first, last):                           # edge e reads state e - first; the kernel accumulates
                x, slot = tap(e - first)
                if parts is not None:
                    node = self._ops[e](x, parts[2 * e], node, parts[2 * e + 1], selected=True, slot=slot, x1d=x1_of(e - first))
                else:
                    node = self._ops[e](x, weights[e], node, weights2[e])
            states.append(node)
        return torch.cat([tap(i)[0] for i in range(nstates - self._multiplier, nstates)], dim=3)


class Network(nn.Module):

    def __init__(self, C, num_classes, layers, vqa_model=None, steps=4, multiplier=4, stem_multiplier=3):
        super().__init__()
        self._C, self._num_classes, self._layers = C, num_classes, layers
        self._steps, self._multiplier = steps, multiplier
        self._criterion = nn.CrossEntropyLoss()
        self._vqa_model = weakref.ref(vqa_model) if vqa_model is not None else (lambda: None)
        width = stem_multiplier * C
        self.stem = nn.Sequential(nn.Conv2d(3, width, 3, padding=1, bias=False), nn.BatchNorm2d(width))
        self.cells = nn.ModuleList()
        before_last, last, cur, was_reduction = width, width, C, False
        for i in range(layers):
            is_reduction = i in (layers // 3, 2 * layers // 3)     # the channel count doubles where the map halves
            if is_reduction:
                cur *= 2
            self.cells.append(Cell(steps, multiplier, before_last, last, cur, is_reduction, was_reduction))
            before_last, last, was_reduction = last, multiplier * cur, is_reduction
        self.global_pooling = nn.AdaptiveAvgPool2d(7)
        self.output_ch, self.output_size = 256, 7
        self._initialize_alphas()

    # ---- architecture parameters --------------------------------------------------------------------------------
    def _initialize_alphas(self):
        """alpha [edges, primitives] and beta [edges] per cell type, 1e-3 * N(0,1), drawn on the host in the reference's
        order (normal alpha, reduce alpha, normal beta, reduce beta) and kept as plain leaf tensors, not Parameters."""
        dev = _config.active().DEVICE
        edges = len(_edge_inputs(self._steps))
        shapes = {'alphas_normal': (edges, len(PRIMITIVES)), 'alphas_reduce': (edges, len(PRIMITIVES)),
                  'betas_normal': (edges,), 'betas_reduce': (edges,)}
        for name in _ARCH_NAMES:
            setattr(self, name, (1e-3 * torch.randn(*shapes[name])).to(dev).requires_grad_(True))
        self._arch_parameters = [getattr(self, name) for name in _ARCH_NAMES]

    def arch_parameters(self):
        return self._arch_parameters

    def save_arch_parameters(self, path):
        torch.save({name: getattr(self, name) for name in _ARCH_NAMES}, path)

    def load_arch_parameters(self, path):
        stored = torch.load(path, map_location=self.alphas_normal.device)
        for name in _ARCH_NAMES:
            getattr(self, name).data.copy_(stored[name].data)

    def new(self):
        """A fresh network of the same geometry on config.DEVICE carrying a copy of the architecture parameters."""
        twin = Network(self._C, self._num_classes, self._layers, self._vqa_model()).to(_config.active().DEVICE)
        for mine, theirs in zip(self.arch_parameters(), twin.arch_parameters()):
            theirs.data.copy_(mine.data)
        return twin

    def _edge_weights(self, alphas, betas):
        """(softmax over primitives per edge, softmax over the incoming edges of each node)."""
        per_node = [torch.softmax(betas[a:b], dim=-1) for a, b in _node_slices(self._steps)]
        return torch.softmax(alphas, dim=-1), torch.cat(per_node, dim=0)

    # ---- forward --------------------------------------------------------------------------------------------------
    def forward(self, input):
        N, _, H, W = input.shape
        image = input.expand(N, 3, H, W).contiguous()
        stem = D.batchnorm(self.stem[1], D.StemConvFn.apply(image, self.stem[0].weight))
        table = {False: self._edge_weights(self.alphas_normal, self.betas_normal),
                 True: self._edge_weights(self.alphas_reduce, self.betas_reduce)}
        s0 = s1 = stem
        for cell in self.cells:
            s0, s1 = s1, cell(s0, s1, *table[cell.reduction])
        return D.AvgPool7FlattenFn.apply(s1)

    def _loss(self, images, questions, labels):
        from ... import ops
        logits, _ = self._vqa_model()(images, questions)
        return ops.cross_entropy(logits, labels)

    # ---- discretisation -------------------------------------------------------------------------------------------
    def genotype(self):
        """For every node keep the two incoming edges with the strongest non-'none' primitive (strength = alpha-softmax x
        beta-softmax) and that primitive; ties go to the lower index, edges are listed strongest first."""
        none = PRIMITIVES.index('none')

        def decode(alphas, betas):
            w, w2 = self._edge_weights(alphas, betas)
            strength = (w * w2.unsqueeze(1)).detach().cpu().numpy().astype(np.float32)
            strength[:, none] = -np.inf
            gene = []
            for first, last in _node_slices(self._steps):
                best_op = strength[first:last].argmax(axis=1)                       # first maximum, like a strict '>' scan
                best_val = strength[first:last].max(axis=1)
                for j in sorted(range(last - first), key=lambda e: -best_val[e])[:2]:   # stable: lower edge wins a tie
                    gene.append((PRIMITIVES[int(best_op[j])], j))
            return gene

        concat = range(2 + self._steps - self._multiplier, self._steps + 2)
        return Genotype(normal=decode(self.alphas_normal, self.betas_normal), normal_concat=concat,
                        reduce=decode(self.alphas_reduce, self.betas_reduce), reduce_concat=concat)
