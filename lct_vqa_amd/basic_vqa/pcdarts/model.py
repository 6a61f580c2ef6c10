"""Derived (discrete) PC-DARTS network: the architecture that `Network.genotype()` decodes after the search
(reference basic_vqa/pcdarts/model_search.py:205-250, genotypes.py:3,78-82) — one primitive per edge, the two
strongest input edges per intermediate node, full channel width (no partial-channel bypass, no α/β).

The reference repository stops at `genotype()`; it never instantiates the derived cell.  The construction below is
the standard one of the DARTS family for that genotype format (every intermediate node = sum of its two selected
`op(state[index])`, stride 2 on edges from the cell inputs in a reduction cell, output = channel concat of the
`concat` nodes), with the SAME stem, channel schedule, reduction positions and 7x7 pooled output as the search network
(`model_search.Network`), so that it is a drop-in `ImgEncoder.darts` and runs on the same HIP primitives
(operations.py -> darts_ops) without the weighted sum.  SURVEY.md section 8 row f-4.
"""
import torch
import torch.nn as nn

from ... import darts_ops as D
from .genotypes import Genotype
from .operations import OPS, FactorizedReduce, Identity, ReLUConvBN


def _apply_op(op, x):
    if isinstance(op, Identity):
        return x
    if isinstance(op, (nn.Sequential, nn.AvgPool2d, nn.MaxPool2d)):
        return D.run_seq(op if isinstance(op, nn.Sequential) else nn.Sequential(op), x)
    return op(x)


class Cell(nn.Module):

    def __init__(self, genotype, C_prev_prev, C_prev, C, reduction, reduction_prev):
        super(Cell, self).__init__()
        self.reduction = reduction
        if reduction_prev:
            self.preprocess0 = FactorizedReduce(C_prev_prev, C)
        else:
            self.preprocess0 = ReLUConvBN(C_prev_prev, C, 1, 1, 0)
        self.preprocess1 = ReLUConvBN(C_prev, C, 1, 1, 0)
        if reduction:
            op_names, indices = zip(*genotype.reduce)
            concat = genotype.reduce_concat
        else:
            op_names, indices = zip(*genotype.normal)
            concat = genotype.normal_concat
        self._compile(C, op_names, indices, concat, reduction)

    def _compile(self, C, op_names, indices, concat, reduction):
        assert len(op_names) == len(indices) and len(op_names) % 2 == 0
        self._steps = len(op_names) // 2
        self._concat = list(concat)
        self.multiplier = len(self._concat)
        self._ops = nn.ModuleList()
        for name, index in zip(op_names, indices):
            stride = 2 if reduction and index < 2 else 1
            self._ops.append(OPS[name](C, stride, True))
        self._indices = list(indices)

    def forward(self, s0, s1):
        s0 = self.preprocess0(s0)
        s1 = self.preprocess1(s1)
        states = [s0, s1]
        for i in range(self._steps):
            h1 = _apply_op(self._ops[2 * i], states[self._indices[2 * i]])
            h2 = _apply_op(self._ops[2 * i + 1], states[self._indices[2 * i + 1]])
            states.append(h1 + h2)
        return torch.cat([states[i] for i in self._concat], dim=3)


class NetworkDerived(nn.Module):
    """Same constructor geometry as `model_search.Network(C, num_classes, layers, ...)` plus the genotype."""

    def __init__(self, C, num_classes, layers, genotype, stem_multiplier=3):
        super(NetworkDerived, self).__init__()
        if not isinstance(genotype, Genotype):
            genotype = Genotype(*genotype)
        self._C, self._num_classes, self._layers, self._genotype = C, num_classes, layers, genotype
        C_curr = stem_multiplier * C
        self.stem = nn.Sequential(
            nn.Conv2d(3, C_curr, 3, padding=1, bias=False),
            nn.BatchNorm2d(C_curr))
        C_prev_prev, C_prev, C_curr = C_curr, C_curr, C
        self.cells = nn.ModuleList()
        reduction_prev = False
        for i in range(layers):
            if i in [layers // 3, 2 * layers // 3]:
                C_curr *= 2
                reduction = True
            else:
                reduction = False
            cell = Cell(genotype, C_prev_prev, C_prev, C_curr, reduction, reduction_prev)
            reduction_prev = reduction
            self.cells += [cell]
            C_prev_prev, C_prev = C_prev, cell.multiplier * C_curr
        self.global_pooling = nn.AdaptiveAvgPool2d(7)
        self.output_ch = C_prev
        self.output_size = 7

    def genotype(self):
        return self._genotype

    def forward(self, input):
        N, _, H, W = input.shape
        input = input.expand(N, 3, H, W).contiguous()
        s = D.StemConvFn.apply(input, self.stem[0].weight)
        s0 = s1 = D.batchnorm(self.stem[1], s)
        for cell in self.cells:
            s0, s1 = s1, cell(s0, s1)
        return D.AvgPool7FlattenFn.apply(s1)
