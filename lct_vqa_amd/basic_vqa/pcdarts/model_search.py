"""PC-DARTS search network (reference basic_vqa/pcdarts/model_search.py:11-250; darts_vqa adds
save/load_arch_parameters, darts_vqa/pcdarts/model_search.py:204-217).  Activations are NHWC internally; the
network takes the reference's NCHW image and returns the reference's [B, 256*7*7] feature (NCHW flatten order)."""
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import darts_ops as D
from .. import config as _config
from .genotypes import PRIMITIVES, Genotype
from .operations import OPS, FactorizedReduce, Identity, ReLUConvBN, Zero  # noqa: F401


def channel_shuffle(x, groups):
    """Reference semantics on an NHWC tensor (only used by tests; MixedOp fuses the shuffle)."""
    B, H, W, C = x.shape
    return x.view(B, H, W, groups, C // groups).transpose(3, 4).reshape(B, H, W, C)


class MixedOp(nn.Module):

    def __init__(self, C, stride):
        super(MixedOp, self).__init__()
        self._ops = nn.ModuleList()
        self.mp = nn.MaxPool2d(2, 2)
        self.k = 4
        self.stride = stride
        for primitive in PRIMITIVES:
            op = OPS[primitive](C // self.k, stride, False)
            if 'pool' in primitive:
                op = nn.Sequential(op, nn.BatchNorm2d(C // self.k, affine=False))
            self._ops.append(op)

    def branches(self, x):
        """-> (list of branch tensors on the first C/4 channels, indices of their primitives)."""
        C = x.shape[3]
        xtemp = D.SliceChannelsFn.apply(x, 0, C // self.k)
        outs, idx = [], []
        for k, op in enumerate(self._ops):
            if isinstance(op, Zero):
                continue
            if isinstance(op, Identity):
                outs.append(xtemp)
            elif isinstance(op, nn.Sequential):
                outs.append(D.run_seq(op, xtemp))
            else:
                outs.append(op(xtemp))
            idx.append(k)
        return outs, idx

    def forward(self, x, weights, state=None, beta=None):
        """state (+)= beta * channel_shuffle(cat(sum_k w_k op_k(x[:C/4]), bypass(x[C/4:])), 4).
        With state=None and beta=None this is the reference's MixedOp.forward(x, weights)."""
        outs, idx = self.branches(x)
        wsel = weights[torch.as_tensor(idx, device=weights.device)]
        if beta is None:
            beta = torch.ones(1, dtype=torch.float32, device=x.device)
        return D.MixedCombineFn.apply(state, x, wsel, beta.reshape(1), self.stride == 2, *outs)


class Cell(nn.Module):

    def __init__(self, steps, multiplier, C_prev_prev, C_prev, C, reduction, reduction_prev):
        super(Cell, self).__init__()
        self.reduction = reduction
        if reduction_prev:
            self.preprocess0 = FactorizedReduce(C_prev_prev, C, affine=False)
        else:
            self.preprocess0 = ReLUConvBN(C_prev_prev, C, 1, 1, 0, affine=False)
        self.preprocess1 = ReLUConvBN(C_prev, C, 1, 1, 0, affine=False)
        self._steps = steps
        self._multiplier = multiplier
        self._ops = nn.ModuleList()
        self._bns = nn.ModuleList()
        for i in range(self._steps):
            for j in range(2 + i):
                stride = 2 if reduction and j < 2 else 1
                self._ops.append(MixedOp(C, stride))

    def forward(self, s0, s1, weights, weights2):
        s0 = self.preprocess0(s0)
        s1 = self.preprocess1(s1)
        states = [s0, s1]
        offset = 0
        for i in range(self._steps):
            s = None
            for j, h in enumerate(states):
                s = self._ops[offset + j](h, weights[offset + j], s, weights2[offset + j])
            offset += len(states)
            states.append(s)
        return torch.cat(states[-self._multiplier:], dim=3)


class Network(nn.Module):

    def __init__(self, C, num_classes, layers, vqa_model=None, steps=4, multiplier=4, stem_multiplier=3):
        super(Network, self).__init__()
        self._C = C
        self._num_classes = num_classes
        self._layers = layers
        self._criterion = nn.CrossEntropyLoss()
        self._vqa_model = weakref.ref(vqa_model) if vqa_model is not None else (lambda: None)
        self._steps = steps
        self._multiplier = multiplier
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
            cell = Cell(steps, multiplier, C_prev_prev, C_prev, C_curr, reduction, reduction_prev)
            reduction_prev = reduction
            self.cells += [cell]
            C_prev_prev, C_prev = C_prev, multiplier * C_curr
        self.global_pooling = nn.AdaptiveAvgPool2d(7)
        self.output_ch = 256
        self.output_size = 7
        self._initialize_alphas()

    def new(self):
        model_new = Network(self._C, self._num_classes, self._layers, self._vqa_model()).to(_config.active().DEVICE)
        for x, y in zip(model_new.arch_parameters(), self.arch_parameters()):
            x.data.copy_(y.data)
        return model_new

    def _edge_weights(self, alphas, betas):
        weights = F.softmax(alphas, dim=-1)
        n, start = 3, 2
        weights2 = F.softmax(betas[0:2], dim=-1)
        for _ in range(self._steps - 1):
            end = start + n
            weights2 = torch.cat([weights2, F.softmax(betas[start:end], dim=-1)], dim=0)
            start = end
            n += 1
        return weights, weights2

    def forward(self, input):
        N, _, H, W = input.shape
        input = input.expand(N, 3, H, W).contiguous()
        s = D.StemConvFn.apply(input, self.stem[0].weight)
        s0 = s1 = D.batchnorm(self.stem[1], s)
        for cell in self.cells:
            if cell.reduction:
                weights, weights2 = self._edge_weights(self.alphas_reduce, self.betas_reduce)
            else:
                weights, weights2 = self._edge_weights(self.alphas_normal, self.betas_normal)
            s0, s1 = s1, cell(s0, s1, weights, weights2)
        return D.AvgPool7FlattenFn.apply(s1)

    def _loss(self, images, questions, labels):
        from ... import ops
        logits, _ = self._vqa_model()(images, questions)
        return ops.cross_entropy(logits, labels)

    def _initialize_alphas(self):
        dev = _config.active().DEVICE
        k = sum(1 for i in range(self._steps) for n in range(2 + i))
        num_ops = len(PRIMITIVES)
        self.alphas_normal = (1e-3 * torch.randn(k, num_ops)).to(dev).requires_grad_(True)
        self.alphas_reduce = (1e-3 * torch.randn(k, num_ops)).to(dev).requires_grad_(True)
        self.betas_normal = (1e-3 * torch.randn(k)).to(dev).requires_grad_(True)
        self.betas_reduce = (1e-3 * torch.randn(k)).to(dev).requires_grad_(True)
        self._arch_parameters = [self.alphas_normal, self.alphas_reduce, self.betas_normal, self.betas_reduce]

    def arch_parameters(self):
        return self._arch_parameters

    def save_arch_parameters(self, path):
        torch.save({'alphas_normal': self.alphas_normal, 'alphas_reduce': self.alphas_reduce,
                    'betas_normal': self.betas_normal, 'betas_reduce': self.betas_reduce}, path)

    def load_arch_parameters(self, path):
        st = torch.load(path, map_location=self.alphas_normal.device)
        for name in ('alphas_normal', 'alphas_reduce', 'betas_normal', 'betas_reduce'):
            getattr(self, name).data.copy_(st[name].data)

    def genotype(self):
        none_idx = PRIMITIVES.index('none')

        def _parse(weights, weights2):
            gene, n, start = [], 2, 0
            for i in range(self._steps):
                end = start + n
                Wt = weights[start:end].copy()
                W2 = weights2[start:end].copy()
                for j in range(n):
                    Wt[j, :] = Wt[j, :] * W2[j]
                edges = sorted(range(i + 2),
                               key=lambda x: -max(Wt[x][k] for k in range(len(Wt[x])) if k != none_idx))[:2]
                for j in edges:
                    k_best = None
                    for k in range(len(Wt[j])):
                        if k != none_idx and (k_best is None or Wt[j][k] > Wt[j][k_best]):
                            k_best = k
                    gene.append((PRIMITIVES[k_best], j))
                start = end
                n += 1
            return gene

        wn, wn2 = self._edge_weights(self.alphas_normal, self.betas_normal)
        wr, wr2 = self._edge_weights(self.alphas_reduce, self.betas_reduce)
        gene_normal = _parse(wn.data.cpu().numpy(), wn2.data.cpu().numpy())
        gene_reduce = _parse(wr.data.cpu().numpy(), wr2.data.cpu().numpy())
        concat = range(2 + self._steps - self._multiplier, self._steps + 2)
        return Genotype(normal=gene_normal, normal_concat=concat, reduce=gene_reduce, reduce_concat=concat)
