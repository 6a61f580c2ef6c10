"""PC-DARTS primitives on the HIP kernels, shaped like reference basic_vqa/pcdarts/operations.py:4-104 where it matters to a
user of the reference: the names (``OPS``, ``ReLUConvBN``, ``DilConv``, ``SepConv``, ``Identity``, ``Zero``,
``FactorizedReduce``), the constructor arguments, and the ``state_dict`` keys / parameter creation order (``op.<index>`` inside a
Sequential, ``conv_1`` / ``conv_2`` / ``bn``), so that checkpoints and the seed-for-seed initialisation carry over.

Everything else is this package's own design.  A primitive is DATA: a tuple of layer specs (``_relu``, ``_dw``, ``_pw``,
``_bn``) from which one builder makes the holder Sequential; the torch modules in it never run -- they hold parameters
and running statistics, while ``darts_ops.run_seq`` walks the Sequential and launches the HIP kernels on NHWC tensors
(a ReLU is folded into the convolution that follows it).
"""
import torch
import torch.nn as nn

from ... import darts_ops as D


# ---- layer specs -> holder modules (creation order = parameter order = RNG consumption order) ----------------------
def _relu():
    return nn.ReLU(inplace=False)


def _dw(C, k, stride, pad, dil=1):
    """depthwise k x k"""
    return nn.Conv2d(C, C, kernel_size=k, stride=stride, padding=pad, dilation=dil, groups=C, bias=False)


def _pw(C_in, C_out, stride=1):
    """1 x 1"""
    return nn.Conv2d(C_in, C_out, kernel_size=1, stride=stride, padding=0, bias=False)


def _bn(C, affine):
    return nn.BatchNorm2d(C, affine=affine)


class _SeqOp(nn.Module):
    """A primitive whose layers sit in ``self.op`` (the reference's attribute name) and run through run_seq."""

    def __init__(self, *layers):
        super().__init__()
        self.op = nn.Sequential(*layers)

    def forward(self, x):
        return D.run_seq(self.op, x)


class ReLUConvBN(_SeqOp):
    def __init__(self, C_in, C_out, kernel_size, stride, padding, affine=True):
        conv = nn.Conv2d(C_in, C_out, kernel_size, stride=stride, padding=padding, bias=False)
        super().__init__(_relu(), conv, _bn(C_out, affine))


class DilConv(_SeqOp):
    def __init__(self, C_in, C_out, kernel_size, stride, padding, dilation, affine=True):
        super().__init__(_relu(), _dw(C_in, kernel_size, stride, padding, dilation), _pw(C_in, C_out), _bn(C_out, affine))


class SepConv(_SeqOp):
    def __init__(self, C_in, C_out, kernel_size, stride, padding, affine=True):
        super().__init__(_relu(), _dw(C_in, kernel_size, stride, padding), _pw(C_in, C_in), _bn(C_in, affine),
                         _relu(), _dw(C_in, kernel_size, 1, padding), _pw(C_in, C_out), _bn(C_out, affine))


class Identity(nn.Module):
    def forward(self, x):
        return x


class Zero(nn.Module):
    """The 'none' primitive: zeros of the (strided) input's shape.  MixedOp never runs it -- it contributes nothing and its
    architecture weight gets a zero gradient -- so this forward only serves direct callers."""

    def __init__(self, stride):
        super().__init__()
        self.stride = stride

    def forward(self, x):                                         # NHWC
        s = self.stride
        return torch.zeros_like(x if s == 1 else x[:, ::s, ::s, :])


class FactorizedReduce(nn.Module):
    """Stride-2 reduction that keeps every pixel in play: two 1x1 convs read relu(x) at even and at odd positions, their
    outputs are concatenated over channels, then BatchNorm."""

    def __init__(self, C_in, C_out, affine=True):
        super().__init__()
        if C_out % 2:
            raise ValueError('FactorizedReduce needs an even C_out')
        self.relu = _relu()
        self.conv_1 = _pw(C_in, C_out // 2, stride=2)
        self.conv_2 = _pw(C_in, C_out // 2, stride=2)
        self.bn = _bn(C_out, affine)

    def forward(self, x):
        even, odd = D.Subsample2Fn.apply(x, True)                  # relu(x)[::2, ::2], relu(x)[1::2, 1::2]
        halves = [D.PwConvFn.apply(even, self.conv_1.weight, False), D.PwConvFn.apply(odd, self.conv_2.weight, False)]
        return D.batchnorm(self.bn, torch.cat(halves, dim=3))


def _pool(kind):
    mk = {'avg': lambda s: nn.AvgPool2d(3, stride=s, padding=1, count_include_pad=False),
          'max': lambda s: nn.MaxPool2d(3, stride=s, padding=1)}[kind]
    return lambda C, stride, affine: mk(stride)


def _conv(cls, k, pad, *extra):
    return lambda C, stride, affine: cls(C, C, k, stride, pad, *extra, affine=affine)


# name -> factory(C, stride, affine), the reference's table
OPS = {
    'none': lambda C, stride, affine: Zero(stride),
    'avg_pool_3x3': _pool('avg'),
    'max_pool_3x3': _pool('max'),
    'skip_connect': lambda C, stride, affine: Identity() if stride == 1 else FactorizedReduce(C, C, affine=affine),
    'sep_conv_3x3': _conv(SepConv, 3, 1),
    'sep_conv_5x5': _conv(SepConv, 5, 2),
    'sep_conv_7x7': _conv(SepConv, 7, 3),
    'dil_conv_3x3': _conv(DilConv, 3, 2, 2),
    'dil_conv_5x5': _conv(DilConv, 5, 4, 2),
}
