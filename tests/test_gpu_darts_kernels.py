"""Shape sweep of the PC-DARTS streaming kernels through the C ABI against torch on the CPU: every channel-quad count
(C = 4 .. 64), kernel size (3, 5, 7), stride and dilation of the depthwise kernels, and every (Cin, Cout) pair of the
register-outer-product 1x1 weight gradient, on odd sizes that leave ragged tails.  (The model-level tests only reach the
shapes of the search network: C <= 16, K <= 5.)"""
import pytest
import torch
import torch.nn.functional as F

from _util import rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('C', [4, 8, 16, 32, 64, 6, 12])
@pytest.mark.parametrize('K,s,d', [(3, 1, 1), (3, 2, 1), (3, 1, 2), (5, 1, 1), (5, 2, 1), (5, 1, 2), (7, 1, 1)])
def test_depthwise_kernels_vs_torch(hip_lib, C, K, s, d):
    torch.manual_seed(100 * C + 10 * K + s + d)
    B, H, W = 3, 13, 11
    p = d * (K - 1) // 2
    x = torch.randn(B, C, H, W)
    w = torch.randn(C, 1, K, K)
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    yr = F.conv2d(torch.relu(xr), wr, stride=s, padding=p, dilation=d, groups=C)
    dy = torch.randn_like(yr)
    gx, gw = torch.autograd.grad(yr, [xr, wr], dy)
    Ho, Wo = yr.shape[2], yr.shape[3]
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wd = w.reshape(C, K, K).contiguous().to(DEV)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = torch.empty(B, Ho, Wo, C, device=DEV)
    dx = torch.empty_like(xd)
    dw = torch.empty_like(wd)
    scratch = torch.empty(hip_lib.query('dwconv_bwd_weight_scratch_elems', B, H, W, C, K, s, p, d), device=DEV)
    hip_lib('dwconv_fwd', xd, wd, y, B, H, W, C, K, s, p, d, 1)
    hip_lib('dwconv_bwd_input', dyd, wd, xd, dx, B, H, W, C, K, s, p, d, 1)
    hip_lib('dwconv_bwd_weight', dyd, xd, dw, scratch, B, H, W, C, K, s, p, d, 1)
    assert rel_err(y.permute(0, 3, 1, 2), yr) < 1e-5
    assert rel_err(dx.permute(0, 3, 1, 2), gx) < 1e-5
    assert rel_err(dw, gw[:, 0]) < 1e-4


@pytest.mark.parametrize('Cin', [4, 8, 16, 12])
@pytest.mark.parametrize('Cout', [4, 8, 16])
@pytest.mark.parametrize('relu', [0, 1])
def test_pointwise_weight_gradient_vs_torch(hip_lib, Cin, Cout, relu):
    torch.manual_seed(7 * Cin + Cout + relu)
    P = 3 * 37 * 29                                        # odd pixel count: ragged last block
    x = torch.randn(P, Cin)
    dy = torch.randn(P, Cout)
    ref = dy.double().t() @ (torch.relu(x) if relu else x).double()
    assert hip_lib.query('pwconv_wgrad_small_supported', P, Cout, Cin)
    dw = torch.empty(Cout, Cin, device=DEV)
    scratch = torch.empty(hip_lib.query('pwconv_wgrad_small_scratch_elems', P, Cout, Cin), device=DEV)
    hip_lib('pwconv_wgrad_small', dy.to(DEV), x.to(DEV), dw, scratch, P, Cout, Cin, relu)
    assert rel_err(dw.double(), ref) < 1e-5


@pytest.mark.parametrize('Cout,Cin', [(16, 48), (32, 48), (32, 64), (64, 128), (64, 256), (8, 16), (64, 16), (4, 256)])
@pytest.mark.parametrize('relu', [0, 1])
def test_pointwise_weight_gradient_blocked_kernel_vs_torch(hip_lib, Cout, Cin, relu):
    """The preprocess layers' shapes (ReLUConvBN C_prev -> C of every cell, operations.py:24-33): register-blocked reduction kernel
    (pwconv_wgrad_blk_kernel) instead of a split-K TN GEMM; a pixel count that is not a multiple of anything, and one larger than a
    workgroup's slice."""
    torch.manual_seed(3 * Cin + Cout + relu)
    for P in (3 * 37 * 29, 70001):
        x = torch.randn(P, Cin)
        dy = torch.randn(P, Cout)
        ref = dy.double().t() @ (torch.relu(x) if relu else x).double()
        assert hip_lib.query('pwconv_wgrad_small_supported', P, Cout, Cin)
        dw = torch.full((Cout, Cin), float('nan'), device=DEV)
        scratch = torch.empty(hip_lib.query('pwconv_wgrad_small_scratch_elems', P, Cout, Cin), device=DEV)
        hip_lib('pwconv_wgrad_small', dy.to(DEV), x.to(DEV), dw, scratch, P, Cout, Cin, relu)
        assert rel_err(dw.double(), ref) < 1e-5, (P, rel_err(dw.double(), ref))
