"""The minimal-filtering identities the convolution kernels are built on, in float64 on the CPU (no GPU, no library):
F(4x4,3x3) of the forward / dgrad kernels (conv3x3_wino4.hip) and its transpose F(3x3,4x4) of the Winograd-domain weight gradient
(conv3x3_wino_wgrad.hip).  The matrices below are the ones the kernels hard-code (same interpolation points 0, +-1, +-2, inf)."""
import numpy as np

BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
               [0, 4, 0, -5, 0, 1]], float)
G3 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
               [0, 0, 1]])
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], float)
G4 = np.array([[1 / 4, 0, 0, 0], [-1 / 6, -1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6, 1 / 6], [1 / 24, 1 / 12, 1 / 6, 1 / 3],
               [1 / 24, -1 / 12, 1 / 6, -1 / 3], [0, 0, 0, 1]])
AT3 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 1]], float)


def corr2d(d, g, m):
    r = g.shape[0]
    return np.array([[sum(d[i + a, j + b] * g[a, b] for a in range(r) for b in range(r)) for j in range(m)] for i in range(m)])


def test_forward_f4x4_3x3():
    rng = np.random.default_rng(0)
    d, g = rng.standard_normal((6, 6)), rng.standard_normal((3, 3))
    y = AT4 @ ((G3 @ g @ G3.T) * (BT @ d @ BT.T)) @ AT4.T
    assert np.abs(y - corr2d(d, g, 4)).max() < 1e-12


def test_weight_gradient_f3x3_4x4():
    rng = np.random.default_rng(1)
    x, dz = rng.standard_normal((6, 6)), rng.standard_normal((4, 4))       # 6x6 input tile, 4x4 tile of the output gradient
    dw = AT3 @ ((G4 @ dz @ G4.T) * (BT @ x @ BT.T)) @ AT3.T
    assert np.abs(dw - corr2d(x, dz, 3)).max() < 1e-12


def test_weight_gradient_sums_over_tiles_before_the_output_transform():
    """The kernel accumulates the 36 transformed-domain products over all tiles (the batched GEMM) and applies A'^T . A' once."""
    rng = np.random.default_rng(2)
    xs, dzs = rng.standard_normal((5, 6, 6)), rng.standard_normal((5, 4, 4))
    m = sum((G4 @ dz @ G4.T) * (BT @ x @ BT.T) for x, dz in zip(xs, dzs))
    want = sum(corr2d(x, dz, 3) for x, dz in zip(xs, dzs))
    assert np.abs(AT3 @ m @ AT3.T - want).max() < 1e-11
