"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/lctvqa.h declares;
the host layer fails loudly without a GPU (no fallback)."""
import re
import os

import pytest
import torch

from _util import ROOT


def test_header_symbols_exported(hip_lib):
    from lct_vqa_amd._lib import parse_header, HEADER
    sigs = parse_header()
    declared = set(re.findall(r'\b(lctvqa_\w+)\s*\(', open(HEADER).read()))
    assert declared == set(sigs), declared ^ set(sigs)
    assert len(sigs) >= 34
    dll = hip_lib.load()
    for name in sigs:
        assert hasattr(dll, name), name
    assert hip_lib.query('version') >= 100
    assert hip_lib.query('conv3x3_packed_elems', 64, 64) == 64 * 64 * 9


def test_tiling_table(hip_lib):
    q = lambda *a: hip_lib.query('conv3x3_uses_mfma', *a)
    # every VGG19 layer after the first at 224x224 runs on the MFMA kernel
    for (H, Cin, Cout, pool) in [(224, 64, 64, 1), (112, 64, 128, 0), (112, 128, 128, 1), (56, 128, 256, 0),
                                 (56, 256, 256, 1), (28, 256, 512, 0), (28, 512, 512, 1), (14, 512, 512, 0),
                                 (14, 512, 512, 1)]:
        for B in (1, 4, 256):
            assert q(B, H, H, Cin, Cout, pool) == 1, (B, H, Cin, Cout)
    assert q(4, 4, 4, 512, 512, 0) == 0      # tiny maps: direct fallback
    assert q(4, 224, 224, 3, 64, 0) == 0     # first layer has its own kernel


def test_no_cpu_fallback():
    from lct_vqa_amd import ops
    with pytest.raises(RuntimeError):
        ops.linear(torch.zeros(2, 4), torch.zeros(3, 4))
    from lct_vqa_amd.basic_vqa import models
    m = models.VqaModel(32, 50, 40, 16, 1, 32)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 32, 32), torch.zeros(1, 5, dtype=torch.long))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'lct_vqa_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dp, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f
