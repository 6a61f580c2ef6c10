"""Helpers shared by the tests (test infrastructure)."""
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


def load(name):
    return torch.load(os.path.join(GOLD, name), map_location='cpu', weights_only=False)


def checksum(t):
    t = t.detach().double()
    return [float(t.sum()), float(t.abs().sum())]


def close_sum(a, b, rel=1e-6):
    return all(abs(x - y) <= rel * max(1.0, abs(y)) for x, y in zip(a, b))


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def vgg_params_from_seed(seed, prefix='img_encoder.model.'):
    """Regenerate the VGG19 tensors of a fixture: the reference constructs vgg19 FIRST after
    torch.manual_seed(seed) (models.py:23), so the restated architecture seeded the same way
    yields the same tensors (verified against the fixture's checksums by the caller)."""
    from oracle import tv_vgg19
    torch.manual_seed(seed)
    m = tv_vgg19.vgg19()
    sd = m.state_dict()
    return {prefix + k: v for k, v in sd.items() if not k.startswith('classifier.6.')}


def full_params(fx):
    P = dict(fx['state'])
    V = vgg_params_from_seed(fx['seed'])
    for k, cs in fx['vgg_checksums'].items():
        assert close_sum(checksum(V[k]), cs), k
    P.update(V)
    return P
