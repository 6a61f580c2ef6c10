"""Adam (+ fused global-norm clip) on the HIP library.

``FusedAdam`` is call-compatible with ``torch.optim.Adam`` as the reference uses it
(basic_vqa/model_factory.py:20-21,41-42: ``optim.Adam(model.parameters(), lr=...)``; StepLR reads
``param_groups[i]['lr']``).  With ``flatten=True`` the trainable parameters, their gradients and the
Adam moments live in ONE contiguous fp32 arena each (parameters/gradients become views), so that
  * ``step(max_grad_norm=5)`` = one grad-norm reduction + one fused clip+Adam kernel
    (16 B/param read, 12 B/param written: HBM-bound),
  * the data-parallel all-reduce works on the flat gradient buffer directly (parallel.py),
  * the architects' ``_concat``/axpy vector ops (architect_lct.py:112-116,194-207) are zero-copy.
"""
import torch

from ._lib import lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, flatten=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.flat = None
        if flatten:
            self._flatten()

    # ------------------------------------------------------------------ flat arena
    ALIGN = 4                  # floats: every parameter starts on a 16-byte boundary of the arena (the kernels that read
                               # parameters as 16-byte vectors -- LSTM weights, GEMM operands -- require that of a pointer)

    def _flatten(self):
        if len(self.param_groups) != 1:
            raise ValueError('flatten=True supports a single param group')
        ps = [p for p in self.param_groups[0]['params'] if p.requires_grad]
        if not ps:
            raise ValueError('no trainable parameters')
        dev = ps[0].device
        offs, n = [], 0
        for p in ps:
            offs.append(n)
            n += -(-p.numel() // self.ALIGN) * self.ALIGN
        fp = torch.zeros(n, dtype=torch.float32, device=dev)      # the padding stays zero (zero gradient, zero update)
        fg = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, o in zip(ps, offs):
            k = p.numel()
            fp[o:o + k].copy_(p.data.reshape(-1))
            p.data = fp[o:o + k].view_as(p.data)
            p.grad = fg[o:o + k].view_as(p.data)
        self.flat = dict(params=ps, offsets=offs, p=fp, g=fg, m=torch.zeros_like(fp), v=torch.zeros_like(fp),
                         steps=[0] * len(ps), touched=set(), hooks=[],
                         scratch=torch.empty(1024, dtype=torch.float32, device=dev),
                         norm=torch.zeros(1, dtype=torch.float32, device=dev))
        # torch.optim.Adam leaves a parameter whose .grad is None alone (no moment decay, no step count).  In the arena a
        # gradient is never None, so "backward reached this parameter" is recorded by a hook instead.
        index = {p: i for i, p in enumerate(ps)}
        for p in ps:
            self.flat['hooks'].append(p.register_post_accumulate_grad_hook(lambda q, _t=self.flat['touched'], _i=index: _t.add(_i[q])))
        self._publish_state()

    def _publish_state(self):
        """Expose the arena's moments in torch.optim.Adam's per-parameter state format ('step', 'exp_avg',
        'exp_avg_sq' as VIEWS of the flat buffers) so that state_dict() / checkpoints (reference
        basic_vqa/experiment.py:398-417: {'ef_opt': optimizer.state_dict(), ...}) are wire-compatible."""
        f = self.flat
        for p, o, st in zip(f['params'], f['offsets'], f['steps']):
            k = p.numel()
            self.state[p] = {'step': torch.tensor(float(st)), 'exp_avg': f['m'][o:o + k].view_as(p.data),
                             'exp_avg_sq': f['v'][o:o + k].view_as(p.data)}

    def load_state_dict(self, state_dict):
        """Accepts a torch.optim.Adam (or FusedAdam) state_dict; in flat mode the moments are copied into the arena."""
        import copy
        state_dict = copy.deepcopy(state_dict)        # never alias (and later mutate in place) the caller's tensors
        if self.flat is None:
            return super().load_state_dict(state_dict)
        f = self.flat
        super().load_state_dict(state_dict)          # fills self.state[p] with (new) tensors
        for i, (p, o) in enumerate(zip(f['params'], f['offsets'])):
            k = p.numel()
            st = self.state.get(p, {})
            if 'exp_avg' in st:
                f['m'][o:o + k].copy_(st['exp_avg'].reshape(-1))
                f['v'][o:o + k].copy_(st['exp_avg_sq'].reshape(-1))
                f['steps'][i] = int(float(st['step']))
        self._publish_state()

    @property
    def flat_grad(self):
        return self.flat['g'] if self.flat else None

    @property
    def flat_param(self):
        return self.flat['p'] if self.flat else None

    def zero_grad(self, set_to_none=False):
        if self.flat is not None:
            self.flat['g'].zero_()
            self.flat['touched'].clear()
            return
        super().zero_grad(set_to_none=set_to_none)

    def _runs(self):
        """[(first, last, step)]: maximal runs of consecutive arena parameters that this step updates with the same step
        count -- one kernel launch each; normally ONE run = the whole arena.  Parameters backward did not reach since
        the last zero_grad() are skipped like torch.optim.Adam skips ``grad is None`` (if no hook fired at all, e.g. the
        gradients were written into the arena by hand, every parameter is taken as reached)."""
        f = self.flat
        live = f['touched'] if f['touched'] else range(len(f['params']))
        runs = []
        for i in sorted(live):
            f['steps'][i] += 1
            self.state[f['params'][i]]['step'] = torch.tensor(float(f['steps'][i]))
            if runs and runs[-1][1] == i - 1 and runs[-1][2] == f['steps'][i]:
                runs[-1][1] = i
            else:
                runs.append([i, i, f['steps'][i]])
        return runs

    def grad_norm(self):
        """Device scalar ||g||_2 over the arena (clip_grad_norm_'s total_norm)."""
        f = self.flat
        lib('grad_norm', f['g'], f['g'].numel(), f['scratch'], f['norm'])
        return f['norm']

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None, max_grad_norm=None):
        loss = closure() if closure is not None else None
        if self.flat is not None:
            f, g = self.flat, self.param_groups[0]
            if f['p'].device.type != 'cuda':
                raise RuntimeError('FusedAdam.step: the arena is not on the ROCm device (lct_vqa_amd has no CPU path)')
            norm = self.grad_norm() if max_grad_norm is not None else None
            for first, last, step in self._runs():
                o, e = f['offsets'][first], f['offsets'][last] + -(-f['params'][last].numel() // self.ALIGN) * self.ALIGN
                lib('adam_step', f['p'][o:e], f['g'][o:e], f['m'][o:e], f['v'][o:e], e - o, norm,
                    float(max_grad_norm or 0.0), float(g['lr']), float(g['betas'][0]), float(g['betas'][1]),
                    float(g['eps']), step, float(g['weight_decay']))
            f['touched'].clear()
            return loss
        if max_grad_norm is not None:
            raise ValueError('fused clipping needs flatten=True (use nn.utils.clip_grad_norm_ otherwise)')
        for g in self.param_groups:
            for p in g['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p)
                    st['exp_avg_sq'] = torch.zeros_like(p)
                st['step'] = st['step'] + 1
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if not p.is_contiguous():
                    raise RuntimeError('FusedAdam: non-contiguous parameter')
                lib('adam_step', p, grad, st['exp_avg'], st['exp_avg_sq'], p.numel(), None, 0.0,
                    float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']),
                    int(st['step']), float(g['weight_decay']))
        return loss
