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
    """``flatten``: False = one ``adam_step`` launch per parameter; True = build the arena now from every parameter that
    requires a gradient (they must already be on the ROCm device); ``'auto'`` = what the factories hand to the reference's
    driver: the driver builds its optimizers BEFORE it moves the models to the device (experiment.py:59-62 vs :120-123), so
    the arena is built on the first ``step()``, from the parameters that carry a gradient at that moment (a frozen trunk --
    36 tensors, 575 MB in the W model -- never enters it).  Parameters that only receive a gradient later are updated one by
    one next to the arena, like ``flatten=False`` does."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, flatten=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.flat = None
        self._auto = flatten == 'auto'
        if flatten is True:
            self._flatten([p for p in self.param_groups[0]['params'] if p.requires_grad])

    # ------------------------------------------------------------------ flat arena
    ALIGN = 4                  # floats: every parameter starts on a 16-byte boundary of the arena (the kernels that read
                               # parameters as 16-byte vectors -- LSTM weights, GEMM operands -- require that of a pointer)

    def _flatten(self, ps, all_touched=False):
        if len(self.param_groups) != 1:
            raise ValueError('the flat arena supports a single param group')
        if not ps:
            raise ValueError('no trainable parameters')
        dev = ps[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in ps):
            raise ValueError('the flat arena needs fp32 parameters on one device')
        offs, n = [], 0
        for p in ps:
            offs.append(n)
            n += -(-p.numel() // self.ALIGN) * self.ALIGN
        fp = torch.zeros(n, dtype=torch.float32, device=dev)      # the padding stays zero (zero gradient, zero update)
        fg = torch.zeros(n, dtype=torch.float32, device=dev)
        fm, fv = torch.zeros_like(fp), torch.zeros_like(fp)
        steps, gviews = [], []
        for p, o in zip(ps, offs):
            k = p.numel()
            fp[o:o + k].copy_(p.data.reshape(-1))
            p.data = fp[o:o + k].view_as(p.data)
            gv = fg[o:o + k].view_as(p.data)
            if p.grad is not None:                                # a gradient that exists already (first step of 'auto') moves in
                gv.copy_(p.grad)
            p.grad = gv
            gviews.append(gv)
            st = self.state.get(p, {})                            # moments loaded before the arena existed (a resumed checkpoint)
            if 'exp_avg' in st:
                fm[o:o + k].copy_(st['exp_avg'].reshape(-1))
                fv[o:o + k].copy_(st['exp_avg_sq'].reshape(-1))
            steps.append(int(float(st['step'])) if 'step' in st else 0)
        self.flat = dict(params=ps, offsets=offs, p=fp, g=fg, m=fm, v=fv, steps=steps, gviews=gviews,
                         touched=set(range(len(ps))) if all_touched else set(), hooks=[], member={id(p) for p in ps},
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
        """Accepts a torch.optim.Adam (or FusedAdam) state_dict; in flat mode the moments are copied into the arena (before
        the arena of ``flatten='auto'`` exists they wait in the per-parameter state and move in when it is built)."""
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

    def zero_grad(self, set_to_none=True):
        if self.flat is None:
            return super().zero_grad(set_to_none=set_to_none)
        f = self.flat
        f['g'].zero_()
        f['touched'].clear()
        for p, gv in zip(f['params'], f['gviews']):
            if p.grad is not gv:                                  # someone dropped or replaced it (model.zero_grad()): back in place
                p.grad = gv
        for p in self.param_groups[0]['params']:
            if id(p) not in f['member'] and p.grad is not None:
                p.grad = None

    def _reclaim(self):
        """Arena parameters whose ``.grad`` is no longer the arena's view (``model.zero_grad()`` set it to None and backward
        made a fresh tensor, or a caller assigned one): the gradient is copied into the arena and the view restored."""
        f = self.flat
        for i, (p, gv) in enumerate(zip(f['params'], f['gviews'])):
            g = p.grad
            if g is gv:
                continue
            if g is not None and g.data_ptr() != gv.data_ptr():
                gv.copy_(g)
                f['touched'].add(i)
            elif g is None:
                f['touched'].discard(i)
                gv.zero_()
            p.grad = gv

    def _plan(self):
        """[[first, last, step]]: maximal runs of consecutive arena parameters that this step updates with the same step
        count -- one kernel launch each; normally ONE run = the whole arena.  Parameters backward did not reach since
        the last zero_grad() are skipped like torch.optim.Adam skips ``grad is None``.  'Reached' is recorded by
        post-accumulate-grad hooks; where no hook fired at all -- gradients written into the arena by hand
        (``p.grad.copy_``, ``bilevel.set_grads`` after ``autograd.grad``) or a step replayed from a HIP graph -- every
        parameter is taken as reached.  Nothing is mutated here: ``_commit`` advances the step counts once the launches are out."""
        f = self.flat
        live = f['touched'] if f['touched'] else range(len(f['params']))
        runs = []
        for i in sorted(live):
            step = f['steps'][i] + 1
            if runs and runs[-1][1] == i - 1 and runs[-1][2] == step:
                runs[-1][1] = i
            else:
                runs.append([i, i, step])
        return runs

    def _commit(self, runs):
        f = self.flat
        for first, last, step in runs:
            for i in range(first, last + 1):
                f['steps'][i] = step
                self.state[f['params'][i]]['step'] = torch.tensor(float(step))

    def _runs(self):
        """Plan and commit in one call (tests, callers that launch nothing)."""
        runs = self._plan()
        self._commit(runs)
        return runs

    def grad_norm(self):
        """Device scalar ||g||_2 over the arena (clip_grad_norm_'s total_norm)."""
        f = self.flat
        lib('grad_norm', f['g'], f['g'].numel(), f['scratch'], f['norm'])
        return f['norm']

    # ------------------------------------------------------------------ step
    def _step_one(self, p, g):
        st = self.state[p]
        if not st:
            st['step'] = 0
            st['exp_avg'] = torch.zeros_like(p)
            st['exp_avg_sq'] = torch.zeros_like(p)
        grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
        if not p.is_contiguous():
            raise RuntimeError('FusedAdam: non-contiguous parameter')
        lib('adam_step', p, grad, st['exp_avg'], st['exp_avg_sq'], p.numel(), None, 0.0,
            float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']),
            int(st['step']) + 1, float(g['weight_decay']))
        st['step'] = st['step'] + 1

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm=None):
        loss = closure() if closure is not None else None
        if self.flat is None and self._auto:
            ps = [p for p in self.param_groups[0]['params'] if p.requires_grad and p.grad is not None]
            if ps and all(p.is_cuda and p.dtype == torch.float32 and p.grad.dtype == torch.float32 for p in ps) and len(self.param_groups) == 1:
                self._flatten(ps, all_touched=True)
        if self.flat is not None:
            f, g = self.flat, self.param_groups[0]
            if f['p'].device.type != 'cuda':
                raise RuntimeError('FusedAdam.step: the arena is not on the ROCm device (lct_vqa_amd has no CPU path)')
            self._reclaim()
            norm = self.grad_norm() if max_grad_norm is not None else None
            runs = self._plan()
            for first, last, step in runs:
                o, e = f['offsets'][first], f['offsets'][last] + -(-f['params'][last].numel() // self.ALIGN) * self.ALIGN
                lib('adam_step', f['p'][o:e], f['g'][o:e], f['m'][o:e], f['v'][o:e], e - o, norm,
                    float(max_grad_norm or 0.0), float(g['lr']), float(g['betas'][0]), float(g['betas'][1]),
                    float(g['eps']), step, float(g['weight_decay']))
            self._commit(runs)
            f['touched'].clear()
            for p in g['params']:                                 # late joiners: outside the arena, one launch each
                if id(p) not in f['member'] and p.grad is not None:
                    if max_grad_norm is not None:
                        raise ValueError('fused clipping covers the arena only; a parameter outside it carries a gradient')
                    self._step_one(p, g)
            return loss
        if max_grad_norm is not None:
            raise ValueError('fused clipping needs the flat arena (use nn.utils.clip_grad_norm_ otherwise)')
        for g in self.param_groups:
            for p in g['params']:
                if p.grad is not None:
                    self._step_one(p, g)
        return loss
