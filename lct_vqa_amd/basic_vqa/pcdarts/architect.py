"""2-level PC-DARTS architect, same constructor and ``step`` signature as reference basic_vqa/pcdarts/architect.py:12-109.

    alpha.grad = d L_val(theta', alpha) / d alpha  -  eta * d/d alpha [ grad_theta L_train(theta, alpha) . v ],
    theta' = theta - eta * grad_theta L_train,   v = grad_theta' L_val(theta', alpha)

with the second term as a central finite difference (lct_vqa_amd.bilevel.fd_grad), then one Adam step on alpha.
Host logic only: every forward / backward it triggers runs in the HIP library."""
from ... import bilevel
from .. import config as _config


class Architect(object):

    def __init__(self, model):
        cfg = _config.active()
        self.model = model
        self.optimizer = bilevel.arch_optimizer(model.arch_parameters(), cfg.ARCH_LEARNING_RATE, cfg.ARCH_WEIGHT_DECAY)
        self.trace = None      # set to a dict to have the next step leave its intermediates there (parity tests)
        self._twins = {}       # the virtual-step twins of the previous step, re-used (bilevel.virtual_step)

    def step(self, img_train, qst_train, label_train, img_valid, qst_valid, label_valid, eta=None,
             network_optimizer=None, unrolled=True):
        """One update of the architecture parameters.  ``network_optimizer`` is accepted for call compatibility: the
        reference reads a momentum buffer from it and multiplies it by zero."""
        self.optimizer.zero_grad()
        self.compute_grads(img_train, qst_train, label_train, img_valid, qst_valid, label_valid, eta, unrolled)
        self.optimizer.step()

    @bilevel.architecture_pass
    def compute_grads(self, img_train, qst_train, label_train, img_valid, qst_valid, label_valid, eta=None, unrolled=True):
        """Leave d L_val / d alpha (second order when ``unrolled``) in the ``.grad`` of the architecture parameters."""
        alphas = self.model.arch_parameters()
        if unrolled:
            dalpha = self._second_order(img_train, qst_train, label_train, img_valid, qst_valid, label_valid, eta)
        else:
            dalpha = bilevel.grads(self.model._loss(img_valid, qst_valid, label_valid), alphas)
        bilevel.set_grads(alphas, dalpha)

    def _second_order(self, img_train, qst_train, label_train, img_valid, qst_valid, label_valid, eta):
        model = self.model
        weights = list(model.parameters())
        ahead = bilevel.virtual_step(model, bilevel.grads(model._loss(img_train, qst_train, label_train), weights), eta, self._twins)
        ahead_alphas, ahead_weights = ahead.arch_parameters(), list(ahead.parameters())
        g = bilevel.grads(ahead._loss(img_valid, qst_valid, label_valid), ahead_alphas + ahead_weights)
        dalpha, v = g[:len(ahead_alphas)], g[len(ahead_alphas):]
        implicit = bilevel.fd_grad(weights, v, lambda: bilevel.grads(model._loss(img_train, qst_train, label_train),
                                                                    model.arch_parameters(), weight_grads=False))
        if self.trace is not None:
            self.trace.update(dalpha_direct=dalpha, v=v, implicit=implicit)
        return [d - eta * i for d, i in zip(dalpha, implicit)]
