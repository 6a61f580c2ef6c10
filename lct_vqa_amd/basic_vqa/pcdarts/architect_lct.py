"""3-level "Learning by Creating Tests" architect: ``ArchitectLct(ef_model, w_model, ef_optimizer, w_optimizer).step(...)``
as in reference basic_vqa/pcdarts/architect_lct.py:13-235.

    EF' = EF - ef_lr * grad L_EF(train)                                                  virtual step of the examiner
    (q~, a~) = EF'.generate(train images),  a~ <- softmax(a~ / TEMPERATURE)               the tests it creates
    W'  = W - w_lr * grad [ CE(W(img, qst), label) + W_LAMBDA * softXEnt(W(img, q~), a~) ]  virtual step of the student
    v     = grad_W' L_W'(valid)
    kappa = d/dEF' [ grad_W (soft loss) . v ]      finite difference in W; the tests are re-generated at both probes
    gamma = d/dalpha [ grad_EF L_EF(train) . kappa ]      finite difference in EF
    alpha.grad = gamma * ef_lr * w_lr,  then Adam on alpha

The expected numbers of unreached tensors (36: W's frozen trunk; 2: EF'.qst_encoder.fc2 under W's soft loss) are the
reference's assertions.  Host logic only; every forward / backward / generate runs in the HIP library."""
import logging

from ... import bilevel, ops
from .. import config as _config


class ArchitectLct(object):

    W_FROZEN = 36          # tensors of the W model that never receive a gradient (VGG19 under no_grad + its holder)
    EF_UNREACHED = 2       # EF'.qst_encoder.fc2.{weight,bias}: the token head does not feed generate()'s outputs

    def __init__(self, ef_model, w_model, ef_optimizer, w_optimizer):
        cfg = _config.active()
        self.ef_model, self.w_model = ef_model, w_model
        self.ef_optimizer, self.w_optimizer = ef_optimizer, w_optimizer
        self.optimizer = bilevel.arch_optimizer(ef_model.arch_parameters(), cfg.ARCH_LEARNING_RATE, cfg.ARCH_WEIGHT_DECAY)
        self.trace = None      # set to a dict to have the next step leave its intermediates there (parity tests)
        self._twins = {}       # the virtual-step twins of the previous step, re-used (bilevel.virtual_step)

    def step(self, img_train, qst_train, label_train, img_valid, qst_valid, label_valid, ef_lr, w_lr):
        for opt in (self.ef_optimizer, self.w_optimizer, self.optimizer):
            opt.zero_grad()
        self.compute_grads(img_train, qst_train, label_train, img_valid, qst_valid, label_valid, ef_lr, w_lr)
        self.optimizer.step()

    @bilevel.architecture_pass
    def compute_grads(self, img_train, qst_train, label_train, img_valid, qst_valid, label_valid, ef_lr, w_lr):
        """Leave gamma * ef_lr * w_lr in the ``.grad`` of the examiner's architecture parameters (no optimizer step)."""
        gamma, val_loss = self._alpha_direction(img_train, qst_train, label_train, img_valid, qst_valid, label_valid, ef_lr, w_lr)
        bilevel.set_grads(self.ef_model.arch_parameters(), [g * ef_lr * w_lr for g in gamma])
        logging.info("| TRAIN SET | STAGE3 | W'-Val-Loss: {:.4f}".format(val_loss.item()))

    @staticmethod
    def _tests(examiner, img):
        """Questions the examiner generates for ``img`` and its answer distribution at TEMPERATURE (differentiable in
        the examiner's parameters through the answer logits)."""
        qst, ans = examiner.generate(img)
        return qst, ops.SoftmaxRowsFn.apply(ans.reshape(ans.shape[0], -1), 1.0 / _config.active().TEMPERATURE)

    def _alpha_direction(self, img_train, qst_train, label_train, img_valid, qst_valid, label_valid, ef_lr, w_lr):
        ef, w = self.ef_model, self.w_model
        ef_weights, w_weights = list(ef.parameters()), list(w.parameters())

        ef_ahead = bilevel.virtual_step(ef, bilevel.grads(ef._loss(img_train, qst_train, label_train), ef_weights, 0), ef_lr, self._twins, 'ef')
        pseudo_qst, pseudo_ans = self._tests(ef_ahead, img_train)
        soft = w._soft_loss(img_train, qst_train, label_train, pseudo_qst, pseudo_ans)
        w_ahead = bilevel.virtual_step(w, bilevel.grads(soft, w_weights, self.W_FROZEN), w_lr, self._twins, 'w')

        val_loss = w_ahead._loss(img_valid, qst_valid, label_valid)
        v = bilevel.grads(val_loss, w_ahead.parameters(), self.W_FROZEN)
        if self.trace is not None:
            self.trace.update(ef_ahead=ef_ahead, w_ahead=w_ahead, pseudo_qst=pseudo_qst, pseudo_ans=pseudo_ans.detach(),
                              val_loss=val_loss.detach(), grad_wprime=v)

        def soft_loss_grad_wrt_examiner():
            q, a = self._tests(ef_ahead, img_train)                  # regenerated: the probes move W, not EF'
            return bilevel.grads(w._soft_loss(img_train, qst_train, label_train, q, a), ef_ahead.parameters(), self.EF_UNREACHED)

        kappa = bilevel.fd_grad(w_weights, v, soft_loss_grad_wrt_examiner)
        gamma = bilevel.fd_grad(ef_weights, kappa,
                                lambda: bilevel.grads(ef._loss(img_train, qst_train, label_train), ef.arch_parameters(), 0, weight_grads=False))
        if self.trace is not None:
            self.trace.update(kappa=kappa, gamma=gamma)
        return gamma, val_loss
