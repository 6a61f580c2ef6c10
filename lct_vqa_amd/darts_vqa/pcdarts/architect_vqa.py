"""2-level PC-DARTS architect of the darts_vqa tree: ``Architect(model, args).step(...)`` as in reference
darts_vqa/pcdarts/architect_vqa.py:10-145.  Same estimator as basic_vqa's Architect; the losses take the tree's
``qst_only`` switch (``args.qst_only``: answer head and image-feature fc are unreached, 6 tensors without gradient, a count
the reference asserts on the training loss), learning rate / weight decay come from ``args``.  Host logic only."""
from ... import bilevel


class Architect(object):

    def __init__(self, model, args):
        self.model, self.args = model, args
        self.exp_zero_grad = 6 if args.qst_only else 0
        self.optimizer = bilevel.arch_optimizer(model.arch_parameters(), args.arch_learn_rate, args.arch_wt_decay)
        self.trace = None      # set to a dict to have the next step leave its intermediates there (parity tests)
        self._twins = {}       # the virtual-step twins of the previous step, re-used (bilevel.virtual_step)

    def step(self, img_train, qst_train, label_train, img_valid, qst_valid, label_valid, eta, network_optimizer=None,
             unrolled=True):
        self.optimizer.zero_grad()
        self.compute_grads(img_train, qst_train, label_train, img_valid, qst_valid, label_valid, eta, unrolled)
        self.optimizer.step()

    @bilevel.architecture_pass
    def compute_grads(self, img_train, qst_train, label_train, img_valid, qst_valid, label_valid, eta, unrolled=True):
        """Leave the architecture gradient in the ``.grad`` of the architecture parameters (no optimizer step)."""
        alphas = self.model.arch_parameters()
        if unrolled:
            dalpha = self._second_order(img_train, qst_train, label_train, img_valid, qst_valid, label_valid, eta)
        else:                                     # first order: the reference evaluates this loss WITHOUT qst_only
            dalpha = bilevel.grads(self.model._loss(img_valid, qst_valid, label_valid), alphas)
        bilevel.set_grads(alphas, dalpha)

    def _train_loss(self, model, img, qst, label):
        return model._loss(img, qst, label, self.args.qst_only)

    def _second_order(self, img_train, qst_train, label_train, img_valid, qst_valid, label_valid, eta):
        model = self.model
        weights = list(model.parameters())
        g_train = bilevel.grads(self._train_loss(model, img_train, qst_train, label_train), weights, self.exp_zero_grad)
        ahead = bilevel.virtual_step(model, g_train, eta, self._twins)
        ahead_alphas, ahead_weights = ahead.arch_parameters(), list(ahead.parameters())
        g = bilevel.grads(self._train_loss(ahead, img_valid, qst_valid, label_valid), ahead_alphas + ahead_weights)
        dalpha, v = g[:len(ahead_alphas)], g[len(ahead_alphas):]
        implicit = bilevel.fd_grad(weights, v, lambda: bilevel.grads(self._train_loss(model, img_train, qst_train, label_train),
                                                                    model.arch_parameters(), weight_grads=False))
        if self.trace is not None:
            self.trace.update(dalpha_direct=dalpha, v=v, implicit=implicit)
        return [d - eta * i for d, i in zip(dalpha, implicit)]
