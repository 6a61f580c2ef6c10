"""3-level "Learning by Creating Tests" architect (reference basic_vqa/pcdarts/architect_lct.py:13-235):
  1. EF' = EF - ef_lr * grad L_EF(train)                      (_compute_unrolled_model)
  2. pseudo questions/answers from EF'.generate(train images); answers -> softmax(. / TEMPERATURE)
  3. W' = W - w_lr * grad [CE(W(img,qst),label) + W_LAMBDA softXEnt(W(img,pseudo_qst), pseudo_ans)]
  4. grad_W' = grad L_W'(valid)
  5. kappa  = d/dEF' of W's soft loss along grad_W'   (finite difference, pseudo-QA regenerated each time)
  6. gamma  = d/dalpha of EF's loss along kappa          (finite difference)
  7. alpha.grad = gamma * ef_lr * w_lr ; Adam(betas=(.5,.999), weight_decay) step
Host logic only; every forward/backward/generate it triggers runs in the HIP library."""
import logging

import numpy as np
import torch

from ... import ops
from .. import config as _config


def _concat(xs):
    return torch.cat([x.reshape(-1) for x in xs])


class ArchitectLct(object):

    def __init__(self, ef_model, w_model, ef_optimizer, w_optimizer):
        config = _config.active()
        self.ef_model = ef_model
        self.w_model = w_model
        self.ef_momentum = 0
        self.ef_weight_decay = 0
        self.optimizer = torch.optim.Adam(self.ef_model.arch_parameters(), lr=config.ARCH_LEARNING_RATE,
                                          betas=(0.5, 0.999), weight_decay=config.ARCH_WEIGHT_DECAY)
        self.ef_optimizer = ef_optimizer
        self.w_optimizer = w_optimizer

    def step(self, img_train, qst_train, label_train, img_valid, qst_valid, label_valid, ef_lr, w_lr):
        self.ef_optimizer.zero_grad()
        self.w_optimizer.zero_grad()
        self.optimizer.zero_grad()
        self._backward_step_unrolled(img_train, qst_train, label_train, img_valid, qst_valid, label_valid, ef_lr, w_lr)
        self.optimizer.step()

    def _pseudo_qa(self, model, img):
        pseudo_qst, pseudo_ans = model.generate(img)
        n = pseudo_ans.shape[0]
        inv_t = 1.0 / _config.active().TEMPERATURE
        pseudo_ans = ops.SoftmaxRowsFn.apply(pseudo_ans.reshape(n, -1), inv_t)       # F.softmax(ans / T, dim=1)
        return pseudo_qst, pseudo_ans

    def _backward_step_unrolled(self, img_train, qst_train, label_train, img_valid, qst_valid, label_valid, ef_lr, w_lr):
        unrolled_ef_model = self._compute_unrolled_model(img_train, qst_train, label_train, ef_lr, self.ef_optimizer,
                                                         self.ef_model, self.ef_model._loss)
        pseudo_qst, pseudo_ans = self._pseudo_qa(unrolled_ef_model, img_train)
        unrolled_w_model = self._compute_unrolled_model_2(img_train, qst_train, label_train, pseudo_qst, pseudo_ans, w_lr,
                                                          self.w_optimizer, self.w_model, self.w_model._soft_loss,
                                                          exp_zero_grad=36)
        unrolled_loss = unrolled_w_model._loss(img_valid, qst_valid, label_valid)
        grad_wprime = self._calc_grad(unrolled_loss, unrolled_w_model.parameters, exp_zero_grad=36)

        def pseudo_qa_fn():
            return self._pseudo_qa(unrolled_ef_model, img_train)

        def qa_fn():
            return qst_train, label_train

        kappa = self._hessian_vector_product_2(grad_wprime, img_train, qa_fn, pseudo_qa_fn, self.w_model,
                                               self.w_model._soft_loss, unrolled_ef_model.parameters, exp_zero_grad=2)
        gamma = self._hessian_vector_product(kappa, img_train, qa_fn, self.ef_model, self.ef_model._loss,
                                             self.ef_model.arch_parameters, exp_zero_grad=0)
        for v, g in zip(self.ef_model.arch_parameters(), gamma):
            if v.grad is None:
                v.grad = (g.data * ef_lr * w_lr).clone()
            else:
                v.grad.data.copy_(g.data * ef_lr * w_lr)
        logging.info('| TRAIN SET | STAGE3 | W\'-Val-Loss: {:.4f}'.format(unrolled_loss.item()))

    def _compute_unrolled_model(self, img, qst, label, eta, optimizer, model, loss_fn, exp_zero_grad=0,
                                weight_decay=0, momentum=0):
        loss = loss_fn(img, qst, label)
        theta = _concat(model.parameters()).data
        try:
            moment = _concat(optimizer.state[v]['momentum_buffer'] for v in model.parameters()).mul_(self.network_momentum)
        except Exception:                     # the reference never defines network_momentum: always zeros
            moment = torch.zeros_like(theta)
        grads = self._calc_grad(loss, model.parameters, exp_zero_grad)
        dtheta = _concat(grads).data + weight_decay * theta
        return self._construct_model_from_theta(theta - eta * (moment + dtheta), model)

    def _compute_unrolled_model_2(self, img, qst, label, pseudo_qst, pseudo_label, eta, optimizer, model, loss_fn,
                                  exp_zero_grad=0, weight_decay=0, momentum=0):
        loss = loss_fn(img, qst, label, pseudo_qst, pseudo_label)
        theta = _concat(model.parameters()).data
        try:
            moment = _concat(optimizer.state[v]['momentum_buffer'] for v in model.parameters()).mul_(self.network_momentum)
        except Exception:
            moment = torch.zeros_like(theta)
        grads = self._calc_grad(loss, model.parameters, exp_zero_grad)
        dtheta = _concat(grads).data + weight_decay * theta
        return self._construct_model_from_theta(theta - eta * (moment + dtheta), model)

    def _construct_model_from_theta(self, theta, model):
        model_new = model.new()
        model_dict = model.state_dict()
        params, offset = {}, 0
        for k, v in model.named_parameters():
            v_length = int(np.prod(v.size()))
            params[k] = theta[offset: offset + v_length].view(v.size())
            offset += v_length
        assert offset == len(theta)
        model_dict.update(params)
        model_new.load_state_dict(model_dict)
        return model_new.to(_config.active().DEVICE)

    def _calc_grad(self, loss, param_fn, exp_zero_grad=0):
        grads = list(torch.autograd.grad(loss, param_fn(), allow_unused=True))
        num_zero_grad = 0
        for i, p in enumerate(param_fn()):
            if grads[i] is None:
                grads[i] = torch.zeros_like(p)
                num_zero_grad += 1
            else:
                assert grads[i].shape == p.shape
        assert num_zero_grad == exp_zero_grad, (num_zero_grad, exp_zero_grad)
        return grads

    def _hessian_vector_product(self, vector, img, qa_fn, model, loss_fn, param_fn, r=1e-2, exp_zero_grad=0):
        R = r / _concat(vector).norm()
        for p, v in zip(model.parameters(), vector):
            p.data.add_(v, alpha=R)
        qst, ans = qa_fn()
        grads_p = self._calc_grad(loss_fn(img, qst, ans), param_fn, exp_zero_grad)
        for p, v in zip(model.parameters(), vector):
            p.data.sub_(v, alpha=2 * R)
        qst, ans = qa_fn()
        grads_n = self._calc_grad(loss_fn(img, qst, ans), param_fn, exp_zero_grad)
        for p, v in zip(model.parameters(), vector):
            p.data.add_(v, alpha=R)
        return [(x - y).div_(2 * R) for x, y in zip(grads_p, grads_n)]

    def _hessian_vector_product_2(self, vector, img, qa_fn, pseudo_qa_fn, model, loss_fn, param_fn, r=1e-2,
                                  exp_zero_grad=0):
        R = r / _concat(vector).norm()
        for p, v in zip(model.parameters(), vector):
            p.data.add_(v, alpha=R)
        qst, ans = qa_fn()
        pseudo_qst, pseudo_ans = pseudo_qa_fn()
        grads_p = self._calc_grad(loss_fn(img, qst, ans, pseudo_qst, pseudo_ans), param_fn, exp_zero_grad)
        for p, v in zip(model.parameters(), vector):
            p.data.sub_(v, alpha=2 * R)
        qst, ans = qa_fn()
        pseudo_qst, pseudo_ans = pseudo_qa_fn()
        grads_n = self._calc_grad(loss_fn(img, qst, ans, pseudo_qst, pseudo_ans), param_fn, exp_zero_grad)
        for p, v in zip(model.parameters(), vector):
            p.data.add_(v, alpha=R)
        return [(x - y).div_(2 * R) for x, y in zip(grads_p, grads_n)]
