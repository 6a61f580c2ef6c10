"""Validation helpers of the reference's driver (basic_vqa/misc.py:42-58, experiment.py:296-340) on the device.

Same names and argument meaning as the reference (`num_correct(pred, multi_choice)`, `num_correct_qst(qst_pred, qst)`),
plus `Evaluator`, which keeps the epoch counters in HBM so that the validation loop has no per-batch `.item()` /
`.cpu()` synchronisation (the reference moves every prediction to the host, experiment.py:322-323).
"""
import torch

from .. import ops


def _stack_choices(multi_choice, device):
    """The loader yields `answer_multi_choice` as a list of K LongTensors [B] (data_loader_v2.py collate); accept that
    or an already stacked [K,B] tensor."""
    if isinstance(multi_choice, (list, tuple)):
        multi_choice = torch.stack([torch.as_tensor(a) for a in multi_choice])
    return multi_choice.to(device=device, dtype=torch.int64, non_blocking=True)


def num_correct(pred, multi_choice):
    """#samples whose prediction equals one of its multiple-choice answers (misc.py:42-45).  pred [B] int64."""
    if not pred.is_cuda:
        raise RuntimeError('lct_vqa_amd: num_correct expects device predictions (no CPU fallback)')
    ch = _stack_choices(multi_choice, pred.device)
    return int((ch == pred.unsqueeze(0)).any(dim=0).sum().item())


def num_correct_qst(qst_pred, qst):
    """(#questions with 0, <=3, <=5 wrong tokens): qst_pred [B,T,QV] logits against qst[:, 1:] (misc.py:47-58)."""
    B, T, QV = qst_pred.shape
    tok = ops.argmax_rows(qst_pred.reshape(B * T, QV)).reshape(B, T)[:, :-1]
    err_cnt = (qst[:, 1:] != tok).sum(dim=1)
    return int((err_cnt == 0).sum().item()), int((err_cnt <= 3).sum().item()), int((err_cnt <= 5).sum().item())


class Evaluator(object):
    """Device-resident counters of the reference's `val()` (experiment.py:296-340): running CE loss, Exp1 / Exp2
    multiple-choice hits ('<unk>' accepted / rejected).  `update` launches one fused kernel (argmax + unk masking +
    choice match) and the CE kernel; nothing is read back until `result()`."""

    def __init__(self, ans_unk_idx, device):
        self.unk = int(ans_unk_idx)
        self.counts = torch.zeros(2, dtype=torch.int64, device=device)
        self.loss = torch.zeros((), dtype=torch.float32, device=device)
        self.batches = 0
        self.samples = 0

    @torch.no_grad()
    def update(self, output, label, multi_choice):
        ch = _stack_choices(multi_choice, output.device)
        pred, _ = ops.eval_multi_choice(output, ch, self.unk, self.counts)
        self.loss += ops.cross_entropy(output, label)
        self.batches += 1
        self.samples += output.shape[0]
        return pred

    def result(self):
        c = self.counts.tolist()
        return {'loss': float(self.loss.item()) / max(1, self.batches), 'acc_exp1': c[0] / max(1, self.samples),
                'acc_exp2': c[1] / max(1, self.samples), 'corr_exp1': c[0], 'corr_exp2': c[1]}
