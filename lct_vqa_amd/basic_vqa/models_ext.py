"""Extensions by analogy named by BASELINE.json but absent from the reference:
the GRU question encoder of configs[2] ("ResNet-152+GRU").  The reference pins nothing here; the oracle is
torch.nn.GRU + the same surrounding arithmetic as basic_vqa/models.py:45-68 with (h) in place of cat(h,c).
ResNet-152 is NOT built (the image encoder of these classes is the VGG19 trunk)."""
import torch.nn as nn

from .. import ops
from .models import VqaModel


class QstEncoderGRU(nn.Module):
    """Embedding -> tanh -> GRU(zero init) -> tanh(h) -> Linear(L*H, E)."""

    def __init__(self, qst_vocab_size, word_embed_size, embed_size, num_layers, hidden_size):
        super(QstEncoderGRU, self).__init__()
        if num_layers != 1:
            raise NotImplementedError('lct_vqa_amd: the HIP GRU path implements num_layers == 1')
        self.word2vec = nn.Embedding(qst_vocab_size, word_embed_size)
        self.tanh = nn.Tanh()
        self.gru = nn.GRU(word_embed_size, hidden_size, num_layers)
        self.fc = nn.Linear(num_layers * hidden_size, embed_size)

    def forward(self, question):
        qst_vec = ops.embed(question, self.word2vec.weight, True)
        g = self.gru
        _, hidden = ops.gru(qst_vec, g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0)
        return ops.linear(ops.TanhFn.apply(hidden), self.fc.weight, self.fc.bias)


class VqaModelGRU(VqaModel):
    """models.VqaModel with the GRU question encoder."""

    def __init__(self, embed_size, qst_vocab_size, ans_vocab_size, word_embed_size, num_layers, hidden_size):
        super(VqaModelGRU, self).__init__(embed_size, qst_vocab_size, ans_vocab_size, word_embed_size, num_layers, hidden_size)
        self.qst_encoder = QstEncoderGRU(qst_vocab_size, word_embed_size, embed_size, num_layers, hidden_size)
