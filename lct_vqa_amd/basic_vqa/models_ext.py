"""Extensions by analogy named by BASELINE.json but absent from the reference:
the GRU question encoder of configs[2] ("ResNet-152+GRU").  The reference pins nothing here; the oracle is
torch.nn.GRU + the same surrounding arithmetic as basic_vqa/models.py:45-68 with (h) in place of cat(h,c).
``ImgEncoderResNet`` / ``VqaModelResNetGRU``: the same model with a FROZEN ResNet-152 trunk (lct_vqa_amd/resnet.py) in place of
the VGG19 -- the reference's ImgEncoder pattern (models.py:12-42: pretrained trunk under no_grad -> fc -> / detached L2
norm; models_lct.py:62-63: ``pretrained=False`` trains the trunk) applied to the encoder configs[2] names."""
import torch
import torch.nn as nn

from .. import ops, vgg
from ..resnet import ResNet152Trunk
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


class ImgEncoderResNet(nn.Module):
    """ResNet-152 (minus fc) -> fc -> / detached L2 norm: reference models.py:12-42 (pretrained=True: frozen trunk under
    no_grad) and models_lct.py:35-69 (pretrained=False: the trunk trains) with the other trunk."""

    def __init__(self, embed_size, pretrained=True):
        super().__init__()
        self.model = ResNet152Trunk()
        self.fc = nn.Linear(self.model.out_features, embed_size)
        self.pretrained = pretrained

    def trunk(self, image):
        with torch.set_grad_enabled(torch.is_grad_enabled() and not self.pretrained):
            return self.model(image)

    def forward(self, image, trunk_feature=None, conv_feature=None):
        feat = self.trunk(image) if trunk_feature is None else trunk_feature
        return vgg.embed_feature(feat, self.fc)


class VqaModelResNetGRU(VqaModelGRU):
    """BASELINE.json configs[2]'s model shape: ResNet-152 image encoder (frozen) + GRU question encoder + the fusion head."""

    def __init__(self, embed_size, qst_vocab_size, ans_vocab_size, word_embed_size, num_layers, hidden_size, pretrained=True):
        super().__init__(embed_size, qst_vocab_size, ans_vocab_size, word_embed_size, num_layers, hidden_size)
        self.img_encoder = ImgEncoderResNet(embed_size, pretrained)

    def trainable_parameters(self):
        frozen = self.img_encoder.pretrained
        return [p for n, p in self.named_parameters() if not (frozen and n.startswith('img_encoder.model.'))]
