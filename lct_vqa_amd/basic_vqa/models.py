"""W model (plain VQA model): same classes, constructor signatures, attribute names and
state_dict keys as reference basic_vqa/models.py:7-125; forward/backward run in the HIP library."""
import torch
import torch.nn as nn

from .. import ops, vgg
from . import config as _config


def softXEnt(pred, target):
    """Soft cross entropy loss (reference models.py:7-10)."""
    return ops.soft_xent(pred, target)


class ImgEncoder(nn.Module):
    """Frozen VGG19 (minus last fc) -> fc -> / detached L2 norm (reference models.py:12-42)."""

    def __init__(self, embed_size):
        super().__init__()
        self.model, width = vgg.truncated_vgg19(pretrained=True)
        self.fc = nn.Linear(width, embed_size)

    def trunk(self, image, conv_feature=None, stem=None):
        """The frozen part only (VGG19 minus its last fc, under no_grad): [B,4096].  It does not depend on any
        trainable parameter, so a caller may run it for batch i+1 on a second HIP stream while batch i's
        head/backward/optimizer runs (bench.py)."""
        with torch.no_grad():
            return self.model(image, conv_feature, stem)

    def conv_features(self, image):
        """vgg19.features + avgpool of the frozen trunk (no dropout in it): shareable between forwards of one batch."""
        return self.model.conv_features(image)

    def forward(self, image, trunk_feature=None, conv_feature=None):
        img_feature = self.trunk(image, conv_feature) if trunk_feature is None else trunk_feature
        return vgg.embed_feature(img_feature, self.fc)


def _single_layer(lstm):
    if lstm.num_layers != 1:
        raise NotImplementedError('lct_vqa_amd: the HIP LSTM path implements num_layers == 1 '
                                  '(the reference default, config.py:22)')
    return lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0


class QstEncoder(nn.Module):
    """Embedding -> tanh -> LSTM -> cat(h,c) -> tanh -> fc (reference models.py:45-68)."""

    def __init__(self, qst_vocab_size, word_embed_size, embed_size, num_layers, hidden_size):
        super(QstEncoder, self).__init__()
        self.word2vec = nn.Embedding(qst_vocab_size, word_embed_size)
        self.tanh = nn.Tanh()
        self.lstm = nn.LSTM(word_embed_size, hidden_size, num_layers)
        self.fc = nn.Linear(2 * num_layers * hidden_size, embed_size)

    def forward(self, question):
        qst_vec = ops.embed(question, self.word2vec.weight, True)        # [T,B,We], tanh fused
        _, hidden, cell = ops.lstm(qst_vec, *_single_layer(self.lstm))
        qst_feature = ops.CatTanhFn.apply(hidden, cell)
        return ops.linear(qst_feature, self.fc.weight, self.fc.bias)


class VqaModel(nn.Module):

    def __init__(self, embed_size, qst_vocab_size, ans_vocab_size, word_embed_size, num_layers, hidden_size):
        super(VqaModel, self).__init__()
        self.img_encoder = ImgEncoder(embed_size)
        self.qst_encoder = QstEncoder(qst_vocab_size, word_embed_size, embed_size, num_layers, hidden_size)
        self.tanh = nn.Tanh()
        self.dropout = nn.Dropout(0.5)
        self.fc1 = nn.Linear(embed_size, ans_vocab_size)
        self.fc2 = nn.Linear(ans_vocab_size, ans_vocab_size)
        self.embed_size = embed_size
        self.qst_vocab_size = qst_vocab_size
        self.ans_vocab_size = ans_vocab_size
        self.word_embed_size = word_embed_size
        self.num_layers = num_layers
        self.hidden_size = hidden_size
        self.criterion = nn.CrossEntropyLoss()

    def _head(self, img_feature, qst_feature):
        """mul -> tanh -> dropout -> fc1 -> tanh -> dropout -> fc2 (reference models.py:96-102)."""
        p = self.dropout.p
        train = ops.dropout_active(self.dropout.training, p)
        dev = img_feature.device
        scale = 1.0 / (1.0 - p) if train else 1.0
        m1 = ops.dropout_mask(tuple(img_feature.shape), dev, p) if train else None
        m2 = ops.dropout_mask((img_feature.shape[0], self.fc1.out_features), dev, p) if train else None
        # mul -> tanh -> dropout fused into the A-operand load of the fc1 GEMM (+ tanh + dropout epilogue)
        z = ops.FusionFc1Fn.apply(img_feature, qst_feature, self.fc1.weight, self.fc1.bias, m1, scale, m2, scale)
        return ops.linear(z, self.fc2.weight, self.fc2.bias)

    def forward(self, img, qst, trunk_feature=None, conv_feature=None):
        img_feature = self.img_encoder(img, trunk_feature, conv_feature)
        qst_feature = self.qst_encoder(qst)
        return self._head(img_feature, qst_feature)

    def trainable_parameters(self):
        """Parameters that can receive a gradient: everything outside the VGG19 trunk, which runs under
        no_grad (models.py:35-36).  torch.optim.Adam skips grad-None parameters; a flat arena
        (optim.FusedAdam(flatten=True)) and the DP all-reduce should only hold these."""
        return [p for n, p in self.named_parameters() if not n.startswith('img_encoder.model.')]

    def new(self):
        new_model = VqaModel(self.embed_size, self.qst_vocab_size, self.ans_vocab_size,
                             self.word_embed_size, self.num_layers, self.hidden_size)
        new_model.to(_config.active().DEVICE)
        return new_model

    def _loss(self, images, questions, labels):
        ans_out = self(images, questions)
        return ops.cross_entropy(ans_out, labels)

    def _soft_loss(self, images, questions, labels, pseudo_qst, pseudo_labels, conv_feature=None):
        """reference models.py:118-125.  Both forwards see the same images through the same frozen trunk: its
        convolutional part (deterministic, 98 % of a forward) is evaluated once; the classifier with its two live
        dropouts and everything after it run per forward, in the reference's order (so do the mask draws).  A driver that
        has already evaluated it for this batch (on another stream, say) passes it as ``conv_feature``."""
        conv = conv_feature if conv_feature is not None else self.img_encoder.conv_features(images)
        ans_out_1 = self(images, questions, conv_feature=conv)
        loss_1 = ops.cross_entropy(ans_out_1, labels)
        ans_out_2 = self(images, pseudo_qst, conv_feature=conv)
        loss_2 = softXEnt(ans_out_2, pseudo_labels)
        return loss_1 + _config.active().W_LAMBDA * loss_2
