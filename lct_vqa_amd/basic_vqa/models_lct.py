"""EF (encoder-decoder, "LCT") model: same classes / signatures / state_dict keys as reference
basic_vqa/models_lct.py:9-267; arithmetic runs in the HIP library."""
import torch
import torch.nn as nn

from .. import ops, vgg
from .._lib import lib
from . import config as _config
from .models import VqaModel as WModel, softXEnt, _single_layer  # noqa: F401


class ImgEncoder(nn.Module):
    """PC-DARTS image encoder: Network(16, E, 4) -> fc -> / detached L2 norm (reference models_lct.py:9-33)."""

    def __init__(self, embed_size, vqa_model, init_ch=16, layers=4):
        super(ImgEncoder, self).__init__()
        from .pcdarts.model_search import Network
        self.darts = Network(init_ch, embed_size, layers, vqa_model)
        in_features = self.darts.output_ch * (self.darts.output_size ** 2)
        self.fc = nn.Linear(in_features, embed_size)

    def forward(self, image):
        return vgg.embed_feature(self.darts(image), self.fc)


class ImgEncoderFixed(nn.Module):
    """VGG19 (minus last fc) -> fc -> / detached L2 norm (reference models_lct.py:35-69).
    pretrained=True => trunk frozen under no_grad."""

    def __init__(self, embed_size, pretrained=True):
        super().__init__()
        self.model, width = vgg.truncated_vgg19(pretrained=pretrained)
        self.fc = nn.Linear(width, embed_size)
        self.pretrained = pretrained

    def conv_features(self, image):
        """vgg19.features + avgpool of a FROZEN trunk (deterministic, no dropout in it): a driver that owns the loop may
        evaluate it once per batch and hand it to forward / generate of that batch (stage 1 and stage 2 of
        experiment.py:216-245 see the same images through the same frozen weights)."""
        if not self.pretrained:
            raise RuntimeError('conv_features: the trunk is trainable (pretrained=False); its features are not shareable')
        return self.model.conv_features(image)

    def forward(self, image, conv_feature=None):
        if conv_feature is not None and not self.pretrained:
            raise RuntimeError('conv_feature given for a trainable trunk')
        with torch.set_grad_enabled(torch.is_grad_enabled() and not self.pretrained):      # frozen iff "pretrained"
            img_feature = self.model(image, conv_feature)
        return vgg.embed_feature(img_feature, self.fc)


class QstEncoder(nn.Module):
    """LSTM encoder-decoder: encodes a question (initial state = image embedding) and emits
    per-token vocabulary logits; generates questions greedily or by sampling
    (reference models_lct.py:72-170)."""

    def __init__(self, qst_vocab_size, word_embed_size, embed_size, num_layers, hidden_size,
                 deterministic=True, temperature=0.1, max_length=30):
        super(QstEncoder, self).__init__()
        self.hidden_size = hidden_size
        self.deterministic = deterministic
        self.temperature = temperature
        self.max_length = max_length
        self.word2vec = nn.Embedding(qst_vocab_size, word_embed_size)
        self.tanh = nn.Tanh()
        self.lstm = nn.LSTM(word_embed_size, hidden_size, num_layers)
        self.fc1 = nn.Linear(2 * num_layers * hidden_size, embed_size)
        self.fc2 = nn.Linear(hidden_size, qst_vocab_size)
        self.softmax = torch.nn.Softmax(dim=2)
        torch.nn.init.xavier_uniform_(self.fc1.weight.data)
        torch.nn.init.xavier_uniform_(self.fc2.weight.data)
        torch.nn.init.zeros_(self.fc1.bias)
        torch.nn.init.zeros_(self.fc2.bias)

    def forward(self, question, image_embedding, token_loss=False, feature_only=False):
        """-> (question feature [B,E], token logits [B,T,QV]); with ``token_loss`` the second item is instead the scalar
        CE(logits[:, :-1], question[:, 1:]) of models_lct.py:253-255, computed without materialising the logits."""
        B = image_embedding.shape[0]
        h0 = image_embedding.reshape(B, self.hidden_size)          # view(1,-1,H), L = 1
        qst_vec = ops.embed(question, self.word2vec.weight, True)  # [T,B,We]
        out, hidden, cell = ops.lstm(qst_vec, *_single_layer(self.lstm), h0, h0)   # teacher forcing
        qst_feature = ops.CatTanhFn.apply(hidden, cell)
        qst_feature = ops.linear(qst_feature, self.fc1.weight, self.fc1.bias)
        if feature_only:                                            # generate()'s re-encoding: the token logits are not used
            return qst_feature, None
        tout = ops.TanhFn.apply(out)                                # [T,B,H]
        if token_loss:
            return qst_feature, ops.token_head_loss(tout, self.fc2.weight, self.fc2.bias, question)
        qst_out = ops.linear(tout, self.fc2.weight, self.fc2.bias)  # [T,B,QV]
        return qst_feature, qst_out.transpose(0, 1)                 # [B,T,QV] view

    def generate(self, image_embedding):
        """Generate questions for the given image embeddings (reference models_lct.py:124-157).
        Step 0 feeds tanh(embedding(<start>=2)); later steps feed embedding(pred) WITHOUT tanh."""
        B = len(image_embedding)
        dev = image_embedding.device
        h = image_embedding.reshape(B, self.hidden_size)
        c = h
        if self.deterministic and lib.query('lstm_step_fused_supported', B, self.hidden_size):
            # greedy decoding: the whole token loop in one C call (same kernels, identical token ids)
            return ops.lstm_decode_greedy(self.word2vec.weight, *_single_layer(self.lstm), self.fc2.weight, self.fc2.bias,
                                          h, c, self.max_length, start_token=2)
        start_word = torch.full((B, 1), 2, dtype=torch.long, device=dev)
        current_word = ops.embed(start_word, self.word2vec.weight, True)      # [1,B,We]
        qst_out = torch.zeros((B, self.max_length), dtype=torch.long, device=dev)
        params = _single_layer(self.lstm)
        for t in range(self.max_length):
            out, h, c = ops.lstm(current_word, *params, h, c)
            tout = ops.TanhFn.apply(out)                                       # [1,B,H]
            prob = ops.linear(tout, self.fc2.weight, self.fc2.bias)            # [1,B,QV]
            pred = self.sample(prob.transpose(0, 1))                           # [B,1]
            current_word = ops.embed(pred, self.word2vec.weight, False)
            qst_out[:, t] = pred[:, 0]
        return qst_out

    def sample(self, prob):
        """prob [B,1,QV] -> token ids [B,1] (reference models_lct.py:159-170)."""
        B = prob.shape[0]
        if self.deterministic:
            return ops.argmax_rows(prob.reshape(B, -1)).reshape(B, 1)
        soft = ops.softmax_rows(prob.reshape(B, -1), 1.0 / self.temperature)
        return torch.multinomial(soft, 1)      # host-side sampling in the reference; off the hot path


class VqaModel(nn.Module):

    def __init__(self, embed_size, qst_vocab_size, ans_vocab_size, word_embed_size, num_layers,
                 hidden_size, pretrained=True):
        super(VqaModel, self).__init__()
        self.img_encoder = None
        if _config.active().ARCH_TYPE == 'darts':
            self.img_encoder = ImgEncoder(embed_size, self)
        else:
            self.img_encoder = ImgEncoderFixed(embed_size, pretrained)
        self.qst_encoder = QstEncoder(qst_vocab_size, word_embed_size, embed_size, num_layers, hidden_size)
        self.tanh = nn.Tanh()
        self.dropout = nn.Dropout(0.5)
        self.fc1 = nn.Linear(embed_size, ans_vocab_size)
        self.fc2 = nn.Linear(ans_vocab_size, ans_vocab_size)
        self.criterion = nn.CrossEntropyLoss()
        self.embed_size = embed_size
        self.qst_vocab_size = qst_vocab_size
        self.ans_vocab_size = ans_vocab_size
        self.word_embed_size = word_embed_size
        self.num_layers = num_layers
        self.hidden_size = hidden_size
        self.pretrained = pretrained

    _head = WModel._head

    def _encode_image(self, img, conv_feature):
        return self.img_encoder(img) if conv_feature is None else self.img_encoder(img, conv_feature)

    def forward(self, img, qst, conv_feature=None):
        img_feature = self._encode_image(img, conv_feature)
        qst_feature, qst_out = self.qst_encoder(qst, img_feature)
        return self._head(img_feature, qst_feature), qst_out

    def generate(self, img, conv_feature=None):
        """(generated questions, answer logits for them) -- reference models_lct.py:220-241."""
        img_feature = self._encode_image(img, conv_feature)
        qst = self.qst_encoder.generate(img_feature)
        qst_feature, _ = self.qst_encoder(qst, img_feature, feature_only=True)
        return qst, self._head(img_feature, qst_feature)

    def trainable_parameters(self):
        """See models.VqaModel.trainable_parameters: the trunk is frozen only when pretrained."""
        frozen = isinstance(self.img_encoder, ImgEncoderFixed) and self.img_encoder.pretrained
        return [p for n, p in self.named_parameters() if not (frozen and n.startswith('img_encoder.model.'))]

    def genotype(self):
        return self.img_encoder.darts.genotype()

    def arch_parameters(self):
        return self.img_encoder.darts.arch_parameters()

    def _loss(self, images, questions, labels, conv_feature=None):
        """CE(ans) + CE(qst_out[:, :-1], qst[:, 1:]); <pad>=0 is not ignored (models_lct.py:250-257)."""
        if ops.linear_xent_fused() and questions.shape[1] > 1:      # token head fused with its cross entropy: no [B,T,QV] tensor
            img_feature = self._encode_image(images, conv_feature)
            qst_feature, qst_loss = self.qst_encoder(questions, img_feature, token_loss=True)
            return ops.cross_entropy(self._head(img_feature, qst_feature), labels) + qst_loss
        ans_out, qst_out = self(images, questions, conv_feature)
        ans_loss = ops.cross_entropy(ans_out, labels)
        qst = questions[:, 1:].flatten()
        qst_out = qst_out[:, :-1].flatten(end_dim=1)
        qst_loss = ops.cross_entropy(qst_out, qst)
        return ans_loss + qst_loss

    def new(self):
        new_vqa_model = VqaModel(self.embed_size, self.qst_vocab_size, self.ans_vocab_size,
                                 self.word_embed_size, self.num_layers, self.hidden_size, self.pretrained)
        if hasattr(self.img_encoder, 'darts'):
            new_vqa_model.img_encoder.darts = self.img_encoder.darts.new()
        new_vqa_model.to(_config.active().DEVICE)
        return new_vqa_model
