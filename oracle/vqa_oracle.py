"""TEST INFRASTRUCTURE -- CPU oracle for the VQA hot path.  NOT the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  The product path (lct_vqa_amd/) never imports it and fails loudly
when its HIP library is missing.

A plain fp32 restatement (functional, explicit formulas, torch CPU tensors) of
the arithmetic of the reference's hot path.  Every function cites the reference
lines it follows (paths relative to /root/reference).  Parameters are passed as
a dict keyed by the reference's own state_dict names so that a reference
checkpoint is directly usable.

Pinned (tests/test_oracle_golden.py) against outputs of the reference's own
modules, generated in the build container by tests/make_golden.py and committed
under tests/golden/.  The one piece not pinned by the reference is the VGG19
architecture itself (torchvision, un-vendored; see oracle/tv_vgg19.py).

Dropout: the reference draws Bernoulli masks from torch's CPU generator.  The
oracle takes the masks explicitly (``masks`` = list of 0/1 float tensors in
call order); ``draw_masks`` reproduces the reference's draw order so that a
seeded reference run and the oracle see identical masks.
"""
import math
import torch
import torch.nn.functional as F

VGG_CFG_E = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M',
             512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']


# ----------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------
def soft_xent(pred, target):
    """basic_vqa/models.py:7-10  softXEnt."""
    logp = pred - torch.logsumexp(pred, dim=1, keepdim=True)
    return -(target * logp).sum() / pred.shape[0]


def cross_entropy(logits, labels):
    """nn.CrossEntropyLoss() (mean) as used at basic_vqa/models.py:89,115."""
    logp = logits - torch.logsumexp(logits, dim=1, keepdim=True)
    return -logp.gather(1, labels.view(-1, 1)).mean()


# ----------------------------------------------------------------------------
# VGG19 image encoder
# ----------------------------------------------------------------------------
def _drop(x, masks, p=0.5):
    """nn.Dropout(p) in train mode with an explicit 0/1 mask; identity if masks is None."""
    if masks is None:
        return x
    m = masks.pop(0)
    assert m.shape == x.shape, (m.shape, x.shape)
    return x * m * (1.0 / (1.0 - p))


def vgg19_trunk(image, P, prefix, masks=None):
    """torchvision vgg19 minus its last fc, as built at basic_vqa/models.py:23-28
    and called at models.py:36.  ``prefix`` e.g. 'img_encoder.model.'"""
    x = image
    idx = 0
    for v in VGG_CFG_E:
        if v == 'M':
            x = F.max_pool2d(x, 2, 2)
            idx += 1
        else:
            x = F.conv2d(x, P[f'{prefix}features.{idx}.weight'],
                         P[f'{prefix}features.{idx}.bias'], padding=1)
            x = torch.relu(x)
            idx += 2
    x = F.adaptive_avg_pool2d(x, (7, 7))
    x = x.flatten(1)
    x = torch.relu(F.linear(x, P[f'{prefix}classifier.0.weight'], P[f'{prefix}classifier.0.bias']))
    x = _drop(x, masks)
    x = torch.relu(F.linear(x, P[f'{prefix}classifier.3.weight'], P[f'{prefix}classifier.3.bias']))
    x = _drop(x, masks)
    return x


def img_encoder(image, P, masks=None, frozen=True, prefix='img_encoder.'):
    """basic_vqa/models.py:31-42 (ImgEncoder.forward) and
    models_lct.py:55-69 (ImgEncoderFixed.forward; frozen == pretrained)."""
    if frozen:
        with torch.no_grad():
            feat = vgg19_trunk(image, P, prefix + 'model.', masks)
    else:
        feat = vgg19_trunk(image, P, prefix + 'model.', masks)
    y = F.linear(feat, P[prefix + 'fc.weight'], P[prefix + 'fc.bias'])
    l2 = y.norm(p=2, dim=1, keepdim=True).detach()          # models.py:39 -- detached
    return y / l2


# ----------------------------------------------------------------------------
# LSTM (nn.LSTM, 1 direction, gate order i,f,g,o)
# ----------------------------------------------------------------------------
def lstm(x, P, prefix, h0=None, c0=None, num_layers=1):
    """nn.LSTM(word_embed, hidden, num_layers) forward (models.py:52,61).
    x [T,B,I] -> out [T,B,H], (h [L,B,H], c [L,B,H])."""
    T, B, _ = x.shape
    hs, cs = [], []
    inp = x
    for l in range(num_layers):
        w_ih = P[f'{prefix}weight_ih_l{l}']
        w_hh = P[f'{prefix}weight_hh_l{l}']
        b = P[f'{prefix}bias_ih_l{l}'] + P[f'{prefix}bias_hh_l{l}']
        H = w_hh.shape[1]
        h = x.new_zeros(B, H) if h0 is None else h0[l]
        c = x.new_zeros(B, H) if c0 is None else c0[l]
        outs = []
        for t in range(T):
            g = inp[t] @ w_ih.t() + h @ w_hh.t() + b
            i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
            i, f, gg, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)
            c = f * c + i * gg
            h = o * torch.tanh(c)
            outs.append(h)
        inp = torch.stack(outs, 0)
        hs.append(h)
        cs.append(c)
    return inp, (torch.stack(hs, 0), torch.stack(cs, 0))


def _hc_feature(hidden, cell):
    """models.py:62-65: cat(h,c,dim=2) -> transpose(0,1) -> reshape(B,-1) -> tanh."""
    f = torch.cat((hidden, cell), 2).transpose(0, 1)
    f = f.reshape(f.shape[0], -1)
    return torch.tanh(f)


def qst_encoder(question, P, num_layers=1, prefix='qst_encoder.'):
    """basic_vqa/models.py:55-68 (QstEncoder.forward of the W model)."""
    v = torch.tanh(P[prefix + 'word2vec.weight'][question])      # [B,T,We]
    v = v.transpose(0, 1)
    _, (h, c) = lstm(v, P, prefix + 'lstm.', num_layers=num_layers)
    f = _hc_feature(h, c)
    return F.linear(f, P[prefix + 'fc.weight'], P[prefix + 'fc.bias'])


def lct_qst_encoder(question, img_feat, P, num_layers=1, prefix='qst_encoder.',
                    feat_head='fc1', tok_head='fc2'):
    """basic_vqa/models_lct.py:96-122 (QstEncoder.forward of the EF model).
    darts_vqa/vqa_model.py swaps the roles of fc1/fc2 (feat_head='fc2', tok_head='fc1')."""
    H = P[prefix + 'lstm.weight_hh_l0'].shape[1]
    hs = img_feat.view(1, -1, H)                                   # models_lct.py:99
    v = torch.tanh(P[prefix + 'word2vec.weight'][question]).transpose(0, 1)
    out, (h, c) = lstm(v, P, prefix + 'lstm.', h0=hs, c0=hs, num_layers=num_layers)
    f = _hc_feature(h, c)
    qf = F.linear(f, P[f'{prefix}{feat_head}.weight'], P[f'{prefix}{feat_head}.bias'])
    out = torch.tanh(out.transpose(0, 1))
    qst_out = F.linear(out, P[f'{prefix}{tok_head}.weight'], P[f'{prefix}{tok_head}.bias'])
    return qf, qst_out


def lct_generate_tokens(img_feat, P, max_length=30, prefix='qst_encoder.', tok_head='fc2', margins=None):
    """basic_vqa/models_lct.py:124-157 (QstEncoder.generate), deterministic=True.
    Step-0 input = tanh(embedding(2)); later inputs = embedding(pred) WITHOUT tanh
    (models_lct.py:137-138 vs :153-154).  ``margins``: a list that receives the [B] top-2 gap of every step's logits
    (fixture generation: how decisive each greedy choice was)."""
    B = img_feat.shape[0]
    H = P[prefix + 'lstm.weight_hh_l0'].shape[1]
    emb = P[prefix + 'word2vec.weight']
    h = img_feat.view(B, H)
    c = img_feat.view(B, H)
    w_ih, w_hh = P[prefix + 'lstm.weight_ih_l0'], P[prefix + 'lstm.weight_hh_l0']
    b = P[prefix + 'lstm.bias_ih_l0'] + P[prefix + 'lstm.bias_hh_l0']
    cur = torch.tanh(emb[torch.full((B,), 2, dtype=torch.long)])
    toks = torch.zeros(B, max_length, dtype=torch.long)
    for t in range(max_length):
        g = cur @ w_ih.t() + h @ w_hh.t() + b
        i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        prob = F.linear(torch.tanh(h), P[f'{prefix}{tok_head}.weight'], P[f'{prefix}{tok_head}.bias'])
        pred = torch.argmax(prob, 1)
        if margins is not None:
            top2 = prob.topk(2, dim=1).values
            margins.append(top2[:, 0] - top2[:, 1])
        toks[:, t] = pred
        cur = emb[pred]
    return toks


# ----------------------------------------------------------------------------
# fusion head + full models
# ----------------------------------------------------------------------------
def fusion_head(img_f, qst_f, P, masks=None):
    """basic_vqa/models.py:96-102."""
    z = torch.tanh(img_f * qst_f)
    z = _drop(z, masks)
    z = torch.tanh(F.linear(z, P['fc1.weight'], P['fc1.bias']))
    z = _drop(z, masks)
    return F.linear(z, P['fc2.weight'], P['fc2.bias'])


def w_model_forward(image, question, P, masks=None, num_layers=1):
    """basic_vqa/models.py:91-104 (VqaModel.forward).  masks (train mode): in call order
    [vgg drop1 [B,4096], vgg drop2 [B,4096], head drop1 [B,E], head drop2 [B,A]]."""
    masks = None if masks is None else list(masks)
    img_f = img_encoder(image, P, masks, frozen=True)
    qst_f = qst_encoder(question, P, num_layers)
    return fusion_head(img_f, qst_f, P, masks)


def ef_model_forward(image, question, P, masks=None, pretrained=True, num_layers=1):
    """basic_vqa/models_lct.py:200-218 (VqaModel.forward, ARCH_TYPE='fixed')."""
    masks = None if masks is None else list(masks)
    img_f = img_encoder(image, P, masks, frozen=pretrained)
    qst_f, qst_out = lct_qst_encoder(question, img_f, P, num_layers)
    return fusion_head(img_f, qst_f, P, masks), qst_out


def ef_loss(image, question, label, P, masks=None, pretrained=True):
    """basic_vqa/models_lct.py:250-257 (_loss): CE(ans) + CE(qst_out[:, :-1], qst[:, 1:]);
    <pad>=0 is NOT ignored."""
    ans, qst_out = ef_model_forward(image, question, P, masks, pretrained)
    l1 = cross_entropy(ans, label)
    l2 = cross_entropy(qst_out[:, :-1].flatten(end_dim=1), question[:, 1:].flatten())
    return l1 + l2


def ef_generate(image, P, masks=None, pretrained=True, max_length=30):
    """basic_vqa/models_lct.py:220-241 (VqaModel.generate): returns (qst, answer logits)."""
    masks = None if masks is None else list(masks)
    img_f = img_encoder(image, P, masks, frozen=pretrained)
    qst = lct_generate_tokens(img_f, P, max_length)
    qst_f, _ = lct_qst_encoder(qst, img_f, P)
    return qst, fusion_head(img_f, qst_f, P, masks)


def w_soft_loss(image, question, label, pseudo_qst, pseudo_label, P, w_lambda=1.0,
                masks1=None, masks2=None):
    """basic_vqa/models.py:117-125 (_soft_loss)."""
    l1 = cross_entropy(w_model_forward(image, question, P, masks1), label)
    l2 = soft_xent(w_model_forward(image, pseudo_qst, P, masks2), pseudo_label)
    return l1 + w_lambda * l2


def draw_masks(shapes, p=0.5):
    """Draw dropout keep-masks exactly as nn.Dropout does on CPU (same generator
    consumption for the same shape/dtype), in the given order."""
    out = []
    for s in shapes:
        out.append((F.dropout(torch.ones(s), p, True) != 0).float())
    return out


def w_mask_shapes(B, A, E=512):
    return [(B, 4096), (B, 4096), (B, E), (B, A)]


# ----------------------------------------------------------------------------
# optimiser step as used on the path
# ----------------------------------------------------------------------------
def clip_grad_norm(grads, max_norm):
    """nn.utils.clip_grad_norm_(params, 5) at basic_vqa/experiment.py:226,256."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return [g * coef for g in grads], total


def adam_step(p, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam(lr=1e-3) defaults (basic_vqa/model_factory.py:20-21,41-42)."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


def w_train_step(image, question, label, P, opt_state, masks=None, lr=1e-3, clip=5.0):
    """One STAGE-style weight step of the W model: forward, CE, backward, clip, Adam
    (basic_vqa/experiment.py:248-257 without the pseudo-QA term).
    P: dict of tensors (trainable ones are everything outside 'img_encoder.model.').
    opt_state: dict name -> (m, v), plus 'step'.  Returns (loss, logits, new P, grads)."""
    names = [k for k in P if not k.startswith('img_encoder.model.')]
    Q = dict(P)
    for k in names:
        Q[k] = P[k].detach().clone().requires_grad_(True)
    logits = w_model_forward(image, question, Q, masks)
    loss = cross_entropy(logits, label)
    grads = torch.autograd.grad(loss, [Q[k] for k in names])
    cg, total = clip_grad_norm(list(grads), clip)
    opt_state['step'] = opt_state.get('step', 0) + 1
    newP = dict(P)
    for k, g in zip(names, cg):
        m, v = opt_state.get(k, (torch.zeros_like(g), torch.zeros_like(g)))
        p, m, v = adam_step(P[k].detach(), g, m, v, opt_state['step'], lr)
        opt_state[k] = (m, v)
        newP[k] = p
    return loss.detach(), logits.detach(), newP, dict(zip(names, grads))


# ----------------------------------------------------------------------------- validation counters
def num_correct(pred, multi_choice):
    """reference basic_vqa/misc.py:42-45: a prediction counts when it equals any of the sample's choices."""
    res = torch.stack([(ans == pred) for ans in multi_choice])
    return res.any(dim=0).sum().item()


def num_correct_qst(qst_pred, qst):
    """reference basic_vqa/misc.py:47-58: questions with 0 / <=3 / <=5 mismatching tokens."""
    tok = qst_pred.argmax(dim=2)[:, :-1]
    err_cnt = (~(qst[:, 1:] == tok)).sum(dim=1)
    return (err_cnt == 0).sum().item(), (err_cnt <= 3).sum().item(), (err_cnt <= 5).sum().item()


def val_counters(output, label, multi_choice, ans_unk_idx):
    """reference basic_vqa/experiment.py:314-324: (CE loss, corr with '<unk>' accepted, corr with '<unk>' -> -9999)."""
    _, pred1 = torch.max(output, 1)
    _, pred2 = torch.max(output, 1)
    loss = torch.nn.functional.cross_entropy(output, label)
    pred2[pred2 == ans_unk_idx] = -9999
    return loss.item(), num_correct(pred1, multi_choice), num_correct(pred2, multi_choice)
