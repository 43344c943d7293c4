"""CPU ORACLE (test infrastructure, NOT product code) for the encoder/decoder networks.

Restates with torch-CPU tensor ops (float32 or float64) the TensorFlow-1.9 arithmetic the
reference's model code lowers to:
  /root/reference/modules.py:39-356   (bn, conv1d, conv1d_banks, gru, prenet, highwaynet, CBHG)
  /root/reference/encoder.py:78-123   (encoder_spec_phn._build_model)
  /root/reference/decoder.py:75-199   (decoder_specs._build_model/_build_loss)
  /root/reference/decoder.py:227-263  (Adam as tf.train.AdamOptimizer applies it)

TensorFlow 1.9.0-rc0 (pinned by meta_info_def in enc_14_ckpt/encoder-136512.meta) is a
third-party dependency that is absent from /root/reference and not installable here, so
its published op semantics are restated:
  * tf.layers.dense on [N,T,C] = matmul over the last axis + bias,
  * tf.layers.conv1d (NHWC Conv2D, stride 1, "SAME", no bias), kernel stored [k,Cin,Cout];
    SAME pads left (k-1)//2 and right k-1-left,
  * FusedBatchNorm eps 1e-3; inference uses moving stats; training normalises with the
    biased batch variance and feeds the Bessel-corrected one to the moving average
    (decay 0.999),
  * max_pooling1d(2, stride 1, "same"): out[t] = max(x[t], x[t+1]), out[T-1] = x[T-1],
  * GRUCell: g = sigmoid([x,h]Wg + bg); r,u = split(g) (r first);
             c = tanh([x, r*h]Wc + bc); h' = u*h + (1-u)*c,
  * bidirectional_dynamic_rnn: zero initial state, backward = reverse-run-reverse, concat,
  * tf.layers.dropout: x / keep * floor(keep + U[0,1)) in training, identity otherwise.

PARITY STATUS: **parity unpinned** against TensorFlow's own kernels (the reference has no tests or
golden activations, TensorFlow cannot run here).  Pinned by the reference's artefacts: the real
trained weights of enc_14_ckpt (per-tensor CRC32C verified by the bundle reader), the variable
names/shapes of SURVEY.md section 8c, and the graph the reference's Saver wrote next to those weights
(enc_14_ckpt/encoder-136512.meta -> tests/golden/enc_14_graph.json): tests/test_graph_pins_cpu.py
checks every attribute used here against it and evaluates the saved forward graph (GRU cell and
highway block node by node; the whole training-mode forward on the trained weights) with a small
interpreter -- this module returns the same numbers to 1e-10.  Only the primitive kernels the
interpreter has to supply (SAME padding arithmetic, FusedBatchNorm's variances) remain a reading of
TensorFlow's documentation.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.

Weights are a dict  TF variable name -> torch tensor  in TF layout.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = float(np.float32(1e-3))          # FusedBatchNorm's epsilon attribute as the saved graph holds it (float32)
BN_DECAY = 0.999


# --------------------------------------------------------------------------- building blocks
def dense(x, w, scope, act=None, relu_on=None):
    """tf.layers.dense(x, units, activation, name=scope) (modules.py:291-293,315-317).
    ``relu_on``: GIVEN routing of a relu (0/1 tensor of the output's shape: y * relu_on instead of max(y, 0)); see
    conv1d_banks."""
    y = x @ w[scope + '/kernel'] + w[scope + '/bias']
    if act == 'relu':
        y = torch.relu(y) if relu_on is None else y * relu_on
    elif act == 'sigmoid':
        y = torch.sigmoid(y)
    return y


def dropout(x, rate, mask=None):
    """tf.layers.dropout in training mode with an explicit keep-mask (0/1); identity when
    mask is None (inference)."""
    if mask is None:
        return x
    keep = float(np.float32(1.0 - rate))     # the saved graph holds keep_prob as a float32 constant (0.6 -> 0.60000002)
    return x / keep * mask


def prenet(x, w, scope, dropout_rate=0.5, masks=None, relu_on=None):
    """modules.py:274-295.  masks = (mask1, mask2) or None.  relu_on = (on1, on2) or None: given relu routing."""
    y = dense(x, w, scope + '/dense1', 'relu', None if relu_on is None else relu_on[0])
    y = dropout(y, dropout_rate, None if masks is None else masks[0])
    y = dense(y, w, scope + '/dense2', 'relu', None if relu_on is None else relu_on[1])
    y = dropout(y, dropout_rate, None if masks is None else masks[1])
    return y


def conv1d(x, kernel):
    """modules.py:104-140 with padding SAME, rate 1, no bias.  x [N,T,Cin], kernel
    [k,Cin,Cout] -> [N,T,Cout]."""
    k = kernel.shape[0]
    pad_l = (k - 1) // 2
    pad_r = k - 1 - pad_l
    xp = F.pad(x.transpose(1, 2), (pad_l, pad_r))               # [N,Cin,T+k-1]
    wt = kernel.permute(2, 1, 0).contiguous()                   # [Cout,Cin,k]
    return F.conv1d(xp, wt).transpose(1, 2)


def bn(x, w, scope, is_training=False, stats_out=None):
    """modules.py:39-102 -> tf.contrib.layers.batch_norm(fused=True, center, scale).
    Normalises over all axes but the last.  When ``stats_out`` is a dict and training, the
    new moving statistics are stored in it (updates_collections=None => updated in place)."""
    gamma, beta = w[scope + '/gamma'], w[scope + '/beta']
    if is_training:
        red = tuple(range(x.dim() - 1))
        n = x.numel() // x.shape[-1]
        mean = x.mean(dim=red)
        var = ((x - mean) ** 2).mean(dim=red)                   # biased
        if stats_out is not None:
            unb = var * (n / max(n - 1, 1))
            d = float(np.float32(1.0 - BN_DECAY))        # AssignMovingAvg: moving -= (moving - batch) * decay (float32 constant)
            stats_out[scope + '/moving_mean'] = w[scope + '/moving_mean'] - (w[scope + '/moving_mean'] - mean.detach()) * d
            stats_out[scope + '/moving_variance'] = w[scope + '/moving_variance'] - (w[scope + '/moving_variance'] - unb.detach()) * d
    else:
        mean, var = w[scope + '/moving_mean'], w[scope + '/moving_variance']
    return (x - mean) * torch.rsqrt(var + BN_EPS) * gamma + beta


def conv1d_banks(x, w, scope, K, is_training=False, stats_out=None, taps=None, relu_on=None):
    """modules.py:144-166: K convs of width 1..K (each 128 filters in every shipped model --
    called without embed_size at modules.py:328), concat, bn, relu.
    ``taps``: optional dict that receives 'banks_pre', the normalised pre-activations.
    ``relu_on``: optional 0/1 tensor of the output's shape -- GIVEN routing: the relu passes exactly these elements
    (pre * relu_on instead of max(pre, 0)).  A float32 kernel and this float64 restatement legitimately disagree about
    the side of the kink for the handful of pre-activations that lie within float32 rounding of zero; with the
    device's own decisions handed in, values change by at most that rounding and the gradients follow the same route
    on both sides, so they can be compared channel by channel without excuses."""
    outs = [conv1d(x, w[scope + '/conv1d/conv1d/kernel'])]
    for k in range(2, K + 1):
        outs.append(conv1d(x, w[scope + '/num_%d/conv1d/conv1d/kernel' % k]))
    y = torch.cat(outs, dim=-1)
    pre = bn(y, w, scope + '/bn', is_training, stats_out)
    if taps is not None:
        taps['banks_pre'] = pre.detach()
    if relu_on is not None:
        return pre * relu_on
    return torch.relu(pre)


def max_pool_2_same(x, winners=None):
    """tf.layers.max_pooling1d(pool_size=2, strides=1, padding='same') (modules.py:331).
    ``winners`` = (own, prev), optional 0/1 tensors of x's shape -- GIVEN routing: own[t] says x[t] is the maximum of
    its own output frame t, prev[t] says x[t] is the maximum of output frame t - 1; out[t] = x[t] own[t] +
    x[t+1] prev[t+1].  Exactly one of the two is set wherever either operand is positive (both clear where the
    output is 0), so the value equals the maximum up to the rounding that made the decision arbitrary."""
    if winners is not None:
        own, prev = winners
        nxt = x * prev
        return x * own + torch.cat([nxt[:, 1:], torch.zeros_like(nxt[:, :1])], dim=1)
    nxt = torch.cat([x[:, 1:], x[:, -1:]], dim=1)
    return torch.maximum(x, nxt)


def highwaynet(x, w, scope, relu_on=None):
    """modules.py:297-319.  relu_on: given routing of dense1's relu (see dense)."""
    H = dense(x, w, scope + '/dense1', 'relu', relu_on)
    Tg = dense(x, w, scope + '/dense2', 'sigmoid')
    return H * Tg + x * (1.0 - Tg)


def gru_direction(x, w, scope, reverse=False):
    """One tf.nn.dynamic_rnn over a GRUCell (modules.py:197-203).  x [N,T,C] -> [N,T,H]."""
    Wg, bg = w[scope + '/gru_cell/gates/kernel'], w[scope + '/gru_cell/gates/bias']
    Wc, bc = w[scope + '/gru_cell/candidate/kernel'], w[scope + '/gru_cell/candidate/bias']
    N, T, C = x.shape
    H = Wc.shape[1]
    h = x.new_zeros((N, H))
    outs = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        xt = x[:, t]
        g = torch.sigmoid(torch.cat([xt, h], dim=1) @ Wg + bg)
        r, u = g[:, :H], g[:, H:]
        c = torch.tanh(torch.cat([xt, r * h], dim=1) @ Wc + bc)
        h = u * h + (1.0 - u) * c
        outs[t] = h
    return torch.stack(outs, dim=1)


def gru_bidirectional(x, w, scope):
    """modules.py:198-201: concat(fw, bw) on the channel axis."""
    fw = gru_direction(x, w, scope + '/bidirectional_rnn/fw', reverse=False)
    bw = gru_direction(x, w, scope + '/bidirectional_rnn/bw', reverse=True)
    return torch.cat([fw, bw], dim=2)


def lstm_direction(x, w, scope, reverse=False):
    """One tf.nn.dynamic_rnn over tf.contrib.rnn.LSTMCell(H) with its TF-1.9 defaults (modules.py:236-243):
    no peepholes, no projection, forget_bias = 1.0, state zero; lstm_matrix = [x, h] W + b,
    i, j, f, o = split(lstm_matrix, 4); c' = sigmoid(f + 1) c + sigmoid(i) tanh(j); h' = sigmoid(o) tanh(c').
    **Parity unpinned**: recalled from TensorFlow's published rnn_cell_impl.LSTMCell (third-party, absent from the
    reference); no artefact of the reference contains an LSTM graph (its saved graph uses GRU cells)."""
    W, b = w[scope + '/lstm_cell/kernel'], w[scope + '/lstm_cell/bias']
    N, T, C = x.shape
    H = W.shape[1] // 4
    h = x.new_zeros((N, H))
    c = x.new_zeros((N, H))
    outs = [None] * T
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        z = torch.cat([x[:, t], h], dim=1) @ W + b
        i, j, f, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
        c = torch.sigmoid(f + 1.0) * c + torch.sigmoid(i) * torch.tanh(j)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs[t] = h
    return torch.stack(outs, dim=1)


def lstm_bidirectional(x, w, scope):
    fw = lstm_direction(x, w, scope + '/bidirectional_rnn/fw', reverse=False)
    bw = lstm_direction(x, w, scope + '/bidirectional_rnn/bw', reverse=True)
    return torch.cat([fw, bw], dim=2)


def routing_from_bits(bits, dtype=torch.float64):
    """Unpack vc_bn_post_routing's bytes (include/vc_hip.h: bit 0 relu passes, bit 1 winner of its own pool frame,
    bit 2 winner of the previous frame's) into the (relu_on, own, prev) tensors conv1d_banks / max_pool_2_same take."""
    b = torch.as_tensor(bits).to(torch.uint8)
    return tuple(((b >> k) & 1).to(dtype) for k in range(3))


def cbhg(x, w, scope, K, n_highway, is_training=False, stats_out=None, taps=None, routing=None):
    """modules.py:323-356.  ``taps``: optional dict that receives intermediate tensors.  ``routing``: optional dict of
    GIVEN relu / max-pool decisions (see conv1d_banks): 'banks' -> (relu_on, own, prev) for the filter bank's relu and
    the max-pool behind it, 'conv1d_1' -> relu_on of the first projection's relu, 'highway' -> list of relu_on, one
    per highway block; a missing key leaves that layer to its own max(x, 0)."""
    routing = routing or {}
    rb = routing.get('banks')
    y = conv1d_banks(x, w, scope + '/conv1d_banks', K, is_training, stats_out, taps=taps,
                     relu_on=None if rb is None else rb[0])
    if taps is not None:
        taps['banks'] = y
    y = max_pool_2_same(y, None if rb is None else rb[1:])
    y = conv1d(y, w[scope + '/conv1d_1/conv1d/kernel'])
    y = bn(y, w, scope + '/conv1d_1', is_training, stats_out)
    y = torch.relu(y) if routing.get('conv1d_1') is None else y * routing['conv1d_1']
    if taps is not None:
        taps['proj1'] = y
    y = conv1d(y, w[scope + '/conv1d_2/conv1d/kernel'])
    y = bn(y, w, scope + '/conv1d_2', is_training, stats_out)
    y = y + x
    if taps is not None:
        taps['proj2_res'] = y
    for i in range(n_highway):
        y = highwaynet(y, w, scope + '/highwaynet_%d' % i, None if routing.get('highway') is None else routing['highway'][i])
    if taps is not None:
        taps['highway'] = y
    if (scope + '/lstm/bidirectional_rnn/fw/lstm_cell/kernel') in w:          # use_lstm (modules.py:347-350)
        return lstm_bidirectional(y, w, scope + '/lstm')
    y = gru_bidirectional(y, w, scope + '/gru')
    return y


# --------------------------------------------------------------------------- models
def encoder_forward(x, w, cfg, scope=None, taps=None, is_training=False, masks=None, stats_out=None, routing=None):
    """encoder.py:78-123.  Inference mode by default; ``is_training`` switches batch norm to batch
    statistics and ``masks`` = (mask1, mask2) are the prenet dropout keep-masks.  ``routing``: optional dict of given
    relu / pool decisions ('prenet' -> (on1, on2) and the keys cbhg takes; see conv1d_banks).
    Returns (y_logits, y_pred, y_pred_class, CBHG_out)."""
    scope = scope or cfg.get('model_name', 'encoder')
    routing = routing or {}
    pre = prenet(x, w, scope + '/prenet', cfg['dropout_rate'], masks, routing.get('prenet'))
    if taps is not None:
        taps['prenet'] = pre
    out = cbhg(pre, w, scope + '/CBHG', cfg['num_conv_banks'], cfg['num_highwaynet_blocks'],
               is_training, stats_out, taps=taps, routing=routing)
    logits = dense(out, w, scope + '/y_logits')
    pred = torch.softmax(logits, dim=-1)
    cls = torch.argmax(logits, dim=-1).to(torch.int32)
    return logits, pred, cls, out


def decoder_forward(ppg, w, cfg, is_training=False, masks=None, stats_out=None, taps=None, target_mel=None,
                    f_mel_pred=None, routing=None):
    """decoder.py:75-182.  ppg = encoder softmax [N,T,61].
    ``masks``: dict 'step1'/'step2' -> (mask1, mask2) dropout keep-masks for training.
    ``target_mel`` / ``f_mel_pred``: with cfg['use_target_mel_step2'] the second stage is fed
    f_mel_pred * y_mel + (1 - f_mel_pred) * target_mel (decoder.py:148-152).
    ``routing``: optional dict 'step1'/'step2' -> dict of given relu / pool decisions: 'prenet' -> (on1, on2) and the keys
    cbhg takes ('banks', 'conv1d_1', 'highway').
    Returns (y_mel, y_stft)."""
    scope = cfg.get('model_name', 'decoder')
    x = ppg
    ys = []
    for i, sd in enumerate(cfg['steps_v']):
        s = '%s/step%d' % (scope, i + 1)
        m = None if masks is None else masks['step%d' % (i + 1)]
        rt = {} if routing is None else routing.get('step%d' % (i + 1), {})
        pre = prenet(x, w, s + '/prenet', cfg['dropout_rate'], m, rt.get('prenet'))
        t = None
        if taps is not None:
            t = {}
            taps['step%d' % (i + 1)] = t
            t['prenet'] = pre
        out = cbhg(pre, w, s + '/CBHG', sd['num_conv_banks'], sd['num_highwaynet_blocks'],
                   is_training, stats_out, taps=t, routing=rt)
        if t is not None:
            t['cbhg'] = out
        y = dense(out, w, s + '/y_logits')
        ys.append(y)
        x = y
        if i == 0 and cfg.get('use_target_mel_step2', False):
            x = f_mel_pred * y + (1.0 - f_mel_pred) * target_mel
    return ys[0], ys[1]


def decoder_loss(y_mel, y_stft, t_mel, t_stft, cfg):
    """decoder.py:185-199."""
    mel_loss = cfg['mel_loss_weight'] * ((y_mel - t_mel) ** 2).mean()
    stft_loss = cfg['stft_loss_weight'] * ((y_stft - t_stft) ** 2).mean()
    if cfg['loss_type'] == 'log':
        loss = torch.log(mel_loss) + torch.log(stft_loss)
    elif cfg['loss_type'] == 'sum':
        loss = mel_loss + stft_loss
    else:
        raise Exception('- ERROR, _build_loss, loss_type not understood.')
    return mel_loss, stft_loss, loss


def encoder_loss(logits, target):
    """encoder.py:134-137: mean softmax cross-entropy with (one-hot) float labels."""
    return -(target * torch.log_softmax(logits, dim=-1)).sum(-1).mean()


def encoder_metrics(logits, target):
    """encoder.py:143-150: accuracy of argmax(pred) vs argmax(target), mse of the posteriors."""
    pred = torch.softmax(logits, dim=-1)
    acc = (pred.argmax(-1) == target.argmax(-1)).double().mean()
    mse = ((pred - target) ** 2).mean()
    return acc, mse


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer (decoder.py:236-246): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    m,v EMAs; p -= lr_t * m / (sqrt(v) + eps)  (eps OUTSIDE the bias correction)."""
    lr_t = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    p = p - lr_t * m / (torch.sqrt(v) + eps)
    return p, m, v


# --------------------------------------------------------------------------- weights
def trainable_names(w):
    return sorted(k for k in w if not k.endswith('/moving_mean') and not k.endswith('/moving_variance'))


def _glorot(rng, shape):
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:                                   # conv kernel [k, Cin, Cout]
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def _cbhg_shapes(scope, cin_prenet, E, K, n_hw, n_out):
    """(name, shape, kind) for prenet + CBHG + y_logits under ``scope`` (names follow the TF
    variable scoping in modules.py / decoder.py:97-180)."""
    H = E // 2
    out = [(scope + '/prenet/dense1/kernel', (cin_prenet, E), 'glorot'),
           (scope + '/prenet/dense1/bias', (E,), 'zeros'),
           (scope + '/prenet/dense2/kernel', (E, H), 'glorot'),
           (scope + '/prenet/dense2/bias', (H,), 'zeros')]
    b = scope + '/CBHG/conv1d_banks'
    out.append((b + '/conv1d/conv1d/kernel', (1, H, 128), 'glorot'))
    for k in range(2, K + 1):
        out.append((b + '/num_%d/conv1d/conv1d/kernel' % k, (k, H, 128), 'glorot'))
    for bnscope, C in ((b + '/bn', 128 * K), (scope + '/CBHG/conv1d_1', H), (scope + '/CBHG/conv1d_2', H)):
        out += [(bnscope + '/beta', (C,), 'zeros'), (bnscope + '/gamma', (C,), 'ones'),
                (bnscope + '/moving_mean', (C,), 'zeros'), (bnscope + '/moving_variance', (C,), 'ones')]
    out.append((scope + '/CBHG/conv1d_1/conv1d/kernel', (3, 128 * K, H), 'glorot'))
    out.append((scope + '/CBHG/conv1d_2/conv1d/kernel', (3, H, H), 'glorot'))
    for i in range(n_hw):
        hs = scope + '/CBHG/highwaynet_%d' % i
        out += [(hs + '/dense1/kernel', (H, H), 'glorot'), (hs + '/dense1/bias', (H,), 'zeros'),
                (hs + '/dense2/kernel', (H, H), 'glorot'), (hs + '/dense2/bias', (H,), 'minus1')]
    for d in ('fw', 'bw'):
        gs = scope + '/CBHG/gru/bidirectional_rnn/%s/gru_cell' % d
        out += [(gs + '/gates/kernel', (2 * H, 2 * H), 'glorot'), (gs + '/gates/bias', (2 * H,), 'ones'),
                (gs + '/candidate/kernel', (2 * H, H), 'glorot'), (gs + '/candidate/bias', (H,), 'zeros')]
    out += [(scope + '/y_logits/kernel', (E, n_out), 'glorot'), (scope + '/y_logits/bias', (n_out,), 'zeros')]
    return out


def model_variable_shapes(cfg, kind):
    """List of (name, shape, init-kind) for an encoder cfg (kind='encoder') or a decoder cfg."""
    if kind == 'encoder':
        E = cfg['embed_size'] or cfg['input_shape'][-1]
        return _cbhg_shapes(cfg['model_name'], cfg['input_shape'][-1], E,
                            cfg['num_conv_banks'], cfg['num_highwaynet_blocks'], cfg['n_output'])
    out = []
    cin = cfg['input_shape'][-1]
    prevE = None
    for i, sd in enumerate(cfg['steps_v']):
        E = sd['embed_size'] or (cin if i == 0 else prevE)
        out += _cbhg_shapes('%s/step%d' % (cfg['model_name'], i + 1), cin, E,
                            sd['num_conv_banks'], sd['num_highwaynet_blocks'], sd['n_output'])
        cin = sd['n_output']
        prevE = E
    return out


def init_weights(cfg, kind, seed=2, perturb_bn=False):
    """SURVEY.md section 8d synthetic weights: Glorot-uniform kernels, zero biases except
    highway dense2 bias -1 and GRU gate bias +1, BN gamma 1 / beta 0 / mean 0 / var 1.
    ``perturb_bn`` randomises the BN tensors so parity tests exercise them.
    Returns dict name -> np.float32 array."""
    rng = np.random.RandomState(seed)
    w = {}
    for name, shape, how in model_variable_shapes(cfg, kind):
        if how == 'glorot':
            w[name] = _glorot(rng, shape)
        elif how == 'zeros':
            w[name] = np.zeros(shape, np.float32)
        elif how == 'ones':
            w[name] = np.ones(shape, np.float32)
        elif how == 'minus1':
            w[name] = -np.ones(shape, np.float32)
    if perturb_bn:
        for name in list(w):
            if name.endswith('/gamma'):
                w[name] = rng.uniform(0.5, 1.5, w[name].shape).astype(np.float32)
            elif name.endswith('/beta') or name.endswith('/moving_mean'):
                w[name] = rng.uniform(-0.2, 0.2, w[name].shape).astype(np.float32)
            elif name.endswith('/moving_variance'):
                w[name] = rng.uniform(0.5, 2.0, w[name].shape).astype(np.float32)
            elif name.endswith('/bias'):
                w[name] = (w[name] + rng.uniform(-0.1, 0.1, w[name].shape)).astype(np.float32)
    return w


def to_torch(w, dtype=torch.float32, requires_grad=False):
    out = {}
    for k, v in w.items():
        t = torch.from_numpy(np.asarray(v)).to(dtype).clone()
        if requires_grad and not (k.endswith('/moving_mean') or k.endswith('/moving_variance')):
            t.requires_grad_(True)
        out[k] = t
    return out
