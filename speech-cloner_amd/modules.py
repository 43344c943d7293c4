"""Network building blocks with the reference's ``modules`` surface, run on the MI355X.

Mirrors /root/reference/modules.py:39-356 -- same function names and leading argument order
(``bn``, ``conv1d``, ``conv1d_banks``, ``gru``, ``prenet``, ``highwaynet``, ``CBHG``) -- but
eager: every function takes/returns a ``torch.cuda`` tensor of shape [N, T, C] and launches
hand-written HIP kernels through the C ABI (include/vc_hip.h: ``vc_conv_gemm``,
``vc_gru_bidir``).  torch only owns device memory and does one-time weight re-layout.

TensorFlow's ``variable_scope`` / ``get_variable`` pair is replaced by a tiny equivalent:
a ``VariableStore`` (name -> float32 master tensor in TF layout, names identical to the
reference's checkpoints) and a scope stack (``variable_scope``).  Kernel-format copies
(transposed kernels, folded BatchNorm, packed GRU weights, optional bf16) are derived lazily
and cached; ``VariableStore.invalidate()`` drops them after a weight update.

``embed`` and ``attention_decoder`` (modules.py:10-36, 246-272) are dead code in the reference
(never called) and are not provided; ``lstm`` (modules.py:207-243) is reachable only with
``use_lstm`` true, which no shipped configuration sets: any-size recurrence kernels, not tuned ones
(inference here; the training step's LSTM lives in training.py / vc_lstm_train_forward, vc_lstm_backward).
"""
import contextlib
import ctypes as C
import math
import threading

import numpy as np

import _vc

# Host-side fusion choices (one launch vs several for the same arithmetic).  Not read from the environment: set them
# here (modules.OPTIONS['prenet_chain'] = False) or per model through the optional config key 'kernel_options'.
# Kernel selectors inside the library are _vc.set_option / vc_set_option (include/vc_hip.h).
OPTIONS = {'prenet_chain': True,      # both prenet layers in one launch (vc_prenet_chain)
           'highway_chain': True,     # all highway layers + the GRU input projection in one launch (vc_highway_chain)
           'cbhg_front': True}        # the encoder's whole pre-recurrence chain in one launch (vc_cbhg_front)


def apply_options(d):
    """Optional config key 'kernel_options' of encoder_spec_phn / decoder_specs: {name: value} with names from
    OPTIONS (bool) or vc_set_option (int).  Process-global, like the library's own table."""
    for k, v in (d or {}).items():
        if k in OPTIONS:
            OPTIONS[k] = bool(v)
        else:
            _vc.set_option(k, int(v))

BN_EPS = 1e-3           # tf.contrib.layers.batch_norm epsilon (read from the reference's .meta)
BN_DECAY = 0.999        # moving-average decay
BANK_FILTERS = 128      # conv1d_banks is called without embed_size (modules.py:328) => 256 // 2


def _torch():
    import torch
    return torch


# ------------------------------------------------------------------------------- variables
class VariableStore:
    """Named float32 master variables (TF layout) + cached kernel-format derivatives."""

    def __init__(self, compute_dtype='float32', device='cuda', seed=0):
        torch = _torch()
        if compute_dtype in ('float32', 'f32', torch.float32):
            self.dtype, self.vc_dtype = torch.float32, _vc.VC_F32
        elif compute_dtype in ('bfloat16', 'bf16', torch.bfloat16):
            self.dtype, self.vc_dtype = torch.bfloat16, _vc.VC_BF16
        else:
            raise ValueError(' - ERROR, compute_dtype {} not understood'.format(compute_dtype))
        self.device = torch.device(device)
        self.vars = {}
        self.order = []                      # creation order (== TF graph order)
        self.non_trainable = set()
        self._cache = {}
        self._rng = np.random.RandomState(seed)

    # -- creation (tf.get_variable with the layer's default initializer)
    def get(self, name, shape, init):
        v = self.vars.get(name)
        if v is not None:
            if tuple(v.shape) != tuple(shape):
                raise ValueError(' - ERROR, variable {} has shape {}, requested {}'.format(name, tuple(v.shape), shape))
            return v
        torch = _torch()
        if init == 'glorot':
            if len(shape) == 2:
                fan_in, fan_out = shape
            else:
                rf = int(np.prod(shape[:-2]))
                fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            a = self._rng.uniform(-lim, lim, size=shape).astype(np.float32)
        elif isinstance(init, (int, float)):
            a = np.full(shape, float(init), dtype=np.float32)
        else:
            raise ValueError(init)
        v = torch.from_numpy(a).to(self.device)
        self.vars[name] = v
        self.order.append(name)
        if name.endswith('/moving_mean') or name.endswith('/moving_variance'):
            self.non_trainable.add(name)
        return v

    def assign(self, name, value):
        torch = _torch()
        t = value if torch.is_tensor(value) else torch.from_numpy(np.ascontiguousarray(value, dtype=np.float32))
        if name not in self.vars:
            raise KeyError(name)
        if tuple(t.shape) != tuple(self.vars[name].shape):
            raise ValueError(' - ERROR, assign {}: shape {} != {}'.format(name, tuple(t.shape), tuple(self.vars[name].shape)))
        self.vars[name].copy_(t.to(self.device, dtype=torch.float32))
        self.invalidate()

    def load_dict(self, d, strict=True):
        """Copy every variable this store knows from ``d`` (name -> array)."""
        missing = [n for n in self.vars if n not in d]
        if strict and missing:
            raise KeyError('variables missing from checkpoint: {}'.format(missing[:4]))
        torch = _torch()
        for n, v in self.vars.items():
            if n in d:
                a = np.asarray(d[n], dtype=np.float32)
                if v.dim() == 0 and a.size == 1:             # scalars (np.ascontiguousarray would make them 1-d)
                    a = a.reshape(())
                a = np.ascontiguousarray(a) if a.ndim else a
                if tuple(a.shape) != tuple(v.shape):
                    raise ValueError(' - ERROR, checkpoint tensor {} has shape {}, model wants {}'.format(n, a.shape, tuple(v.shape)))
                v.copy_(torch.from_numpy(np.array(a)) if a.ndim == 0 else torch.from_numpy(a))
        self.invalidate()

    def to_numpy(self):
        return {n: v.detach().cpu().numpy() for n, v in self.vars.items()}

    def trainable_names(self, prefix=''):
        return [n for n in self.order if n.startswith(prefix) and n not in self.non_trainable]

    def invalidate(self, prefix=None):
        """Drop the kernel-layout copies derived from the weights -- all of them, or (``prefix``) only those of the
        variables under one scope: a decoder that trains on top of a FROZEN encoder in the same store must not make the
        encoder rebuild its copies (dozens of small launches at the head of every training step)."""
        if prefix is None:
            self._cache.clear()
        else:
            pre = prefix.rstrip('/')
            for k in [k for k in self._cache if isinstance(k, tuple) and any(
                    isinstance(e, str) and (e == pre or e.startswith(pre + '/')) for e in k[1:])]:
                del self._cache[k]
        self.version = getattr(self, 'version', 0) + 1      # anything derived from the weights outside this cache checks it

    def cached(self, key, fn):
        """Kernel-layout copy of weights, built on first use.  The build runs on the CURRENT stream, later
        users may run on any stream: a copy built on a side stream (first decoder.predict chunk after a
        restore, with n_streams > 1) is waited for here, once, so that a chunk issued right afterwards on
        another stream cannot read it half-built.  Builds on the default stream need no wait: every side
        stream is ordered behind the default stream before it gets work."""
        v = self._cache.get(key)
        if v is None:
            v = fn()
            self._cache[key] = v
            torch = _torch()
            if torch.cuda.is_available():
                cur = torch.cuda.current_stream()
                if cur != torch.cuda.default_stream():
                    cur.synchronize()
        return v


_CTX = threading.local()


def _ctx():
    if not hasattr(_CTX, 'stack'):
        _CTX.stack = []
        _CTX.store = None
    return _CTX


@contextlib.contextmanager
def variable_store(store):
    """Make ``store`` the current variable store (the stand-in for the TF default graph)."""
    c = _ctx()
    prev = c.store
    c.store = store
    try:
        yield store
    finally:
        c.store = prev


@contextlib.contextmanager
def variable_scope(name, reuse=None):
    """tf.variable_scope: pushes ``name`` on the scope stack."""
    c = _ctx()
    c.stack.append(name)
    try:
        yield '/'.join(c.stack)
    finally:
        c.stack.pop()


def _scope(*names):
    return '/'.join(list(_ctx().stack) + [n for n in names if n])


def _store():
    s = _ctx().store
    if s is None:
        raise RuntimeError(' - ERROR, no current VariableStore (use `with variable_store(store):`)')
    return s


# ------------------------------------------------------------------------------- launch helpers
def _pad8(n):
    return (n + 7) // 8 * 8


def _as3(x):
    if x.dim() != 3:
        raise ValueError(' - ERROR, expected a [N, T, C] tensor, got shape {}'.format(tuple(x.shape)))
    return x


def gemm_desc(X, M, T, Cin, ldx, N, groups, Cout, ldc, vc_dtype, mode=_vc.GEMM_PLAIN,
              pro_scale=None, pro_shift=None, pro_relu=0, pro_pool=0,
              epi_scale=None, epi_shift=None, act=_vc.ACT_NONE, R=None, ldr=0, out_f32=False, epi_pool=0, sum_groups=False):
    """Fill a vc_gemm_desc.  groups: list of (Bt tensor [N, K], K, taps, pad_l, c_off)."""
    d = _vc.GemmDesc()
    d.dtype, d.mode = vc_dtype, mode
    d.d_X = X.data_ptr()
    d.M, d.T, d.Cin, d.ldx, d.N, d.n_groups = M, T, Cin, ldx, N, len(groups)
    for i, (Bt, K, taps, pad_l, c_off) in enumerate(groups):
        g = d.groups[i]
        g.d_Bt, g.K, g.taps, g.pad_l, g.c_off = Bt.data_ptr(), K, taps, pad_l, c_off
    d.d_pro_scale = pro_scale.data_ptr() if pro_scale is not None else None
    d.d_pro_shift = pro_shift.data_ptr() if pro_shift is not None else None
    d.pro_relu, d.pro_pool = int(pro_relu), int(pro_pool)
    d.d_epi_scale = epi_scale.data_ptr() if epi_scale is not None else None
    d.d_epi_shift = epi_shift.data_ptr() if epi_shift is not None else None
    d.act = act
    d.d_R = R.data_ptr() if R is not None else None
    d.ldr = ldr
    d.d_C, d.ldc, d.out_f32 = Cout.data_ptr(), ldc, int(bool(out_f32))
    d.epi_pool = int(epi_pool)
    d.sum_groups = int(sum_groups)
    return d


def gemm_launch(X, M, T, Cin, ldx, N, groups, Cout, ldc, vc_dtype, **kw):
    """Fill a vc_gemm_desc (see gemm_desc) and launch vc_conv_gemm on the current stream."""
    d = gemm_desc(X, M, T, Cin, ldx, N, groups, Cout, ldc, vc_dtype, **kw)
    if vc_dtype == _vc.VC_BF16 and len(groups) == 1 and N == 256:
        # the one launch shape that can split K over workgroups (include/vc_hip.h, vc_gemm_desc.d_workspace): scratch from
        # the caching allocator of the current stream, private to this call
        nbytes = _vc.lib().vc_conv_gemm_workspace_bytes(C.byref(d))
        if nbytes:
            ws = _torch().empty(nbytes, dtype=_torch().uint8, device=Cout.device)
            d.d_workspace, d.workspace_bytes = ws.data_ptr(), nbytes
    _vc.check(_vc.lib().vc_conv_gemm(C.byref(d), _vc.current_stream()))
    return Cout


def convert(x, dtype):
    """dtype conversion through vc_convert (f32 <-> bf16)."""
    torch = _torch()
    if x.dtype == dtype:
        return x
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    code = {torch.float32: _vc.VC_F32, torch.bfloat16: _vc.VC_BF16}
    _vc.check(_vc.lib().vc_convert(x.data_ptr(), code[x.dtype], y.data_ptr(), code[dtype], x.numel(),
                                   _vc.current_stream()))
    return y


def axpby(x, a, y, b, out=None):
    """out = a * x + b * y on float32 [N, T, C] tensors (vc_axpby; out may be x or y)."""
    torch = _torch()
    if x.dtype != torch.float32 or y.dtype != torch.float32 or x.shape != y.shape:
        raise ValueError(' - ERROR, axpby wants two float32 tensors of one shape')
    x, y = x.contiguous(), y.contiguous()
    if out is None:
        out = torch.empty_like(x)
    Cn = x.shape[-1]
    _vc.check(_vc.lib().vc_axpby(x.data_ptr(), Cn, float(a), y.data_ptr(), Cn, float(b), out.data_ptr(), Cn,
                                 x.numel() // Cn, Cn, _vc.current_stream()))
    return out


# ------------------------------------------------------------------------------- weight prep
def _prep_dense(store, scope, cin, units, bias_init=0.0):
    k = store.get(scope + '/kernel', (cin, units), 'glorot')
    b = store.get(scope + '/bias', (units,), bias_init)

    def build():
        torch = _torch()
        cp = _pad8(cin)
        bt = torch.zeros((units, cp), dtype=torch.float32, device=store.device)
        bt[:, :cin] = k.t()
        return bt.to(store.dtype).contiguous(), b.clone()
    return store.cached(('dense', scope), build)


def _prep_conv(store, scope, size, cin, filters):
    k = store.get(scope + '/conv1d/kernel', (size, cin, filters), 'glorot')

    def build():
        return k.reshape(size * cin, filters).t().contiguous().to(store.dtype)
    return store.cached(('conv', scope), build)


def _bn_vars(store, scope, C_):
    return (store.get(scope + '/beta', (C_,), 0.0), store.get(scope + '/gamma', (C_,), 1.0),
            store.get(scope + '/moving_mean', (C_,), 0.0), store.get(scope + '/moving_variance', (C_,), 1.0))


def _prep_bn(store, scope, C_):
    beta, gamma, mean, var = _bn_vars(store, scope, C_)

    def build():
        torch = _torch()
        s = gamma * torch.rsqrt(var + BN_EPS)
        return s.contiguous(), (beta - mean * s).contiguous()
    return store.cached(('bn', scope), build)


def _prep_highway(store, scope, H):
    k1 = store.get(scope + '/dense1/kernel', (H, H), 'glorot')
    b1 = store.get(scope + '/dense1/bias', (H,), 0.0)
    k2 = store.get(scope + '/dense2/kernel', (H, H), 'glorot')
    b2 = store.get(scope + '/dense2/bias', (H,), -1.0)

    def build():
        torch = _torch()
        nb = (H + 31) // 32
        if H % 32 == 0:       # (the training step rebuilds this after every update: three launches, not 4 per 32 units)
            bt = torch.stack([k1.t().reshape(nb, 32, H), k2.t().reshape(nb, 32, H)], dim=1).reshape(64 * nb, H)
            bias = torch.stack([b1.view(nb, 32), b2.view(nb, 32)], dim=1).reshape(64 * nb)
            return bt.to(store.dtype).contiguous(), bias
        bt = torch.zeros((64 * nb, H), dtype=torch.float32, device=store.device)
        bias = torch.zeros((64 * nb,), dtype=torch.float32, device=store.device)
        for q in range(nb):
            lo, hi = 32 * q, min(H, 32 * q + 32)
            bt[64 * q:64 * q + (hi - lo)] = k1[:, lo:hi].t()
            bt[64 * q + 32:64 * q + 32 + (hi - lo)] = k2[:, lo:hi].t()
            bias[64 * q:64 * q + (hi - lo)] = b1[lo:hi]
            bias[64 * q + 32:64 * q + 32 + (hi - lo)] = b2[lo:hi]
        return bt.to(store.dtype).contiguous(), bias
    return store.cached(('highway', scope), build)


def _prep_gru(store, scope, cin, H):
    w = {}
    for d in ('fw', 'bw'):
        s = '{}/bidirectional_rnn/{}/gru_cell'.format(scope, d)
        w[d] = (store.get(s + '/gates/kernel', (cin + H, 2 * H), 'glorot'),
                store.get(s + '/gates/bias', (2 * H,), 1.0),
                store.get(s + '/candidate/kernel', (cin + H, H), 'glorot'),
                store.get(s + '/candidate/bias', (H,), 0.0))

    def build():
        torch = _torch()
        btx = torch.cat([torch.cat([w[d][0][:cin].t(), w[d][2][:cin].t()], 0) for d in ('fw', 'bw')], 0)
        bx = torch.cat([torch.cat([w[d][1], w[d][3]], 0) for d in ('fw', 'bw')], 0)
        wh = [torch.cat([w[d][0][cin:], w[d][2][cin:]], 1).contiguous().to(store.dtype) for d in ('fw', 'bw')]
        return btx.contiguous().to(store.dtype), bx.contiguous(), wh[0], wh[1]
    return store.cached(('gru', scope), build)


def _prep_gru_uni(store, scope, cin, H):
    """Kernel-format weights of a unidirectional GRU (tf.nn.dynamic_rnn default scope 'rnn')."""
    s = '{}/rnn/gru_cell'.format(scope)
    wg = store.get(s + '/gates/kernel', (cin + H, 2 * H), 'glorot')
    bg = store.get(s + '/gates/bias', (2 * H,), 1.0)
    wc = store.get(s + '/candidate/kernel', (cin + H, H), 'glorot')
    bc = store.get(s + '/candidate/bias', (H,), 0.0)

    def build():
        torch = _torch()
        one = torch.cat([wg[:cin].t(), wc[:cin].t()], 0)
        b1 = torch.cat([bg, bc], 0)
        wh = torch.cat([wg[cin:], wc[cin:]], 1).contiguous().to(store.dtype)
        return torch.cat([one, one], 0).contiguous().to(store.dtype), torch.cat([b1, b1], 0).contiguous(), wh
    return store.cached(('gru_uni', scope), build)


# ------------------------------------------------------------------------------- blocks
def dense(inputs, units, activation_fn=None, name='dense', bias_init=0.0, out_f32=False, in_features=None):
    """tf.layers.dense(inputs, units, activation, name=name) on [N, T, C] (modules.py:291-293).
    ``in_features``: true input width when ``inputs`` carries zero padding columns up to a
    multiple of 8 (the 61-class posteriors are stored 64 wide)."""
    torch = _torch()
    store = _store()
    x = _as3(inputs)
    N_, T_, Cx = x.shape
    scope = _scope(name)
    cin = Cx if in_features is None else in_features
    kname = scope + '/kernel'
    if kname in store.vars:
        cin = store.vars[kname].shape[0]
    bt, b = _prep_dense(store, scope, cin, units, bias_init)
    if Cx != _pad8(cin):
        raise ValueError(' - ERROR, dense {}: input width {} does not match kernel rows {}'.format(scope, Cx, cin))
    out = torch.empty((N_, T_, units), dtype=torch.float32 if out_f32 else store.dtype, device=x.device)
    act = {None: _vc.ACT_NONE, 'relu': _vc.ACT_RELU, 'sigmoid': _vc.ACT_SIGMOID, 'tanh': _vc.ACT_TANH}[activation_fn]
    gemm_launch(x, N_ * T_, T_, Cx, Cx, units, [(bt, Cx, 1, 0, 0)], out, units, store.vc_dtype,
                epi_shift=b, act=act, out_f32=out_f32)
    return out


def bn(inputs, is_training=True, activation_fn=None, scope="bn", reuse=None):
    """modules.py:39-102 in inference mode as a stand-alone op: the folded scale/shift is run
    through the GEMM kernel's epilogue with an identity kernel only in tests; the hot path
    never calls this -- conv1d()/conv1d_banks() fuse the normalisation (see CBHG)."""
    if is_training:
        raise NotImplementedError(' - ERROR, bn: training-mode batch statistics are fused into the training step')
    torch = _torch()
    store = _store()
    x = _as3(inputs)
    N_, T_, C_ = x.shape
    s, sh = _prep_bn(store, _scope(scope), C_)
    eye = store.cached(('eye', C_), lambda: torch.eye(C_, dtype=store.dtype, device=store.device))
    out = torch.empty_like(x)
    act = {None: _vc.ACT_NONE, 'relu': _vc.ACT_RELU}[activation_fn]
    gemm_launch(x, N_ * T_, T_, C_, C_, C_, [(eye, C_, 1, 0, 0)], out, C_, store.vc_dtype,
                epi_scale=s, epi_shift=sh, act=act)
    return out


def _f32_split_ok(store):
    """float32 inference: the filter banks and the post-bank projection as three float16 MFMA products of exactly split
    operands (gemm16.py / csrc/vc_gemm16.hip: the error against float64 equals the f32-MFMA kernels', at 2-3x their
    rate).  vc_set_option('f32_f16x3', 0) keeps them on the f32-input MFMA kernels (A/B, tests)."""
    return store.dtype == _torch().float32 and _vc.get_option('f32_f16x3') != 0


def conv1d(inputs, filters=None, size=1, rate=1, padding="SAME", use_bias=False, activation_fn=None,
           scope="conv1d", reuse=None, bn_scope=None, residual=None, pool_input=False):
    """modules.py:104-140 (tf.layers.conv1d, stride 1, no bias).  Extra keyword arguments fuse
    what the reference applies right after/before the convolution inside CBHG:
      bn_scope    inference batch norm of that scope folded into the epilogue (modules.py:335,338)
      activation_fn 'relu' after the norm
      residual    tensor added after the norm (modules.py:340)
      pool_input  max_pooling1d(2, 1, 'same') applied to ``inputs`` on the fly (modules.py:331);
                  2 = same, with the promise that ``inputs`` >= 0 (post-ReLU) so the max runs on
                  the integer bit patterns"""
    if rate != 1 or padding.upper() != "SAME" or use_bias:
        raise NotImplementedError(' - ERROR, conv1d: only rate=1, padding=SAME, use_bias=False are used by the reference')
    torch = _torch()
    store = _store()
    x = _as3(inputs)
    N_, T_, Cin = x.shape
    if filters is None:
        filters = Cin
    with variable_scope(scope):
        bt = _prep_conv(store, _scope(), size, Cin, filters)
    s = sh = None
    if bn_scope is not None:
        s, sh = _prep_bn(store, _scope(bn_scope), filters)
    out = torch.empty((N_, T_, filters), dtype=store.dtype, device=x.device)
    act = {None: _vc.ACT_NONE, 'relu': _vc.ACT_RELU}[activation_fn]
    if _f32_split_ok(store) and residual is None and filters in (128, 256) and Cin % 64 == 0 and 1024 <= Cin <= 4096 and size <= 33:
        # long-K projection (conv1d_1 behind the banks, modules.py:331-335): float32 result from float16 products
        import gemm16
        kscope = _scope(scope)

        def build():
            w16 = gemm16.Weights16(x.device)
            pairs, cs = gemm16.conv_forward_operands(w16, store.vars[kscope + '/conv1d/kernel'])
            w16.refresh()
            return w16, pairs, (cs * s if s is not None else cs)
        w16, pairs, scale = store.cached(('g16conv', kscope, bn_scope), build)
        M = N_ * T_
        x16, rs = gemm16.split16(x.contiguous().view(M, Cin), M, Cin, Cin, T_, pool=1 if pool_input else 0)
        gemm16.gemm16(x16, rs, M, T_, Cin, pairs, out, filters, col_scale=scale, col_shift=sh, act=act)
        return out
    gemm_launch(x, N_ * T_, T_, Cin, Cin, filters, [(bt, size * Cin, size, (size - 1) // 2, 0)], out, filters,
                store.vc_dtype, pro_pool=int(pool_input), epi_scale=s, epi_shift=sh, act=act,
                R=residual, ldr=filters if residual is not None else 0)
    return out


def conv1d_banks(inputs, K=16, embed_size=256, is_training=True, scope="conv1d_banks", reuse=None, pool_output=None):
    """modules.py:144-166: K convolutions of width 1..K (embed_size//2 filters each), concat,
    batch norm, relu -- ONE grouped launch, heaviest bank first, norm+relu in the epilogue.
    pool_output='auto' lets the launch also apply the max_pooling1d(2, 1, 'same') that follows the
    banks in CBHG (modules.py:331) where the kernel supports it; the return value is then
    (output, pooled: bool)."""
    if is_training:
        raise NotImplementedError(' - ERROR, conv1d_banks: training mode runs through the fused training step')
    torch = _torch()
    store = _store()
    x = _as3(inputs)
    N_, T_, Cin = x.shape
    F_ = embed_size // 2
    if K > _vc.GEMM_MAX_GROUPS:
        raise ValueError(' - ERROR, conv1d_banks: K={} > {}'.format(K, _vc.GEMM_MAX_GROUPS))
    groups = []
    with variable_scope(scope):
        for k in range(1, K + 1):
            sub = 'conv1d' if k == 1 else 'num_{}/conv1d'.format(k)
            with variable_scope(sub):
                bt = _prep_conv(store, _scope(), k, Cin, F_)
            groups.append((bt, k * Cin, k, (k - 1) // 2, F_ * (k - 1)))
        s, sh = _prep_bn(store, _scope('bn'), F_ * K)
    out = torch.empty((N_, T_, F_ * K), dtype=store.dtype, device=x.device)
    if _f32_split_ok(store) and K % 2 == 0 and K <= 32 and Cin % 64 == 0 and F_ == BANK_FILTERS == 128:
        # float32 result from float16 products; pairs of filter widths as in the bf16 bank kernel.  The pool stays with
        # the next convolution's operand (pooled = False)
        import gemm16
        bscope = _scope(scope)

        def build():
            w16 = gemm16.Weights16(x.device)
            kern = [store.vars[bscope + ('/conv1d' if k == 1 else '/num_{}/conv1d'.format(k)) + '/conv1d/kernel'] for k in range(1, K + 1)]
            pairs, cs = gemm16.bank_forward_operands(w16, kern, Cin)
            w16.refresh()
            return w16, pairs, cs * s
        w16, pairs, scale = store.cached(('g16bank', bscope), build)
        M = N_ * T_
        x16, rs = gemm16.split16(x.contiguous().view(M, Cin), M, Cin, Cin, T_)
        gemm16.gemm16(x16, rs, M, T_, Cin, pairs, out, F_ * K, col_scale=scale, col_shift=sh, act=_vc.ACT_RELU)
        return (out, False) if pool_output == 'auto' else out
    d = gemm_desc(x, N_ * T_, T_, Cin, Cin, F_, groups, out, F_ * K, store.vc_dtype,
                  epi_scale=s, epi_shift=sh, act=_vc.ACT_RELU, epi_pool=1 if pool_output == 'auto' else 0)
    pooled = False
    if pool_output == 'auto':
        pooled = bool(_vc.lib().vc_conv_gemm_epi_pool_supported(C.byref(d)))
        d.epi_pool = int(pooled)
    _vc.check(_vc.lib().vc_conv_gemm(C.byref(d), _vc.current_stream()))
    return (out, pooled) if pool_output == 'auto' else out


def _refuse_cudnn(use_Cudnn, what):
    """modules.py:188-197 / 227-236: with use_Cudnn the reference builds tf.contrib.cudnn_rnn.CudnnGRU / CudnnLSTM,
    whose weights are ONE opaque parameter blob in cuDNN's own layout (and whose GRU applies the reset gate after the
    recurrent product, unlike GRUCell).  Such a checkpoint cannot be restored into the GRUCell / LSTMCell variables
    this package creates, and no shipped configuration sets it: refuse instead of silently building another model."""
    if use_Cudnn:
        raise NotImplementedError(' - ERROR, {}: use_Cudnn=True (cuDNN parameter blob / cell variant) is not built; '
                                  'only the GRUCell / LSTMCell variable layout of use_Cudnn=False is supported'.format(what))


def gru(inputs, num_units=None, bidirection=False, scope="gru", use_Cudnn=False, reuse=None):
    """modules.py:168-204: the x-halves of both directions' cell matmuls are one GEMM, the recurrence one
    persistent launch (bidirection=True is the only form CBHG uses; the unidirectional form is provided too)."""
    _refuse_cudnn(use_Cudnn, 'gru')
    torch = _torch()
    store = _store()
    x = _as3(inputs)
    N_, T_, Cin = x.shape
    H = Cin if num_units is None else num_units
    if not bidirection:
        # modules.py:202-204: tf.nn.dynamic_rnn over one GRUCell (variables <scope>/rnn/gru_cell/...).  No shipped
        # configuration builds it (CBHG asks for the bidirectional form), so it rides on the bidirectional launch with
        # the forward cell in both slots; the second half of the result is dropped.
        btx, bx, wh = _prep_gru_uni(store, _scope(scope), Cin, H)
        xproj = torch.empty((N_ * T_, 6 * H), dtype=torch.float32, device=x.device)
        gemm_launch(x, N_ * T_, T_, Cin, Cin, 6 * H, [(btx, Cin, 1, 0, 0)], xproj, 6 * H, store.vc_dtype,
                    epi_shift=bx, out_f32=True)
        return _gru_recurrence(xproj, N_, T_, H, wh, wh)[:, :, :H].contiguous()
    btx, bx, wh_fw, wh_bw = _prep_gru(store, _scope(scope), Cin, H)
    xproj = torch.empty((N_ * T_, 6 * H), dtype=torch.float32, device=x.device)
    gemm_launch(x, N_ * T_, T_, Cin, Cin, 6 * H, [(btx, Cin, 1, 0, 0)], xproj, 6 * H, store.vc_dtype,
                epi_shift=bx, out_f32=True)
    return _gru_recurrence(xproj, N_, T_, H, wh_fw, wh_bw)


def _gru_recurrence(xproj, N_, T_, H, wh_fw, wh_bw):
    """The recurrent half of gru(): xproj [N*T, 6H] float32 = input projections of both directions."""
    torch = _torch()
    store = _store()
    x = xproj
    out = torch.empty((N_, T_, 2 * H), dtype=store.dtype, device=x.device)
    if store.dtype == torch.float32 and 128 < H <= 1024 and _vc.get_option('gru_f32_wide') != 0:
        # float32 weights of more than 128 units do not fit a CU's registers: the inference kernels stream all 786 KB of
        # them every step (15.8 ms for 64 windows at 256 units).  The TRAINING forward kernel (vc_gru_train_forward) keeps
        # half of them resident and is the same recurrence (3.0 ms, results equal to 5e-7); what it saves for the
        # backward pass goes to scratch.
        gates = torch.empty((2, N_ * T_, 3 * H), dtype=torch.float32, device=x.device)
        rh = torch.empty((2, N_ * T_, H), dtype=torch.float32, device=x.device)
        _vc.check(_vc.lib().vc_gru_train_forward(xproj.data_ptr(), wh_fw.data_ptr(), wh_bw.data_ptr(), N_, T_, H,
                                                 out.data_ptr(), gates.data_ptr(), rh.data_ptr(), _vc.current_stream()))
        return out
    nws = _vc.lib().vc_gru_workspace_bytes(H, store.vc_dtype)
    ws = torch.empty((max(nws, 16),), dtype=torch.uint8, device=x.device)
    _vc.check(_vc.lib().vc_gru_bidir(xproj.data_ptr(), wh_fw.data_ptr(), wh_bw.data_ptr(), store.vc_dtype,
                                     N_, T_, H, out.data_ptr(), store.vc_dtype, ws.data_ptr(), nws,
                                     _vc.current_stream()))
    return out


def _prep_lstm(store, scope, cin, H, bidirection):
    """tf.contrib.rnn.LSTMCell variables: <scope>/bidirectional_rnn/{fw,bw}/lstm_cell/{kernel [cin+H, 4H], bias [4H]}
    (or <scope>/rnn/lstm_cell/... for the unidirectional form)."""
    dirs = ('bidirectional_rnn/fw', 'bidirectional_rnn/bw') if bidirection else ('rnn', 'rnn')
    w = []
    for d in dirs:
        s = '{}/{}/lstm_cell'.format(scope, d)
        w.append((store.get(s + '/kernel', (cin + H, 4 * H), 'glorot'), store.get(s + '/bias', (4 * H,), 0.0)))

    def build():
        torch = _torch()
        btx = torch.cat([k[:cin].t() for k, _ in w], 0).contiguous().to(store.dtype)        # [8H, cin]
        bx = torch.cat([b for _, b in w], 0).contiguous()
        wh = [k[cin:].contiguous().to(store.dtype) for k, _ in w]
        return btx, bx, wh[0], wh[1]
    return store.cached(('lstm', scope, bidirection), build)


def lstm(inputs, num_units=None, bidirection=False, scope="lstm", use_Cudnn=False, reuse=None):
    """modules.py:207-243: LSTMCell (no peepholes, forget_bias 1.0) under (bidirectional_)dynamic_rnn; the input
    halves of both directions are one GEMM, the recurrence one launch (vc_lstm_bidir).  Inference only."""
    _refuse_cudnn(use_Cudnn, 'lstm')
    torch = _torch()
    store = _store()
    x = _as3(inputs)
    N_, T_, Cin = x.shape
    H = Cin if num_units is None else num_units
    btx, bx, wh_fw, wh_bw = _prep_lstm(store, _scope(scope), Cin, H, bidirection)
    xproj = torch.empty((N_ * T_, 8 * H), dtype=torch.float32, device=x.device)
    gemm_launch(x, N_ * T_, T_, Cin, Cin, 8 * H, [(btx, Cin, 1, 0, 0)], xproj, 8 * H, store.vc_dtype,
                epi_shift=bx, out_f32=True)
    out = torch.empty((N_, T_, 2 * H), dtype=store.dtype, device=x.device)
    _vc.check(_vc.lib().vc_lstm_bidir(xproj.data_ptr(), wh_fw.data_ptr(), wh_bw.data_ptr(), store.vc_dtype, N_, T_, H,
                                      out.data_ptr(), store.vc_dtype, _vc.current_stream()))
    return out if bidirection else out[:, :, :H].contiguous()


def _dropout(x, rate, is_training):
    if is_training:
        raise NotImplementedError(' - ERROR, dropout in training mode runs through the fused training step')
    return x


def prenet(inputs, num_units=None, embed_size=256, dropout_rate=0.5, is_training=True, scope="prenet", reuse=None,
           in_features=None):
    """modules.py:274-295: dense(E)+relu, dropout, dense(E/2)+relu, dropout."""
    if num_units is None:
        num_units = [embed_size, embed_size // 2]
    torch = _torch()
    store = _store()
    x = _as3(inputs)
    N_, T_, Cx = x.shape
    if (not is_training and store.dtype == torch.bfloat16 and x.dtype in (torch.bfloat16, torch.float32)
            and OPTIONS['prenet_chain']
            and _vc.lib().vc_prenet_chain_supported(Cx, num_units[0], num_units[1])):
        # both layers in one launch, the intermediate never leaves the registers (vc_prenet_chain)
        with variable_scope(scope):
            sc1, sc2 = _scope('dense1'), _scope('dense2')
            cin = Cx if in_features is None else in_features
            if sc1 + '/kernel' in store.vars:
                cin = store.vars[sc1 + '/kernel'].shape[0]
            if Cx != _pad8(cin):
                raise ValueError(' - ERROR, prenet {}: input width {} does not match kernel rows {}'.format(sc1, Cx, cin))
            bt1, b1 = _prep_dense(store, sc1, cin, num_units[0])
            bt2, b2 = _prep_dense(store, sc2, num_units[0], num_units[1])
        pk1, pk2 = store.cached(('prenet_pk', sc1), lambda: (_mfma_pack(bt1, num_units[0], Cx, 0),
                                                               _mfma_pack(bt2, num_units[1], num_units[0], 1)))
        x = x.contiguous()
        out = torch.empty((N_, T_, num_units[1]), dtype=store.dtype, device=x.device)
        _vc.check(_vc.lib().vc_prenet_chain(x.data_ptr(), int(x.dtype == torch.float32), N_ * T_, Cx, Cx, num_units[0], num_units[1], pk1.data_ptr(),
                                            b1.data_ptr(), pk2.data_ptr(), b2.data_ptr(), out.data_ptr(), num_units[1],
                                            _vc.current_stream()))
        return out
    with variable_scope(scope):
        outputs = dense(convert(x, store.dtype), num_units[0], 'relu', name="dense1", in_features=in_features)
        outputs = _dropout(outputs, dropout_rate, is_training)
        outputs = dense(outputs, num_units[1], 'relu', name="dense2")
        outputs = _dropout(outputs, dropout_rate, is_training)
    return outputs


def highwaynet(inputs, num_units=None, scope="highwaynet", reuse=None):
    """modules.py:297-319: H = relu(x W1 + b1), T = sigmoid(x W2 + b2), out = H*T + x*(1-T);
    both matmuls and the gate are one launch (paired-column epilogue)."""
    torch = _torch()
    store = _store()
    x = _as3(inputs)
    N_, T_, Cx = x.shape
    if not num_units:
        num_units = Cx
    if num_units != Cx:
        raise ValueError(' - ERROR, highwaynet: num_units must equal the input width')
    bt, bias = _prep_highway(store, _scope(scope), Cx)
    out = torch.empty_like(x)
    gemm_launch(x, N_ * T_, T_, Cx, Cx, bt.shape[0], [(bt, Cx, 1, 0, 0)], out, Cx, store.vc_dtype,
                mode=_vc.GEMM_HIGHWAY, epi_shift=bias)
    return out


def highway_chain(inputs, num_units, n_layers, scope_fmt='highwaynet_{}', gru_scope=None):
    """n_layers consecutive highwaynet blocks (modules.py:342-345).  bf16 with 128 or 256 units and at
    least 128 frames runs as ONE launch that keeps the activations on chip (vc_highway_chain);
    anything else is the per-layer launch.  Same results either way (bit-identical).
    gru_scope: also apply the bidirectional GRU of that scope (modules.py:346) -- in the fused form
    its input projection is the tail of the same launch and the highway output never reaches HBM."""
    torch = _torch()
    store = _store()
    x = _as3(inputs)
    N_, T_, Cx = x.shape
    fused = (store.dtype == torch.bfloat16 and Cx == num_units and Cx in (128, 256) and N_ * T_ >= 128
             and 1 <= n_layers <= 8 and OPTIONS['highway_chain'])
    if not fused:
        out = x
        for i in range(n_layers):
            out = highwaynet(out, num_units=num_units, scope=scope_fmt.format(i))
        return out if gru_scope is None else gru(out, num_units=num_units, bidirection=True, scope=gru_scope)
    packed, biases = [], []
    for i in range(n_layers):
        sc = _scope(scope_fmt.format(i))
        bt, bias = _prep_highway(store, sc, Cx)

        def build(bt=bt):
            pk = torch.empty(bt.numel(), dtype=store.dtype, device=store.device)
            _vc.check(_vc.lib().vc_highway_pack(bt.data_ptr(), bt.shape[0], Cx, pk.data_ptr(), _vc.current_stream()))
            return pk
        packed.append(store.cached(('highway_pk', sc), build))
        biases.append(bias)
    x = x.contiguous()
    PA = (C.c_void_p * n_layers)(*[p.data_ptr() for p in packed])
    BA = (C.c_void_p * n_layers)(*[b.data_ptr() for b in biases])
    if gru_scope is None:
        out = torch.empty_like(x)
        _vc.check(_vc.lib().vc_highway_chain(x.data_ptr(), N_ * T_, Cx, Cx, n_layers, PA, BA, out.data_ptr(), Cx,
                                             None, None, 0, None, 0, _vc.current_stream()))
        return out
    H = num_units
    gsc = _scope(gru_scope)
    btx, bx, wh_fw, wh_bw = _prep_gru(store, gsc, Cx, H)

    def build_px():
        pk = torch.empty(btx.numel(), dtype=store.dtype, device=store.device)
        _vc.check(_vc.lib().vc_highway_pack(btx.data_ptr(), btx.shape[0], Cx, pk.data_ptr(), _vc.current_stream()))
        return pk
    px = store.cached(('gru_pk', gsc), build_px)
    xproj = torch.empty((N_ * T_, 6 * H), dtype=torch.float32, device=x.device)
    _vc.check(_vc.lib().vc_highway_chain(x.data_ptr(), N_ * T_, Cx, Cx, n_layers, PA, BA, None, 0,
                                         px.data_ptr(), bx.data_ptr(), 6 * H, xproj.data_ptr(), 6 * H,
                                         _vc.current_stream()))
    return _gru_recurrence(xproj, N_, T_, H, wh_fw, wh_bw)


def CBHG(inputs, embed_size=256, num_conv_banks=16, num_highwaynet_blocks=4, dropout_rate=0.5, is_training=True,
         scope="CBHG", use_Cudnn=False, use_lstm=False, reuse=None):
    """modules.py:323-356.  [N, T, E/2] -> [N, T, E]."""
    _refuse_cudnn(use_Cudnn, 'CBHG')
    if use_lstm and is_training:
        raise NotImplementedError(' - ERROR, CBHG(is_training=True): training runs through exec_train_step (training.py), not through this function')
    with variable_scope(scope):
        # max pooling (modules.py:331) rides on the bank launch's stores where that kernel can do it,
        # else on conv1d_1's operand load (2: operand is post-ReLU (>= 0), integer-ordered max)
        enc, pooled = conv1d_banks(inputs, K=num_conv_banks, is_training=is_training, pool_output='auto')   # (N, T, K*128)
        enc = conv1d(enc, filters=embed_size // 2, size=3, scope="conv1d_1", bn_scope="conv1d_1",
                     activation_fn='relu', pool_input=0 if pooled else 2)                  # (N, T, E/2)
        enc = conv1d(enc, filters=embed_size // 2, size=3, scope="conv1d_2", bn_scope="conv1d_2",
                     residual=inputs)                                                      # + residual
        # highway blocks + bidirectional GRU (modules.py:342-354; LSTM when use_lstm)
        if use_lstm:
            enc = highway_chain(enc, embed_size // 2, num_highwaynet_blocks)
            return lstm(enc, num_units=embed_size // 2, bidirection=True)                       # (N, T, E)
        output = highway_chain(enc, embed_size // 2, num_highwaynet_blocks, gru_scope='gru')    # (N, T, E)
    return output


def _cbhg_front(x, E, num_conv_banks, num_highwaynet_blocks, prenet_scope, scope):
    """The fused launch of prenet_CBHG: features [N, T, C] -> (xproj float32 [N*T, 6H], recurrent weights)."""
    torch = _torch()
    store = _store()
    N_, T_, Cx = x.shape
    Cw, H = E // 2, E // 2
    with variable_scope(prenet_scope):
        bt1, b1 = _prep_dense(store, _scope('dense1'), Cx, E)
        bt2, b2 = _prep_dense(store, _scope('dense2'), E, Cw)
        psc = _scope()
    with variable_scope(scope):
        csc = _scope()
        banks = []
        with variable_scope('conv1d_banks'):
            for k in range(1, num_conv_banks + 1):
                with variable_scope('conv1d' if k == 1 else 'num_{}/conv1d'.format(k)):
                    banks.append(_prep_conv(store, _scope(), k, Cw, 128))
            bs, bsh = _prep_bn(store, _scope('bn'), 128 * num_conv_banks)
        with variable_scope('conv1d_1'):
            _prep_conv(store, _scope(), 3, 128 * num_conv_banks, Cw)
            k1 = store.vars[_scope() + '/conv1d/kernel']
        p1s, p1b = _prep_bn(store, _scope('conv1d_1'), Cw)
        with variable_scope('conv1d_2'):
            bt_p2 = _prep_conv(store, _scope(), 3, Cw, Cw)
        p2s, p2b = _prep_bn(store, _scope('conv1d_2'), Cw)
        hws = [_prep_highway(store, _scope('highwaynet_{}'.format(i)), Cw) for i in range(num_highwaynet_blocks)]
        btx, bx, wh_fw, wh_bw = _prep_gru(store, _scope('gru'), Cw, H)

    def build():
        K_ = num_conv_banks
        coef = torch.zeros((_vc.lib().vc_cbhg_front_coef_floats(),), dtype=torch.float32, device=store.device)
        for off, v in ((0, b1), (96, b2), (160, bs), (1184, bsh), (2208, p1s), (2272, p1b), (2336, p2s), (2400, p2b),
                       (2464, bx)) + tuple((2720 + 128 * i, hb) for i, (_, hb) in enumerate(hws)):
            coef[off:off + v.numel()] = v
        w1 = k1.reshape(3, K_, 4, 2, 16, Cw).permute(5, 1, 2, 0, 3, 4).reshape(Cw, 3 * 128 * K_).to(store.dtype)
        return dict(
            d1=_mfma_pack(bt1, E, Cx, 0), d2=_mfma_pack(bt2, Cw, E, 1),
            bank=torch.cat([_mfma_pack(b, 128, (k + 1) * Cw, 0) for k, b in enumerate(banks)]),
            p1=_mfma_pack(w1, Cw, 3 * 128 * K_, 0), p2=_mfma_pack(bt_p2, Cw, 3 * Cw, 0),
            hw=[_mfma_pack(bt, bt.shape[0], Cw, 1) for bt, _ in hws], gx=_mfma_pack(btx, 6 * H, Cw, 1),
            coef=coef)
    pk = store.cached(('cbhg_front', psc, csc), build)
    x = x.contiguous()
    xproj = torch.empty((N_ * T_, 6 * H), dtype=torch.float32, device=x.device)
    d = _vc.CbhgFrontDesc()
    d.d_x, d.x_f32, d.ldx, d.n_windows, d.T = x.data_ptr(), int(x.dtype == torch.float32), Cx, N_, T_
    d.n_features, d.prenet_units, d.width, d.n_banks, d.bank_filters = Cx, E, Cw, num_conv_banks, 128
    d.n_highway, d.gru_units = num_highwaynet_blocks, H
    d.d_pk_dense1, d.d_pk_dense2, d.d_pk_bank = pk['d1'].data_ptr(), pk['d2'].data_ptr(), pk['bank'].data_ptr()
    d.d_pk_proj1, d.d_pk_proj2, d.d_pk_gru = pk['p1'].data_ptr(), pk['p2'].data_ptr(), pk['gx'].data_ptr()
    for i in range(num_highwaynet_blocks):
        d.d_pk_highway[i] = pk['hw'][i].data_ptr()
    d.d_coef = pk['coef'].data_ptr()
    d.d_xproj, d.ldp = xproj.data_ptr(), 6 * H
    _vc.check(_vc.lib().vc_cbhg_front(C.byref(d), _vc.current_stream()))
    return xproj, H, wh_fw, wh_bw


def _mfma_pack(W, rows, K, chained):
    """Row-major bf16 matrix [rows, >= K] -> MFMA fragment order (vc_mfma_pack, include/vc_hip.h)."""
    torch = _torch()
    W = W.contiguous()
    out = torch.empty(((rows + 31) // 32) * ((K + 15) // 16) * 512, dtype=torch.bfloat16, device=W.device)
    _vc.check(_vc.lib().vc_mfma_pack(W.data_ptr(), rows, K, W.shape[1], int(chained), out.data_ptr(), _vc.current_stream()))
    return out


def prenet_CBHG(inputs, embed_size=256, num_conv_banks=16, num_highwaynet_blocks=4, dropout_rate=0.5, is_training=True,
                prenet_scope="prenet", scope="CBHG", use_Cudnn=False, use_lstm=False, in_features=None):
    """prenet (modules.py:274-295) followed by CBHG (modules.py:323-356), as encoder.py:101-107 and
    decoder.py:100-125 / 134-153 call them.  The shipped encoder shape in bf16 runs everything up to the
    recurrence as ONE launch (vc_cbhg_front: the layers are 40 channels wide, launch- and HBM-latency
    bound one by one); every other shape is prenet() + CBHG().  ``inputs`` may be float32 there (the
    conversion is part of the launch).  OPTIONS['cbhg_front'] = False switches the fused form off (A/B)."""
    _refuse_cudnn(use_Cudnn, 'CBHG')
    torch = _torch()
    store = _store()
    x = _as3(inputs)
    N_, T_, Cx = x.shape
    cin = Cx if in_features is None else in_features
    E = embed_size
    fused = (not is_training and not use_lstm and store.dtype == torch.bfloat16 and cin == Cx
             and OPTIONS['cbhg_front']
             and bool(_vc.lib().vc_cbhg_front_supported(Cx, E, E // 2, num_conv_banks, 128, num_highwaynet_blocks, E // 2, T_)))
    if not fused:
        pre = prenet(convert(x, store.dtype), None, E, dropout_rate, is_training, scope=prenet_scope, in_features=in_features)
        return CBHG(pre, E, num_conv_banks, num_highwaynet_blocks, dropout_rate, is_training, scope=scope,
                    use_Cudnn=use_Cudnn, use_lstm=use_lstm)
    xproj, H, wh_fw, wh_bw = _cbhg_front(x, E, num_conv_banks, num_highwaynet_blocks, prenet_scope, scope)
    return _gru_recurrence(xproj, N_, T_, H, wh_fw, wh_bw)


def softmax_argmax_dual(logits, pad_to):
    """softmax_argmax with a second, zero-padded bf16 copy of the probabilities from the same launch
    (the decoder's input layout).  Returns (prob float32 [N, T, n], class ids, prob bf16 [N, T, pad_to])."""
    torch = _torch()
    x = _as3(logits)
    N_, T_, W = x.shape
    prob = torch.empty((N_, T_, W), dtype=torch.float32, device=x.device)
    prob16 = torch.empty((N_, T_, pad_to), dtype=torch.bfloat16, device=x.device)
    cls = torch.empty((N_, T_), dtype=torch.int32, device=x.device)
    _vc.check(_vc.lib().vc_softmax_argmax_dual(x.data_ptr(), N_ * T_, W, W, prob.data_ptr(), W, prob16.data_ptr(), pad_to,
                                               cls.data_ptr(), _vc.current_stream()))
    return prob, cls, prob16


def create_stage_variables(store, scope, in_features, embed_size, num_conv_banks, num_highwaynet_blocks, n_output,
                           use_lstm=False):
    """Create (if absent) every variable of one prenet -> CBHG -> dense(n_output) stage under
    ``scope`` with TensorFlow's default initialisers, in graph order, WITHOUT launching kernels
    (so checkpoints can be restored before the first forward).  Names as in the reference's
    checkpoints."""
    E, H = embed_size, embed_size // 2
    store.get(scope + '/prenet/dense1/kernel', (in_features, E), 'glorot')
    store.get(scope + '/prenet/dense1/bias', (E,), 0.0)
    store.get(scope + '/prenet/dense2/kernel', (E, H), 'glorot')
    store.get(scope + '/prenet/dense2/bias', (H,), 0.0)
    b = scope + '/CBHG/conv1d_banks'
    for k in range(1, num_conv_banks + 1):
        sub = b + ('/conv1d' if k == 1 else '/num_{}/conv1d'.format(k))
        store.get(sub + '/conv1d/kernel', (k, H, BANK_FILTERS), 'glorot')
    _bn_vars(store, b + '/bn', BANK_FILTERS * num_conv_banks)
    store.get(scope + '/CBHG/conv1d_1/conv1d/kernel', (3, BANK_FILTERS * num_conv_banks, H), 'glorot')
    _bn_vars(store, scope + '/CBHG/conv1d_1', H)
    store.get(scope + '/CBHG/conv1d_2/conv1d/kernel', (3, H, H), 'glorot')
    _bn_vars(store, scope + '/CBHG/conv1d_2', H)
    for i in range(num_highwaynet_blocks):
        hs = scope + '/CBHG/highwaynet_{}'.format(i)
        store.get(hs + '/dense1/kernel', (H, H), 'glorot')
        store.get(hs + '/dense1/bias', (H,), 0.0)
        store.get(hs + '/dense2/kernel', (H, H), 'glorot')
        store.get(hs + '/dense2/bias', (H,), -1.0)
    for d in ('fw', 'bw'):
        if use_lstm:
            ls = scope + '/CBHG/lstm/bidirectional_rnn/{}/lstm_cell'.format(d)
            store.get(ls + '/kernel', (2 * H, 4 * H), 'glorot')
            store.get(ls + '/bias', (4 * H,), 0.0)
            continue
        gs = scope + '/CBHG/gru/bidirectional_rnn/{}/gru_cell'.format(d)
        store.get(gs + '/gates/kernel', (2 * H, 2 * H), 'glorot')
        store.get(gs + '/gates/bias', (2 * H,), 1.0)
        store.get(gs + '/candidate/kernel', (2 * H, H), 'glorot')
        store.get(gs + '/candidate/bias', (H,), 0.0)
    store.get(scope + '/y_logits/kernel', (E, n_output), 'glorot')
    store.get(scope + '/y_logits/bias', (n_output,), 0.0)


def softmax_argmax(logits, n_valid=None, pad_to=None, out_dtype=None):
    """tf.nn.softmax + tf.argmax over the last axis (encoder.py:110-111).  Returns
    (prob [N, T, pad_to] with zero padding columns, class ids int32 [N, T])."""
    torch = _torch()
    x = _as3(logits)
    if x.dtype != torch.float32:
        raise ValueError(' - ERROR, softmax_argmax wants float32 logits')
    N_, T_, W = x.shape
    n = W if n_valid is None else n_valid
    ldp = n if pad_to is None else pad_to
    out_dtype = out_dtype or torch.float32
    prob = torch.empty((N_, T_, ldp), dtype=out_dtype, device=x.device)
    cls = torch.empty((N_, T_), dtype=torch.int32, device=x.device)
    code = {torch.float32: _vc.VC_F32, torch.bfloat16: _vc.VC_BF16}[out_dtype]
    _vc.check(_vc.lib().vc_softmax_argmax(x.data_ptr(), N_ * T_, n, W, prob.data_ptr(), ldp, code,
                                          cls.data_ptr(), _vc.current_stream()))
    return prob, cls
