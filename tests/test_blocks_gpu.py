"""Network blocks on the MI355X vs the CPU oracle (oracle/model_oracle.py, TF-1.9 semantics).

Every test calls the HIP kernels through the C ABI (modules.py -> vc_conv_gemm / vc_gru_bidir /
vc_softmax_argmax).  Tolerances: float32 path abs/rel 2e-5 against a float64 oracle (the f32
MFMA is an exact fp32 fma chain; only the summation order differs); bf16 path 3e-2 abs on O(1)
activations."""
import numpy as np
import pytest
import torch

import _vc

from oracle import model_oracle as mo

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _default_kernel_options():
    """Tests switch between equivalent kernels with _vc.set_option / modules.OPTIONS (never the environment);
    every test starts from, and leaves, the library's defaults."""
    import _vc
    import modules
    names = ('bank256', 'bank256_xcd', 'conv256', 'conv256_min_k', 'conv256_wm', 'proj256', 'wgrad_xcd', 'gru_mfma',
             'cbhg_front_mi')
    saved = dict(modules.OPTIONS)
    yield
    for n in names:
        _vc.set_option(n, -1)
    modules.OPTIONS.update(saved)


def _store(dtype, wdict=None):
    import modules
    st = modules.VariableStore(dtype)
    return st


def _close(dev, ref, tol, what):
    dev = dev.float().cpu().double()
    ref = ref.double()
    assert dev.shape == ref.shape, (what, dev.shape, ref.shape)
    err = (dev - ref).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    assert err <= tol * scale, '%s: max abs err %.3e (scale %.2f) > %.1e' % (what, err, scale, tol)


TOL = {'float32': 2e-5, 'bfloat16': 3e-2}


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
@pytest.mark.parametrize('N,T,cin,units', [(2, 40, 80, 80), (3, 16, 40, 61), (1, 400, 256, 201), (2, 50, 64, 520)])
def test_dense(dtype, N, T, cin, units):
    import modules
    rng = np.random.RandomState(0)
    st = _store(dtype)
    x = torch.from_numpy(rng.standard_normal((N, T, cin)).astype(np.float32))
    with modules.variable_store(st), modules.variable_scope('m'):
        for act in (None, 'relu', 'sigmoid'):
            y = modules.dense(modules.convert(x.cuda(), st.dtype), units, act, name='d_%s' % act, out_f32=True)
            w = {k: v.cpu().double() for k, v in st.vars.items()}
            xr = x.double() if dtype == 'float32' else x.bfloat16().double()
            ref = mo.dense(xr, w, 'm/d_%s' % act, act)
            _close(y, ref, TOL[dtype], 'dense %s' % act)


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
@pytest.mark.parametrize('size', [1, 2, 3, 4, 5, 8])
def test_conv1d_same_padding(dtype, size):
    """TF SAME padding incl. even kernels (left (k-1)//2, right k-1-left), no leakage across
    windows, M / N / K tails."""
    import modules
    rng = np.random.RandomState(size)
    N, T, cin, f = 3, 37, 40, 72
    st = _store(dtype)
    x = torch.from_numpy(rng.standard_normal((N, T, cin)).astype(np.float32))
    with modules.variable_store(st), modules.variable_scope('c'):
        y = modules.conv1d(modules.convert(x.cuda(), st.dtype), filters=f, size=size, scope='cv')
    k = st.vars['c/cv/conv1d/kernel'].cpu().double()
    xr = x.double() if dtype == 'float32' else x.bfloat16().double()
    if dtype == 'bfloat16':
        k = k.float().bfloat16().double()
    _close(y, mo.conv1d(xr, k), TOL[dtype], 'conv1d k=%d' % size)


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_conv1d_fused_bn_relu_pool_residual(dtype):
    import modules
    rng = np.random.RandomState(3)
    N, T, cin, f = 2, 48, 128, 40
    st = _store(dtype)
    x = torch.from_numpy(np.abs(rng.standard_normal((N, T, cin))).astype(np.float32))
    res = torch.from_numpy(rng.standard_normal((N, T, f)).astype(np.float32))
    with modules.variable_store(st), modules.variable_scope('c'):
        modules._bn_vars(st, 'c/p1', f)
        for nm in ('beta', 'gamma', 'moving_mean', 'moving_variance'):
            v = rng.uniform(0.5, 1.5, f) if nm in ('gamma', 'moving_variance') else rng.uniform(-0.3, 0.3, f)
            st.assign('c/p1/' + nm, v.astype(np.float32))
        y = modules.conv1d(modules.convert(x.cuda(), st.dtype), filters=f, size=3, scope='p1', bn_scope='p1',
                           activation_fn='relu', pool_input=True,
                           residual=modules.convert(res.cuda(), st.dtype))
    w = {k: v.cpu().double() for k, v in st.vars.items()}
    cast = (lambda t: t.double()) if dtype == 'float32' else (lambda t: t.bfloat16().double())
    k = cast(st.vars['c/p1/conv1d/kernel'].cpu())
    ref = torch.relu(mo.bn(mo.conv1d(mo.max_pool_2_same(cast(x)), k), w, 'c/p1')) + cast(res)
    _close(y, ref, TOL[dtype], 'fused conv1d')


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
@pytest.mark.parametrize('K,cin', [(6, 40), (5, 32), (16, 128)])
def test_conv1d_banks(dtype, K, cin):
    import modules
    rng = np.random.RandomState(K)
    N, T = 2, 56
    st = _store(dtype)
    x = torch.from_numpy(rng.standard_normal((N, T, cin)).astype(np.float32))
    with modules.variable_store(st), modules.variable_scope('e'):
        y = modules.conv1d_banks(modules.convert(x.cuda(), st.dtype), K=K, is_training=False)
        for nm in ('beta', 'gamma', 'moving_mean', 'moving_variance'):
            v = rng.uniform(0.5, 1.5, 128 * K) if nm in ('gamma', 'moving_variance') else rng.uniform(-0.3, 0.3, 128 * K)
            st.assign('e/conv1d_banks/bn/' + nm, v.astype(np.float32))
        y = modules.conv1d_banks(modules.convert(x.cuda(), st.dtype), K=K, is_training=False)
    assert y.shape == (N, T, 128 * K)
    cast = (lambda t: t.double()) if dtype == 'float32' else (lambda t: t.float().bfloat16().double())
    w = {k: (cast(v.cpu()) if k.endswith('kernel') else v.cpu().double()) for k, v in st.vars.items()}
    ref = mo.conv1d_banks(cast(x), w, 'e/conv1d_banks', K)
    _close(y, ref, TOL[dtype], 'banks K=%d' % K)


@pytest.mark.parametrize('N,T,cin,f,size,pool,with_res', [(2, 400, 512, 128, 3, 2, False), (3, 100, 384, 256, 3, 0, True),
                                                         (1, 333, 1024, 128, 1, 0, False), (5, 77, 256, 256, 5, 2, True),
                                                         # >= 2048 rows x 128 columns: the 256-row blocks (eight waves)
                                                         (6, 400, 1024, 128, 3, 0, False), (9, 250, 512, 128, 3, 2, True),
                                                         (11, 197, 256, 128, 7, 0, True),
                                                         # one tap per slab (dense layers): the slab is needed one section after its request
                                                         (7, 400, 2048, 128, 1, 0, False), (2, 400, 1024, 256, 1, 0, True), (1, 130, 4096, 128, 1, 0, False)])
def test_conv1d_deep_pipeline_kernel(N, T, cin, f, size, pool, with_res):
    """Long-K single-filter bf16 convolutions run on vc_conv256.hip (LDS-direct operand loads, the
    max-pool taken on the fragments).  Bit-identical to conv_kernel / gemm_kernel and within the bf16
    tolerance of the oracle; window edges (SAME padding, last frame pooling with itself) included."""
    import modules
    from conftest import poison_gpu_state
    rng = np.random.RandomState(cin + T)
    st = _store('bfloat16')
    x = torch.from_numpy(np.abs(rng.standard_normal((N, T, cin))).astype(np.float32))
    res = torch.from_numpy(rng.standard_normal((N, T, f)).astype(np.float32))
    xd = modules.convert(x.cuda(), st.dtype)
    rd = modules.convert(res.cuda(), st.dtype) if with_res else None
    with modules.variable_store(st), modules.variable_scope('c'):
        modules._bn_vars(st, 'c/p1', f)
        for nm in ('beta', 'gamma', 'moving_mean', 'moving_variance'):
            v = rng.uniform(0.5, 1.5, f) if nm in ('gamma', 'moving_variance') else rng.uniform(-0.3, 0.3, f)
            st.assign('c/p1/' + nm, v.astype(np.float32))
        kw = dict(filters=f, size=size, scope='p1', bn_scope='p1', activation_fn='relu', pool_input=pool, residual=rd)
        _vc.set_option('conv256', 0)
        y_old = modules.conv1d(xd, **kw)
        _vc.set_option('conv256', -1)
        poison_gpu_state()
        y = modules.conv1d(xd, **kw)
        again = [modules.conv1d(xd, **kw) for _ in range(12)]             # a rare ordering bug shows as a rare mismatch
    torch.cuda.synchronize()
    assert not torch.isnan(y.float()).any()
    assert torch.equal(y, y_old)
    assert all(torch.equal(y, z) for z in again)
    w = {k: v.cpu().double() for k, v in st.vars.items()}
    cast = lambda t: t.bfloat16().double()
    xin = mo.max_pool_2_same(cast(x)) if pool else cast(x)
    ref = torch.relu(mo.bn(mo.conv1d(xin, cast(st.vars['c/p1/conv1d/kernel'].cpu())), w, 'c/p1'))
    if with_res:
        ref = ref + cast(res)
    _close(y, ref, TOL['bfloat16'], 'conv256 cin=%d' % cin)


@pytest.mark.parametrize('N,T,cin,K', [(3, 100, 128, 32), (5, 333, 64, 8), (2, 400, 256, 32), (1, 256, 64, 2)])
def test_conv1d_banks_paired_256_tile_kernel(N, T, cin, K):
    """bf16 banks with >= 256 frames, 64-channel slabs and an even K run on vc_bank256.hip (filter
    widths paired, 256 x 256 tiles).  Checked against the oracle at the bf16 tolerance and, bit for
    bit, against conv_kernel (same products in the same order), with LDS left full of NaN by the
    kernel before it (stale LDS bytes once leaked into the narrower filter's missing tap)."""
    import modules
    import training
    rng = np.random.RandomState(K + T)
    st = _store('bfloat16')
    x = torch.from_numpy(rng.standard_normal((N, T, cin)).astype(np.float32))
    xd = modules.convert(x.cuda(), st.dtype)
    with modules.variable_store(st), modules.variable_scope('e'):
        modules.conv1d_banks(xd, K=K, is_training=False)
        for nm in ('beta', 'gamma', 'moving_mean', 'moving_variance'):
            v = rng.uniform(0.5, 1.5, 128 * K) if nm in ('gamma', 'moving_variance') else rng.uniform(-0.3, 0.3, 128 * K)
            st.assign('e/conv1d_banks/bn/' + nm, v.astype(np.float32))
        _vc.set_option('bank256', 0)
        y_old = modules.conv1d_banks(xd, K=K, is_training=False)
        _vc.set_option('bank256', -1)
        nan = torch.full((4096, 2048), float('nan'), device='cuda')
        out = torch.empty(2048, device='cuda')
        ys = []
        for _ in range(3):
            training._Ops.col_sum(nan, 4096, 2048, 2048, out)          # reduces through LDS on every CU
            ys.append(modules.conv1d_banks(xd, K=K, is_training=False))
    torch.cuda.synchronize()
    for y in ys:
        assert not torch.isnan(y.float()).any()
        assert torch.equal(y, y_old)
    cast = lambda t: t.float().bfloat16().double()
    w = {k: (cast(v.cpu()) if k.endswith('kernel') else v.cpu().double()) for k, v in st.vars.items()}
    ref = mo.conv1d_banks(cast(x), w, 'e/conv1d_banks', K)
    _close(ys[0], ref, TOL['bfloat16'], 'banks256 K=%d' % K)


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
@pytest.mark.parametrize('H', [40, 128, 72])
def test_highwaynet(dtype, H):
    import modules
    rng = np.random.RandomState(H)
    N, T = 2, 44
    st = _store(dtype)
    x = torch.from_numpy(rng.standard_normal((N, T, H)).astype(np.float32))
    with modules.variable_store(st), modules.variable_scope('h'):
        y = modules.highwaynet(modules.convert(x.cuda(), st.dtype), num_units=H, scope='highwaynet_0')
        st.assign('h/highwaynet_0/dense1/bias', rng.uniform(-0.2, 0.2, H).astype(np.float32))
        y = modules.highwaynet(modules.convert(x.cuda(), st.dtype), num_units=H, scope='highwaynet_0')
    assert float(st.vars['h/highwaynet_0/dense2/bias'][0]) == -1.0          # TF initialiser (modules.py:317)
    cast = (lambda t: t.double()) if dtype == 'float32' else (lambda t: t.float().bfloat16().double())
    w = {k: (cast(v.cpu()) if k.endswith('kernel') else v.cpu().double()) for k, v in st.vars.items()}
    _close(y, mo.highwaynet(cast(x), w, 'h/highwaynet_0'), TOL[dtype], 'highway H=%d' % H)


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
@pytest.mark.parametrize('H,T', [(40, 400), (128, 60), (256, 24), (24, 33)])
def test_gru_bidirectional(dtype, H, T):
    import modules
    rng = np.random.RandomState(H + T)
    N = 3
    st = _store(dtype)
    x = torch.from_numpy((0.7 * rng.standard_normal((N, T, H))).astype(np.float32))
    with modules.variable_store(st), modules.variable_scope('g'):
        y = modules.gru(modules.convert(x.cuda(), st.dtype), num_units=H, bidirection=True)
    assert y.shape == (N, T, 2 * H)
    assert float(st.vars['g/gru/bidirectional_rnn/fw/gru_cell/gates/bias'][0]) == 1.0
    cast = (lambda t: t.double()) if dtype == 'float32' else (lambda t: t.float().bfloat16().double())
    w = {k: (cast(v.cpu()) if k.endswith('kernel') else v.cpu().double()) for k, v in st.vars.items()}
    ref = mo.gru_bidirectional(cast(x), w, 'g/gru')
    _close(y, ref, 5e-5 if dtype == 'float32' else 3e-2, 'gru H=%d' % H)


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_gru_unidirectional(dtype):
    """modules.py:202-204: gru(bidirection=False) = tf.nn.dynamic_rnn over one GRUCell, variables <scope>/rnn/gru_cell."""
    import modules
    rng = np.random.RandomState(17)
    N, T, H = 3, 50, 40
    st = _store(dtype)
    x = torch.from_numpy((0.7 * rng.standard_normal((N, T, H))).astype(np.float32))
    with modules.variable_store(st), modules.variable_scope('g'):
        y = modules.gru(modules.convert(x.cuda(), st.dtype), num_units=H, bidirection=False)
    assert y.shape == (N, T, H) and 'g/gru/rnn/gru_cell/gates/kernel' in st.vars
    cast = (lambda t: t.double()) if dtype == 'float32' else (lambda t: t.float().bfloat16().double())
    w = {k: (cast(v.cpu()) if k.endswith('kernel') else v.cpu().double()) for k, v in st.vars.items()}
    _close(y, mo.gru_direction(cast(x), w, 'g/gru/rnn'), 5e-5 if dtype == 'float32' else 3e-2, 'gru unidirectional')


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
@pytest.mark.parametrize('H,T,bidir', [(40, 60, True), (128, 24, True), (72, 33, False)])
def test_lstm(dtype, H, T, bidir):
    """modules.lstm (modules.py:207-243: LSTMCell defaults, forget_bias 1.0, gate order i, j, f, o) vs the oracle's
    restatement (parity unpinned at the TensorFlow boundary: the reference ships no LSTM graph)."""
    import modules
    rng = np.random.RandomState(H + T)
    N = 3
    st = _store(dtype)
    x = torch.from_numpy((0.7 * rng.standard_normal((N, T, H))).astype(np.float32))
    with modules.variable_store(st), modules.variable_scope('l'):
        modules.lstm(modules.convert(x.cuda(), st.dtype), num_units=H, bidirection=bidir)
        for n, v in list(st.vars.items()):
            if n.endswith('bias'):
                st.assign(n, rng.uniform(-0.3, 0.3, tuple(v.shape)).astype(np.float32))
        y = modules.lstm(modules.convert(x.cuda(), st.dtype), num_units=H, bidirection=bidir)
    assert y.shape == (N, T, 2 * H if bidir else H)
    cast = (lambda t: t.double()) if dtype == 'float32' else (lambda t: t.float().bfloat16().double())
    w = {k: (cast(v.cpu()) if k.endswith('kernel') else v.cpu().double()) for k, v in st.vars.items()}
    ref = mo.lstm_bidirectional(cast(x), w, 'l/lstm') if bidir else mo.lstm_direction(cast(x), w, 'l/lstm/rnn')
    _close(y, ref, 5e-5 if dtype == 'float32' else 3e-2, 'lstm H=%d' % H)


def test_cbhg_with_lstm_small_decoder():
    """use_lstm = true through the model objects (decoder_specs -> CBHG -> lstm), inference, float32, vs the oracle."""
    from decoder import decoder_specs
    cfg = {'model_name': 'decoder', 'input_shape': [40, 61], 'dropout_rate': 0.1, 'is_training': False,
           'use_Cudnn': False, 'use_lstm': True, 'use_target_mel_step2': False, 'mel_loss_weight': 400,
           'stft_loss_weight': 400, 'loss_type': 'sum',
           'steps_v': [{'embed_size': 64, 'num_conv_banks': 5, 'num_highwaynet_blocks': 2, 'n_output': 80},
                       {'embed_size': 96, 'num_conv_banks': 4, 'num_highwaynet_blocks': 1, 'n_output': 201}]}
    dec = decoder_specs(cfg, None, None)
    assert 'decoder/step1/CBHG/lstm/bidirectional_rnn/fw/lstm_cell/kernel' in dec.store.vars
    assert not any('/gru/' in n for n in dec.store.vars)
    rng = np.random.RandomState(8)
    for n, v in list(dec.store.vars.items()):
        if n.endswith('bias') or n.endswith('beta') or n.endswith('moving_mean'):
            dec.store.assign(n, rng.uniform(-0.2, 0.2, tuple(v.shape)).astype(np.float32))
        elif n.endswith('gamma') or n.endswith('moving_variance'):
            dec.store.assign(n, rng.uniform(0.5, 1.5, tuple(v.shape)).astype(np.float32))
    ppg = torch.softmax(torch.from_numpy(rng.standard_normal((2, 40, 61)) * 2), -1).float().numpy()
    r = dec.predict(ppg)
    w = {k: v.cpu().double() for k, v in dec.store.vars.items()}
    ym, ys = mo.decoder_forward(torch.from_numpy(ppg).double(), w, cfg)
    assert np.abs(r.y_mel - ym.numpy()).max() < 1e-4 and np.abs(r.y_stft - ys.numpy()).max() < 1e-4


@pytest.mark.parametrize('N,T,cin,K', [(3, 100, 128, 32), (2, 400, 64, 4), (4, 255, 64, 2), (7, 64, 128, 6)])
def test_conv1d_banks_pooled_output(N, T, cin, K):
    """epi_pool: the bank launch stores max_pooling1d(2, 1, 'same') of its result (modules.py:331);
    row tiles then overlap by one frame.  Exactly the pool of the unpooled launch, window ends and
    tile seams (T = 255 puts every window end on a seam) included."""
    import modules
    rng = np.random.RandomState(K * T)
    st = _store('bfloat16')
    xd = modules.convert(torch.from_numpy(rng.standard_normal((N, T, cin)).astype(np.float32)).cuda(), st.dtype)
    with modules.variable_store(st), modules.variable_scope('e'):
        y = modules.conv1d_banks(xd, K=K, is_training=False)
        yp, pooled = modules.conv1d_banks(xd, K=K, is_training=False, pool_output='auto')
    torch.cuda.synchronize()
    assert pooled == (N * T >= 256)
    ref = torch.maximum(y.float(), torch.cat([y.float()[:, 1:], y.float()[:, -1:]], dim=1)) if pooled else y.float()
    assert torch.equal(yp.float(), ref)


def test_epi_pool_is_rejected_where_unsupported():
    import ctypes as C
    import _vc
    import modules
    st = _store('float32')
    x = torch.zeros(2, 200, 64, device='cuda')
    with modules.variable_store(st), modules.variable_scope('e'):
        modules.conv1d_banks(x, K=4, is_training=False)
        with pytest.raises(_vc.VCError, match='epi_pool'):
            bt = st.cached(('conv', 'e/conv1d_banks/conv1d'), lambda: None)
            out = torch.empty(2, 200, 128, device='cuda')
            modules.gemm_launch(x, 400, 200, 64, 64, 128, [(bt, 64, 1, 0, 0)], out, 128, st.vc_dtype,
                                act=_vc.ACT_RELU, epi_pool=1)


@pytest.mark.parametrize('N,T,H,L', [(2, 400, 256, 6), (3, 100, 128, 4), (5, 77, 256, 3), (1, 128, 128, 1), (2, 333, 256, 8)])
def test_highway_chain_single_launch(N, T, H, L):
    """All highwaynet layers of a CBHG block in one launch (activations stay in LDS between layers):
    bit-identical to the per-layer launches, layer by layer within the bf16 tolerance of the oracle."""
    import modules
    from conftest import poison_gpu_state
    rng = np.random.RandomState(H + L)
    st = _store('bfloat16')
    x = torch.from_numpy(rng.standard_normal((N, T, H)).astype(np.float32))
    xd = modules.convert(x.cuda(), st.dtype)
    with modules.variable_store(st), modules.variable_scope('h'):
        modules.OPTIONS['highway_chain'] = False
        modules.highway_chain(xd, H, L)
        for i in range(L):
            st.assign('h/highwaynet_%d/dense1/bias' % i, rng.uniform(-0.2, 0.2, H).astype(np.float32))
            st.assign('h/highwaynet_%d/dense2/bias' % i, rng.uniform(-1.2, 0.2, H).astype(np.float32))
        y_ref = modules.highway_chain(xd, H, L)
        g_ref = modules.highway_chain(xd, H, L, gru_scope='gru')          # per-layer launches + dense + recurrence
        modules.OPTIONS['highway_chain'] = True
        poison_gpu_state()
        y = modules.highway_chain(xd, H, L)
        g = modules.highway_chain(xd, H, L, gru_scope='gru')              # the GRU's input projection rides on the chain
    torch.cuda.synchronize()
    assert not torch.isnan(y.float()).any()
    assert torch.equal(y, y_ref)
    assert g.shape == (N, T, 2 * H) and torch.equal(g, g_ref)
    cast = lambda t: t.float().bfloat16().double()
    w = {k: (cast(v.cpu()) if k.endswith('kernel') else v.cpu().double()) for k, v in st.vars.items()}
    ref = cast(x)
    for i in range(L):
        ref = cast(mo.highwaynet(ref, w, 'h/highwaynet_%d' % i))           # the layers exchange bf16 activations
    _close(y, ref, TOL['bfloat16'], 'highway chain H=%d L=%d' % (H, L))


@pytest.mark.parametrize('H,T,N', [(128, 60, 3), (256, 40, 35), (256, 24, 16), (128, 50, 33)])
def test_gru_mfma_recurrence(H, T, N):
    """The 16-sequences-per-workgroup MFMA recurrence (chosen by itself from 32 sequences up, forced
    here) against the oracle, incl. partly filled sequence groups, and run-to-run identical -- in its eight-wave form
    (the default) and in the four-wave form (gru_mfma4 = 1: all weights in registers, gate weights named as
    accumulator-file operands of hand-placed matrix instructions; kept as a measured alternative), which sums every
    product in the same order and must therefore give the same bits."""
    import modules
    from conftest import poison_gpu_state
    rng = np.random.RandomState(H + T + N)
    st = _store('bfloat16')
    x = torch.from_numpy((0.7 * rng.standard_normal((N, T, H))).astype(np.float32))
    xd = modules.convert(x.cuda(), st.dtype)
    _vc.set_option('gru_mfma', 1)
    try:
        with modules.variable_store(st), modules.variable_scope('g'):
            y = modules.gru(xd, num_units=H, bidirection=True)
            poison_gpu_state()                                # stale LDS (the ring, h) and workspace must not matter
            y2 = modules.gru(xd, num_units=H, bidirection=True)
            _vc.set_option('gru_mfma4', 1)
            y8 = modules.gru(xd, num_units=H, bidirection=True)
            _vc.set_option('gru_mfma4', -1)
            _vc.set_option('gru_mfma', 0)
            yv = modules.gru(xd, num_units=H, bidirection=True)
    finally:
        _vc.set_option('gru_mfma4', -1)
        _vc.set_option('gru_mfma', -1)
    assert not torch.isnan(y.float()).any()
    assert torch.equal(y, y2)
    assert torch.equal(y, y8), (y.float() - y8.float()).abs().max().item()
    cast = lambda t: t.float().bfloat16().double()
    w = {k: (cast(v.cpu()) if k.endswith('kernel') else v.cpu().double()) for k, v in st.vars.items()}
    ref = mo.gru_bidirectional(cast(x), w, 'g/gru')
    _close(y, ref, 3e-2, 'gru mfma H=%d' % H)
    assert (y.float() - yv.float()).abs().max().item() < 2e-2          # the two kernels agree to bf16 rounding


@pytest.mark.parametrize('T,N', [(60, 3), (40, 20), (25, 70)])
def test_gru_small_mfma_recurrence(T, N):
    """The encoder's recurrence (H = 40, bf16; /root/reference/modules.py:168-204) with 16 sequences per WAVE on MFMA
    (csrc/vc_rnn.hip gru_mfma_small_kernel: weights padded to 48 x 64 in registers, wave-private LDS hand-off, no
    barriers; a measured alternative, off by default, forced here) against the oracle, against the one-wave-per-sequence
    kernel it replaces, partly filled groups of 16 included, and run-to-run identical with NaN-poisoned LDS."""
    import modules
    from conftest import poison_gpu_state
    H = 40
    rng = np.random.RandomState(T + N)
    st = _store('bfloat16')
    x = torch.from_numpy((0.7 * rng.standard_normal((N, T, H))).astype(np.float32))
    xd = modules.convert(x.cuda(), st.dtype)
    try:
        with modules.variable_store(st), modules.variable_scope('g'):
            _vc.set_option('gru_small_mfma', 1)
            y = modules.gru(xd, num_units=H, bidirection=True)
            poison_gpu_state()
            y2 = modules.gru(xd, num_units=H, bidirection=True)
            _vc.set_option('gru_small_mfma', 0)
            yw = modules.gru(xd, num_units=H, bidirection=True)
    finally:
        _vc.set_option('gru_small_mfma', -1)
    assert not torch.isnan(y.float()).any()
    assert torch.equal(y, y2)
    cast = lambda t: t.float().bfloat16().double()
    w = {k: (cast(v.cpu()) if k.endswith('kernel') else v.cpu().double()) for k, v in st.vars.items()}
    ref = mo.gru_bidirectional(cast(x), w, 'g/gru')
    _close(y, ref, 3e-2, 'gru small mfma')
    assert (y.float() - yw.float()).abs().max().item() < 2e-2          # the two kernels agree to bf16 rounding


def test_softmax_argmax_exact_ties_and_padding():
    import modules
    rng = np.random.RandomState(1)
    lg = rng.standard_normal((2, 50, 61)).astype(np.float32) * 3
    lg[0, 0, 5] = lg[0, 0, 17] = 9.0                     # tie -> first index (tf.argmax)
    lg[1, 3, :] = 0.0                                    # all equal -> class 0
    d = torch.from_numpy(lg).cuda()
    p, c = modules.softmax_argmax(d)
    ref = torch.softmax(torch.from_numpy(lg).double(), -1)
    _close(p, ref, 1e-6, 'softmax')
    assert c.dtype == torch.int32
    assert np.array_equal(c.cpu().numpy(), np.argmax(lg, -1).astype(np.int32))    # exact, incl. ties
    p64, _ = modules.softmax_argmax(d, pad_to=64, out_dtype=torch.bfloat16)
    assert p64.shape == (2, 50, 64) and float(p64[:, :, 61:].abs().max()) == 0.0
    _close(p64[:, :, :61], ref, 4e-3, 'softmax bf16')


def test_gemm_argument_errors_are_reported():
    import _vc
    import modules
    st = _store('float32')
    x = torch.zeros((1, 10, 42), device='cuda')          # Cin not a multiple of 4
    with modules.variable_store(st), modules.variable_scope('bad'):
        with pytest.raises((_vc.VCError, ValueError)):
            modules.conv1d(x, filters=8, size=3, scope='cv')


@pytest.mark.parametrize('N,T,L,f32_in,mi', [(3, 400, 1, True, '2'), (2, 250, 2, False, '2'), (1, 97, 0, True, '2'),
                                              (5, 400, 1, True, '4'), (2, 123, 3, False, '4'), (1, 8, 1, True, '2')])
def test_encoder_front_single_launch(N, T, L, f32_in, mi):
    """prenet + CBHG of the shipped encoder shape as ONE launch up to the recurrence (vc_cbhg_front) against
    the per-layer launches and the oracle: window edges (SAME padding of every convolution, the pool's last
    frame), tiles that do not divide the window, 0..3 highway layers, float32 / bf16 features, both tile
    heights, NaN-poisoned LDS."""
    import modules
    from conftest import poison_gpu_state
    rng = np.random.RandomState(T + L)
    st = _store('bfloat16')
    x = torch.from_numpy((0.5 * rng.standard_normal((N, T, 80))).astype(np.float32)).cuda()
    xin = x if f32_in else modules.convert(x, st.dtype)
    args = dict(embed_size=80, num_conv_banks=6, num_highwaynet_blocks=L, dropout_rate=0.4, is_training=False)
    with modules.variable_store(st), modules.variable_scope('e'):
        modules.OPTIONS['cbhg_front'] = False
        modules.prenet_CBHG(xin, **args)                                   # creates the variables
        for n, v in list(st.vars.items()):                                   # non-trivial norms / biases
            if n.endswith('gamma') or n.endswith('moving_variance'):
                st.assign(n, rng.uniform(0.5, 1.5, tuple(v.shape)).astype(np.float32))
            elif n.endswith('beta') or n.endswith('moving_mean') or n.endswith('bias'):
                st.assign(n, rng.uniform(-0.3, 0.3, tuple(v.shape)).astype(np.float32))
        y_ref = modules.prenet_CBHG(xin, **args)
        modules.OPTIONS['cbhg_front'] = True
        _vc.set_option('cbhg_front_mi', int(mi))
        assert modules._vc.lib().vc_cbhg_front_supported(80, 80, 40, 6, 128, L, 40, T)
        poison_gpu_state()
        y = modules.prenet_CBHG(xin, **args)
        xp = modules._cbhg_front(xin, 80, 6, L, 'prenet', 'CBHG')[0]
        poison_gpu_state()
        y2 = modules.prenet_CBHG(xin, **args)
    torch.cuda.synchronize()
    assert y.shape == (N, T, 80) and not torch.isnan(y.float()).any() and not torch.isnan(xp).any()
    assert torch.equal(y, y2)                                               # deterministic (fixed-order partial sums)
    # same roundings as the per-layer path, float32 sums in another order
    d = (y.float() - y_ref.float()).abs()
    assert d.max().item() < 2e-2 and d.mean().item() < 2e-4, (d.max().item(), d.mean().item())
    cast = lambda t: t.float().bfloat16().double()
    w = {k: (cast(v.cpu()) if k.endswith('kernel') else v.cpu().double()) for k, v in st.vars.items()}
    ref = mo.cbhg(mo.prenet(cast(x.cpu()), w, 'e/prenet'), w, 'e/CBHG', 6, L)
    _close(y, ref, TOL['bfloat16'], 'fused encoder front T=%d L=%d' % (T, L))


@pytest.mark.parametrize('N,T,cin,E', [(2, 400, 61, 256), (3, 77, 80, 512), (1, 1, 61, 256), (64, 400, 80, 512)])
def test_prenet_single_launch(N, T, cin, E):
    """The decoder stages' prenet as one launch (vc_prenet_chain: intermediate in registers) against the two
    dense launches and the oracle; ragged last block, NaN-poisoned LDS."""
    import modules
    from conftest import poison_gpu_state
    rng = np.random.RandomState(cin + E + T)
    st = _store('bfloat16')
    cp = modules._pad8(cin)
    x = np.zeros((N, T, cp), np.float32)
    x[:, :, :cin] = 0.7 * rng.standard_normal((N, T, cin))
    xd = modules.convert(torch.from_numpy(x).cuda(), st.dtype)
    with modules.variable_store(st), modules.variable_scope('p'):
        modules.OPTIONS['prenet_chain'] = False
        modules.prenet(xd, None, E, 0.1, False, in_features=cin)
        for n, v in list(st.vars.items()):
            if n.endswith('bias'):
                st.assign(n, rng.uniform(-0.3, 0.3, tuple(v.shape)).astype(np.float32))
        y_ref = modules.prenet(xd, None, E, 0.1, False, in_features=cin)
        modules.OPTIONS['prenet_chain'] = True
        assert modules._vc.lib().vc_prenet_chain_supported(cp, E, E // 2)
        poison_gpu_state()
        y = modules.prenet(xd, None, E, 0.1, False, in_features=cin)
        # the form in which every wave streams the weights itself: same products in the same order
        with modules._vc.options(prenet_lds=0):
            poison_gpu_state()
            y_own = modules.prenet(xd, None, E, 0.1, False, in_features=cin)
    torch.cuda.synchronize()
    assert torch.equal(y, y_own)
    assert y.shape == (N, T, E // 2) and not torch.isnan(y.float()).any()
    d = (y.float() - y_ref.float()).abs()
    assert d.max().item() < 1e-2, d.max().item()
    cast = lambda t: t.float().bfloat16().double()
    w = {k: (cast(v.cpu()) if k.endswith('kernel') else v.cpu().double()) for k, v in st.vars.items()}
    h = cast(torch.relu(cast(torch.from_numpy(x[:, :, :cin])) @ w['p/prenet/dense1/kernel'] + w['p/prenet/dense1/bias']))
    ref = torch.relu(h @ w['p/prenet/dense2/kernel'] + w['p/prenet/dense2/bias'])
    _close(y, ref, TOL['bfloat16'], 'fused prenet %d -> %d' % (cin, E))


@pytest.mark.parametrize('N,T,cin,k', [(4, 400, 4096, 3), (5, 250, 1024, 5), (3, 400, 2048, 2)])
def test_wide_projection_on_the_bank_tiles(N, T, cin, k):
    """A single 256-channel convolution with a long K (decoder stage 2's first k = 3 projection) runs on the bank
    kernel's 256 x 256 tiles (its halves as a pair of EQUAL width): bit-identical to conv256_kernel / conv_kernel
    and within the bf16 tolerance of the oracle; window edges and a ragged last tile included, NaN-poisoned LDS."""
    import modules
    from conftest import poison_gpu_state
    rng = np.random.RandomState(cin + k)
    st = _store('bfloat16')
    x = torch.from_numpy(rng.standard_normal((N, T, cin)).astype(np.float32) * 0.5)
    xd = modules.convert(x.cuda(), st.dtype)
    with modules.variable_store(st), modules.variable_scope('p'):
        _vc.set_option('proj256', 0)
        kw = dict(filters=256, size=k, scope='c', bn_scope='c', activation_fn='relu')
        modules.conv1d(xd, **kw)
        for n, v in list(st.vars.items()):
            if n.endswith('gamma') or n.endswith('moving_variance'):
                st.assign(n, rng.uniform(0.5, 1.5, tuple(v.shape)).astype(np.float32))
            elif n.endswith('beta') or n.endswith('moving_mean'):
                st.assign(n, rng.uniform(-0.3, 0.3, tuple(v.shape)).astype(np.float32))
        y_ref = modules.conv1d(xd, **kw)
        _vc.set_option('proj256', -1)
        _vc.set_option('proj256_split', 0)               # one workgroup per row tile: the same summation order
        poison_gpu_state()
        y = modules.conv1d(xd, **kw)
        _vc.set_option('proj256_split', -1)
    torch.cuda.synchronize()
    assert not torch.isnan(y.float()).any()
    assert torch.equal(y, y_ref)
    # and against the oracle (modules.py:334-335: conv1d k, SAME, no bias -> bn -> relu) on the bf16-rounded operands
    w = {n: v.cpu().double() for n, v in st.vars.items()}
    cast = lambda t: t.bfloat16().double()
    ref = torch.relu(mo.bn(mo.conv1d(cast(x), cast(st.vars['p/c/conv1d/kernel'].cpu())), w, 'p/c'))
    _close(y, ref, TOL['bfloat16'], 'proj256 cin=%d k=%d' % (cin, k))


@pytest.mark.parametrize('N,T,cin,k', [(64, 400, 4096, 3), (4, 400, 4096, 3), (5, 250, 1024, 5), (3, 400, 2048, 2)])
def test_wide_projection_with_k_split_over_two_workgroups(N, T, cin, k):
    """The same projection with its K split over two workgroups per row tile (vc_bank256.hip "split K": the first to
    finish publishes its float32 accumulators, the second adds them and runs the epilogue) -- what a launch with at most
    128 row tiles uses when the caller hands it a workspace (modules.gemm_launch does).  (64, 400, 4096, 3) is decoder
    stage 2's first projection at the benchmarked batch (/root/reference/modules.py:334-335).
      * against the oracle at the block tolerance, and against the unsplit launch within one bf16 rounding of the
        result (the two differ only in where the float32 sum over K is associated);
      * deterministic: a + b does not depend on which half arrives last -- repeated launches, NaN-poisoned LDS and
        workspace, an unrelated stream keeping CUs busy (uneven arrival), must be bit-identical;
      * without a workspace the unsplit form runs (same result as proj256_split = 0)."""
    import ctypes as C
    import modules
    from conftest import poison_gpu_state
    rng = np.random.RandomState(cin + k + N)
    st = _store('bfloat16')
    x = torch.from_numpy(rng.standard_normal((N, T, cin)).astype(np.float32) * 0.5)
    xd = modules.convert(x.cuda(), st.dtype)
    kw = dict(filters=256, size=k, scope='c', bn_scope='c', activation_fn='relu')
    with modules.variable_store(st), modules.variable_scope('p'):
        modules.conv1d(xd, **kw)
        for n, v in list(st.vars.items()):
            if n.endswith('gamma') or n.endswith('moving_variance'):
                st.assign(n, rng.uniform(0.5, 1.5, tuple(v.shape)).astype(np.float32))
            elif n.endswith('beta') or n.endswith('moving_mean'):
                st.assign(n, rng.uniform(-0.3, 0.3, tuple(v.shape)).astype(np.float32))
        _vc.set_option('proj256_split', 0)
        y_one = modules.conv1d(xd, **kw)
        _vc.set_option('proj256_split', -1)
        # the library asks for a workspace for this launch, i.e. the split form is what runs next
        d = modules.gemm_desc(xd, N * T, T, cin, cin, 256, [(modules._prep_conv(st, 'p/c', k, cin, 256), k * cin, k, (k - 1) // 2, 0)],
                              torch.empty((N * T, 256), dtype=torch.bfloat16, device='cuda'), 256, _vc.VC_BF16)
        need = _vc.lib().vc_conv_gemm_workspace_bytes(C.byref(d))
        ntm = (N * T + 255) // 256
        assert need == ((ntm * 8 + 255) // 256) * 256 + ntm * 262144, need
        noise_stream = torch.cuda.Stream()
        noise_store = modules.VariableStore('bfloat16')
        noise_x = torch.randn(8, 400, 256, device='cuda').to(torch.bfloat16)
        ys = []
        for rep in range(4):
            poison_gpu_state()
            if rep & 1:
                with torch.cuda.stream(noise_stream), modules.variable_store(noise_store), modules.variable_scope('noise'):
                    for _ in range(3):
                        modules.conv1d_banks(noise_x, K=32, is_training=False)
            ys.append(modules.conv1d(xd, **kw))
        torch.cuda.synchronize()
    y = ys[0]
    assert not torch.isnan(y.float()).any()
    for other in ys[1:]:
        assert torch.equal(y, other)
    # one bf16 rounding of the stored value: |a - b| <= 2^-7 max(|a|, |b|) (+ a float32-sum allowance near zero)
    a_, b_ = y.float(), y_one.float()
    assert bool(((a_ - b_).abs() <= 2.0 ** -7 * torch.maximum(a_.abs(), b_.abs()) + 1e-3).all())
    assert float((a_ != b_).float().mean()) < 0.2          # and most values do not even move
    w = {n: v.cpu().double() for n, v in st.vars.items()}
    cast = lambda t: t.bfloat16().double()
    sub = slice(0, min(N, 4))                            # the oracle on a few windows is enough (and seconds, not minutes)
    ref = torch.relu(mo.bn(mo.conv1d(cast(x[sub]), cast(st.vars['p/c/conv1d/kernel'].cpu())), w, 'p/c'))
    _close(y[sub], ref, TOL['bfloat16'], 'proj256 split cin=%d k=%d' % (cin, k))
    if N > 4:
        ref = torch.relu(mo.bn(mo.conv1d(cast(x[-2:]), cast(st.vars['p/c/conv1d/kernel'].cpu())), w, 'p/c'))
        _close(y[-2:], ref, TOL['bfloat16'], 'proj256 split, last windows')


def test_softmax_dual_output_equals_two_launches():
    """vc_softmax_argmax_dual: float32 posteriors + zero-padded bf16 copy in one launch == two launches."""
    import modules
    rng = np.random.RandomState(12)
    lg = torch.from_numpy((3.0 * rng.standard_normal((3, 77, 61))).astype(np.float32)).cuda()
    p, c = modules.softmax_argmax(lg)
    p16, c16 = modules.softmax_argmax(lg, pad_to=64, out_dtype=torch.bfloat16)
    q, d, q16 = modules.softmax_argmax_dual(lg, 64)
    torch.cuda.synchronize()
    assert torch.equal(p, q) and torch.equal(c, d) and torch.equal(c, c16) and torch.equal(p16, q16)
    assert q16.shape == (3, 77, 64) and float(q16[..., 61:].abs().max()) == 0.0


@pytest.mark.parametrize('N,T,K,H,S', [(4, 400, 8, 256, 1), (4, 400, 8, 256, 4), (3, 97, 5, 128, 2), (32, 400, 32, 128, 16)])
def test_summed_groups_is_the_banks_data_gradient(N, T, K, H, S):
    """vc_gemm_desc.sum_groups: group k convolves input channels [128(k-1), 128k) with its own taps and all groups
    accumulate in ONE output -- the data gradient of conv1d_banks (tf.gradients through modules.py:144-166).  Against a
    float64 torch restatement (sum over k of conv1d(dZ_k, W_k) with TF SAME padding per window), for one block per tile
    (S = 1, fixed order, residual in the epilogue) and for the banks dealt to S blocks that add partial tiles with
    atomics (the residual then is C's starting value)."""
    import modules
    rng = np.random.RandomState(K * 1000 + H + S)
    F = 128
    M = N * T
    dZ = torch.from_numpy(rng.standard_normal((N, T, F * K)).astype(np.float32)).cuda()
    res = torch.from_numpy(rng.standard_normal((M, H)).astype(np.float32)).cuda()
    groups, ref = [], res.double().cpu().view(N, T, H).clone()
    for k in range(1, K + 1):
        w = (rng.standard_normal((H, k * F)) / np.sqrt(k * F)).astype(np.float32)      # operand layout [N = H, taps * Cin]
        wt = torch.from_numpy(w).cuda()
        pad_l = k - 1 - (k - 1) // 2
        groups.append((wt, k * F, k, pad_l, F * (k - 1)))
        x = dZ[:, :, F * (k - 1):F * k].double().cpu().permute(0, 2, 1)                 # [N, Cin, T]
        wk = torch.from_numpy(w).double().view(H, k, F).permute(0, 2, 1)                # [H, Cin, taps]
        xp = torch.nn.functional.pad(x, (pad_l, k - 1 - pad_l))
        ref += torch.nn.functional.conv1d(xp, wk).permute(0, 2, 1)
    if S == 1:
        out = torch.empty((M, H), dtype=torch.float32, device='cuda')
        modules.gemm_launch(dZ.view(M, F * K), M, T, F, F * K, H, groups, out, H, _vc.VC_F32, R=res, ldr=H, out_f32=True, sum_groups=1)
    else:
        out = res.clone()
        modules.gemm_launch(dZ.view(M, F * K), M, T, F, F * K, H, groups, out, H, _vc.VC_F32, out_f32=True, sum_groups=S)
    torch.cuda.synchronize()
    d = (out.double().cpu().view(N, T, H) - ref).abs().max().item()
    assert d < 2e-5 * max(1.0, ref.abs().max().item()), d
    with pytest.raises(_vc.VCError):                      # partial tiles carry no epilogue terms
        modules.gemm_launch(dZ.view(M, F * K), M, T, F, F * K, H, groups, out, H, _vc.VC_F32, R=res, ldr=H, out_f32=True, sum_groups=2)


@pytest.mark.parametrize('H,N', [(256, 6), (192, 3)])
def test_wide_float32_recurrence_on_the_training_forward_kernel_equals_the_streaming_kernel(H, N):
    """float32 inference recurrences of more than 128 units (modules.gru, /root/reference/modules.py:168-204) run on the
    training step's forward kernel, which keeps half of the weights resident (option gru_f32_wide = 0: the streaming
    kernel that re-reads all of them every step).  Same GRUCell arithmetic in float32: the two agree to 2e-6."""
    import modules
    T = 400
    st = modules.VariableStore('float32')
    x = (torch.randn(N, T, H, generator=torch.Generator().manual_seed(H + N)) * 0.5).cuda()
    outs = {}
    for opt in (-1, 0):
        with _vc.options(gru_f32_wide=opt), modules.variable_store(st), modules.variable_scope('g'):
            outs[opt] = modules.gru(x, num_units=H, bidirection=True)
    assert outs[-1].shape == (N, T, 2 * H)
    d = float((outs[-1] - outs[0]).abs().max())
    assert d < 2e-6, d
