"""Training convolutions on split-float16 operands (csrc/vc_gemm16.hip, include/vc_hip.h "f16x3") against float64
restatements of tf.layers.conv1d(padding='same') (/root/reference/modules.py:104-140) and of its data gradient (autograd
through the same restatement).  The claim under test: three float16 products of exactly split operands are a float32
convolution -- the error against float64 is the float32 MFMA kernel's own, not a reduced-precision one.

Tolerances: relative L2 error <= 1.5e-6 and max error <= 4e-6 of the tensor's largest magnitude (a float32 GEMM at these K
measures 2e-7 .. 6e-7; bf16 operands would give 3e-3, a single float16 product 3e-4), and never more than 3x the error
of the float32-MFMA kernel on the same inputs."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu

REL_L2, REL_MAX = 1.5e-6, 4e-6


def _conv_same64(x, W):
    """x [N, T, cin] float64, W [k, cin, cout] (TF layout) -> [N, T, cout]; TF SAME: left pad (k - 1) // 2."""
    k = W.shape[0]
    pl = (k - 1) // 2
    xp = F.pad(x.transpose(1, 2), (pl, k - 1 - pl))
    return F.conv1d(xp, W.permute(2, 1, 0)).transpose(1, 2)


def _err(got, ref):
    ref = ref.double()
    d = got.double().cpu() - ref
    return float(d.norm() / ref.norm()), float(d.abs().max() / ref.abs().max())


def _check(got, ref, what, f32_err=None, loose=1.0):
    l2, mx = _err(got, ref)
    print('%s: rel L2 %.2e, max %.2e%s' % (what, l2, mx, '' if f32_err is None else '   (float32 MFMA kernel: %.2e / %.2e)' % f32_err))
    assert l2 < REL_L2 * loose and mx < REL_MAX * loose, (what, l2, mx)
    if f32_err is not None:
        assert l2 < 3 * f32_err[0] + 1e-7, (what, l2, f32_err)


def _rand_acts(N, T, Cn, seed, heavy=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, T, Cn, generator=g)
    if heavy:
        # per-frame magnitudes over several decades, an exact zero row and relu-like sparsity
        x = torch.relu(x) * torch.exp(2.3 * torch.randn(N, T, 1, generator=g))
        x[0, 3] = 0.0
    return x.float()


def test_split16_is_exact_to_22_bits_and_applies_the_prologue():
    import gemm16
    N, T, Cn = 2, 50, 256
    x = _rand_acts(N, T, Cn, 1, heavy=True)
    xd = x.cuda().view(N * T, Cn)
    x16, rs = gemm16.split16(xd, N * T, Cn, Cn, T)
    rs = rs[:N * T]
    rec = (x16[:, :Cn].double() + x16[:, Cn:].double()) * rs.double()[:, None]
    ref = x.view(N * T, Cn).double()
    winmax = ref.view(N, T * Cn).abs().max(dim=1).values.view(N, 1, 1).expand(N, T, 1).reshape(N * T, 1)
    # every element to 2^-22 of itself, or 2^-38 of its window's maximum (float16 subnormal spacing after scaling)
    bound = torch.maximum(ref.abs() * 2.0 ** -21.5, winmax * 2.0 ** -37)
    assert bool(((rec.cpu() - ref).abs() <= bound).all())
    assert bool((x16[3] == 0).all())                                             # the zero row
    lg = torch.log2(rs.double().cpu())
    assert bool((lg == lg.round()).all())                                        # powers of two ...
    assert bool((rs.view(N, T) == rs.view(N, T)[:, :1]).all())                   # ... one per window
    hi_max = x16[:, :Cn].abs().view(N, -1).max(dim=1).values.float().cpu()
    assert bool(((hi_max >= 2.0 ** 14) & (hi_max <= 2.0 ** 15)).all())
    # prologue: affine, relu, pool(2, 1, same) inside each window
    sc = (torch.rand(Cn) + 0.5).cuda()
    sh = torch.randn(Cn).cuda()
    x2 = _rand_acts(N, T, Cn, 2).cuda().view(N * T, Cn)
    x16, rs = gemm16.split16(x2, N * T, Cn, Cn, T, scale=sc, shift=sh, relu=1, pool=1)
    a = torch.relu(x2.double() * sc.double() + sh.double()).view(N, T, Cn)
    pooled = torch.cat([torch.maximum(a[:, :-1], a[:, 1:]), a[:, -1:]], dim=1).view(N * T, Cn)
    rec = (x16[:, :Cn].double() + x16[:, Cn:].double()) * rs[:N * T].double()[:, None]
    assert float((rec - pooled).abs().max() / pooled.abs().max()) < 2e-7         # float32 affine vs float64: 1 ulp


@pytest.mark.parametrize('H,K,N,heavy', [(128, 8, 3, False), (256, 32, 2, False), (128, 32, 3, True)])
def test_bank_forward_pairs_match_float64(H, K, N, heavy):
    import gemm16, modules, _vc
    T = 400
    M = N * T
    x = _rand_acts(N, T, H, 10 + K, heavy)
    g = torch.Generator().manual_seed(77)
    Ws = [(torch.randn(k, H, 128, generator=g) * (0.3 / (k * H) ** 0.5) * (1 + k % 5)).float() for k in range(1, K + 1)]
    ref = torch.cat([_conv_same64(x.double(), W.double()) for W in Ws], dim=2).reshape(M, 128 * K)
    dev = torch.device('cuda')
    Wd = [W.to(dev).contiguous() for W in Ws]
    w16 = gemm16.Weights16(dev)
    pairs, cs = gemm16.bank_forward_operands(w16, Wd, H)
    w16.refresh()
    xd = x.to(dev).view(M, H)
    x16, rs = gemm16.split16(xd, M, H, H, T)
    out = torch.full((M, 128 * K), float('nan'), device=dev)
    gemm16.gemm16(x16, rs, M, T, H, pairs, out, 128 * K, col_scale=cs)
    # the float32-MFMA kernel on the same inputs
    groups = [(W.permute(2, 0, 1).reshape(128, -1).contiguous(), k * H, k, (k - 1) // 2, 128 * (k - 1)) for k, W in enumerate(Wd, 1)]
    o32 = torch.empty((M, 128 * K), device=dev)
    modules.gemm_launch(xd, M, T, H, H, 128, groups, o32, 128 * K, _vc.VC_F32, out_f32=True)
    _check(out, ref, 'bank forward H=%d K=%d%s' % (H, K, ' heavy-tailed' if heavy else ''), _err(o32, ref))


@pytest.mark.parametrize('H', [256, 128])
def test_projection_split_k_is_deterministic_and_matches_float64(H):
    import gemm16, modules, _vc
    N, T, CB = 3, 400, 1024
    M = N * T
    x = _rand_acts(N, T, CB, 5)
    g = torch.Generator().manual_seed(6)
    W = (torch.randn(3, CB, H, generator=g) * 0.02).float()
    bias = torch.randn(H, generator=g).float()
    ref = (_conv_same64(x.double(), W.double()) + bias.double()).reshape(M, H)
    dev = torch.device('cuda')
    Wd = W.to(dev)
    w16 = gemm16.Weights16(dev)
    pairs, cs = gemm16.conv_forward_operands(w16, Wd)
    w16.refresh()
    xd = x.to(dev).view(M, CB)
    x16, rs = gemm16.split16(xd, M, CB, CB, T)
    assert _vc.lib().vc_gemm16_workspace_bytes(M, CB, 1) > 0
    outs = []
    for ws in (True, True, False):
        out = torch.full((M, H), float('nan'), device=dev)
        gemm16.gemm16(x16, rs, M, T, CB, pairs, out, H, col_scale=cs, col_shift=bias.to(dev), workspace=ws)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])                                         # split K: bit-identical run to run
    o32 = torch.empty((M, H), device=dev)
    modules.gemm_launch(xd, M, T, CB, CB, H, [(Wd.permute(2, 0, 1).reshape(H, -1).contiguous(), 3 * CB, 3, 1, 0)], o32, H,
                        _vc.VC_F32, epi_shift=bias.to(dev), out_f32=True)
    e32 = _err(o32, ref)
    _check(outs[0], ref, 'projection H=%d, K split over 8' % H, e32)
    _check(outs[2], ref, 'projection H=%d, one workgroup per row tile' % H, e32)


def test_projection_data_gradient_matches_autograd():
    import gemm16
    N, T, CB, H = 2, 400, 4096, 128
    M = N * T
    g = torch.Generator().manual_seed(8)
    W = (torch.randn(3, CB, H, generator=g) * 0.02).float()
    dq = _rand_acts(N, T, H, 9)
    p = torch.zeros(N, T, CB, dtype=torch.float64, requires_grad=True)
    (ref,) = torch.autograd.grad(_conv_same64(p, W.double()), p, dq.double())
    dev = torch.device('cuda')
    w16 = gemm16.Weights16(dev)
    pairs, cs = gemm16.conv_dgrad_operands(w16, W.to(dev))
    assert len(pairs) == 16
    w16.refresh()
    d16, rs = gemm16.split16(dq.to(dev).view(M, H), M, H, H, T)
    out = torch.full((M, CB), float('nan'), device=dev)
    gemm16.gemm16(d16, rs, M, T, H, pairs, out, CB, col_scale=cs)
    _check(out, ref.reshape(M, CB), 'projection data gradient (16 pairs)')


@pytest.mark.parametrize('K,N,H', [(8, 3, 256), (32, 2, 256), (32, 2, 128)])
def test_bank_data_gradient_ragged_walk_matches_autograd(K, N, H):
    import gemm16
    T = 400
    M = N * T
    g = torch.Generator().manual_seed(20 + K)
    Ws = [(torch.randn(k, H, 128, generator=g) * (0.3 / (k * H) ** 0.5)).float() for k in range(1, K + 1)]
    dz = _rand_acts(N, T, 128 * K, 30 + K)
    res = _rand_acts(N, T, H, 31 + K)
    x = torch.zeros(N, T, H, dtype=torch.float64, requires_grad=True)
    z = torch.cat([_conv_same64(x, W.double()) for W in Ws], dim=2)
    (ref,) = torch.autograd.grad(z, x, dz.double())
    ref = ref + res.double()
    dev = torch.device('cuda')
    w16 = gemm16.Weights16(dev)
    pairs, cs = gemm16.bank_dgrad_operands(w16, [W.to(dev) for W in Ws], H)
    w16.refresh()
    d16, rs = gemm16.split16(dz.to(dev).view(M, 128 * K), M, 128 * K, 128 * K, T)
    outs = []
    for ws in (True, True, False):
        out = res.to(dev).view(M, H).clone()
        gemm16.gemm16(d16, rs, M, T, 128 * K, pairs, out, H, col_scale=cs, ragged=True, accumulate=True, workspace=ws)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    _check(outs[0], ref.reshape(M, H), 'bank data gradient K=%d H=%d (ragged walk, K split)' % (K, H))
    _check(outs[2], ref.reshape(M, H), 'bank data gradient K=%d H=%d (ragged walk, unsplit)' % (K, H),
           loose=2.0)         # ONE float32 accumulation chain over up to 203,000 products: the chain's own rounding


@pytest.mark.parametrize('H,K', [(128, 8), (256, 32)])
def test_weight_gradients_match_autograd(H, K):
    """Filter gradients of the bank and of the width-3 projection (contraction over 1,600 frames in 4 windows) against
    autograd through the float64 restatement; the K ranges add with float atomics, so two runs agree to rounding only."""
    import gemm16
    N, T = 4, 400
    M, CB = N * T, 128 * K
    g = torch.Generator().manual_seed(40 + K)
    x = _rand_acts(N, T, H, 41, heavy=True)
    dz = _rand_acts(N, T, CB, 42)
    Ws = [torch.zeros(k, H, 128, dtype=torch.float64, requires_grad=True) for k in range(1, K + 1)]
    z = torch.cat([_conv_same64(x.double(), W) for W in Ws], dim=2)
    refs = torch.autograd.grad(z, Ws, dz.double())
    dev = torch.device('cuda')
    arena = torch.zeros(sum(W.numel() for W in Ws) + 61, dtype=torch.float32, device=dev)
    grads, off = [], 61                                   # an odd offset: the atomic form needs no alignment
    for W in Ws:
        grads.append(arena[off:off + W.numel()].view(W.shape))
        off += W.numel()
    XT, rsX = gemm16.transpose_split16(x.to(dev).view(M, H), M, H, H, T, shift0=-(K // 2 - 1), n_shifts=K)
    ZT, rsZ = gemm16.transpose_split16(dz.to(dev).view(M, CB), M, CB, CB, T)
    gemm16.bank_wgrad(XT, rsX, ZT, rsZ, H, K, M, grads, arena)
    worst = 0.0
    for k, (gk, rk) in enumerate(zip(grads, refs), 1):
        l2, mx = _err(gk, rk)
        worst = max(worst, l2)
        assert l2 < REL_L2 and mx < REL_MAX, ('bank %d' % k, l2, mx)
    print('bank filter gradients H=%d K=%d: worst rel L2 %.2e' % (H, K, worst))
    assert float(arena[:61].abs().max()) == 0.0
    # projection: P = pool(relu(bn(Zb))) built by the prologue, dQ [M, H]
    zb = _rand_acts(N, T, CB, 43)
    sc, sh = torch.rand(CB, generator=g) + 0.5, torch.randn(CB, generator=g)
    dq = _rand_acts(N, T, H, 44)
    a = torch.relu(zb.double() * sc.double() + sh.double())
    P = torch.cat([torch.maximum(a[:, :-1], a[:, 1:]), a[:, -1:]], dim=1)
    W1 = torch.zeros(3, CB, H, dtype=torch.float64, requires_grad=True)
    (ref1,) = torch.autograd.grad(_conv_same64(P, W1), W1, dq.double())
    PT, rsP = gemm16.transpose_split16(zb.to(dev).view(M, CB), M, CB, CB, T, scale=sc.to(dev), shift=sh.to(dev), relu=1, pool=1)
    QT, rsQ = gemm16.transpose_split16(dq.to(dev).view(M, H), M, H, H, T, shift0=-1, n_shifts=3)
    dW1 = torch.zeros((3, CB, H), dtype=torch.float32, device=dev)
    gemm16.conv3_wgrad(QT, rsQ, PT, rsP, H, CB, M, dW1)
    _check(dW1, ref1, 'projection filter gradient H=%d' % H)


def test_f32_inference_blocks_on_the_split_path_agree_with_the_f32_mfma_kernels():
    """float32 INFERENCE (modules.conv1d_banks / the post-bank conv1d with folded batch norm, relu and the pooled operand:
    /root/reference/modules.py:144-166, 331-335) takes the split-float16 path by default; vc_set_option('f32_f16x3', 0)
    keeps the f32-input MFMA kernels.  Same function, float32 accuracy both ways: the two agree to 6e-6 of the
    tensor's maximum (at this K the f32-MFMA bank kernel alone measures 2.8e-6 from float64, the split path 1.1e-6:
    test_bank_forward_pairs_match_float64)."""
    import modules, _vc
    N, T, H, K = 2, 400, 256, 32
    x = _rand_acts(N, T, H, 50).cuda()
    st = modules.VariableStore('float32')
    outs = {}
    for opt in (-1, 0):
        with _vc.options(f32_f16x3=opt), modules.variable_store(st), modules.variable_scope('blk'):
            b, pooled = modules.conv1d_banks(x, K=K, embed_size=256, is_training=False, pool_output='auto')
            assert not pooled or opt == 0
            y = modules.conv1d(b, filters=H, size=3, scope='conv1d_1', bn_scope='conv1d_1', activation_fn='relu',
                               pool_input=0 if pooled else 2)
        outs[opt] = (b, y, pooled)
    assert ('g16bank', 'blk/conv1d_banks') in st._cache            # the split path ran and cached its operands
    b1, y1, _ = outs[-1]
    b0, y0, pooled0 = outs[0]
    if not pooled0:
        assert float((b1 - b0).abs().max() / b0.abs().max()) < 6e-6
    assert float((y1 - y0).abs().max() / y0.abs().max()) < 6e-6
    assert float(b1.min()) >= 0.0 and float(y1.min()) >= 0.0       # relu in the epilogue


@pytest.mark.parametrize('H,K,N,T', [(64, 4, 5, 24), (192, 6, 3, 100), (256, 32, 7, 20), (128, 2, 1, 400)])
def test_odd_shapes_match_float64(H, K, N, T):
    """Shapes the shipped configuration never produces but the trainer would hand over (any H % 64 == 0, even K <= 32):
    fewer rows than one 256-row tile, windows shorter than the widest filter (every tap of it masked somewhere), one
    channel slab, a single window.  Forward, the projection's data gradient, and -- where the trainer uses it -- the
    bank's data gradient, against float64."""
    import gemm16
    M, CB = N * T, 128 * K
    g = torch.Generator().manual_seed(1000 + H + K)
    x = _rand_acts(N, T, H, 60 + K)
    Ws = [(torch.randn(k, H, 128, generator=g) * (0.3 / (k * H) ** 0.5)).float() for k in range(1, K + 1)]
    W1 = (torch.randn(3, CB, H, generator=g) * 0.05).float()
    dev = torch.device('cuda')
    w16 = gemm16.Weights16(dev)
    Wd = [W.to(dev) for W in Ws]
    fp, fcs = gemm16.bank_forward_operands(w16, Wd, H)
    dp, dcs = gemm16.conv_dgrad_operands(w16, W1.to(dev))
    if H in (128, 256):
        bp, bcs = gemm16.bank_dgrad_operands(w16, Wd, H)
    w16.refresh()
    # forward
    ref = torch.cat([_conv_same64(x.double(), W.double()) for W in Ws], dim=2).reshape(M, CB)
    x16, rs = gemm16.split16(x.to(dev).view(M, H), M, H, H, T)
    out = torch.full((M, CB), float('nan'), device=dev)
    gemm16.gemm16(x16, rs, M, T, H, fp, out, CB, col_scale=fcs)
    _check(out, ref, 'odd shape forward H=%d K=%d N=%d T=%d' % (H, K, N, T))
    # projection data gradient
    dq = _rand_acts(N, T, H, 61)
    p = torch.zeros(N, T, CB, dtype=torch.float64, requires_grad=True)
    (refp,) = torch.autograd.grad(_conv_same64(p, W1.double()), p, dq.double())
    q16, qrs = gemm16.split16(dq.to(dev).view(M, H), M, H, H, T)
    outp = torch.full((M, CB), float('nan'), device=dev)
    gemm16.gemm16(q16, qrs, M, T, H, dp, outp, CB, col_scale=dcs)
    _check(outp, refp.reshape(M, CB), 'odd shape projection data gradient')
    if H in (128, 256):
        dz = _rand_acts(N, T, CB, 62)
        xx = torch.zeros(N, T, H, dtype=torch.float64, requires_grad=True)
        (refb,) = torch.autograd.grad(torch.cat([_conv_same64(xx, W.double()) for W in Ws], dim=2), xx, dz.double())
        z16, zrs = gemm16.split16(dz.to(dev).view(M, CB), M, CB, CB, T)
        outb = torch.zeros((M, H), device=dev)
        gemm16.gemm16(z16, zrs, M, T, CB, bp, outb, H, col_scale=bcs, ragged=True, accumulate=True)
        _check(outb, refb.reshape(M, H), 'odd shape bank data gradient')


def test_bad_arguments_are_refused():
    import gemm16, _vc
    dev = torch.device('cuda')
    x = torch.zeros((400, 96), device=dev)
    with pytest.raises(_vc.VCError):
        gemm16.split16(x, 400, 96, 96, 400)                                      # channels not a multiple of 64
    with pytest.raises(_vc.VCError):
        gemm16.split16(torch.zeros((401, 128), device=dev), 401, 128, 128, 400)  # rows not whole windows
