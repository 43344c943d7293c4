"""Training convolutions on split-float16 operands (include/vc_hip.h "f16x3", csrc/vc_gemm16.hip).

Host side of vc_split16 / vc_weights16 / vc_gemm16: the float32 convolutions of the decoder's training step
(/root/reference/modules.py:144-166 conv1d_banks, :331-337 the projections, and their data gradients under
tf.gradients, /root/reference/decoder.py:236-246) computed as three float16 MFMA products of exactly split operands --
float32 accuracy (the result's error against float64 equals a float32 GEMM's) at the 16-bit matrix rate.
"""
import ctypes as C

import _vc

BANK_FILTERS = 128      # output channels of one bank filter (modules.BANK_FILTERS)


def _torch():
    import torch
    return torch


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def split16(X, M, Cn, ldx, T, scale=None, shift=None, relu=0, pool=0):
    """-> (X16 [M, 2*Cn] float16 = [hi | lo] of pro(X) * s_window, row_scale [M] = 1 / s_window (+ M/T words of scratch))."""
    torch = _torch()
    out = torch.empty((M, 2 * Cn), dtype=torch.float16, device=X.device)
    rs = torch.empty(M + M // T, dtype=torch.float32, device=X.device)
    _vc.check(_vc.lib().vc_split16(_p(X), M, Cn, ldx, T, _p(scale), _p(shift), int(relu), int(pool), _p(out), _p(rs),
                                   _vc.current_stream()))
    return out, rs


def transpose_split16(X, M, Cn, ldx, T, shift0=0, n_shifts=1, scale=None, shift=None, relu=0, pool=0):
    """Weight-gradient operand: -> (rows (si * Cn + c) of [hi (M) | lo (M)] float16 = channel c of pro(X) read
    shift0 + si frames later (0 outside the window), row_scale [n_shifts * Cn] = 1 / the channel's scale (+ scratch))."""
    torch = _torch()
    out = torch.empty((n_shifts * Cn, 2 * M), dtype=torch.float16, device=X.device)
    rs = torch.empty((n_shifts + 1) * Cn, dtype=torch.float32, device=X.device)
    _vc.check(_vc.lib().vc_transpose_split16(_p(X), M, Cn, ldx, T, _p(scale), _p(shift), int(relu), int(pool), shift0,
                                             n_shifts, _p(out), _p(rs), _vc.current_stream()))
    return out, rs


def gemm16(X16, rs, M, T, Cn, pairs, out, ldc, col_scale=None, col_shift=None, ragged=False, accumulate=False,
           workspace=True, atomic_splits=0, act=0):
    """pairs: list of (Bt0, Bt1, taps0, extra, pad_l, c_off0, c_off1[, row0, nrows0, nrows1, s_off0, s_off1]) (ragged:
    taps / pad ignored).  ``workspace``: True = allocate what a split-K launch wants, False = none (one workgroup per
    row tile), or a uint8 tensor.  ``atomic_splits`` n >= 1: the weight-gradient form (include/vc_hip.h)."""
    torch = _torch()
    lib = _vc.lib()
    d = _vc.Gemm16Desc()
    d.d_X16, d.d_row_scale = X16.data_ptr(), (rs.data_ptr() if rs is not None else None)
    d.M, d.T, d.C, d.ldx = M, T, Cn, X16.shape[1]
    d.n_pairs, d.ragged = len(pairs), int(bool(ragged))
    for i, pr in enumerate(pairs):
        b0, b1, taps0, extra, pad_l, c0, c1 = pr[:7]
        p = d.pairs[i]
        p.d_Bt0, p.d_Bt1, p.taps0, p.extra, p.pad_l, p.c_off0, p.c_off1 = b0.data_ptr(), b1.data_ptr(), taps0, extra, pad_l, c0, c1
        if len(pr) > 7:
            p.row0, p.nrows0, p.nrows1, p.s_off0, p.s_off1 = pr[7:12]
    d.d_col_scale = col_scale.data_ptr() if col_scale is not None else None
    d.d_col_shift = col_shift.data_ptr() if col_shift is not None else None
    d.d_C, d.ldc, d.accumulate, d.atomic_splits, d.act = out.data_ptr(), ldc, int(bool(accumulate)), int(atomic_splits), int(act)
    ws = None
    if atomic_splits:
        pass
    elif workspace is True:
        nbytes = lib.vc_gemm16_workspace_bytes(M, Cn, len(pairs))
        if nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=out.device)
    elif workspace is not False and workspace is not None:
        ws = workspace
    if ws is not None:
        d.d_workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    _vc.check(lib.vc_gemm16(C.byref(d), _vc.current_stream()))
    return out


class Weights16:
    """The float16 operand copies of a set of convolution kernels, rewritten from the float32 weights by ONE
    vc_weights16 call (absmax per scale group, split, per-channel un-scale vectors)."""

    def __init__(self, device):
        self.device = device
        self.items = []
        self.keep = []
        self.n_groups = 0
        self._tab = None

    def new_group(self):
        self.n_groups += 1
        return self.n_groups - 1

    def add(self, W, mode, dst, row_len, tap_stride, plane_stride, base, group, scale_dst=None, scale_n=0):
        """W: float32 [k, cin, cout] (TF layout, contiguous; the view must stay where it is: arena slices do)."""
        k, cin, cout = W.shape
        assert W.is_contiguous() and cin % 32 == 0
        it = _vc.W16Item(W.data_ptr(), dst.data_ptr(), scale_dst.data_ptr() if scale_dst is not None else None,
                         k, cin, cout, mode, row_len, tap_stride, plane_stride, base, group, scale_n)
        self.items.append(it)
        self.keep += [W, dst, scale_dst]
        self._tab = None

    def refresh(self):
        torch = _torch()
        if not self.items:
            return
        if self._tab is None:
            arr = (_vc.W16Item * len(self.items))(*self.items)
            self._tab = torch.frombuffer(bytearray(arr), dtype=torch.uint8).to(self.device)
            self._gmax = torch.empty(self.n_groups, dtype=torch.int32, device=self.device)
        _vc.check(_vc.lib().vc_weights16(_p(self._tab), len(self.items), _p(self._gmax), self.n_groups, _vc.current_stream()))


# ---------------------------------------------------------------------------------------------------------------------
# operand builders for the shapes of a CBHG stage

def bank_forward_operands(w16, kernels, H):
    """kernels: the K float32 [k, H, 128] filters of conv1d_banks, k = 1..K (K even).  -> (pairs, col_scale [128 K])."""
    torch = _torch()
    K = len(kernels)
    assert K % 2 == 0 and K <= 32
    dev = w16.device
    col_scale = torch.empty(BANK_FILTERS * K, dtype=torch.float32, device=dev)
    bts = []
    for k, W in enumerate(kernels, 1):
        bt = torch.empty((BANK_FILTERS, k * 2 * H), dtype=torch.float16, device=dev)
        w16.add(W, 0, bt, k * 2 * H, 2 * H, H, 0, w16.new_group(), col_scale[BANK_FILTERS * (k - 1):], BANK_FILTERS)
        bts.append(bt)
    pairs = []
    for p in range(K // 2):
        k0 = 2 * p + 1
        pairs.append((bts[k0 - 1], bts[k0], k0, 1, (k0 - 1) // 2, BANK_FILTERS * (k0 - 1), BANK_FILTERS * k0))
    return pairs, col_scale


def conv_forward_operands(w16, W):
    """One float32 [k, cin, cout] kernel, cout = 256 (the two 128-column halves of a pair) or 128 (a single-filter
    pair).  -> (pairs, col_scale [cout])."""
    torch = _torch()
    k, cin, cout = W.shape
    assert cout in (128, 256)
    dev = w16.device
    col_scale = torch.empty(cout, dtype=torch.float32, device=dev)
    bt = torch.empty((cout, k * 2 * cin), dtype=torch.float16, device=dev)
    w16.add(W, 0, bt, k * 2 * cin, 2 * cin, cin, 0, w16.new_group(), col_scale, cout)
    if cout == 128:
        return [(bt, bt, k, 0, (k - 1) // 2, 0, 0, 0, 0, -1, 0, 0)], col_scale
    return [(bt[:128], bt[128:], k, 0, (k - 1) // 2, 0, 128)], col_scale


def conv_dgrad_operands(w16, W):
    """Data gradient of conv(X [.., cin], W [k, cin, cout]) w.r.t. X: a convolution of dY [.., cout] with the taps
    reversed, left padding k - 1 - (k - 1) // 2, output channels cin (a multiple of 256).  -> (pairs, col_scale [cin])."""
    torch = _torch()
    k, cin, cout = W.shape
    assert cin % 256 == 0 and cin // 256 <= 16
    dev = w16.device
    col_scale = torch.empty(cin, dtype=torch.float32, device=dev)
    bt = torch.empty((cin, k * 2 * cout), dtype=torch.float16, device=dev)
    w16.add(W, 1, bt, k * 2 * cout, 2 * cout, cout, 0, w16.new_group(), col_scale, cin)
    pad = k - 1 - (k - 1) // 2
    pairs = [(bt[256 * i:256 * i + 128], bt[256 * i + 128:256 * i + 256], k, 0, pad, 256 * i, 256 * i + 128)
             for i in range(cin // 256)]
    return pairs, col_scale


def bank_dgrad_operands(w16, kernels, H):
    """Data gradient of conv1d_banks w.r.t. its input (H = 256 channels: one pair; 128: a single-filter pair): ONE
    ragged launch over dZ [M, 128 K].  -> (pairs, col_scale [H])."""
    torch = _torch()
    K = len(kernels)
    assert H in (128, 256) and K <= 32
    dev = w16.device
    PL = BANK_FILTERS * K * (K + 1) // 2
    col_scale = torch.empty(H, dtype=torch.float32, device=dev)
    bt = torch.empty((H, 2 * PL), dtype=torch.float16, device=dev)
    g = w16.new_group()
    for k, W in enumerate(kernels, 1):
        w16.add(W, 1, bt, 2 * PL, BANK_FILTERS, PL, BANK_FILTERS * k * (k - 1) // 2, g, col_scale if k == 1 else None,
                H if k == 1 else 0)
    if H == 128:
        return [(bt, bt, 0, 0, 0, 0, 0, 0, 0, -1, 0, 0)], col_scale
    return [(bt[:128], bt[128:], 0, 0, 0, 0, 128)], col_scale


# ---------------------------------------------------------------------------------------------------------------------
# weight gradients (contraction over the frames; include/vc_hip.h vc_gemm16_desc.atomic_splits)

def bank_wgrad(XT16, rsX, dZT16, rsZ, H, K, M, grads, grad_base, splits=2):
    """Filter gradients of conv1d_banks, ONE launch: dW_k[j, c, o] += sum_m X[m + j - (k-1)//2, c] dZ[m, 128 (k-1) + o].
    XT16 / rsX: transpose_split16 of the bank input X [M, H] over the K shifts -(K/2 - 1) .. K/2; dZT16 / rsZ: of dZ [M, 128 K]
    (one shift, 0).  grads[k - 1]: the float32 [k, H, 128] gradient of bank k, a view into ``grad_base`` (pre-zeroed: the
    K ranges add with float atomics)."""
    shift0 = -(K // 2 - 1)
    base = grad_base.data_ptr()
    pairs = []
    for p in range(K // 2):
        k0 = 2 * p + 1
        o0, o1 = (grads[k0 - 1].data_ptr() - base) // 4, (grads[k0].data_ptr() - base) // 4
        pairs.append((dZT16[BANK_FILTERS * (k0 - 1):], dZT16[BANK_FILTERS * k0:], 1, 0, 0, o0, o1,
                      (-p - shift0) * H, k0 * H, (k0 + 1) * H, BANK_FILTERS * (k0 - 1), BANK_FILTERS * k0))
    rows = K * H
    gemm16(XT16, rsX, rows, rows, M, pairs, grad_base, BANK_FILTERS, col_scale=rsZ, atomic_splits=splits)


def conv3_wgrad(dQT16, rsQ, PT16, rsP, H, CB, M, dW, splits=5):
    """Filter gradient of the width-3 projection: dW[j, c, o] += sum_m P[m + j - 1, c] dQ[m, o].  dQT16 / rsQ:
    transpose_split16 of dQ [M, H] over the shifts -1, 0, 1; PT16 / rsP: of the projection's input P [M, CB] (shift 0).
    The small operand carries the shifts, so the tile comes out as [(shift, o), c] and is transposed into dW [3, CB, H]."""
    torch = _torch()
    scratch = torch.zeros((3 * H, CB), dtype=torch.float32, device=dW.device)
    pairs = [(PT16[256 * i:], PT16[256 * i + 128:], 1, 0, 0, 256 * i, 256 * i + 128, 0, 3 * H, 3 * H, 256 * i, 256 * i + 128)
             for i in range(CB // 256)]
    gemm16(dQT16, rsQ, 3 * H, 3 * H, M, pairs, scratch, CB, col_scale=rsP, atomic_splits=splits)
    # row (si, o): shift s' = si - 1 of dQ <=> tap j = 1 - s' = 2 - si
    dW.add_(scratch.view(3, H, CB).flip(0).permute(0, 2, 1))
