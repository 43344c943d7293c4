"""Decoder training step on the MI355X (float32), behind ``decoder_specs.exec_train_step``.

Implements /root/reference/decoder.py:185-263 + 327-345 -- forward in training mode (dropout,
batch statistics, moving-average updates), the weighted MSE losses, back-propagation through both
decoder stages (the encoder is frozen and runs in inference mode, decoder.py:581-582,637), the
TensorFlow-style Adam update and the per-epoch learning-rate schedule -- as explicit kernel
launches (include/vc_hip.h "Training step" section).  Data-parallel: one process per GPU, each with
its own batch; the flat gradient buffer (33,186,713 floats at the shipped sizes) is all-reduced over
RCCL before Adam; batch-norm statistics stay per replica like in the reference.

All decoder variables are re-homed into ONE flat float32 arena (plus matching gradient / m / v
arenas) so that Adam is a single launch and the all-reduce a single call; every weight-gradient
kernel writes straight into its TF-layout slice of the gradient arena.
"""
import contextlib
import ctypes as C
import math

import numpy as np

import _vc
import gemm16
import modules
from modules import BN_DECAY, BN_EPS, BANK_FILTERS, gemm_launch

MARGIN = 32          # zero margin (frames) around transposed operands (vc_conv_wgrad)


def _torch():
    import torch
    return torch


def _lib():
    return _vc.lib()


def _st():
    return _vc.current_stream()


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class _Ops:
    """Thin wrappers over the training entry points (float32 device tensors)."""

    # vc_wgrad_desc.splits_allowed: the frame reduction of a weight gradient may be split over several workgroups that
    # add their partial sums with atomics (the gradient arena is zeroed at the start of every step); likewise the banks
    # of a filter bank's data gradient (vc_gemm_desc.sum_groups > 1).  0 = one workgroup per tile, fixed summation
    # order (tests compare the two).
    splits_allowed = 1

    # Weight gradients (their transposed operands, the filter-gradient launches, the bias column sums) are needed only
    # by the optimiser, not by the rest of the backward pass: with a side stream set they run THERE, behind whatever
    # the main stream had queued when they were issued, and fill the CUs the latency-bound recurrence kernels leave
    # idle.  StageTrainer sets / joins it; None = everything on the current stream.
    side_stream = None
    # A second one for the SMALL optimiser-only jobs (recurrence and highway weight gradients): behind the multi-millisecond
    # filter-gradient launches of the first they would start late, and the main stream was seen to stall at the point
    # where it hands work to a backlogged stream until that stream had nearly caught up (train_timeline.log).
    side_stream2 = None

    @staticmethod
    @contextlib.contextmanager
    def side(*inputs, small=False):
        """Run the body on the side stream (if any), ordered behind the current stream's queue; `inputs` are tensors of
        the current stream that the body reads or writes (kept alive for the side stream)."""
        ws = _Ops.side_stream2 if (small and _Ops.side_stream2 is not None) else _Ops.side_stream
        if ws is None:
            yield
            return
        torch = _torch()
        ws.wait_stream(torch.cuda.current_stream())
        for t in inputs:
            if t is not None:
                t.record_stream(ws)
        with torch.cuda.stream(ws):
            yield

    @staticmethod
    def join():
        """The current stream waits for everything issued to the side stream so far."""
        if _Ops.side_stream is not None:
            _torch().cuda.current_stream().wait_stream(_Ops.side_stream)

    @staticmethod
    def transpose(X, M, Cn, ld, T, scale=None, shift=None, relu=0, pool=0, row_shift=0):
        """-> (buffer [Cn + 1, M + 2*MARGIN], ldt).  Data starts at column MARGIN; the launch itself zeroes the margins
        and the slack row."""
        torch = _torch()
        ldt = M + 2 * MARGIN
        with _Ops.side(X, scale, shift):
            buf = torch.empty((Cn + 1, ldt), dtype=torch.float32, device=X.device)    # + one slack row
            _vc.check(_lib().vc_transpose_pad(_p(X), M, Cn, ld, T, _p(scale), _p(shift), int(relu), int(pool),
                                              int(row_shift), _p(buf), ldt, MARGIN, 1, _st()))
        return buf, ldt

    @staticmethod
    def wgrad(XT, ldxt, Cin, M, T, dYT, ldyt, groups):
        """groups: list of (dYT_row_offset, N, taps, shift0, dW tensor/view, ldw)."""
        d = _vc.WgradDesc()
        d.d_XT = XT.data_ptr() + MARGIN * 4
        d.ldxt, d.ldyt, d.Cin, d.M, d.T, d.margin, d.n_groups = ldxt, ldyt, Cin, M, T, MARGIN, len(groups)
        d.splits_allowed = int(_Ops.splits_allowed)
        for i, (roff, N, taps, shift0, dW, ldw) in enumerate(groups):
            g = d.groups[i]
            g.d_dYT = dYT.data_ptr() + (roff * ldyt + MARGIN) * 4
            g.d_dW, g.N, g.taps, g.shift0, g.ldw = dW.data_ptr(), N, taps, shift0, ldw
        with _Ops.side(XT, dYT):
            _vc.check(_lib().vc_conv_wgrad(C.byref(d), _st()))

    @staticmethod
    def bn_stats(X, M, Cn, gamma, beta, mmean, mvar):
        torch = _torch()
        dev = X.device
        scale, shift, mean, rstd = (torch.empty(Cn, dtype=torch.float32, device=dev) for _ in range(4))
        ws = torch.empty(_lib().vc_stats_workspace_floats(M, Cn), dtype=torch.float32, device=dev)
        _vc.check(_lib().vc_bn_train_stats(_p(X), M, Cn, Cn, _p(gamma), _p(beta), _p(mmean), _p(mvar),
                                           BN_DECAY, BN_EPS, _p(scale), _p(shift), _p(mean), _p(rstd), _p(ws), _st()))
        return scale, shift, mean, rstd

    @staticmethod
    def bn_backward(G, X, M, Cn, T, gamma, st, mode, dgamma, dbeta):
        torch = _torch()
        scale, shift, mean, rstd = st
        dX = torch.empty((M, Cn), dtype=torch.float32, device=X.device)
        ws = torch.empty(_lib().vc_stats_workspace_floats(M, Cn), dtype=torch.float32, device=X.device)
        _vc.check(_lib().vc_bn_backward(_p(G), _p(X), M, Cn, Cn, T, _p(gamma), _p(scale), _p(shift), _p(mean), _p(rstd),
                                        mode, _p(dX), _p(dgamma), _p(dbeta), _p(ws), _st()))
        return dX

    @staticmethod
    def col_sum(X, M, Cn, ld, out, accumulate=0):
        torch = _torch()
        with _Ops.side(X, out):
            ws = torch.empty(64 * Cn, dtype=torch.float32, device=X.device)
            _vc.check(_lib().vc_col_sum(_p(X), M, Cn, ld, _p(out), int(accumulate), _p(ws), _st()))


class StageTrainer:
    """Flat parameter / gradient / Adam arenas for the variables under one model scope, plus the
    train-mode forward and the backward of a prenet -> CBHG -> dense stage (shared by the decoder's
    two stages and the encoder)."""

    opt_scope = 'dec_opt'

    def __init__(self, model):
        torch = _torch()
        self.dec = model
        self.store = model.store
        if self.store.dtype != torch.float32:
            raise NotImplementedError(' - ERROR, training runs in float32 (compute_dtype must be float32)')
        c = model.cfg_d
        self.cfg = c
        scope = model._scope
        names = self.store.trainable_names(scope + '/')
        total = sum(self.store.vars[n].numel() for n in names)
        dev = self.store.device
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.names, self.offsets = names, {}
        off = 0
        for n in names:
            v = self.store.vars[n]
            k = v.numel()
            view = self.flat[off:off + k].view(v.shape)
            view.copy_(v)
            self.store.vars[n] = view                      # re-home the variable into the arena
            self.offsets[n] = (off, k, tuple(v.shape))
            off += k
        self.store.invalidate()
        self.total = total
        self.step_count = int(model.opt_state[self.opt_scope + '/global_step'])
        self.seed = int(c.get('dropout_seed', 1234))
        self.keep = 1.0 - float(c['dropout_rate'])
        self.loss_ws = torch.empty(256, dtype=torch.float32, device=dev)
        self._side = torch.cuda.Stream(device=dev) if torch.cuda.is_available() else None    # weight gradients (_Ops.side)
        self._side2 = torch.cuda.Stream(device=dev) if torch.cuda.is_available() else None   # the small ones (_Ops.side_stream2)
        self._pending = []                           # gradient buckets already being all-reduced (data parallel)
        self.overlap_allreduce = True                # False: one blocking all-reduce in apply_gradients (tests)
        self.losses = torch.zeros(2, dtype=torch.float32, device=dev)
        # the two scalars a training step returns to its caller are known after the FORWARD pass: copied to pinned host
        # memory right there, with an event, so that exec_train_step waits for that copy only -- not for the backward pass,
        # Adam and the weight re-layouts behind it, which the next step's launches simply queue behind
        self._loss_host = torch.zeros(2, dtype=torch.float32).pin_memory() if torch.cuda.is_available() else None
        self._loss_event = torch.cuda.Event() if torch.cuda.is_available() else None
        self.export_routing = False                  # True: keep every relu / max-pool decision of the step (parity tests)
        # The big convolutions (filter bank, post-bank projection, and their data gradients) as three float16 MFMA products
        # of exactly split float32 operands (gemm16.py; float32 accuracy, 2-3x the f32-MFMA rate).  'train_f16x3': false
        # in the configuration keeps every GEMM on the f32-input MFMA kernels.
        self.f16x3 = bool(c.get('train_f16x3', True))
        self._g16 = {}
        self._w16_stream = torch.cuda.Stream(device=dev) if torch.cuda.is_available() else None    # float16 weight copies after Adam
        self._prep_wgrad = False                     # forward pass: build the filter-gradient operands of its activations early
        self.routing = {}

    def load_slots(self, ckpt):
        """Resume Adam state from a checkpoint dict (TF slot names <opt>/<var>/Adam, /Adam_1)."""
        torch = _torch()
        for n, (m, v) in self.adam_slots().items():
            km, kv = '{}/{}/Adam'.format(self.opt_scope, n), '{}/{}/Adam_1'.format(self.opt_scope, n)
            if km in ckpt and kv in ckpt:
                m.copy_(torch.from_numpy(np.ascontiguousarray(ckpt[km], dtype=np.float32)))
                v.copy_(torch.from_numpy(np.ascontiguousarray(ckpt[kv], dtype=np.float32)))

    def resume(self, ckpt):
        """Everything tf.train.Saver.restore brings back besides the weights: Adam slots and the step the bias
        correction counts from (the model's opt_state was already updated from the same checkpoint)."""
        self.load_slots(ckpt)
        self.step_count = int(self.dec.opt_state[self.opt_scope + '/global_step'])

    def slot_dict(self):
        """Adam slots + beta powers under TF's names, for save()."""
        c = self.cfg
        d = {}
        for n, (m, v) in self.adam_slots().items():
            d['{}/{}/Adam'.format(self.opt_scope, n)] = m.cpu().numpy()
            d['{}/{}/Adam_1'.format(self.opt_scope, n)] = v.cpu().numpy()
        d[self.opt_scope + '/beta1_power'] = np.float32(float(c['beta1']) ** (self.step_count + 1))
        d[self.opt_scope + '/beta2_power'] = np.float32(float(c['beta2']) ** (self.step_count + 1))
        return d

    # ---------------------------------------------------------------- helpers
    def g(self, name):
        """Gradient view (TF layout) of variable ``name``."""
        off, k, shape = self.offsets[name]
        return self.grad[off:off + k].view(shape)

    def w(self, name):
        return self.store.vars[name]

    def adam_slots(self):
        """name -> (m, v) views, for checkpointing (TF slot names <var>/Adam, <var>/Adam_1)."""
        out = {}
        for n, (off, k, shape) in self.offsets.items():
            out[n] = (self.m[off:off + k].view(shape), self.v[off:off + k].view(shape))
        return out

    # ---------------------------------------------------------------- split-float16 operand plans
    def _g16_plan(self, s, H, K, M, T_):
        """Operands of the stage's convolutions for vc_gemm16, or None where the shapes do not fit its tiles (then the
        f32-MFMA kernels take them).  The float16 copies follow the weights: rewritten (one vc_weights16 call per stage)
        whenever the store's version moved -- after every Adam step, after a restore."""
        if not self.f16x3 or K % 2 or K > 32 or H % 64 or M % T_ or modules.BANK_FILTERS != 128:
            return None
        pl = self._g16.get(s)
        if pl is None:
            b = s + '/CBHG/conv1d_banks'
            kern = [self.w(b + ('/conv1d' if k == 1 else '/num_{}/conv1d'.format(k)) + '/conv1d/kernel') for k in range(1, K + 1)]
            W1 = self.w(s + '/CBHG/conv1d_1/conv1d/kernel')                    # [3, 128 K, H]
            w16 = gemm16.Weights16(self.store.device)
            pl = {'w16': w16, 'version': None}
            pl['bank_fwd'] = gemm16.bank_forward_operands(w16, kern, H)
            pl['p1_dgrad'] = gemm16.conv_dgrad_operands(w16, W1)
            if H in (128, 256):                                                 # one pair, or a single-filter pair
                pl['p1_fwd'] = gemm16.conv_forward_operands(w16, W1)
                pl['bank_dgrad'] = gemm16.bank_dgrad_operands(w16, kern, H)
            self._g16[s] = pl
        if pl['version'] != self.store.version:
            pl['w16'].refresh()
            pl['version'] = self.store.version
            pl['event'] = None
        elif pl.get('event') is not None:
            # rewritten behind the last Adam step on a stream of its own (_refresh_w16_async): first use waits for it
            _torch().cuda.current_stream().wait_event(pl['event'])
            pl['event'] = None
        return pl

    def _refresh_w16_async(self):
        """After Adam: rewrite the float16 operand copies of every stage on their own stream, behind the update and beside
        whatever the main stream does next (the step's host read-back, the next step's encoder and prenet); the next
        forward pass waits for the event at its first vc_gemm16 launch of the stage."""
        torch = _torch()
        if not self._g16 or self._w16_stream is None:
            return
        ws = self._w16_stream
        ws.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ws):
            for pl in self._g16.values():
                pl['w16'].refresh()
                pl['version'] = self.store.version
                ev = torch.cuda.Event()
                ev.record(ws)
                pl['event'] = ev

    # ---------------------------------------------------------------- one stage forward
    def _stage_forward(self, s, X0, cin0, E, K, n_hw, n_out, seed_base):
        torch = _torch()
        H = E // 2
        N_, T_, Cp = X0.shape
        M = N_ * T_
        dev = X0.device
        st = self.store
        sv = {'X0': X0, 'cin0': cin0, 'E': E, 'K': K, 'n_hw': n_hw, 'n_out': n_out, 'M': M, 'T': T_, 'N': N_}
        f32 = _vc.VC_F32
        keep = self.keep if self.keep < 1.0 else 0.0

        def dense(X, ldx, cin_p, scope, units, act, seed=None, out_ld=None):
            bt, b = modules._prep_dense(st, scope, self.w(scope + '/kernel').shape[0], units)
            ldc = units if out_ld is None else out_ld
            out = torch.zeros((M, ldc), dtype=torch.float32, device=dev) if ldc != units else \
                torch.empty((M, ldc), dtype=torch.float32, device=dev)
            d = _vc.GemmDesc()
            d.dtype, d.mode, d.d_X = f32, _vc.GEMM_PLAIN, X.data_ptr()
            d.M, d.T, d.Cin, d.ldx, d.N, d.n_groups = M, T_, cin_p, ldx, units, 1
            g0 = d.groups[0]
            g0.d_Bt, g0.K, g0.taps, g0.pad_l, g0.c_off = bt.data_ptr(), cin_p, 1, 0, 0
            d.d_epi_shift = b.data_ptr()
            d.act = act
            d.d_C, d.ldc, d.out_f32 = out.data_ptr(), ldc, 1
            if seed is not None and keep > 0.0:
                d.drop_keep, d.drop_seed = keep, seed
            _vc.check(_lib().vc_conv_gemm(C.byref(d), _st()))
            return out

        # prenet (modules.py:274-295), dropout fused in the epilogue
        D1 = dense(X0, Cp, Cp, s + '/prenet/dense1', E, _vc.ACT_RELU, seed_base + 1)
        D2 = dense(D1, E, E, s + '/prenet/dense2', H, _vc.ACT_RELU, seed_base + 2)
        sv['D1'], sv['D2'] = D1, D2
        # conv banks: raw outputs, batch statistics (modules.py:144-166, is_training)
        CB = BANK_FILTERS * K
        b = s + '/CBHG/conv1d_banks'
        Zb = torch.empty((M, CB), dtype=torch.float32, device=dev)
        g16 = self._g16_plan(s, H, K, M, T_)
        sv['g16'] = g16
        if g16 is not None:
            d16, drs = gemm16.split16(D2, M, H, H, T_)
            gemm16.gemm16(d16, drs, M, T_, H, g16['bank_fwd'][0], Zb, CB, col_scale=g16['bank_fwd'][1])
            del d16, drs
        else:
            groups = []        # (float32 kernel layouts: built -- and re-laid out after every update -- only on this path)
            for k in range(1, K + 1):
                sub = b + ('/conv1d' if k == 1 else '/num_{}/conv1d'.format(k))
                groups.append((modules._prep_conv(st, sub, k, H, BANK_FILTERS), k * H, k, (k - 1) // 2, BANK_FILTERS * (k - 1)))
            gemm_launch(D2, M, T_, H, H, BANK_FILTERS, groups, Zb, CB, f32, out_f32=True)
        sb = _Ops.bn_stats(Zb, M, CB, self.w(b + '/bn/gamma'), self.w(b + '/bn/beta'),
                           self.w(b + '/bn/moving_mean'), self.w(b + '/bn/moving_variance'))
        if self._prep_wgrad and g16 is not None and M % 64 == 0 and self._side is not None:
            # the filter-gradient operands that depend on forward activations only (the bank input over its K tap shifts,
            # the projection input): built NOW on the side stream, under the rest of the forward pass and the recurrence,
            # instead of at the tail of the backward pass
            ws = self._side
            ws.wait_stream(torch.cuda.current_stream())
            for t_ in (D2, Zb, sb[0], sb[1]):
                t_.record_stream(ws)
            with torch.cuda.stream(ws):
                sv['XT'] = gemm16.transpose_split16(D2, M, H, H, T_, shift0=-(K // 2 - 1), n_shifts=K)
                sv['PT'] = gemm16.transpose_split16(Zb, M, CB, CB, T_, scale=sb[0], shift=sb[1], relu=1, pool=1)
        if self.export_routing:
            # parity tests: the relu / pool-winner decisions the backward pass will take on this tensor, for the oracle
            bits = torch.empty((M, CB), dtype=torch.uint8, device=dev)
            _vc.check(_lib().vc_bn_post_routing(_p(Zb), M, CB, CB, T_, _p(sb[0]), _p(sb[1]), _p(bits), _st()))
            self.routing[s] = {'banks': bits.view(N_, T_, CB),
                               # vc_relu_dropout_backward passes the gradient where the stored output is > 0
                               'prenet': ((D1 > 0).view(N_, T_, E), (D2 > 0).view(N_, T_, H)), 'highway': [None] * n_hw}
        # conv1d_1 on pool(relu(bn(Zb))) -- normalisation, relu and pool in the operand prologue
        p1 = s + '/CBHG/conv1d_1'
        Q1 = torch.empty((M, H), dtype=torch.float32, device=dev)
        if g16 is not None and 'p1_fwd' in g16:
            z16, zrs = gemm16.split16(Zb, M, CB, CB, T_, scale=sb[0], shift=sb[1], relu=1, pool=1)
            gemm16.gemm16(z16, zrs, M, T_, CB, g16['p1_fwd'][0], Q1, H, col_scale=g16['p1_fwd'][1])
            del z16, zrs
        else:
            gemm_launch(Zb, M, T_, CB, CB, H, [(modules._prep_conv(st, p1, 3, CB, H), 3 * CB, 3, 1, 0)], Q1, H, f32,
                        pro_scale=sb[0], pro_shift=sb[1], pro_relu=1, pro_pool=1, out_f32=True)
        s1 = _Ops.bn_stats(Q1, M, H, self.w(p1 + '/gamma'), self.w(p1 + '/beta'), self.w(p1 + '/moving_mean'),
                           self.w(p1 + '/moving_variance'))
        if self.export_routing:                    # vc_bn_backward mode 1: bn(Q1) > 0 (bit 0 of the same export)
            bits = torch.empty((M, H), dtype=torch.uint8, device=dev)
            _vc.check(_lib().vc_bn_post_routing(_p(Q1), M, H, H, T_, _p(s1[0]), _p(s1[1]), _p(bits), _st()))
            self.routing[s]['conv1d_1'] = (bits & 1).bool().view(N_, T_, H)
        p2 = s + '/CBHG/conv1d_2'
        Q2 = torch.empty((M, H), dtype=torch.float32, device=dev)
        gemm_launch(Q1, M, T_, H, H, H, [(modules._prep_conv(st, p2, 3, H, H), 3 * H, 3, 1, 0)], Q2, H, f32,
                    pro_scale=s1[0], pro_shift=s1[1], pro_relu=1, pro_pool=0, out_f32=True)
        s2 = _Ops.bn_stats(Q2, M, H, self.w(p2 + '/gamma'), self.w(p2 + '/beta'), self.w(p2 + '/moving_mean'),
                           self.w(p2 + '/moving_variance'))
        Y = torch.empty((M, H), dtype=torch.float32, device=dev)
        _vc.check(_lib().vc_affine_act(_p(Q2), _p(s2[0]), _p(s2[1]), 0, _p(D2), _p(Y), M * H, H, _st()))
        sv.update(Zb=Zb, sb=sb, Q1=Q1, s1=s1, Q2=Q2, s2=s2)
        # highways
        Ys = [Y]
        for i in range(n_hw):
            bt, bias = modules._prep_highway(st, s + '/CBHG/highwaynet_{}'.format(i), H)
            Yn = torch.empty((M, H), dtype=torch.float32, device=dev)
            gemm_launch(Ys[-1], M, T_, H, H, bt.shape[0], [(bt, H, 1, 0, 0)], Yn, H, f32,
                        mode=_vc.GEMM_HIGHWAY, epi_shift=bias)
            Ys.append(Yn)
        sv['Ys'] = Ys
        G = torch.empty((M, 2 * H), dtype=torch.float32, device=dev)
        if self.cfg.get('use_lstm', False):
            # bidirectional LSTM (modules.py:347-350), saved gates and cell states
            ls = s + '/CBHG/lstm'
            btx, bx, wh_fw, wh_bw = modules._prep_lstm(st, ls, H, H, True)
            xproj = torch.empty((M, 8 * H), dtype=torch.float32, device=dev)
            gemm_launch(Ys[-1], M, T_, H, H, 8 * H, [(btx, H, 1, 0, 0)], xproj, 8 * H, f32, epi_shift=bx, out_f32=True)
            gates = torch.empty((2, M, 4 * H), dtype=torch.float32, device=dev)
            cst = torch.empty((2, M, H), dtype=torch.float32, device=dev)
            _vc.check(_lib().vc_lstm_train_forward(_p(xproj), _p(wh_fw), _p(wh_bw), N_, T_, H, _p(G), _p(gates), _p(cst), _st()))
            sv.update(G=G, gates=gates, cst=cst, wh=(wh_fw, wh_bw), btx=btx, lstm=True)
        else:
            # bidirectional GRU with saved gates
            gs = s + '/CBHG/gru'
            btx, bx, wh_fw, wh_bw = modules._prep_gru(st, gs, H, H)
            xproj = torch.empty((M, 6 * H), dtype=torch.float32, device=dev)
            gemm_launch(Ys[-1], M, T_, H, H, 6 * H, [(btx, H, 1, 0, 0)], xproj, 6 * H, f32, epi_shift=bx, out_f32=True)
            gates = torch.empty((2, M, 3 * H), dtype=torch.float32, device=dev)
            rh = torch.empty((2, M, H), dtype=torch.float32, device=dev)
            _vc.check(_lib().vc_gru_train_forward(_p(xproj), _p(wh_fw), _p(wh_bw), N_, T_, H, _p(G), _p(gates), _p(rh), _st()))
            sv.update(G=G, gates=gates, rh=rh, wh=(wh_fw, wh_bw), btx=btx)
        # output projection, padded row stride so the next stage / backward get 16-byte rows
        y = dense(G, 2 * H, 2 * H, s + '/y_logits', n_out, _vc.ACT_NONE, None, out_ld=modules._pad8(n_out))
        sv['y'] = y
        return y, sv

    # ---------------------------------------------------------------- one stage backward
    def _dgrad_dense(self, dY, ld, Kp, W_tf, M, T_, out=None, R=None):
        """dX [M, Cin] = dY [M, Kp(ld)] @ W_tf[Cin, Cout]^T  (W_tf rows are the transposed operand)."""
        torch = _torch()
        cin, cout = W_tf.shape
        bt = W_tf
        if Kp != cout:
            bt = torch.zeros((cin, Kp), dtype=torch.float32, device=W_tf.device)
            bt[:, :cout] = W_tf
        bt = bt.contiguous()
        if out is None:
            out = torch.empty((M, cin), dtype=torch.float32, device=W_tf.device)
        gemm_launch(dY, M, T_, Kp, ld, cin, [(bt, Kp, 1, 0, 0)], out, out.shape[1], _vc.VC_F32,
                    R=R, ldr=(R.shape[1] if R is not None else 0), out_f32=True)
        return out

    def _dgrad_conv_weight(self, W_tf):
        """[k, Cin, Cout] -> transposed operand of the data-gradient conv: [Cin, k*Cout], taps flipped.  From the second
        step on this is the copy _refresh_conv_layouts wrote after the last update (valid while nothing else changed
        the weights: store.version)."""
        if getattr(self, '_layout_version', -1) == self.store.version:
            dg = self._dgrad_w.get(W_tf.data_ptr())
            if dg is not None:
                return dg
        k, cin, cout = W_tf.shape
        return W_tf.flip(0).permute(1, 0, 2).reshape(cin, k * cout).contiguous()

    def _stage_backward(self, s, sv, dY, need_dx):
        """dY: gradient w.r.t. the stage output, [M, pad8(n_out)] (padding columns zero).
        Fills the gradient arena; returns dX0 [M, Cp] if ``need_dx``."""
        torch = _torch()
        f32 = _vc.VC_F32
        M, T_, N_ = sv['M'], sv['T'], sv['N']
        E, K, n_hw, n_out = sv['E'], sv['K'], sv['n_hw'], sv['n_out']
        H = E // 2
        dev = dY.device
        ldy = dY.shape[1]
        inv_keep = 1.0 / self.keep if self.keep < 1.0 else 1.0

        # ---- output dense: y = G Wo + bo
        o = s + '/y_logits'
        _Ops.col_sum(dY, M, n_out, ldy, self.g(o + '/bias'))
        GT, ldg = _Ops.transpose(sv['G'], M, 2 * H, 2 * H, T_)
        dYT, ldyt = _Ops.transpose(dY, M, n_out, ldy, T_)
        _Ops.wgrad(GT, ldg, 2 * H, M, T_, dYT, ldyt, [(0, n_out, 1, 0, self.g(o + '/kernel'), n_out)])
        dG = self._dgrad_dense(dY, ldy, ldy, self.w(o + '/kernel'), M, T_)
        del GT, dYT

        # ---- recurrence
        if sv.get('lstm'):
            dYc = self._lstm_backward(s, sv, dG)
        else:
            dYc = self._gru_backward(s, sv, dG)
        del dG
        return self._stage_backward_front(s, sv, dYc, need_dx)

    def _lstm_backward(self, s, sv, dG):
        """BPTT of the bidirectional LSTM (use_lstm; modules.py:207-243) + its weight gradients; returns the gradient
        w.r.t. the last highway block's output."""
        torch = _torch()
        M, T_, N_ = sv['M'], sv['T'], sv['N']
        H = sv['E'] // 2
        dev = dG.device
        ls = s + '/CBHG/lstm'
        wh_fw, wh_bw = sv['wh']
        dpre = torch.empty((M, 8 * H), dtype=torch.float32, device=dev)
        whT_fw, whT_bw = wh_fw.t().contiguous(), wh_bw.t().contiguous()
        _vc.check(_lib().vc_lstm_backward(_p(dG), _p(sv['gates']), _p(sv['cst']), _p(whT_fw), _p(whT_bw), N_, T_, H, _p(dpre), _st()))
        dbx = torch.empty(8 * H, dtype=torch.float32, device=dev)
        _Ops.col_sum(dpre, M, 8 * H, 8 * H, dbx)
        dpT, ldp = _Ops.transpose(dpre, M, 8 * H, 8 * H, T_)
        YT, ldt = _Ops.transpose(sv['Ys'][-1], M, H, H, T_)
        grp_x = []
        for d, dn in enumerate(('fw', 'bw')):
            cell = '{}/bidirectional_rnn/{}/lstm_cell'.format(ls, dn)
            with _Ops.side(dbx):
                self.g(cell + '/bias').copy_(dbx[d * 4 * H:(d + 1) * 4 * H])
            grp_x.append((d * 4 * H, 4 * H, 1, 0, self.g(cell + '/kernel')[:H], 4 * H))          # x rows of the cell kernel
        _Ops.wgrad(YT, ldt, H, M, T_, dpT, ldp, grp_x)
        for d, dn in enumerate(('fw', 'bw')):
            cell = '{}/bidirectional_rnn/{}/lstm_cell'.format(ls, dn)
            Gd = sv['G'][:, d * H:(d + 1) * H]                                                  # view, ld = 2H
            HpT, ldh = _Ops.transpose(Gd, M, H, 2 * H, T_, row_shift=(-1 if d == 0 else 1))   # h_{prev}
            _Ops.wgrad(HpT, ldh, H, M, T_, dpT, ldp, [(d * 4 * H, 4 * H, 1, 0, self.g(cell + '/kernel')[H:], 4 * H)])
        return self._dgrad_dense(dpre, 8 * H, 8 * H, sv['btx'].t().contiguous(), M, T_)

    def _gru_backward(self, s, sv, dG):
        """BPTT of the bidirectional GRU + its weight gradients; returns the gradient w.r.t. the last highway block's
        output."""
        torch = _torch()
        M, T_, N_ = sv['M'], sv['T'], sv['N']
        H = sv['E'] // 2
        dev = dG.device
        gs = s + '/CBHG/gru'
        wh_fw, wh_bw = sv['wh']
        dpre = torch.empty((M, 6 * H), dtype=torch.float32, device=dev)
        whT_fw, whT_bw = wh_fw.t().contiguous(), wh_bw.t().contiguous()
        _vc.check(_lib().vc_gru_backward(_p(dG), _p(sv['G']), _p(sv['gates']), _p(wh_fw), _p(wh_bw), _p(whT_fw), _p(whT_bw),
                                         N_, T_, H, _p(dpre), _st()))
        # input gradient first (the critical path): dYn = dpre @ Btx  (Btx [6H, H] -> transposed operand [H, 6H]) ...
        dYn = self._dgrad_dense(dpre, 6 * H, 6 * H, sv['btx'].t().contiguous(), M, T_)
        # ... then the weight gradients, one stretch of the side stream
        Yn = sv['Ys'][-1]
        with _Ops.side(dpre, Yn, sv['G'], sv['rh'], small=True):
            inner, _Ops.side_stream, _Ops.side_stream2 = (_Ops.side_stream, _Ops.side_stream2), None, None    # already there: no re-entry
            try:
                dbx = torch.empty(6 * H, dtype=torch.float32, device=dev)
                _Ops.col_sum(dpre, M, 6 * H, 6 * H, dbx)
                dpT, ldp = _Ops.transpose(dpre, M, 6 * H, 6 * H, T_)
                YT, ldt = _Ops.transpose(Yn, M, H, H, T_)
                grp_x = []
                for d, dn in enumerate(('fw', 'bw')):
                    cell = '{}/bidirectional_rnn/{}/gru_cell'.format(gs, dn)
                    gk, ck = self.g(cell + '/gates/kernel'), self.g(cell + '/candidate/kernel')       # [2H,2H], [2H,H]
                    self.g(cell + '/gates/bias').copy_(dbx[d * 3 * H:d * 3 * H + 2 * H])
                    self.g(cell + '/candidate/bias').copy_(dbx[d * 3 * H + 2 * H:(d + 1) * 3 * H])
                    grp_x.append((d * 3 * H, 2 * H, 1, 0, gk[:H], 2 * H))                              # x rows of gates/kernel
                    grp_x.append((d * 3 * H + 2 * H, H, 1, 0, ck[:H], H))                              # x rows of candidate/kernel
                _Ops.wgrad(YT, ldt, H, M, T_, dpT, ldp, grp_x)
                for d, dn in enumerate(('fw', 'bw')):
                    cell = '{}/bidirectional_rnn/{}/gru_cell'.format(gs, dn)
                    gk, ck = self.g(cell + '/gates/kernel'), self.g(cell + '/candidate/kernel')
                    Gd = sv['G'][:, d * H:(d + 1) * H]                                                  # view, ld = 2H
                    HpT, ldh = _Ops.transpose(Gd, M, H, 2 * H, T_, row_shift=(-1 if d == 0 else 1))   # h_{prev}
                    _Ops.wgrad(HpT, ldh, H, M, T_, dpT, ldp, [(d * 3 * H, 2 * H, 1, 0, gk[H:], 2 * H)])
                    RT, ldr = _Ops.transpose(sv['rh'][d], M, H, H, T_)
                    _Ops.wgrad(RT, ldr, H, M, T_, dpT, ldp, [(d * 3 * H + 2 * H, H, 1, 0, ck[H:], H)])
                del dpT, YT, HpT, RT, dbx
            finally:
                _Ops.side_stream, _Ops.side_stream2 = inner
        return dYn

    def _stage_backward_front(self, s, sv, dYc, need_dx):
        """Everything in front of the recurrence: highway blocks, projections, filter bank, prenet."""
        torch = _torch()
        f32 = _vc.VC_F32
        M, T_, N_ = sv['M'], sv['T'], sv['N']
        E, K, n_hw, n_out = sv['E'], sv['K'], sv['n_hw'], sv['n_out']
        H = E // 2
        dev = dYc.device
        inv_keep = 1.0 / self.keep if self.keep < 1.0 else 1.0

        # ---- highways (reverse)
        side_jobs = []
        for i in range(n_hw - 1, -1, -1):
            hs = s + '/CBHG/highwaynet_{}'.format(i)
            bt, bias = modules._prep_highway(self.store, hs, H)
            NP = bt.shape[0]
            Xi = sv['Ys'][i]
            pre = torch.empty((M, NP), dtype=torch.float32, device=dev)
            gemm_launch(Xi, M, T_, H, H, NP, [(bt, H, 1, 0, 0)], pre, NP, f32, epi_shift=bias, out_f32=True)
            if self.export_routing:        # vc_highway_backward: dense1's relu passes where its re-computed pre-activation > 0
                j = torch.arange(H, device=dev)            # unit j sits at column 64 (j / 32) + j % 32 of the paired layout
                self.routing[s]['highway'][i] = (pre[:, 64 * (j // 32) + j % 32] > 0).view(N_, T_, H)
            dp = torch.empty((M, NP), dtype=torch.float32, device=dev)
            dXd = torch.empty((M, H), dtype=torch.float32, device=dev)
            _vc.check(_lib().vc_highway_backward(_p(pre), NP, _p(Xi), _p(dYc), M, H, _p(dp), _p(dXd), _st()))
            # dX = dp @ bt (paired) + direct path: the critical path goes on first ...
            dYc = self._dgrad_dense(dp, NP, NP, bt.t().contiguous(), M, T_, R=dXd)
            side_jobs.append((hs, Xi, dp, NP))
            del pre, dp
        def flush_highway_side():
            # ... then everything only the optimiser needs, for ALL the blocks as one stretch of the side stream (one
            # cross-stream dependency for the chain instead of several per block: the main stream ran at most ~3 blocks
            # ahead of the side stream's backlog otherwise and sat idle for the rest)
            if side_jobs:
                with _Ops.side(*[t for job in side_jobs for t in job[1:3]], small=True):
                    inner, _Ops.side_stream, _Ops.side_stream2 = (_Ops.side_stream, _Ops.side_stream2), None, None    # already there: no re-entry
                    try:
                        for hs, Xi, dp, NP in side_jobs:
                            g1, g2 = self.g(hs + '/dense1/kernel'), self.g(hs + '/dense2/kernel')
                            b1, b2 = self.g(hs + '/dense1/bias'), self.g(hs + '/dense2/bias')
                            grp = []
                            for q in range((H + 31) // 32):
                                n = min(32, H - 32 * q)
                                grp.append((64 * q, n, 1, 0, g1[:, 32 * q:], H))
                                grp.append((64 * q + 32, n, 1, 0, g2[:, 32 * q:], H))
                            XT, ldx = _Ops.transpose(Xi, M, H, H, T_)
                            dpT, ldp = _Ops.transpose(dp, M, NP, NP, T_)
                            dbp = torch.empty(NP, dtype=torch.float32, device=dev)
                            _Ops.col_sum(dp, M, NP, NP, dbp)
                            if H % 32 == 0:    # paired order [32 x dense1 | 32 x dense2] per 32 units -> the two bias vectors: two strided copies
                                pr = dbp.view(H // 32, 2, 32)
                                b1.view(H // 32, 32).copy_(pr[:, 0])
                                b2.view(H // 32, 32).copy_(pr[:, 1])
                            else:
                                for q in range((H + 31) // 32):
                                    n = min(32, H - 32 * q)
                                    b1[32 * q:32 * q + n].copy_(dbp[64 * q:64 * q + n])
                                    b2[32 * q:32 * q + n].copy_(dbp[64 * q + 32:64 * q + 32 + n])
                            for j in range(0, len(grp), 32):
                                _Ops.wgrad(XT, ldx, H, M, T_, dpT, ldp, grp[j:j + 32])
                            del XT, dpT, dbp
                    finally:
                        _Ops.side_stream, _Ops.side_stream2 = inner
                side_jobs.clear()

        # ---- Y0 = bn2(Q2) + D2
        p2 = s + '/CBHG/conv1d_2'
        dQ2 = _Ops.bn_backward(dYc, sv['Q2'], M, H, T_, self.w(p2 + '/gamma'), sv['s2'], 0,
                               self.g(p2 + '/gamma'), self.g(p2 + '/beta'))
        dD2_res = dYc
        # conv1d_2 on relu(bn1(Q1))
        W2c = self.w(p2 + '/conv1d/kernel')
        R1T, ldr1 = _Ops.transpose(sv['Q1'], M, H, H, T_, scale=sv['s1'][0], shift=sv['s1'][1], relu=1)
        dQ2T, ldq2 = _Ops.transpose(dQ2, M, H, H, T_)
        _Ops.wgrad(R1T, ldr1, H, M, T_, dQ2T, ldq2, [(0, H, 3, -1, self.g(p2 + '/conv1d/kernel'), H)])
        dR1 = torch.empty((M, H), dtype=torch.float32, device=dev)
        gemm_launch(dQ2, M, T_, H, H, H, [(self._dgrad_conv_weight(W2c), 3 * H, 3, 1, 0)], dR1, H, f32, out_f32=True)
        del R1T, dQ2T
        p1 = s + '/CBHG/conv1d_1'
        dQ1 = _Ops.bn_backward(dR1, sv['Q1'], M, H, T_, self.w(p1 + '/gamma'), sv['s1'], 1,
                               self.g(p1 + '/gamma'), self.g(p1 + '/beta'))
        # conv1d_1 on pool(relu(bnb(Zb)))
        CB = BANK_FILTERS * K
        W1c = self.w(p1 + '/conv1d/kernel')
        g16 = sv.get('g16')
        wg16 = g16 is not None and M % 64 == 0      # filter gradients on split-float16 operands too (gemm16.py)
        if wg16:
            with _Ops.side(sv['Zb'], sv['sb'][0], sv['sb'][1], dQ1):
                PT, rsP = sv.pop('PT') if 'PT' in sv else gemm16.transpose_split16(
                    sv['Zb'], M, CB, CB, T_, scale=sv['sb'][0], shift=sv['sb'][1], relu=1, pool=1)
                QT, rsQ = gemm16.transpose_split16(dQ1, M, H, H, T_, shift0=-1, n_shifts=3)
                gemm16.conv3_wgrad(QT, rsQ, PT, rsP, H, CB, M, self.g(p1 + '/conv1d/kernel'),
                                   splits=max(1, min(8, 256 // (16 * ((3 * H + 255) // 256)))) if _Ops.splits_allowed else 1)
                del PT, rsP, QT, rsQ
        else:
            PT, ldpt = _Ops.transpose(sv['Zb'], M, CB, CB, T_, scale=sv['sb'][0], shift=sv['sb'][1], relu=1, pool=1)
            dQ1T, ldq1 = _Ops.transpose(dQ1, M, H, H, T_)
            _Ops.wgrad(PT, ldpt, CB, M, T_, dQ1T, ldq1, [(0, H, 3, -1, self.g(p1 + '/conv1d/kernel'), H)])
            del PT, dQ1T
        dP = torch.empty((M, CB), dtype=torch.float32, device=dev)
        if g16 is not None:
            q16, qrs = gemm16.split16(dQ1, M, H, H, T_)
            gemm16.gemm16(q16, qrs, M, T_, H, g16['p1_dgrad'][0], dP, CB, col_scale=g16['p1_dgrad'][1])
            del q16, qrs
        else:
            gemm_launch(dQ1, M, T_, H, H, CB, [(self._dgrad_conv_weight(W1c), 3 * H, 3, 1, 0)], dP, CB, f32, out_f32=True)
        b = s + '/CBHG/conv1d_banks'
        dZb = _Ops.bn_backward(dP, sv['Zb'], M, CB, T_, self.w(b + '/bn/gamma'), sv['sb'], 2,
                               self.g(b + '/bn/gamma'), self.g(b + '/bn/beta'))
        del dP
        # banks: filter gradients (one grouped launch) and data gradient accumulated bank by bank
        kg = [self.g(b + ('/conv1d' if k == 1 else '/num_{}/conv1d'.format(k)) + '/conv1d/kernel') for k in range(1, K + 1)]
        if wg16:
            with _Ops.side(sv['D2'], dZb):
                XT, rsX = sv.pop('XT') if 'XT' in sv else gemm16.transpose_split16(
                    sv['D2'], M, H, H, T_, shift0=-(K // 2 - 1), n_shifts=K)
                ZT, rsZ = gemm16.transpose_split16(dZb, M, CB, CB, T_)
                gemm16.bank_wgrad(XT, rsX, ZT, rsZ, H, K, M, kg, self.grad, splits=2 if _Ops.splits_allowed else 1)      # (1 .. 8 ranges measure the same: ab_gemm16.log)
                del XT, rsX, ZT, rsZ
        else:
            D2T, ldd2 = _Ops.transpose(sv['D2'], M, H, H, T_)
            dZbT, ldzb = _Ops.transpose(dZb, M, CB, CB, T_)
            grp = [(BANK_FILTERS * (k - 1), BANK_FILTERS, k, -((k - 1) // 2), kg[k - 1], BANK_FILTERS) for k in range(1, K + 1)]
            _Ops.wgrad(D2T, ldd2, H, M, T_, dZbT, ldzb, grp)
            del D2T, dZbT
        # data gradient of the banks: sum over k of conv(dZb[:, bank k], W_k^T flipped) + the residual path -- ONE launch
        # whose groups accumulate in the same tile (vc_gemm_desc.sum_groups), not K short-K launches chained through dD2
        grp = []
        if g16 is not None and 'bank_dgrad' in g16:
            # one launch walking bank after bank (its own taps and padding each) into the same accumulators, K split over
            # workgroups in a fixed order; dD2 starts as the residual path's gradient
            z16, zrs = gemm16.split16(dZb, M, CB, CB, T_)
            dD2 = dD2_res.clone()
            gemm16.gemm16(z16, zrs, M, T_, CB, g16['bank_dgrad'][0], dD2, H, col_scale=g16['bank_dgrad'][1], ragged=True,
                          accumulate=True)
            del z16, zrs
        else:
            for k in range(1, K + 1):
                sub = b + ('/conv1d' if k == 1 else '/num_{}/conv1d'.format(k))
                Wk = self.w(sub + '/conv1d/kernel')                                      # [k, H, 128]
                grp.append((self._dgrad_conv_weight(Wk), k * BANK_FILTERS, k, k - 1 - (k - 1) // 2, BANK_FILTERS * (k - 1)))
        if grp and M >= 128 and BANK_FILTERS % 32 == 0:
            # (M/128 x H/128 tiles alone would leave most of the chip idle: the banks are dealt to S blocks per tile
            # whose partial sums are added to dD2, which starts as the residual path's gradient)
            tiles = ((M + 127) // 128) * ((H + 127) // 128)
            S = 1
            while _Ops.splits_allowed and tiles * S < 2048 and 4 * S <= K + 1 and S < 16:        # several rounds of resident blocks: short tail
                S *= 2
            if S > 1:
                dD2 = dD2_res.clone()
                gemm_launch(dZb, M, T_, BANK_FILTERS, CB, H, grp, dD2, H, f32, out_f32=True, sum_groups=S)
            else:
                dD2 = torch.empty((M, H), dtype=torch.float32, device=dev)
                gemm_launch(dZb, M, T_, BANK_FILTERS, CB, H, grp, dD2, H, f32, R=dD2_res, ldr=H, out_f32=True, sum_groups=1)
        elif grp:                              # shapes below the convolution kernel's tile: one launch per bank, chained
            dD2 = dD2_res.clone()
            for k, (wk, Kk, taps, pad, off) in enumerate(grp, 1):
                gemm_launch(dZb[:, off:], M, T_, BANK_FILTERS, CB, H, [(wk, Kk, taps, pad, 0)], dD2, H, f32, R=dD2, ldr=H,
                            out_f32=True)
        del dZb

        flush_highway_side()
        # ---- prenet
        pn = s + '/prenet'
        D1, D2, X0 = sv['D1'], sv['D2'], sv['X0']
        Cp = X0.shape[2]
        cin0 = sv['cin0']
        dZ2 = torch.empty((M, H), dtype=torch.float32, device=dev)
        _vc.check(_lib().vc_relu_dropout_backward(_p(dD2), _p(D2), inv_keep, _p(dZ2), M * H, _st()))
        _Ops.col_sum(dZ2, M, H, H, self.g(pn + '/dense2/bias'))
        D1T, ld1 = _Ops.transpose(D1, M, E, E, T_)
        dZ2T, ldz2 = _Ops.transpose(dZ2, M, H, H, T_)
        _Ops.wgrad(D1T, ld1, E, M, T_, dZ2T, ldz2, [(0, H, 1, 0, self.g(pn + '/dense2/kernel'), H)])
        dD1 = self._dgrad_dense(dZ2, H, H, self.w(pn + '/dense2/kernel'), M, T_)
        del D1T, dZ2T
        dZ1 = torch.empty((M, E), dtype=torch.float32, device=dev)
        _vc.check(_lib().vc_relu_dropout_backward(_p(dD1), _p(D1), inv_keep, _p(dZ1), M * E, _st()))
        _Ops.col_sum(dZ1, M, E, E, self.g(pn + '/dense1/bias'))
        X0v = X0.view(M, Cp)
        X0T, ldx0 = _Ops.transpose(X0v, M, cin0, Cp, T_)
        dZ1T, ldz1 = _Ops.transpose(dZ1, M, E, E, T_)
        _Ops.wgrad(X0T, ldx0, cin0, M, T_, dZ1T, ldz1, [(0, E, 1, 0, self.g(pn + '/dense1/kernel'), E)])
        dX0 = None
        if need_dx:
            W1 = self.w(pn + '/dense1/kernel')                                       # [cin0, E]
            W1p = torch.zeros((Cp, E), dtype=torch.float32, device=dev)
            W1p[:cin0] = W1
            dX0 = self._dgrad_dense(dZ1, E, E, W1p, M, T_)
        return dX0

    def losses_on_host(self):
        """The last forward pass's [mel_loss, stft_loss] as numpy float32, waiting for nothing but their own copy."""
        if self._loss_host is None:
            return self.losses.cpu().numpy()
        self._loss_event.synchronize()
        return self._loss_host.numpy().copy()

    def _join_side(self):
        """The current stream waits for everything this trainer issued to its side stream."""
        if self._side is not None:
            _torch().cuda.current_stream().wait_stream(self._side)
            _torch().cuda.current_stream().wait_stream(self._side2)

    def _drain_pending(self):
        """Buckets of an earlier forward_backward that no apply_gradients consumed (gradient accumulation, a retry
        after an exception): wait for them before the arena is zeroed and written again."""
        pending, self._pending = self._pending, []
        for _, _, work in pending:
            work.wait()
        self._join_side()

    def _slice_of(self, prefix):
        """[lo, hi) of the arena covered by the variables under ``prefix`` (contiguous: creation order)."""
        offs = [self.offsets[n] for n in self.names if n.startswith(prefix)]
        lo = min(o for o, _, _ in offs)
        hi = max(o + k for o, k, _ in offs)
        assert hi - lo == sum(k for _, k, _ in offs), 'variables of %s are not contiguous in the arena' % prefix
        return lo, hi

    def _start_allreduce(self, lo, hi):
        """Data parallel: start summing grad[lo:hi] over the ranks NOW (asynchronously: the process group runs it on its
        own stream behind everything already queued on this one), so that it travels while the rest of the backward
        pass computes.  apply_gradients() waits for the buckets before Adam reads the arena."""
        torch = _torch()
        if not self.overlap_allreduce or not (torch.distributed.is_available() and torch.distributed.is_initialized()) \
                or torch.distributed.get_world_size() < 2:
            return
        if _Ops.side_stream is not None and _Ops.side_stream2 is not None:
            _Ops.side_stream.wait_stream(_Ops.side_stream2)            # the small weight gradients of the bucket too
        with _Ops.side():        # behind the main stream's queue AND the weight gradients issued to the side streams so far
            self._pending.append((lo, hi, torch.distributed.all_reduce(self.grad[lo:hi], op=torch.distributed.ReduceOp.SUM,
                                                                        async_op=True)))

    def _complete_exchange(self):
        """Data parallel: after this every element of the gradient arena has been summed over the ranks exactly once
        and the current stream may read it.  Buckets forward_backward already put on the wire are waited for; every
        stretch of [0, total) they do not cover is all-reduced here (for the encoder trainer, which starts no bucket,
        that is the whole arena in one call).  Whether there is anything to exchange is the process group's business,
        not the caller's: ``world`` in apply_gradients only scales Adam."""
        torch = _torch()
        self._join_side()                             # weight gradients still running on the side stream
        dist_world = torch.distributed.get_world_size() if (torch.distributed.is_available() and
                                                             torch.distributed.is_initialized()) else 1
        if dist_world > 1 or self._pending:
            pending, self._pending = self._pending, []
            covered = sorted((lo, hi) for lo, hi, _ in pending)
            pos = 0
            for lo, hi in covered + [(self.total, self.total)]:
                if lo > pos and dist_world > 1:              # a stretch no bucket covered
                    torch.distributed.all_reduce(self.grad[pos:lo], op=torch.distributed.ReduceOp.SUM)
                pos = max(pos, hi)
            for _, _, work in pending:
                work.wait()

    def apply_gradients(self, world=1):
        """All-reduce (data parallel), Adam, bookkeeping.  decoder.py:236-246 / encoder.py:171-181.  Buckets that
        forward_backward already put on the wire are waited for; whatever they do not cover is summed here."""
        torch = _torch()
        c = self.cfg
        self._complete_exchange()
        self.step_count += 1
        t = self.step_count
        lr = float(self.dec.opt_state[self.opt_scope + '/learning_rate'])
        b1, b2, eps = float(c['beta1']), float(c['beta2']), float(c['epsilon'])
        lr_t = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
        _vc.check(_lib().vc_adam_step(_p(self.flat), _p(self.grad), _p(self.m), _p(self.v), self.total, lr_t, b1, b2, eps,
                                      1.0 / world, _st()))
        if getattr(self, '_layout_tab', None) is None:
            self._conv_seen = {k_: v_ for k_, v_ in self.store._cache.items() if isinstance(k_, tuple) and k_[0] == 'conv'
                               and str(k_[1]).startswith(self.dec._scope + '/')}
        self.store.invalidate(self.dec._scope)    # this model's kernel-layout copies are stale now (a frozen encoder's in the same store are not) ...
        self._refresh_conv_layouts()              # ... except the convolutions', rewritten in place by one launch
        self._refresh_w16_async()
        self.dec.opt_state[self.opt_scope + '/global_step'] = np.int32(t)
        return t

    def _refresh_conv_layouts(self):
        """The forward ([cout, k*cin]) and data-gradient ([cin, k*cout], taps reversed) copies of every convolution
        kernel, rewritten from the updated weights by ONE launch (vc_weight_layouts) instead of ~200 torch transposes
        and flips per step.  The table is built after the first step from what the forward pass cached."""
        torch = _torch()
        if getattr(self, '_layout_tab', None) is None:
            items, keep, dgrad = [], {}, {}
            for key, bt in self._conv_seen.items():
                W = self.store.vars.get(key[1] + '/conv1d/kernel')
                if W is None or W.dim() != 3 or bt.dtype != torch.float32:
                    continue
                k, cin, cout = W.shape
                if tuple(bt.shape) != (cout, k * cin) or not bt.is_contiguous():
                    continue
                dg = torch.empty((cin, k * cout), dtype=torch.float32, device=W.device)
                keep[key] = bt
                dgrad[W.data_ptr()] = dg
                items.append(_vc.LayoutItem(W.data_ptr(), bt.data_ptr(), k, cin, cout, 0))
                items.append(_vc.LayoutItem(W.data_ptr(), dg.data_ptr(), k, cin, cout, 1))
            if not items:
                return
            arr = (_vc.LayoutItem * len(items))(*items)
            tab = torch.frombuffer(bytearray(arr), dtype=torch.uint8).to(self.store.device)
            self._layout_tab, self._conv_keep, self._dgrad_w = (tab, len(items)), keep, dgrad
        tab, n = self._layout_tab
        _vc.check(_lib().vc_weight_layouts(_p(tab), n, _st()))
        self.store._cache.update(self._conv_keep)
        self._layout_version = self.store.version


class DecoderTrainer(StageTrainer):
    """One training step of ``decoder_specs`` (decoder.py:185-263, 327-345)."""

    opt_scope = 'dec_opt'

    # ---------------------------------------------------------------- whole step
    def forward_backward(self, x, target_mel, target_stft, backward=True):
        """Forward + backward of one batch.  Returns the device tensor [mel_loss, stft_loss].
        ``backward=False`` is the reference's exec_calc_metrics on a training graph (decoder.py:349-376): the
        train-mode forward (dropout, batch statistics, moving averages move: updates_collections=None) and the two
        losses; no gradient fill, no backward, no all-reduce."""
        torch = _torch()
        dec, c = self.dec, self.cfg
        self._drain_pending()
        if c['loss_type'] not in ('sum', 'log'):
            raise Exception('- ERROR, _build_loss, loss_type not understood.')
        n_in = c['input_shape'][-1]
        pad = modules._pad8(n_in)
        if dec.encoder is not None:
            ppg = dec.encoder.forward(x, ppg_pad_to=pad, ppg_dtype=torch.float32)['ppg']
        else:
            ppg = torch.zeros((x.shape[0], x.shape[1], pad), dtype=torch.float32, device=x.device)
            ppg[:, :, :n_in] = x
        self.last_ppg = ppg
        M = ppg.shape[0] * ppg.shape[1]
        if backward:
            _vc.check(_lib().vc_fill(_p(self.grad), 0.0, self.total, _st()))     # wgrad accumulates with atomics
        sd1, sd2 = c['steps_v'][0], c['steps_v'][1]
        seed = self.seed + 1000 * self.step_count
        self._prep_wgrad = bool(backward)
        with modules.variable_store(self.store):
            s1, s2 = dec._scope + '/step1', dec._scope + '/step2'
            y1, sv1 = self._stage_forward(s1, ppg, n_in, dec._E[0], sd1['num_conv_banks'], sd1['num_highwaynet_blocks'],
                                          sd1['n_output'], seed)
            n1 = sd1['n_output']
            x2 = y1.view(ppg.shape[0], ppg.shape[1], y1.shape[1])
            f_mel = 1.0
            if c.get('use_target_mel_step2', False):          # decoder.py:152: teacher-forced stage-2 input
                f_mel = float(dec.f_mel_pred)
                ld1 = y1.shape[1]
                x2 = torch.zeros_like(x2) if ld1 != n1 else torch.empty_like(x2)
                _vc.check(_lib().vc_axpby(_p(y1), ld1, f_mel, _p(target_mel), n1, 1.0 - f_mel, _p(x2), ld1, M, n1, _st()))
            y2, sv2 = self._stage_forward(s2, x2, n1, dec._E[1], sd2['num_conv_banks'], sd2['num_highwaynet_blocks'],
                                          sd2['n_output'], seed + 10)
            n2 = sd2['n_output']
            self.y_mel = y1[:, :n1].contiguous() if y1.shape[1] != n1 else y1
            self.y_stft = y2[:, :n2].contiguous() if y2.shape[1] != n2 else y2
            # losses + output gradients (decoder.py:187-195)
            dY1 = torch.zeros_like(y1) if backward else None
            dY2 = torch.zeros_like(y2) if backward else None
            wm, ws = float(c['mel_loss_weight']), float(c['stft_loss_weight'])
            _vc.check(_lib().vc_mse_loss(_p(self.y_mel), _p(target_mel), M * n1, wm, _p(dY1), n1, y1.shape[1],
                                         _p(self.losses[0:1]), _p(self.loss_ws), _st()))
            _vc.check(_lib().vc_mse_loss(_p(self.y_stft), _p(target_stft), M * n2, ws, _p(dY2), n2, y2.shape[1],
                                         _p(self.losses[1:2]), _p(self.loss_ws), _st()))
            if self._loss_host is not None:
                self._loss_host.copy_(self.losses, non_blocking=True)
                self._loss_event.record()
            if not backward:
                return self.losses
            if c['loss_type'] == 'log':          # d log(L) = dL / L  (host round trip for the two scalars)
                lm, ls = (float(v) for v in self.losses.cpu())
                dY1.mul_(1.0 / lm)
                dY2.mul_(1.0 / ls)
            _Ops.side_stream, _Ops.side_stream2 = self._side, self._side2    # weight gradients leave the critical path (see _Ops.side)
            try:
                dX2 = self._stage_backward(s2, sv2, dY2, need_dx=True)
                del sv2
                # stage 2's gradients (22.5 M of the 33.2 M floats) are final once its weight gradients have run: their
                # all-reduce is queued behind those and travels under stage 1's backward
                self._start_allreduce(*self._slice_of(s2 + '/'))
                # y_mel feeds step 2 (decoder.py:155; through the blend with weight f_mel_pred when teacher-forced, :152)
                _vc.check(_lib().vc_axpby(_p(dY1), dY1.shape[1], 1.0, _p(dX2), dX2.shape[1], f_mel, _p(dY1), dY1.shape[1], M,
                                          dY1.shape[1], _st()))
                self._stage_backward(s1, sv1, dY1, need_dx=False)
                self._start_allreduce(*self._slice_of(s1 + '/'))
            finally:
                _Ops.side_stream = _Ops.side_stream2 = None
                self._join_side()                     # whoever reads the gradient arena next finds it complete (also
                                                      # after an exception: nothing keeps writing the arena)
        return self.losses


class EncoderTrainer(StageTrainer):
    """One training step of ``encoder_spec_phn`` (encoder.py:134-194, 256-270): train-mode forward of
    the single prenet -> CBHG -> dense(n_output) stage, softmax cross-entropy with float labels,
    accuracy / mse metrics, backward, Adam."""

    opt_scope = 'opt'

    def forward_backward(self, x, target, backward=True):
        """Returns the device tensor [loss, acc, mse]."""
        torch = _torch()
        enc, c = self.dec, self.cfg
        self._drain_pending()
        N_, T_, Cx = x.shape
        M = N_ * T_
        n_out = c['n_output']
        seed = self.seed + 1000 * self.step_count
        if backward:
            _vc.check(_lib().vc_fill(_p(self.grad), 0.0, self.total, _st()))
        with modules.variable_store(self.store):
            y, sv = self._stage_forward(enc._scope, x, Cx, enc._embed_size, c['num_conv_banks'],
                                        c['num_highwaynet_blocks'], n_out, seed)
            self.y_logits = y
            dY = torch.zeros_like(y) if backward else None
            out3 = torch.empty(3, dtype=torch.float32, device=x.device)
            ws = torch.empty(3 * M, dtype=torch.float32, device=x.device)
            _vc.check(_lib().vc_softmax_ce(_p(y), _p(target), M, n_out, y.shape[1], _p(dY), y.shape[1], _p(out3), _p(ws), _st()))
            if backward:
                _Ops.side_stream, _Ops.side_stream2 = self._side, self._side2
                try:
                    self._stage_backward(enc._scope, sv, dY, need_dx=False)
                finally:
                    _Ops.side_stream = _Ops.side_stream2 = None
                    self._join_side()
        return out3
