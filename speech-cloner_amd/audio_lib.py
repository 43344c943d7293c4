"""Signal front-end with the reference's ``audio_lib`` surface, computed on the MI355X.

Mirrors /root/reference/audio_lib.py: same function names, argument order, defaults and
return types for the functions on the conversion hot path:

  calc_MFCC_input(y, ...) -> (MFCC [F, n_mfcc*(1|2)], M_dB [F, n_mels], P_dB [F, 1+n_fft//2])
                              np.float32, time-major, F = 1 + len(y)//hop_length
                              (audio_lib.py:89-244)
  calc_preemphasis / calc_inv_preemphasis                      (audio_lib.py:12-47)
  calc_PHN_target                                              (audio_lib.py:51-85)

plus a batched entry point the reference does not have (its per-utterance call is a
special case of it):

  calc_MFCC_input_batch(wav [B, L], lens=None, ...) -> three torch.cuda tensors [B, Fmax, C]

All arithmetic runs in hand-written HIP kernels (csrc/vc_frontend.hip) through the C ABI
``vc_frontend_f32`` (include/vc_hip.h); torch only owns the device buffers.  There is no
CPU path: without the native library or a GPU the calls raise.
"""
import ctypes as C

import numpy as np

import _vc

_PLANS = {}


class _Plan:
    def __init__(self, cfg, window):
        from scipy import signal
        lib = _vc.lib()
        self.cfg = cfg
        if isinstance(window, str) or isinstance(window, tuple):
            w = signal.get_window(window, cfg.win_length, fftbins=True)
        else:
            w = np.asarray(window, dtype=np.float64)
            if w.shape != (cfg.win_length,):
                raise ValueError(' - ERROR, window array must have win_length samples')
        self._w = np.ascontiguousarray(w, dtype=np.float64)
        h = C.c_void_p()
        _vc.check(lib.vc_frontend_plan_create(C.byref(cfg), _vc.ptr(self._w), C.byref(h)))
        self.handle = h
        self.mfcc_width = lib.vc_frontend_mfcc_width(h)
        self.n_bins = lib.vc_frontend_power_width(h)
        self._ws = None

    def workspace(self, batch, max_samples, device):
        import torch
        need = _vc.lib().vc_frontend_workspace_bytes(self.handle, batch, max_samples)
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws


def _get_plan(sr, pre_emphasis, hop_length, win_length, n_mels, n_mfcc, n_fft, window,
              mfcc_normaleze_first_mfcc, mfcc_norm_factor, calc_mfcc_derivate,
              M_dB_norm_factor, P_dB_norm_factor, mean_abs_amp_norm, clip_output):
    if n_fft is None:
        n_fft = win_length
    wkey = window if isinstance(window, (str, tuple)) else np.asarray(window).tobytes()
    key = (sr, pre_emphasis, hop_length, win_length, n_mels, n_mfcc, n_fft, wkey,
           bool(mfcc_normaleze_first_mfcc), mfcc_norm_factor, bool(calc_mfcc_derivate),
           M_dB_norm_factor, P_dB_norm_factor, mean_abs_amp_norm, bool(clip_output))
    p = _PLANS.get(key)
    if p is None:
        cfg = _vc.FrontendCfg(int(sr), int(hop_length), int(win_length), int(n_fft), int(n_mels),
                              int(n_mfcc), float(pre_emphasis), float(mean_abs_amp_norm),
                              float(mfcc_norm_factor), float(M_dB_norm_factor),
                              float(P_dB_norm_factor), int(bool(mfcc_normaleze_first_mfcc)),
                              int(bool(calc_mfcc_derivate)), int(bool(clip_output)))
        p = _Plan(cfg, window)
        _PLANS[key] = p
    return p


def calc_MFCC_input_batch(wav, lens=None,
                          sr=16000,
                          pre_emphasis=0.97,
                          hop_length=40,
                          win_length=400,
                          n_mels=128,
                          n_mfcc=40,
                          n_fft=None,
                          window='hann',
                          mfcc_normaleze_first_mfcc=True,
                          mfcc_norm_factor=0.01,
                          calc_mfcc_derivate=False,
                          M_dB_norm_factor=0.01,
                          P_dB_norm_factor=0.01,
                          mean_abs_amp_norm=0.003,
                          clip_output=True,
                          out=None,
                          stage_mask=7):
    """Batched calc_MFCC_input on the GPU.

    wav  : float32 [B, L] torch.cuda tensor (or numpy array, uploaded).
    lens : optional per-utterance sample counts (sequence / int32 tensor); None = all L.
    out  : optional (mfcc, mel, pow) preallocated cuda tensors to write into.
    stage_mask : measurement hook (bench.py): subset of the three launches, see vc_hip.h.
    Returns (MFCC [B, Fmax, W], M_dB [B, Fmax, n_mels], P_dB [B, Fmax, 1+n_fft//2]) cuda float32
    with Fmax = 1 + L//hop_length; rows beyond an utterance's own frame count are zero."""
    import torch
    if not torch.cuda.is_available():
        raise _vc.VCError('calc_MFCC_input needs a GPU (no CPU fallback)')
    plan = _get_plan(sr, pre_emphasis, hop_length, win_length, n_mels, n_mfcc, n_fft, window,
                     mfcc_normaleze_first_mfcc, mfcc_norm_factor, calc_mfcc_derivate,
                     M_dB_norm_factor, P_dB_norm_factor, mean_abs_amp_norm, clip_output)
    if not torch.is_tensor(wav):
        wav = torch.from_numpy(np.ascontiguousarray(wav, dtype=np.float32))
    if wav.dim() != 2:
        raise ValueError(' - ERROR, calc_MFCC_input_batch: wav must be [B, L]')
    wav = wav.to(device='cuda', dtype=torch.float32).contiguous()
    B, L = wav.shape
    half = plan.cfg.n_fft // 2
    d_lens = None
    if lens is not None:
        if torch.is_tensor(lens):
            d_lens = lens.to(device='cuda', dtype=torch.int32).contiguous()
        else:
            h = np.asarray(lens, dtype=np.int32)
            if h.shape != (B,) or h.min() <= half or h.max() > L:
                raise ValueError(' - ERROR, calc_MFCC_input_batch: lens must be [B] with n_fft//2 < len <= L')
            d_lens = torch.from_numpy(h).to('cuda')
    Fmax = 1 + L // plan.cfg.hop_length
    if out is None:
        mfcc = torch.empty((B, Fmax, plan.mfcc_width), dtype=torch.float32, device=wav.device)
        mel = torch.empty((B, Fmax, plan.cfg.n_mels), dtype=torch.float32, device=wav.device)
        pdb = torch.empty((B, Fmax, plan.n_bins), dtype=torch.float32, device=wav.device)
    else:
        mfcc, mel, pdb = out
    ws = plan.workspace(B, L, wav.device)
    _vc.check(_vc.lib().vc_frontend_stages_f32(plan.handle, _vc.ptr(wav), _vc.ptr(d_lens), B, L, wav.stride(0),
                                               _vc.ptr(mfcc), _vc.ptr(mel), _vc.ptr(pdb),
                                               _vc.ptr(ws), ws.numel(), _vc.current_stream(),
                                               int(stage_mask)))
    return mfcc, mel, pdb


def calc_MFCC_input(y,
                    sr=16000,
                    pre_emphasis=0.97,
                    hop_length=40,
                    win_length=400,
                    n_mels=128,
                    n_mfcc=40,
                    n_fft=None,
                    window='hann',
                    mfcc_normaleze_first_mfcc=True,
                    mfcc_norm_factor=0.01,
                    calc_mfcc_derivate=False,
                    M_dB_norm_factor=0.01,
                    P_dB_norm_factor=0.01,
                    mean_abs_amp_norm=0.003,
                    clip_output=True):
    """Drop-in for audio_lib.calc_MFCC_input (audio_lib.py:89-244): one utterance in (numpy
    1-D array), three np.float32 arrays out, time-major."""
    y = np.ascontiguousarray(np.asarray(y).reshape(1, -1), dtype=np.float32)
    mfcc, mel, pdb = calc_MFCC_input_batch(
        y, None, sr, pre_emphasis, hop_length, win_length, n_mels, n_mfcc, n_fft, window,
        mfcc_normaleze_first_mfcc, mfcc_norm_factor, calc_mfcc_derivate, M_dB_norm_factor,
        P_dB_norm_factor, mean_abs_amp_norm, clip_output)
    return mfcc[0].cpu().numpy(), mel[0].cpu().numpy(), pdb[0].cpu().numpy()


def host_tables(sr, n_fft, n_mels, n_mfcc):
    """(mel [n_mels, 1+n_fft//2], dct [n_mfcc, n_mels]) float64 as the native library builds them
    (host-only; works without a GPU)."""
    cfg = _vc.FrontendCfg(int(sr), 1, int(n_fft), int(n_fft), int(n_mels), int(n_mfcc),
                          0.0, 1.0, 1.0, 1.0, 1.0, 0, 0, 0)
    mel = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float64)
    dct = np.zeros((n_mfcc, n_mels), dtype=np.float64)
    _vc.check(_vc.lib().vc_frontend_host_tables(C.byref(cfg), _vc.ptr(mel), _vc.ptr(dct)))
    return mel, dct


def calc_preemphasis(wav, coeff=0.97):
    """audio_lib.py:12-28 -- y[n] = x[n] - coeff*x[n-1], zero initial state.  Host helper (the
    GPU path fuses this filter into the STFT gather)."""
    wav = np.asarray(wav, dtype=np.float64)
    out = wav.copy()
    out[1:] -= coeff * wav[:-1]
    return out


def calc_inv_preemphasis(preem_wav, coeff=0.97):
    """audio_lib.py:31-47 -- inverse IIR  y[n] = x[n] + coeff*y[n-1]."""
    from scipy import signal
    return signal.lfilter([1], [1, -coeff], preem_wav)


def calc_PHN_target(y, phn_v, phn_conv_d, hop_length=40, win_length=400):
    """audio_lib.py:51-85 -- per-frame phoneme target by larger window overlap (integer host
    logic used when building training caches)."""
    n_samples = int(y.shape[0] / hop_length) + 1
    half_n_fft = win_length // 2
    target_v = []
    i_phn = 0
    for i_s in range(n_samples):
        i_win_s = i_s * hop_length - half_n_fft
        i_win_e = i_win_s + win_length
        while phn_v[i_phn][1] <= i_win_s and i_phn + 1 < len(phn_v):
            i_phn += 1
        cur = phn_v[i_phn]
        pick = cur
        if i_phn + 1 < len(phn_v):
            nxt = phn_v[i_phn + 1]
            d_cur = min(cur[1], i_win_e) - max(cur[0], i_win_s)
            d_nxt = min(nxt[1], i_win_e) - max(nxt[0], i_win_s)
            if d_cur < d_nxt:
                pick = nxt
        target_v.append(phn_conv_d[pick[2]])
    return np.array(target_v, dtype=np.int32)
