"""Signal front-end with the reference's ``audio_lib`` surface, computed on the MI355X.

Mirrors /root/reference/audio_lib.py: same function names, argument order, defaults and
return types for the functions on the conversion hot path:

  calc_MFCC_input(y, ...) -> (MFCC [F, n_mfcc*(1|2)], M_dB [F, n_mels], P_dB [F, 1+n_fft//2])
                              np.float32, time-major, F = 1 + len(y)//hop_length
                              (audio_lib.py:89-244)
  calc_preemphasis / calc_inv_preemphasis                      (audio_lib.py:12-47)
  calc_PHN_target                                              (audio_lib.py:51-85)
  griffin_lim_alg(stft_amp [bins, F], ...) -> wav              (audio_lib.py:249-274)
  from_power_to_wav(P [F, bins], ...) -> wav                   (audio_lib.py:278-308)

plus batched entry points the reference does not have (its per-utterance calls are special
cases of them):

  calc_MFCC_input_batch(wav [B, L], lens=None, ...) -> three torch.cuda tensors [B, Fmax, C]
  from_power_to_wav_batch(P [B, Fmax, bins], n_frames=None, ...) -> torch.cuda tensor [B, hop*(Fmax-1)]

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
                          stage_mask=7,
                          out_frames=None):
    """Batched calc_MFCC_input on the GPU.

    wav  : float32 [B, L] torch.cuda tensor (or numpy array, uploaded).
    lens : optional per-utterance sample counts (sequence / int32 tensor); None = all L.
    out  : optional (mfcc, mel, pow) preallocated cuda tensors to write into.
    stage_mask : measurement hook (bench.py): subset of the three launches, see vc_hip.h.
    out_frames : store only the first out_frames frames of every utterance (the later ones still count for its
                 normalisation statistics): with out_frames a multiple of the window length the result reshapes to the
                 encoder's window batch without a copy.  None = all Fmax = 1 + L//hop_length frames.
    Returns (MFCC [B, R, W], M_dB [B, R, n_mels], P_dB [B, R, 1+n_fft//2]) cuda float32, R = out_frames or Fmax;
    rows beyond an utterance's own frame count are zero."""
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
    R = Fmax if out_frames is None else int(out_frames)
    if not 0 < R <= Fmax:
        raise ValueError(' - ERROR, calc_MFCC_input_batch: out_frames must be in [1, {}]'.format(Fmax))
    if out is None:
        mfcc = torch.empty((B, R, plan.mfcc_width), dtype=torch.float32, device=wav.device)
        mel = torch.empty((B, R, plan.cfg.n_mels), dtype=torch.float32, device=wav.device)
        pdb = torch.empty((B, R, plan.n_bins), dtype=torch.float32, device=wav.device)
    else:
        mfcc, mel, pdb = out
        for t, w in ((mfcc, plan.mfcc_width), (mel, plan.cfg.n_mels), (pdb, plan.n_bins)):
            if tuple(t.shape) != (B, R, w) or not t.is_contiguous() or t.dtype != torch.float32:
                raise ValueError(' - ERROR, calc_MFCC_input_batch: out tensors must be contiguous float32 [B, {}, width]'.format(R))
    ws = plan.workspace(B, L, wav.device)
    _vc.check(_vc.lib().vc_frontend_stages_f32(plan.handle, _vc.ptr(wav), _vc.ptr(d_lens), B, L, wav.stride(0), R,
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


# --------------------------------------------------------------------------- vocoder
_VOC_PLANS = {}


class _VocPlan:
    def __init__(self, win_length, hop_length, n_fft):
        h = C.c_void_p()
        _vc.check(_vc.lib().vc_vocoder_plan_create(int(win_length), int(hop_length), int(n_fft), None, C.byref(h)))
        self.handle = h
        self.win_length, self.hop_length, self.n_fft = int(win_length), int(hop_length), int(n_fft)
        self.n_bins = 1 + self.n_fft // 2
        self._ws = None

    def workspace(self, batch, max_frames, trace, device):
        import torch
        need = _vc.lib().vc_vocoder_workspace_bytes(self.handle, batch, max_frames, int(trace))
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws


def _get_voc_plan(win_length, hop_length, n_fft):
    if n_fft is None:
        n_fft = win_length
    key = (int(win_length), int(hop_length), int(n_fft))
    p = _VOC_PLANS.get(key)
    if p is None:
        p = _VOC_PLANS[key] = _VocPlan(*key)
    return p


def _frames_arg(n_frames, B, Fmax, plan):
    """Validates per-utterance frame counts; returns (device int32 tensor or None, host array)."""
    import torch
    if n_frames is None:
        h = np.full((B,), Fmax, dtype=np.int32)
        d = None
    else:
        h = np.asarray(n_frames.cpu() if torch.is_tensor(n_frames) else n_frames, dtype=np.int32)
        if h.shape != (B,) or h.max() > Fmax:
            raise ValueError(' - ERROR, n_frames must be [B] with values <= {}'.format(Fmax))
        d = torch.from_numpy(h).to('cuda')
    if plan.hop_length * (int(h.min()) - 1) <= plan.n_fft // 2:
        raise ValueError(' - ERROR, griffin_lim: every utterance needs hop_length*(frames-1) > n_fft//2 samples '
                         '(librosa.stft reflect padding)')
    return d, h


def griffin_lim_batch(amp, n_frames=None, win_length=400, hop_length=80, num_iters=300, n_fft=None, phase0=None,
                      trace=False):
    """Batched Griffin-Lim on the GPU.  amp: float32 [B, Fmax, bins] magnitudes (frame-major, the
    decoder's y_stft layout); phase0: same shape, radians (default: pi * np.random.rand drawn per
    utterance in the reference's [bins, F] order, audio_lib.py:255).
    Returns wav [B, hop*(Fmax-1)] cuda float32 (zero beyond an utterance's hop*(frames-1) samples)
    and, with ``trace``, the per-iteration sum of squared waveform changes [num_iters, B]."""
    import torch
    if not torch.cuda.is_available():
        raise _vc.VCError('griffin_lim needs a GPU (no CPU fallback)')
    plan = _get_voc_plan(win_length, hop_length, n_fft)
    if not torch.is_tensor(amp):
        amp = torch.from_numpy(np.ascontiguousarray(amp, dtype=np.float32))
    amp = amp.to(device='cuda', dtype=torch.float32).contiguous()
    if amp.dim() != 3 or amp.shape[2] != plan.n_bins:
        raise ValueError(' - ERROR, griffin_lim_batch: amp must be [B, F, {}]'.format(plan.n_bins))
    B, Fmax, nb = amp.shape
    d_nf, h_nf = _frames_arg(n_frames, B, Fmax, plan)
    if phase0 is None:
        ph = np.zeros((B, Fmax, nb), dtype=np.float32)
        for b in range(B):
            ph[b, :h_nf[b]] = (np.pi * np.random.rand(nb, int(h_nf[b]))).T
        phase0 = torch.from_numpy(ph)
    elif not torch.is_tensor(phase0):
        phase0 = torch.from_numpy(np.ascontiguousarray(phase0, dtype=np.float32))
    phase0 = phase0.to(device='cuda', dtype=torch.float32).contiguous()
    if phase0.shape != amp.shape:
        raise ValueError(' - ERROR, griffin_lim_batch: phase0 must have the shape of amp')
    L = plan.hop_length * (Fmax - 1)
    wav = torch.empty((B, L), dtype=torch.float32, device=amp.device)
    tr = torch.empty((int(num_iters), B), dtype=torch.float32, device=amp.device) if trace else None
    ws = plan.workspace(B, Fmax, trace, amp.device)
    _vc.check(_vc.lib().vc_griffin_lim_f32(plan.handle, _vc.ptr(amp), _vc.ptr(phase0), _vc.ptr(d_nf), B, Fmax,
                                           int(num_iters), _vc.ptr(wav), L, _vc.ptr(tr), _vc.ptr(ws), ws.numel(),
                                           _vc.current_stream()))
    return (wav, tr) if trace else wav


def griffin_lim_alg(stft_amp, win_length, hop_length, num_iters=300, n_fft=None, verbose=True, phase0=None):
    """audio_lib.py:249-274.  stft_amp [1+n_fft//2, F] -> wav [hop*(F-1)].  The initial phase comes
    from the global numpy generator exactly like the reference (``np.random.seed`` makes both
    reproducible) unless ``phase0`` [bins, F] is given.  verbose prints the reference's
    per-iteration ``mrse_delta`` lines (after the run: the iterations are queued asynchronously)."""
    stft_amp = np.asarray(stft_amp)
    if phase0 is None:
        phase0 = np.pi * np.random.rand(*stft_amp.shape)
    amp = np.ascontiguousarray(stft_amp.T, dtype=np.float32)[None]
    ph = np.ascontiguousarray(np.asarray(phase0).T, dtype=np.float32)[None]
    r = griffin_lim_batch(amp, None, win_length, hop_length, num_iters, n_fft, ph, trace=bool(verbose))
    if verbose:
        wav, tr = r
        tr = tr.cpu().numpy()[:, 0]
        for i in range(1, int(num_iters)):
            print(' i={}  mrse_delta = {}'.format(i, np.sqrt(tr[i] / max(wav.shape[1], 1))))
    else:
        wav = r
    return wav[0].cpu().numpy()


def from_power_to_wav_batch(P, n_frames=None, P_dB_norm_factor=0.01, pre_emphasis=0.97, hop_length=40,
                            win_length=800, mean_abs_amp_norm=0.01, n_iter=200, n_fft=None, realse=1.0,
                            phase0=None, trace=False):
    """Batched from_power_to_wav: P [B, Fmax, bins] normalised power dB (the decoder's y_stft) ->
    wav [B, hop*(Fmax-1)] cuda float32; utterance b is valid up to hop*(n_frames[b]-1) samples."""
    import torch
    if not torch.cuda.is_available():
        raise _vc.VCError('from_power_to_wav needs a GPU (no CPU fallback)')
    plan = _get_voc_plan(win_length, hop_length, n_fft)
    if not torch.is_tensor(P):
        P = torch.from_numpy(np.ascontiguousarray(P, dtype=np.float32))
    P = P.to(device='cuda', dtype=torch.float32).contiguous()
    if P.dim() != 3 or P.shape[2] != plan.n_bins:
        raise ValueError(' - ERROR, from_power_to_wav_batch: P must be [B, F, {}]'.format(plan.n_bins))
    B, Fmax, nb = P.shape
    d_nf, _ = _frames_arg(n_frames, B, Fmax, plan)
    amp = torch.empty_like(P)
    _vc.check(_vc.lib().vc_power_to_amp(_vc.ptr(P), _vc.ptr(d_nf), B, Fmax, nb, float(P_dB_norm_factor), float(realse),
                                        _vc.ptr(amp), _vc.current_stream()))
    r = griffin_lim_batch(amp, n_frames, win_length, hop_length, n_iter, n_fft, phase0, trace)
    wav = r[0] if trace else r
    _vc.check(_vc.lib().vc_inv_preemphasis_normalize(plan.handle, _vc.ptr(wav), _vc.ptr(d_nf), B, Fmax, wav.shape[1],
                                                     float(pre_emphasis), float(mean_abs_amp_norm), _vc.current_stream()))
    return r


def from_power_to_wav(P, P_dB_norm_factor=0.01, pre_emphasis=0.97, hop_length=40, win_length=800,
                      mean_abs_amp_norm=0.01, n_iter=200, n_fft=None, realse=1.0, verbose=True, phase0=None):
    """audio_lib.py:278-308.  P [F, bins] -> wav float32 numpy [hop*(F-1)]."""
    P = np.asarray(P)
    if phase0 is None:
        phase0 = np.pi * np.random.rand(P.shape[1], P.shape[0])          # [bins, F] like audio_lib.py:255
    ph = np.ascontiguousarray(np.asarray(phase0).T, dtype=np.float32)[None]
    r = from_power_to_wav_batch(np.ascontiguousarray(P, dtype=np.float32)[None], None, P_dB_norm_factor, pre_emphasis,
                                hop_length, win_length, mean_abs_amp_norm, n_iter, n_fft, realse, ph, trace=bool(verbose))
    if verbose:
        wav, tr = r
        tr = tr.cpu().numpy()[:, 0]
        for i in range(1, int(n_iter)):
            print(' i={}  mrse_delta = {}'.format(i, np.sqrt(tr[i] / max(wav.shape[1], 1))))
    else:
        wav = r
    return wav[0].cpu().numpy()
