"""CPU ORACLE (test infrastructure, NOT product code) for the signal front-end.

This file restates, in plain numpy/scipy, the arithmetic of the reference's
``calc_MFCC_input`` (/root/reference/audio_lib.py:89-244) and of the librosa-0.6.x
functions it lowers to.  librosa is a third-party dependency that is NOT vendored in the
reference and NOT installable here (no network); it is unpinned by the reference (no
requirements file) and its API use (``librosa.filters.dct`` audio_lib.py:176,
``librosa.output.write_wav`` test.py:177) implies >=0.6.0,<0.7.  Its published algorithm is
restated below function by function.

PARITY STATUS: **parity unpinned** at the librosa boundary -- the reference ships no tests,
golden vectors or fixtures for this path.  What pins this file:
  * the shape contract F = 1 + L // hop_length (audio_lib.py:52, ARCTIC_reader.py:161),
  * the structural facts of the mel matrix recorded in SURVEY.md section 2.1
    (391 non-zeros, 1..14 per filter, bins 0 and 200 unused) -- checked in tests/,
  * orthonormality of the DCT basis rows and Parseval for the STFT -- checked in tests/.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (speech-cloner_amd/) never does.

Dtype notes (numpy-1.x value-based casting, which the reference era used, emulated
explicitly because this container runs numpy 2.x):
  * ``librosa.load`` yields float32 audio; ``(c/np.abs(y).mean()) * y`` keeps float32
    (audio_lib.py:125-126),
  * ``scipy.signal.lfilter`` with float64 coefficients returns float64 (audio_lib.py:27),
  * ``librosa.stft`` computes the FFT in float64 and stores complex64,
  * ``np.abs`` / ``** 2`` / ``power_to_db`` then run in float32 (audio_lib.py:150-157),
  * the mel matrix is float64, so ``M @ P`` and everything after it is float64
    (audio_lib.py:169-179) until the final ``astype(np.float32)`` (audio_lib.py:244).
"""
import numpy as np
from scipy import signal


# --------------------------------------------------------------------------- librosa pieces
def hz_to_mel(freq):
    """Slaney-style hz->mel (librosa.core.time_frequency.hz_to_mel, htk=False)."""
    freq = np.asanyarray(freq, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = freq / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    mels = np.where(freq >= min_log_hz,
                    min_log_mel + np.log(np.maximum(freq, 1e-300) / min_log_hz) / logstep,
                    mels)
    return mels


def mel_to_hz(mels):
    """Slaney-style mel->hz (librosa.core.time_frequency.mel_to_hz, htk=False)."""
    mels = np.asanyarray(mels, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * mels
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    freqs = np.where(mels >= min_log_mel,
                     min_log_hz * np.exp(logstep * (mels - min_log_mel)),
                     freqs)
    return freqs


def mel_filterbank(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm=1)
    as called at audio_lib.py:160-166.  Returns float64 [n_mels, 1 + n_fft//2]."""
    if fmax is None:
        fmax = float(sr) / 2
    n_bins = 1 + n_fft // 2
    weights = np.zeros((n_mels, n_bins), dtype=np.float64)
    fftfreqs = np.linspace(0, float(sr) / 2, n_bins, endpoint=True)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])      # norm=1: area normalisation
    weights *= enorm[:, np.newaxis]
    return weights


def dct_basis(n_filters, n_input):
    """librosa.filters.dct(n_filters, n_input) (librosa 0.6.x) as called at audio_lib.py:176:
    orthonormal DCT-II basis, float64 [n_filters, n_input]."""
    basis = np.empty((n_filters, n_input), dtype=np.float64)
    basis[0, :] = 1.0 / np.sqrt(n_input)
    samples = np.arange(1, 2 * n_input, 2) * np.pi / (2.0 * n_input)
    for i in range(1, n_filters):
        basis[i, :] = np.cos(i * samples) * np.sqrt(2.0 / n_input)
    return basis


def fft_window(window, win_length, n_fft):
    """librosa.filters.get_window(window, win_length, fftbins=True) followed by
    util.pad_center(., n_fft), as librosa.core.stft does."""
    w = signal.get_window(window, win_length, fftbins=True).astype(np.float64)
    if win_length < n_fft:
        lpad = (n_fft - win_length) // 2
        w = np.pad(w, (lpad, n_fft - win_length - lpad), mode='constant')
    return w


def stft(y, n_fft, hop_length, win_length, window='hann'):
    """librosa.core.stft(y, n_fft, hop_length, win_length, window, center=True,
    pad_mode='reflect') as called at audio_lib.py:141-147.
    y float64 [L] -> complex64 [1 + n_fft//2, 1 + L//hop_length]."""
    w = fft_window(window, win_length, n_fft).reshape(-1, 1)
    yp = np.pad(y, int(n_fft // 2), mode='reflect')
    n_frames = 1 + (len(yp) - n_fft) // hop_length
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(n_frames)[None, :]
    frames = yp[idx]                                            # [n_fft, n_frames]
    spec = np.fft.fft(w * frames, axis=0)[:1 + n_fft // 2]
    return spec.astype(np.complex64)


def power_to_db(S, amin=1e-10, top_db=80.0):
    """librosa.core.power_to_db(S, ref=1.0, amin, top_db).  Keeps S's float dtype
    (numpy-1.x scalar casting)."""
    dt = S.dtype
    log_spec = (dt.type(10.0) * np.log10(np.maximum(dt.type(amin), S))).astype(dt)
    log_spec = log_spec - dt.type(10.0 * np.log10(max(amin, 1.0)))
    if top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - dt.type(top_db))
    return log_spec


def amplitude_to_db(S, amin=1e-5, top_db=80.0):
    """librosa.core.amplitude_to_db(S, ref=1.0, amin=1e-5, top_db=80) =
    power_to_db(S**2, amin=amin**2)."""
    magnitude = np.abs(S)
    return power_to_db(np.square(magnitude), amin=amin ** 2, top_db=top_db)


# --------------------------------------------------------------------------- reference path
def calc_preemphasis(wav, coeff=0.97):
    """audio_lib.py:12-28: scipy.signal.lfilter([1, -coeff], [1], wav) (zero initial state)."""
    return signal.lfilter([1, -coeff], [1], wav)


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
    """Restatement of audio_lib.py:89-244 (same signature, defaults and return order).

    Returns (MFCC [F, n_mfcc*(1|2)], M_dB [F, n_mels], P_dB [F, 1 + n_fft//2]) float32,
    F = 1 + len(y)//hop_length."""
    y = np.asarray(y)
    if y.dtype != np.float64:
        y = y.astype(np.float32)

    # audio_lib.py:125-126 -- amplitude normalisation (dtype of y is kept: float32 audio)
    if mean_abs_amp_norm != 1.0:
        scale = np.float64(mean_abs_amp_norm) / np.abs(y).mean()
        y = (y.dtype.type(scale) * y).astype(y.dtype)

    # audio_lib.py:129-133 -- pre-emphasis FIR in float64
    if pre_emphasis != 0.0:
        y_preem = calc_preemphasis(y.astype(np.float64), pre_emphasis)
    else:
        y_preem = y.astype(np.float64)

    if n_fft is None:
        n_fft = win_length

    # audio_lib.py:141-150 -- STFT magnitude (complex64 -> float32)
    F = np.abs(stft(y_preem, n_fft, hop_length, win_length, window))
    # audio_lib.py:155-157
    P = F ** 2                                                  # float32
    P_dB = power_to_db(P)                                       # float32
    # audio_lib.py:160-169
    M = mel_filterbank(sr, n_fft, n_mels)
    M_spec = M @ P                                              # float64
    # audio_lib.py:172 -- amplitude_to_db applied to the mel POWER (quirk kept on purpose)
    M_spec_dB = amplitude_to_db(M_spec)
    # audio_lib.py:176-179
    D = dct_basis(n_mfcc, n_mels)
    MFCC = D @ M_spec_dB

    # audio_lib.py:207-216 -- time-major
    MFCC_ret = MFCC.T.copy()
    M_dB_ret = M_spec_dB.T
    P_dB_ret = P_dB.T

    # audio_lib.py:220-228
    if mfcc_normaleze_first_mfcc:
        MFCC_ret[:, 0] -= MFCC_ret[0, 0]
    if mfcc_norm_factor != 1.0:
        MFCC_ret = mfcc_norm_factor * MFCC_ret
    if calc_mfcc_derivate:
        z = np.zeros((1, MFCC_ret.shape[1]), dtype=np.float32)
        d_MFCC = 2 * np.concatenate([z, MFCC_ret[2:] - MFCC_ret[:-2], z], axis=0)
        MFCC_ret = np.concatenate([MFCC_ret, d_MFCC], axis=1)
    # audio_lib.py:230-235
    if P_dB_norm_factor != 1.0:
        P_dB_ret = P_dB_ret.dtype.type(P_dB_norm_factor) * (P_dB_ret - P_dB_ret.min())
    if M_dB_norm_factor != 1.0:
        M_dB_ret = M_dB_norm_factor * (M_dB_ret - M_dB_ret.min())
    # audio_lib.py:237-240
    if clip_output:
        MFCC_ret = np.clip(MFCC_ret, -1.0, 1.0)
        P_dB_ret = np.clip(P_dB_ret, -1.0, 1.0)
        M_dB_ret = np.clip(M_dB_ret, -1.0, 1.0)
    # audio_lib.py:244
    return (MFCC_ret.astype(np.float32), M_dB_ret.astype(np.float32),
            P_dB_ret.astype(np.float32))


def calc_MFCC_input_batch(wav, lengths=None, **kw):
    """Batched convenience used by tests/bench: wav [B, Lmax], lengths [B] (None = all full).
    Returns three lists of per-utterance arrays (ragged in F)."""
    wav = np.asarray(wav)
    B = wav.shape[0]
    outs = ([], [], [])
    for b in range(B):
        L = wav.shape[1] if lengths is None else int(lengths[b])
        r = calc_MFCC_input(wav[b, :L], **kw)
        for o, v in zip(outs, r):
            o.append(v)
    return outs


def calc_PHN_target(y_len, phn_v, phn_conv_d, hop_length=40, win_length=400):
    """audio_lib.py:51-85 -- per-frame phoneme label by larger window overlap.  Integer work;
    only the length contract n_frames = int(L / hop) + 1 matters to the hot path."""
    n_samples = int(y_len / hop_length) + 1
    half = win_length // 2
    target_v = []
    i_phn = 0
    for i_s in range(n_samples):
        ws = i_s * hop_length - half
        we = i_s * hop_length + win_length - half
        while phn_v[i_phn][1] <= ws and i_phn + 1 < len(phn_v):
            i_phn += 1
        da = min(phn_v[i_phn][1], we) - max(phn_v[i_phn][0], ws)
        if i_phn + 1 < len(phn_v):
            db = min(phn_v[i_phn + 1][1], we) - max(phn_v[i_phn + 1][0], ws)
            target_v.append(phn_conv_d[phn_v[i_phn][2]] if da >= db
                            else phn_conv_d[phn_v[i_phn + 1][2]])
        else:
            target_v.append(phn_conv_d[phn_v[i_phn][2]])
    return np.array(target_v, dtype=np.int32)


# --------------------------------------------------------------------------- synthetic audio
def synth_speech(B, L, seed=0, sr=16000):
    """SURVEY.md section 8d synthetic "speech-like" audio: 5 harmonics of f0 in U(80,250) Hz
    with slow AM (2-8 Hz) + N(0, 0.05) noise, scaled to peak 0.5.  float32 [B, L]."""
    rng = np.random.RandomState(seed)
    t = np.arange(L, dtype=np.float64) / sr
    out = np.empty((B, L), dtype=np.float32)
    for b in range(B):
        f0 = rng.uniform(80, 250)
        am_f = rng.uniform(2, 8)
        am_ph = rng.uniform(0, 2 * np.pi)
        x = np.zeros(L)
        for h in range(1, 6):
            x += (1.0 / h) * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 2 * np.pi))
        x *= 0.55 + 0.45 * np.sin(2 * np.pi * am_f * t + am_ph)
        x += rng.normal(0, 0.05, L)
        x *= 0.5 / np.abs(x).max()
        out[b] = x.astype(np.float32)
    return out
