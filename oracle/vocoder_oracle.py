"""CPU ORACLE (test infrastructure, NOT product code) for the Griffin-Lim vocoder.

Restates in numpy (float64 unless noted) the arithmetic of
  /root/reference/audio_lib.py:31-47    calc_inv_preemphasis  (scipy.signal.lfilter([1],[1,-c]))
  /root/reference/audio_lib.py:249-274  griffin_lim_alg       (librosa.istft / librosa.stft loop)
  /root/reference/audio_lib.py:278-308  from_power_to_wav
and of the librosa-0.6.x functions they lower to (librosa is an un-vendored, unpinned third-party
dependency that is absent here, see frontend_oracle.py):
  * librosa.core.istft(stft, hop_length, win_length, window='hann', center=True):
      n_fft = 2*(rows-1); every frame is ifft(hermitian extension).real * padded window, frames are
      overlap-added at i*hop, the sum is divided by window_sumsquare where that exceeds
      np.finfo(float32).tiny, and n_fft//2 samples are trimmed from both ends,
  * librosa.filters.window_sumsquare(window, n_frames, hop, win_length, n_fft, norm=None),
  * librosa.core.stft (frontend_oracle.stft), librosa.magphase + np.angle: phase of 0 is 0,
  * librosa.core.db_to_power(S) = 10 ** (0.1 * S).

The reference draws the initial phase from the unseeded global numpy generator
(audio_lib.py:255: ``np.pi * np.random.rand(*stft_amp.shape)``); here it is an argument
(``phase0``) or drawn from ``np.random.RandomState(seed)`` in the same [bins, frames] order, which
is what the reference computes after ``np.random.seed(seed)``.

PARITY STATUS: **parity unpinned** at the librosa boundary (no fixtures in the reference for this
path).  What pins this file: tests/test_vocoder_cpu.py checks stft/istft against torch.stft /
torch.istft (an independent implementation of the same published algorithm), perfect
reconstruction istft(stft(x)) == x, and lfilter against its closed form.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.
"""
import numpy as np
from scipy import signal

from oracle.frontend_oracle import fft_window, stft

F32_TINY = float(np.finfo(np.float32).tiny)


def window_sumsquare(window, n_frames, hop_length, win_length, n_fft):
    """librosa.filters.window_sumsquare with norm=None."""
    n = n_fft + hop_length * (n_frames - 1)
    x = np.zeros(n, dtype=np.float64)
    win_sq = fft_window(window, win_length, n_fft) ** 2
    for i in range(n_frames):
        s = i * hop_length
        x[s:min(n, s + n_fft)] += win_sq[:max(0, min(n_fft, n - s))]
    return x


def istft(spec, hop_length, win_length, window='hann'):
    """librosa.core.istft(spec [1+n_fft/2, F], hop_length, win_length, window, center=True)."""
    n_fft = 2 * (spec.shape[0] - 1)
    w = fft_window(window, win_length, n_fft)
    n_frames = spec.shape[1]
    y = np.zeros(n_fft + hop_length * (n_frames - 1), dtype=np.float64)
    full = np.concatenate([spec, spec[-2:0:-1].conj()], axis=0)            # hermitian extension
    frames = np.fft.ifft(full, axis=0).real * w[:, None]                    # [n_fft, F]
    for i in range(n_frames):
        y[i * hop_length:i * hop_length + n_fft] += frames[:, i]
    wss = window_sumsquare(window, n_frames, hop_length, win_length, n_fft)
    nz = wss > F32_TINY
    y[nz] /= wss[nz]
    return y[n_fft // 2:-(n_fft // 2)]


def project_phase(spec, amp):
    """stft_amp * exp(1j * np.angle(exp(1j*np.angle(D)))) (audio_lib.py:267-269): keeps the phase
    of D, replaces its magnitude; a zero bin gets phase 0."""
    mag = np.abs(spec)
    unit = np.where(mag > 0, spec / np.where(mag > 0, mag, 1.0), 1.0 + 0.0j)
    return amp * unit


def initial_phase(shape, seed):
    """audio_lib.py:255 after np.random.seed(seed): pi * U[0,1) in [bins, frames] order."""
    return np.pi * np.random.RandomState(seed).rand(*shape)


def griffin_lim_alg(stft_amp, win_length, hop_length, num_iters=300, n_fft=None, phase0=None, seed=0,
                    trace=None):
    """audio_lib.py:249-274.  stft_amp [1+n_fft/2, F].  ``trace``: optional list that receives
    the rms difference between successive waveforms (what verbose mode prints)."""
    if n_fft is None:
        n_fft = win_length
    stft_amp = np.asarray(stft_amp, dtype=np.float64)
    if phase0 is None:
        phase0 = initial_phase(stft_amp.shape, seed)
    spec = stft_amp * np.exp(1j * np.asarray(phase0, dtype=np.float64))
    wav = last = None
    for i in range(num_iters):
        wav = istft(spec, hop_length, win_length)
        if trace is not None and last is not None:
            trace.append(float(np.sqrt(np.mean(np.square(last - wav)))))
        if i != num_iters - 1:
            d = stft(wav, n_fft, hop_length, win_length).astype(np.complex128)
            spec = project_phase(d, stft_amp)
        last = wav
    return wav


def griffin_lim_step(wav, stft_amp, win_length, hop_length, n_fft=None):
    """One projection: wav -> istft(amp * phase(stft(wav))) in float64 without the complex64
    rounding of librosa.stft (used for tight single-step checks)."""
    if n_fft is None:
        n_fft = win_length
    w = fft_window('hann', win_length, n_fft).reshape(-1, 1)
    yp = np.pad(wav, n_fft // 2, mode='reflect')
    n_frames = 1 + (len(yp) - n_fft) // hop_length
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(n_frames)[None, :]
    d = np.fft.fft(w * yp[idx], axis=0)[:1 + n_fft // 2]
    return istft(project_phase(d, stft_amp), hop_length, win_length)


def spectral_convergence(wav, stft_amp, win_length, hop_length, n_fft=None):
    """|| |STFT(wav)| - amp ||_F / || amp ||_F : the quantity Griffin-Lim decreases."""
    if n_fft is None:
        n_fft = win_length
    d = np.abs(stft(np.asarray(wav, dtype=np.float64), n_fft, hop_length, win_length).astype(np.complex128))
    return float(np.linalg.norm(d - stft_amp) / max(np.linalg.norm(stft_amp), 1e-30))


def calc_inv_preemphasis(preem_wav, coeff=0.97):
    """audio_lib.py:31-47."""
    return signal.lfilter([1], [1, -coeff], preem_wav)


def power_to_amp(P, P_dB_norm_factor=0.01, realse=1.0):
    """audio_lib.py:289-298: clamp, optional 'realse' power with mean preserved, dB -> magnitude.
    P [F, bins] -> amplitude [bins, F]."""
    P = np.maximum(0.0, np.asarray(P, dtype=np.float64))
    if realse != 1.0:
        p_mean = P.mean()
        P = P ** realse
        P = (p_mean / P.mean()) * P
    return np.sqrt(np.power(10.0, 0.1 * (P.T / P_dB_norm_factor - 80)))


def from_power_to_wav(P, P_dB_norm_factor=0.01, pre_emphasis=0.97, hop_length=40, win_length=800,
                      mean_abs_amp_norm=0.01, n_iter=200, n_fft=None, realse=1.0, phase0=None, seed=0):
    """audio_lib.py:278-308."""
    amp = power_to_amp(P, P_dB_norm_factor, realse)
    y = griffin_lim_alg(amp, win_length, hop_length, num_iters=n_iter, n_fft=n_fft, phase0=phase0, seed=seed)
    if pre_emphasis != 0:
        y = calc_inv_preemphasis(y, pre_emphasis)
    return y * (mean_abs_amp_norm / np.abs(y).mean())
