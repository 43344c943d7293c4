"""CPU checks that pin oracle/vocoder_oracle.py (the Griffin-Lim restatement of
/root/reference/audio_lib.py:249-308): librosa is absent, so stft / istft are compared with
torch.stft / torch.istft -- an independent implementation of the same published algorithm --
and with the algebraic properties the algorithm must have."""
import numpy as np
import torch

from oracle import frontend_oracle as fo
from oracle import vocoder_oracle as vo


def _speech(L, seed=0):
    return fo.synth_speech(1, L, seed=seed)[0].astype(np.float64)


def test_stft_istft_match_torch():
    y = _speech(8000, 3)
    y = y[:80 * (len(y) // 80)]
    S = vo.stft(y, 400, 80, 400).astype(np.complex128)
    w = torch.hann_window(400, periodic=True, dtype=torch.float64)
    St = torch.stft(torch.from_numpy(y), 400, 80, 400, w, center=True, pad_mode='reflect', return_complex=True).numpy()
    assert S.shape == St.shape == (201, 1 + len(y) // 80)
    assert np.abs(S - St).max() < 1e-6 * np.abs(St).max()          # complex64 storage of librosa.stft
    rng = np.random.RandomState(0)
    Z = St * np.exp(1j * rng.uniform(-0.3, 0.3, St.shape))           # an inconsistent spectrogram
    a = vo.istft(Z, 80, 400)
    b = torch.istft(torch.from_numpy(Z), 400, 80, 400, w, center=True).numpy()
    assert a.shape == b.shape == (80 * (St.shape[1] - 1),)
    assert np.abs(a - b).max() < 1e-10 * max(np.abs(b).max(), 1.0)


def test_istft_inverts_stft_and_padded_window():
    y = _speech(4000, 5)[:3960]
    for n_fft, win, hop in ((400, 400, 80), (512, 400, 128), (800, 800, 40)):
        yy = y[:hop * (len(y) // hop)]
        w = fo.fft_window('hann', win, n_fft).reshape(-1, 1)
        yp = np.pad(yy, n_fft // 2, mode='reflect')
        nfr = 1 + (len(yp) - n_fft) // hop
        idx = np.arange(n_fft)[:, None] + hop * np.arange(nfr)[None, :]
        S = np.fft.fft(w * yp[idx], axis=0)[:1 + n_fft // 2]
        back = vo.istft(S, hop, win)
        assert back.shape == yy.shape
        assert np.abs(back - yy).max() < 1e-10


def test_window_sumsquare_closed_form():
    wss = vo.window_sumsquare('hann', 50, 80, 400, 400)
    assert wss.shape == (400 + 80 * 49,)
    assert np.allclose(wss[400:-400], 1.875)       # hann^2 at 5x overlap sums to 3N/(8 hop) = 1.875
    assert wss[0] == 0.0


def test_project_phase_zero_bin_and_magnitude():
    d = np.array([[0.0 + 0.0j, 3.0 + 4.0j], [-2.0 + 0.0j, 0.0 - 1.0j]])
    amp = np.array([[2.0, 10.0], [1.0, 0.5]])
    s = vo.project_phase(d, amp)
    assert s[0, 0] == 2.0 + 0.0j                                   # np.angle(0) == 0
    assert np.allclose(s[0, 1], 6.0 + 8.0j) and np.allclose(s[1, 0], -1.0) and np.allclose(s[1, 1], -0.5j)


def test_initial_phase_is_the_seeded_global_generator():
    np.random.seed(7)
    ref = np.pi * np.random.rand(201, 13)                           # audio_lib.py:255 after np.random.seed(7)
    assert np.array_equal(vo.initial_phase((201, 13), 7), ref)


def test_griffin_lim_decreases_spectral_distance():
    y = _speech(6400, 1)
    amp = np.abs(vo.stft(y, 400, 80, 400)).astype(np.float64)
    tr = []
    w5 = vo.griffin_lim_alg(amp, 400, 80, num_iters=5, seed=1)
    w40 = vo.griffin_lim_alg(amp, 400, 80, num_iters=40, seed=1, trace=tr)
    c5, c40 = vo.spectral_convergence(w5, amp, 400, 80), vo.spectral_convergence(w40, amp, 400, 80)
    assert w40.shape == (6400,) and len(tr) == 39
    assert c40 < c5 < 1.0 and c40 < 0.25
    assert tr[-1] < tr[0]
    # one explicit step from the 5-iteration waveform equals the 6-iteration run
    w6 = vo.griffin_lim_alg(amp, 400, 80, num_iters=6, seed=1)
    step = vo.griffin_lim_step(w5, amp, 400, 80)
    assert np.abs(step - w6).max() < 1e-5 * np.abs(w6).max()       # complex64 rounding inside the loop


def test_inverse_preemphasis_inverts_preemphasis():
    y = _speech(3000, 2)
    pe = fo.calc_preemphasis(y, 0.97)
    back = vo.calc_inv_preemphasis(pe, 0.97)
    assert np.abs(back - y).max() < 1e-10
    x = np.zeros(50); x[0] = 1.0
    assert np.allclose(vo.calc_inv_preemphasis(x, 0.5), 0.5 ** np.arange(50))


def test_power_to_amp_and_release():
    rng = np.random.RandomState(0)
    P = rng.uniform(-0.1, 0.9, (30, 201))
    a = vo.power_to_amp(P, 0.01, 1.0)
    assert a.shape == (201, 30)
    assert np.allclose(a.T, 10 ** (0.05 * (np.maximum(P, 0) / 0.01 - 80)))
    a2 = vo.power_to_amp(P, 0.01, 1.3)
    Pc = np.maximum(P, 0)
    Pr = Pc ** 1.3
    Pr *= Pc.mean() / Pr.mean()
    assert np.allclose(a2.T, 10 ** (0.05 * (Pr / 0.01 - 80)))
