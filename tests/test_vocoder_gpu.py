"""GPU parity of the Griffin-Lim vocoder (csrc/vc_vocoder.hip through the C ABI) against
oracle/vocoder_oracle.py (restating /root/reference/audio_lib.py:31-47, 249-308).

Floating point: kernels compute in float32, the oracle in float64.  One projection step has to
agree to 2e-5 of the waveform peak; over several iterations phase differences at near-silent bins
grow, so longer runs are compared through the waveform with a looser bound and through the
spectral-convergence figure Griffin-Lim minimises."""
import numpy as np
import pytest
import torch

from oracle import frontend_oracle as fo
from oracle import vocoder_oracle as vo

pytestmark = pytest.mark.gpu


def _amp_of_speech(L, seed, n_fft=400, hop=80, win=400):
    y = fo.synth_speech(1, L, seed=seed)[0].astype(np.float64)
    y = y[:hop * (len(y) // hop)]
    return np.abs(vo.stft(y, n_fft, hop, win)).astype(np.float64)          # [bins, F]


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize('n_iter,tol', [(1, 2e-5), (2, 5e-5), (4, 2e-4)])
def test_griffin_lim_first_iterations_match_oracle(n_iter, tol):
    import audio_lib
    amp = _amp_of_speech(8000, 11)
    ph = vo.initial_phase(amp.shape, 3)
    ref = vo.griffin_lim_alg(amp, 400, 80, num_iters=n_iter, phase0=ph)
    got = audio_lib.griffin_lim_alg(amp, 400, 80, num_iters=n_iter, verbose=False, phase0=ph)
    assert got.dtype == np.float32 and got.shape == ref.shape == (80 * (amp.shape[1] - 1),)
    assert _rel(got, ref) < tol


def test_seeded_global_generator_is_the_default_phase():
    """audio_lib.py:255 draws from np.random; seeding it reproduces the oracle run."""
    import audio_lib
    amp = _amp_of_speech(4000, 2)
    np.random.seed(5)
    got = audio_lib.griffin_lim_alg(amp, 400, 80, num_iters=2, verbose=False)
    ref = vo.griffin_lim_alg(amp, 400, 80, num_iters=2, seed=5)
    assert _rel(got, ref) < 5e-5


def test_single_projection_step_from_common_state():
    """One fused kernel launch == oracle step, starting from a float32 state both sides share."""
    import audio_lib
    amp = _amp_of_speech(16000, 4)
    ph = vo.initial_phase(amp.shape, 1)
    w10 = audio_lib.griffin_lim_alg(amp, 400, 80, num_iters=10, verbose=False, phase0=ph)
    w11 = audio_lib.griffin_lim_alg(amp, 400, 80, num_iters=11, verbose=False, phase0=ph)
    step = vo.griffin_lim_step(w10.astype(np.float64), amp, 400, 80)
    assert _rel(w11, step) < 3e-5


def test_long_run_converges_like_the_oracle(capsys):
    import audio_lib
    amp = _amp_of_speech(16000, 6)
    ph = vo.initial_phase(amp.shape, 2)
    tr = []
    ref = vo.griffin_lim_alg(amp, 400, 80, num_iters=60, phase0=ph, trace=tr)
    got = audio_lib.griffin_lim_alg(amp, 400, 80, num_iters=60, verbose=True, phase0=ph)
    c_ref = vo.spectral_convergence(ref, amp, 400, 80)
    c_got = vo.spectral_convergence(got, amp, 400, 80)
    assert c_got < 0.2 and abs(c_got - c_ref) < 0.02 * max(c_ref, 1e-3) + 2e-4, (c_got, c_ref)
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 2e-2
    # verbose mode prints the reference's per-iteration line (audio_lib.py:263-264)
    lines = [l for l in capsys.readouterr().out.splitlines() if 'mrse_delta' in l]
    assert len(lines) == 59 and lines[0].startswith(' i=1  mrse_delta = ')
    vals = np.array([float(l.split('=')[-1]) for l in lines])
    assert np.allclose(vals[:5], tr[:5], rtol=2e-3) and np.allclose(vals, tr, rtol=0.1, atol=1e-6)


def test_generic_transform_sizes():
    """from_power_to_wav's own defaults (win 800, hop 40: audio_lib.py:281-282) and a padded window."""
    import audio_lib
    for n_fft, win, hop, L in ((800, 800, 40, 4000), (512, 400, 128, 6400)):
        amp = _amp_of_speech(L, 8, n_fft, hop, win)
        ph = vo.initial_phase(amp.shape, 4)
        ref = vo.griffin_lim_alg(amp, win, hop, num_iters=3, n_fft=n_fft, phase0=ph)
        got = audio_lib.griffin_lim_alg(amp, win, hop, num_iters=3, n_fft=n_fft, verbose=False, phase0=ph)
        assert got.shape == ref.shape and _rel(got, ref) < 2e-4, (n_fft, _rel(got, ref))


def test_ragged_batch_equals_single_utterances():
    import audio_lib
    rng = np.random.RandomState(0)
    frames = [37, 120, 64]
    Fmax = max(frames)
    amp = np.zeros((3, Fmax, 201), np.float32)
    ph = np.zeros((3, Fmax, 201), np.float32)
    for b, F in enumerate(frames):
        amp[b, :F] = _amp_of_speech(80 * (F - 1), 20 + b).T
        ph[b, :F] = rng.uniform(0, np.pi, (F, 201))
    amp[1, 100:] += 7.0                      # garbage beyond n_frames must not leak in
    wav = audio_lib.griffin_lim_batch(amp, [37, 100, 64], 400, 80, num_iters=8, phase0=ph).cpu().numpy()
    assert wav.shape == (3, 80 * (Fmax - 1))
    for b, F in enumerate([37, 100, 64]):
        single = audio_lib.griffin_lim_batch(amp[b:b + 1, :F], None, 400, 80, num_iters=8, phase0=ph[b:b + 1, :F])
        single = single.cpu().numpy()[0]
        assert np.array_equal(wav[b, :80 * (F - 1)], single)                 # bit-identical
        assert not wav[b, 80 * (F - 1):].any()
        ref = vo.griffin_lim_alg(amp[b, :F].T.astype(np.float64), 400, 80, num_iters=8, phase0=ph[b, :F].T)
        assert _rel(single, ref) < 1e-3


@pytest.mark.parametrize('realse', [1.0, 1.25])
def test_from_power_to_wav_matches_oracle(realse):
    import audio_lib
    kw = dict(sr=16000, pre_emphasis=0.97, hop_length=80, win_length=400, n_mels=80, n_mfcc=40, n_fft=None,
              window='hann', mfcc_normaleze_first_mfcc=True, mfcc_norm_factor=0.01, calc_mfcc_derivate=True,
              M_dB_norm_factor=0.01, P_dB_norm_factor=0.01, mean_abs_amp_norm=0.003, clip_output=True)
    y = fo.synth_speech(1, 12000, seed=3)[0]
    _, _, P = fo.calc_MFCC_input(y, **kw)                                    # [F, 201] like the decoder's y_stft
    ph = vo.initial_phase((201, P.shape[0]), 9)
    args = dict(P_dB_norm_factor=0.01, pre_emphasis=0.97, hop_length=80, win_length=400, mean_abs_amp_norm=0.045,
                n_iter=3, n_fft=None, realse=realse)
    ref = vo.from_power_to_wav(P, phase0=ph, **args)
    got = audio_lib.from_power_to_wav(P, verbose=False, phase0=ph, **args)
    assert got.shape == ref.shape == (80 * (P.shape[0] - 1),)
    assert abs(np.abs(got).mean() - 0.045) < 1e-6
    assert _rel(got, ref) < 5e-4


def test_inverse_preemphasis_kernel_is_the_iir_filter():
    """vc_inv_preemphasis_normalize on its own: y[n] = x[n] + c*y[n-1] over chunked carries."""
    import ctypes as C
    import _vc
    import audio_lib
    plan = audio_lib._get_voc_plan(400, 80, None)
    rng = np.random.RandomState(1)
    F = 1301
    L = 80 * (F - 1)
    x = rng.standard_normal((2, L)).astype(np.float32)
    d = torch.from_numpy(x).cuda()
    nf = torch.tensor([F, 700], dtype=torch.int32).cuda()
    _vc.check(_vc.lib().vc_inv_preemphasis_normalize(plan.handle, _vc.ptr(d), _vc.ptr(nf), 2, F, L, 0.97, 0.0,
                                                     _vc.current_stream()))
    got = d.cpu().numpy()
    for b, Lb in enumerate([L, 80 * 699]):
        ref = vo.calc_inv_preemphasis(x[b, :Lb].astype(np.float64), 0.97)
        assert _rel(got[b, :Lb], ref) < 2e-6
        assert np.array_equal(got[b, Lb:], x[b, Lb:])                        # untouched beyond the utterance


def test_round_trip_through_the_front_end_full_size():
    """Size-independent property at conversion scale (60 s of audio, 200 iterations as test.py:87):
    features of the vocoded audio reproduce the power spectrogram it was made from."""
    import audio_lib
    kw = dict(sr=16000, pre_emphasis=0.97, hop_length=80, win_length=400, n_mels=80, n_mfcc=40, n_fft=None,
              window='hann', mfcc_normaleze_first_mfcc=True, mfcc_norm_factor=0.01, calc_mfcc_derivate=True,
              M_dB_norm_factor=0.01, P_dB_norm_factor=0.01, mean_abs_amp_norm=0.003, clip_output=True)
    y = fo.synth_speech(1, 16000 * 60, seed=12)[0]
    _, _, P = audio_lib.calc_MFCC_input(y, **kw)
    np.random.seed(0)
    wav = audio_lib.from_power_to_wav(P, 0.01, 0.97, 80, 400, 0.003, n_iter=200, n_fft=None, verbose=False)
    assert wav.shape == (80 * (P.shape[0] - 1),) and np.isfinite(wav).all()
    _, _, P2 = audio_lib.calc_MFCC_input(wav, **kw)
    n = min(len(P), len(P2))
    loud = P[:n] > 0.4                       # bins within 40 dB of the peak (P is dB/100, clipped to [0, 0.8])
    err = np.abs(P2[:n] - P[:n])[loud]
    assert np.median(err) < 0.01 and err.mean() < 0.02, (np.median(err), err.mean())   # < 1 dB / 2 dB


def test_bad_arguments_raise():
    import audio_lib
    import _vc
    amp = np.ones((201, 3))
    with pytest.raises(ValueError, match='reflect padding'):
        audio_lib.griffin_lim_alg(amp, 400, 80, num_iters=2, verbose=False)
    with pytest.raises(ValueError):
        audio_lib.griffin_lim_alg(np.ones((100, 50)), 400, 80, num_iters=2, verbose=False)
    with pytest.raises(_vc.VCError):
        audio_lib.griffin_lim_alg(np.ones((202, 50)), 401, 80, num_iters=2, verbose=False)
