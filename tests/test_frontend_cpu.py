"""CPU-side checks of the front-end: the oracle against the committed golden vectors and the
structural pins SURVEY.md records, the 400-point DFT algebra compiled for the host, and the
native library's host-side tables against the oracle's.  No GPU needed."""
import os
import subprocess

import numpy as np
import pytest

from conftest import FE_KW, FE_KW_GENERIC, ROOT
from oracle import frontend_oracle as fo


def test_mel_matrix_structure_pins():
    """391 non-zeros, 1..14 per filter, bins 0 and 200 unused."""
    M = fo.mel_filterbank(16000, 400, 80)
    nz = M > 0
    assert M.shape == (80, 201) and nz.sum() == 391
    assert nz.sum(1).min() == 1 and nz.sum(1).max() == 14
    assert nz[:, 0].sum() == 0 and nz[:, 200].sum() == 0
    # each filter's support is one contiguous run (the kernel's sparse rows rely on it)
    for m in range(80):
        idx = np.flatnonzero(nz[m])
        assert np.array_equal(idx, np.arange(idx[0], idx[-1] + 1))


def test_dct_orthonormal():
    D = fo.dct_basis(40, 80)
    assert np.abs(D @ D.T - np.eye(40)).max() < 1e-12
    assert np.allclose(D[0], 1 / np.sqrt(80))


def test_stft_contract_and_parseval():
    rng = np.random.RandomState(0)
    for L in (201, 399, 400, 401, 3999, 4000, 4079, 4080):
        y = rng.standard_normal(L)
        S = fo.stft(y, 400, 80, 400)
        assert S.shape == (201, 1 + L // 80) and S.dtype == np.complex64   # F = 1 + L//hop
    # Parseval on one frame: sum |X_k|^2 over the full spectrum == N * sum (w x)^2
    y = rng.standard_normal(4000)
    S = fo.stft(y, 400, 80, 400).astype(np.complex128)
    w = fo.fft_window('hann', 400, 400)
    yp = np.pad(y, 200, mode='reflect')
    f = 7
    seg = w * yp[80 * f: 80 * f + 400]
    full = np.abs(S[0, f]) ** 2 + np.abs(S[200, f]) ** 2 + 2 * (np.abs(S[1:200, f]) ** 2).sum()
    assert abs(full - 400 * (seg ** 2).sum()) / full < 1e-5


def test_oracle_matches_golden_default(golden_dir):
    g = np.load(os.path.join(golden_dir, 'frontend_default.npz'))
    outs = fo.calc_MFCC_input_batch(g['wav'], g['lens'], **FE_KW)
    for b in range(2):
        for name, o in zip(('mfcc', 'mel', 'pdb'), outs):
            ref = g['%s%d' % (name, b)]
            assert o[b].dtype == np.float32 and o[b].shape == ref.shape
            assert o[b].shape[0] == 1 + int(g['lens'][b]) // 80
            assert np.array_equal(o[b], ref), name
    assert outs[0][0].shape[1] == 80 and outs[2][0].shape[1] == 201
    assert outs[0][0].min() >= -1 and outs[0][0].max() <= 1
    # delta features: first and last rows are zero (audio_lib.py:227)
    assert np.all(outs[0][0][0, 40:] == 0) and np.all(outs[0][0][-1, 40:] == 0)
    # frame-0 first coefficient is zeroed by the normalisation (audio_lib.py:221)
    assert outs[0][0][0, 0] == 0


def test_oracle_matches_golden_generic(golden_dir):
    g = np.load(os.path.join(golden_dir, 'frontend_generic.npz'))
    o = fo.calc_MFCC_input(g['wav'][0], **FE_KW_GENERIC)
    assert o[0].shape == (101, 20) and o[1].shape == (101, 64) and o[2].shape == (101, 257)
    for name, a in zip(('mfcc', 'mel', 'pdb'), o):
        assert np.array_equal(a, g[name]), name


def test_gain_invariance():
    """audio_lib.py:125-126 renormalises the amplitude, so features do not depend on input gain."""
    wav = fo.synth_speech(1, 8000, seed=3)[0]
    a = fo.calc_MFCC_input(wav, **FE_KW)
    b = fo.calc_MFCC_input((wav * 0.25).astype(np.float32), **FE_KW)
    for x, y in zip(a, b):
        assert np.abs(x - y).max() < 2e-6


def test_phn_target_length_contract():
    phn_v = [(0, 3000, 'a'), (3000, 6000, 'b'), (6000, 8000, 'c')]
    conv = {'a': [1, 0, 0], 'b': [0, 1, 0], 'c': [0, 0, 1]}
    t = fo.calc_PHN_target(8000, phn_v, conv, hop_length=80, win_length=400)
    assert t.shape == (1 + 8000 // 80, 3) and t.dtype == np.int32
    import audio_lib
    t2 = audio_lib.calc_PHN_target(np.zeros(8000), phn_v, conv, hop_length=80, win_length=400)
    assert np.array_equal(t, t2)


def test_dft400_algebra_on_host(tmp_path):
    """Compiles csrc/fe_dft400.h with g++ and checks the 25x16 split against a naive DFT."""
    exe = str(tmp_path / 'test_dft400')
    subprocess.check_call(['g++', '-O2', '-I', os.path.join(ROOT, 'speech-cloner_amd', 'csrc'),
                           os.path.join(ROOT, 'tests', 'cpp', 'test_dft400.cpp'), '-o', exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_native_host_tables_match_oracle():
    import _vc
    if not os.path.exists(_vc.LIB_PATH):
        pytest.fail('libvc_hip.so not built (run __graft_entry__.build())')
    import audio_lib
    for (sr, n_fft, n_mels, n_mfcc) in ((16000, 400, 80, 40), (16000, 512, 64, 20), (22050, 1024, 128, 13)):
        mel, dct = audio_lib.host_tables(sr, n_fft, n_mels, n_mfcc)
        assert np.abs(mel - fo.mel_filterbank(sr, n_fft, n_mels)).max() < 1e-12
        assert np.abs(dct - fo.dct_basis(n_mfcc, n_mels)).max() < 1e-12


def test_preemphasis_helpers():
    import audio_lib
    x = np.random.RandomState(1).standard_normal(1000)
    y = audio_lib.calc_preemphasis(x, 0.97)
    assert np.allclose(y, fo.calc_preemphasis(x, 0.97), atol=1e-12)
    assert np.allclose(audio_lib.calc_inv_preemphasis(y, 0.97), x, atol=1e-9)
