"""Front-end parity on the MI355X: HIP path (through the C ABI) vs the CPU oracle and the
committed golden vectors.

Tolerances (features live in [-1, 1] after the 0.01 dB scaling):
  MFCC / delta  abs 1e-4      mel dB  abs 1e-4      power dB  abs 2e-4
The device computes in fp32 where librosa mixes f64/f32 (oracle header), so bit equality is
not expected; frame counts and zero padding are exact."""
import os

import numpy as np
import pytest

from conftest import FE_KW, FE_KW_GENERIC
from oracle import frontend_oracle as fo

pytestmark = pytest.mark.gpu

TOL = {'mfcc': 1e-4, 'mel': 1e-4, 'pdb': 2e-4}


def _cmp(dev, ref, tol, what):
    dev = np.asarray(dev)
    assert dev.shape == ref.shape and dev.dtype == np.float32, (what, dev.shape, ref.shape)
    err = np.abs(dev.astype(np.float64) - ref.astype(np.float64)).max()
    assert err <= tol, '%s: max abs err %.3e > %.1e' % (what, err, tol)
    return err


def test_golden_default_ragged(golden_dir):
    import audio_lib
    g = np.load(os.path.join(golden_dir, 'frontend_default.npz'))
    mfcc, mel, pdb = audio_lib.calc_MFCC_input_batch(g['wav'], g['lens'], **FE_KW)
    Fmax = 1 + g['wav'].shape[1] // 80
    assert mfcc.shape == (2, Fmax, 80) and mel.shape == (2, Fmax, 80) and pdb.shape == (2, Fmax, 201)
    for b in range(2):
        F = 1 + int(g['lens'][b]) // 80                       # exact integer contract
        for name, t in (('mfcc', mfcc), ('mel', mel), ('pdb', pdb)):
            _cmp(t[b, :F].cpu().numpy(), g['%s%d' % (name, b)], TOL[name], '%s[%d]' % (name, b))
            assert float(t[b, F:].abs().max()) == 0.0 if F < Fmax else True   # zero padding rows


def test_golden_generic_path(golden_dir):
    import audio_lib
    g = np.load(os.path.join(golden_dir, 'frontend_generic.npz'))
    out = audio_lib.calc_MFCC_input(g['wav'][0], **FE_KW_GENERIC)
    # unit norm factors: values are raw dB (|x| ~ 100) -> scale the tolerance by 100
    for name, o in zip(('mfcc', 'mel', 'pdb'), out):
        _cmp(o, g[name], 100 * TOL[name], name)


def test_single_utterance_api_matches_oracle():
    import audio_lib
    for L, seed in ((16000, 1), (4079, 2), (401, 3), (51200, 4)):
        wav = fo.synth_speech(1, L, seed=seed)[0]
        dev = audio_lib.calc_MFCC_input(wav, **FE_KW)
        ref = fo.calc_MFCC_input(wav, **FE_KW)
        assert all(isinstance(d, np.ndarray) for d in dev)
        for name, d, r in zip(('mfcc', 'mel', 'pdb'), dev, ref):
            assert d.shape[0] == 1 + L // 80
            _cmp(d, r, TOL[name], '%s L=%d' % (name, L))


def test_noise_and_flag_variants():
    import audio_lib
    rng = np.random.RandomState(5)
    wav = rng.standard_normal(12345).astype(np.float32)
    for kw in (dict(FE_KW), dict(FE_KW, calc_mfcc_derivate=False), dict(FE_KW, clip_output=False),
               dict(FE_KW, mfcc_normaleze_first_mfcc=False, pre_emphasis=0.0),
               dict(FE_KW, hop_length=40, n_mels=128), dict(FE_KW, window='hamming')):
        dev = audio_lib.calc_MFCC_input(wav, **kw)
        ref = fo.calc_MFCC_input(wav, **kw)
        for name, d, r in zip(('mfcc', 'mel', 'pdb'), dev, ref):
            _cmp(d, r, TOL[name], '%s %s' % (name, sorted(kw.items())[:0]))


def test_config2_full_size_properties():
    """BASELINE config 2 (batch 32 x 4 s @ 16 kHz): size-independent properties at full size +
    oracle comparison on a sample of utterances."""
    import torch
    import audio_lib
    wav = fo.synth_speech(32, 64000, seed=0)
    d_wav = torch.from_numpy(wav).cuda()
    mfcc, mel, pdb = audio_lib.calc_MFCC_input_batch(d_wav, None, **FE_KW)
    assert mfcc.shape == (32, 801, 80) and mel.shape == (32, 801, 80) and pdb.shape == (32, 801, 201)
    assert torch.isfinite(mfcc).all() and torch.isfinite(mel).all() and torch.isfinite(pdb).all()
    # clip range, min-shift => every utterance's minimum is exactly 0, top_db => max <= 0.8
    assert float(mfcc.abs().max()) <= 1.0
    assert torch.all(pdb.amin(dim=(1, 2)) == 0) and torch.all(mel.amin(dim=(1, 2)) == 0)
    assert float(pdb.max()) <= 0.8 + 1e-6 and float(mel.max()) <= 0.8 + 1e-6
    assert torch.all(mfcc[:, 0, 40:] == 0) and torch.all(mfcc[:, -1, 40:] == 0) and torch.all(mfcc[:, 0, 0] == 0)
    # gain invariance (audio_lib.py:125-126): scaling the audio does not change the features
    m2, l2, p2 = audio_lib.calc_MFCC_input_batch(d_wav * 0.125, None, **FE_KW)
    assert float((m2 - mfcc).abs().max()) < 1e-5 and float((p2 - pdb).abs().max()) < 1e-5
    # batching is transparent: utterance b alone == row b of the batch, bit for bit
    m1, l1, p1 = audio_lib.calc_MFCC_input_batch(d_wav[5:6].contiguous(), None, **FE_KW)
    assert torch.equal(m1[0], mfcc[5]) and torch.equal(l1[0], mel[5]) and torch.equal(p1[0], pdb[5])
    for b in (0, 13, 31):
        ref = fo.calc_MFCC_input(wav[b], **FE_KW)
        for name, t, r in zip(('mfcc', 'mel', 'pdb'), (mfcc, mel, pdb), ref):
            _cmp(t[b].cpu().numpy(), r, TOL[name], '%s[%d]' % (name, b))


def test_ragged_tile_edges_and_narrow_dynamic_range():
    """The two-launch path of the shipped configuration at its seams: utterances whose frame counts sit on and around
    the 14-frame feature tiles and the 16-frame statistics tiles (F = 3, 14, 15, 16, 17, 29), the shortest legal one, and
    white noise -- whose spectrum spans far less than top_db = 80 dB, so the min shift acts on the TRUE minimum (not the
    floor) and must still map it to exactly 0, as `x - x.min()` does in the reference (audio_lib.py:230-235)."""
    import torch
    import audio_lib
    rng = np.random.RandomState(11)
    lens = [201, 1119, 1120, 1279, 1280, 2319, 8000]
    L = max(lens)
    wav = np.zeros((len(lens), L), np.float32)
    for b, n in enumerate(lens):
        wav[b, :n] = rng.standard_normal(n).astype(np.float32) * (0.01 + 0.3 * b)
    mfcc, mel, pdb = audio_lib.calc_MFCC_input_batch(torch.from_numpy(wav).cuda(), lens, **FE_KW)
    Fmax = 1 + L // 80
    for b, n in enumerate(lens):
        F = 1 + n // 80
        ref = fo.calc_MFCC_input(wav[b, :n], **FE_KW)
        for name, t, r in zip(('mfcc', 'mel', 'pdb'), (mfcc, mel, pdb), ref):
            _cmp(t[b, :F].cpu().numpy(), r, TOL[name], '%s len %d' % (name, n))
            if F < Fmax:
                assert float(t[b, F:].abs().max()) == 0.0
        assert float(pdb[b, :F].min()) == 0.0 and float(mel[b, :F].min()) == 0.0, n      # exact, floor or not
        assert float(ref[1].min()) == 0.0 and float(ref[1].max()) < 0.6     # the oracle agrees: the mel range is far inside 80 dB
        assert float(mfcc[b, 0, 0]) == 0.0


def test_stored_rows_can_be_fewer_than_the_frames():
    """out_frames (vc_hip.h out_rows): the first R frames of every utterance, bit-identical to the same rows of the full
    result -- the frames that are not stored still count for the utterance's max / min / mean|x| (the reference computes
    the whole utterance, then reshapes to windows: test.py:121-123) -- at R on and off the 14-frame tile grid, ragged
    batch included; [B, 800, 80] then is the encoder's [2B, 400, 80] window batch without a copy."""
    import torch
    import _vc
    import audio_lib
    rng = np.random.RandomState(5)
    lens = [64000, 64000, 40000, 1119, 63999, 64000]
    wav = np.zeros((len(lens), 64000), np.float32)
    for b, n in enumerate(lens):
        wav[b, :n] = rng.standard_normal(n).astype(np.float32) * 0.05 * (b + 1)
    wav[1, 63960:] *= 40.0                      # utterance 1's maximum sits in frame 800, which R = 800 does not store
    d = torch.from_numpy(wav).cuda()
    full = audio_lib.calc_MFCC_input_batch(d, lens, **FE_KW)
    assert full[0].shape[1] == 801
    for R in (800, 795, 14, 1):
        part = audio_lib.calc_MFCC_input_batch(d, lens, out_frames=R, **FE_KW)
        for name, f, q in zip(('mfcc', 'mel', 'pdb'), full, part):
            assert q.shape[:2] == (len(lens), R) and q.is_contiguous()
            assert torch.equal(q, f[:, :R]), (name, R)
    win = audio_lib.calc_MFCC_input_batch(d, lens, out_frames=800, **FE_KW)[0].view(2 * len(lens), 400, 80)
    assert win.is_contiguous() and torch.equal(win[3], full[0][1, 400:800])
    with pytest.raises(ValueError):
        audio_lib.calc_MFCC_input_batch(d, lens, out_frames=802, **FE_KW)
    with pytest.raises(_vc.VCError):            # the general-configuration path keeps raw tiles in the outputs: all rows or none
        audio_lib.calc_MFCC_input_batch(d, lens, out_frames=400, **FE_KW_GENERIC)


def test_one_launch_form_equals_two_pass_and_never_depends_on_waiting():
    """The shipped configuration runs as ONE launch (csrc/vc_frontend400.hip, fe400_fused_kernel: every frame transformed
    once; a block publishes its tile record, the utterance's last arriver reduces the records to one line, every block
    waits for that line and finishes from LDS).
      * against the two-launch form (option fe_fused = 0) the extremes are identical and sum|x| differs only in
        summation order: features equal to 2e-6, and both within the front-end tolerances of the oracle;
      * a block whose poll runs out computes the utterance's records itself: with fe_fused_spin = 0 EVERY block takes that
        path -- walk all tiles, restore its own -- and the result must be bit-identical to the waiting path;
      * bit-identical from run to run with NaN-poisoned LDS / workspace and an unrelated stream keeping CUs busy (uneven
        arrival of an utterance's tiles)."""
    import torch
    import _vc
    import audio_lib
    import modules
    from conftest import poison_gpu_state
    rng = np.random.RandomState(17)
    lens = [16000, 15999, 1119, 8000, 201, 12345]
    L = max(lens)
    wav = np.zeros((len(lens), L), np.float32)
    for b, n in enumerate(lens):
        wav[b, :n] = fo.synth_speech(1, n, seed=20 + b)[0] * (0.2 + 0.3 * b)
    d = torch.from_numpy(wav).cuda()
    _vc.set_option('fe_fused', 1)
    try:
        _one_launch_checks(d, wav, lens)
    finally:
        _vc.set_option('fe_fused', -1)


def _one_launch_checks(d, wav, lens):
    import torch
    import _vc
    import audio_lib
    import modules
    from conftest import poison_gpu_state
    fused = audio_lib.calc_MFCC_input_batch(d, lens, **FE_KW)
    with _vc.options(fe_fused=0):
        two = audio_lib.calc_MFCC_input_batch(d, lens, **FE_KW)
    for name, f_, t_ in zip(('mfcc', 'mel', 'pdb'), fused, two):
        assert torch.isfinite(f_).all()
        assert float((f_ - t_).abs().max()) < 2e-6, name
    for b in (0, 2, 4):
        F = 1 + lens[b] // 80
        ref = fo.calc_MFCC_input(wav[b, :lens[b]], **FE_KW)
        for name, t, r in zip(('mfcc', 'mel', 'pdb'), fused, ref):
            _cmp(t[b, :F].cpu().numpy(), r, TOL[name], 'one launch %s len %d' % (name, lens[b]))
    # nobody waits: every block walks its utterance
    with _vc.options(fe_fused_spin=0):
        poison_gpu_state()
        alone = audio_lib.calc_MFCC_input_batch(d, lens, **FE_KW)
        part = audio_lib.calc_MFCC_input_batch(d, lens, out_frames=100, **FE_KW)
    for name, f_, a_, p_ in zip(('mfcc', 'mel', 'pdb'), fused, alone, part):
        assert torch.equal(f_, a_), name
        assert torch.equal(f_[:, :100], p_), name
    # the benchmark's batch, under uneven load, repeated
    big = torch.from_numpy(fo.synth_speech(32, 64000, seed=3)).cuda()
    ref = audio_lib.calc_MFCC_input_batch(big, None, out_frames=800, **FE_KW)
    noise_stream = torch.cuda.Stream()
    noise_store = modules.VariableStore('bfloat16')
    noise_x = torch.randn(16, 400, 256, device='cuda').to(torch.bfloat16)
    for rep in range(3):
        poison_gpu_state()
        if rep:
            with torch.cuda.stream(noise_stream), modules.variable_store(noise_store), modules.variable_scope('noise'):
                for _ in range(4):
                    modules.conv1d_banks(noise_x, K=32, is_training=False)
        again = audio_lib.calc_MFCC_input_batch(big, None, out_frames=800, **FE_KW)
        for name, f_, a_ in zip(('mfcc', 'mel', 'pdb'), ref, again):
            assert torch.equal(f_, a_), (name, rep)
    torch.cuda.synchronize()


def test_errors_are_loud():
    import torch
    import _vc
    import audio_lib
    with pytest.raises(_vc.VCError):
        audio_lib.calc_MFCC_input(np.zeros(100, dtype=np.float32), **FE_KW)      # L <= n_fft//2
    with pytest.raises(ValueError):
        audio_lib.calc_MFCC_input_batch(torch.zeros(2, 8000).cuda(), [8000, 9000], **FE_KW)
