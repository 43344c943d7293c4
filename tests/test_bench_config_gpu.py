"""Parity AT THE BENCHMARKED CONFIGURATION (BASELINE.json configs C3 / C4; /root/reference/encoder.py:78-123,
decoder.py:75-182, 447-465): the launches bench.py times -- shipped hyper-parameters (E = 256 / 512, K = 32, T = 400),
64 and 128 windows per launch, bf16 -- against the float64 oracle on a subset of the batch's windows (the oracle needs
seconds per window), and the float32 encoder at batch 64 against the committed golden vectors.

Stated end-to-end bf16 tolerance, derived BEFORE measuring:
  * a block (dense / conv / bank / highway chain / recurrence) in bf16 with float32 accumulation is within 3e-2 of
    max(1, |ref|max) of the oracle (tests/test_blocks_gpu.py) -- a 4-5 sigma bound on an error
    whose standard deviation is ~2^-9 x sqrt(2/3) x |activation| rms per rounding of an operand;
  * the decoder chains D = 14 such blocks (2 stages x {prenet, bank, 2 projections, highway chain, recurrence, dense});
    block errors are independent and pass through BatchNorm-ed, gated layers with gain ~1, so they add in quadrature:
    max |err| <= 3e-2 x sqrt(14) = 0.11 x max(1, |ref|max)   for y_mel and y_stft given the SAME posteriors
    (decoder-only bound), and the error's rms is ~1/5 of its maximum:  rms err <= 2.5e-2 x max(1, |ref| rms) ... (A)
  * end to end the decoder additionally sees the bf16 encoder's posterior drift (mean 2e-3, max 0.15, >= 97 % equal
    argmax: test_model_gpu.py).  A posterior shift d at one frame moves the stage-1 prenet input by d in <= 2 of 61
    columns, i.e. by <= sqrt(2) d |W1| ~ 0.11 d per unit against activations of ~|W1| ~ 0.08 -- up to 1.4 d relative
    AT THAT FRAME -- and the K = 32 bank spreads it over +-16 frames.  With 3 % of frames affected by d ~ 0.1 the
    end-to-end bound is therefore NOT the decoder-only one: stated as
    max |err| <= 0.25 x max(1, |ref|max), rms err <= 5e-2 x max(1, |ref| rms) ................................ (B)
    (the error on frames farther than 16 frames from any argmax flip is printed beside it; the decoder-only test
    below is what isolates (A): same posteriors on both sides).
The measured figures are printed (pytest -s) and recorded in DESIGN.md; (A)/(B) were not widened after measuring.

(A)/(B) say what bf16 arithmetic may cost; they are 40-86x above what the kernels actually deliver (round 2, MI355X:
y_mel max 2.9e-3 / rms 6.1e-4, y_stft max 2.0e-3 / rms 4.2e-4 -- the same decoder-only and end to end, at 64 and at
128 windows), so a change that made the decoder 30x worse would pass them.  Beside each derived bound the tests
therefore assert a REGRESSION bound at <= 5x the measured figures:  max |err| <= 1.5e-2, rms err <= 3e-3 ....... (R)
(absolute: the references' own scale is |ref|max ~0.5, rms ~0.15).  (R) is a tripwire for the kernels, not a claim
about bf16; if a deliberate numerical change moves the figures, re-measure and restate (R) with the change."""

REG_MAX, REG_RMS = 1.5e-2, 3e-3           # (R): <= 5x the round-2 measurement, see the module docstring
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import model_oracle as mo

pytestmark = pytest.mark.gpu

HP = os.path.join(ROOT, 'speech-cloner_amd', 'hp')


def _cfgs(golden_dir, dtype):
    enc_cfg = json.load(open(os.path.join(HP, 'encoder_cfg_d.json')))
    enc_cfg.update(is_training=False, model_path=os.path.join(golden_dir, 'enc_14_ckpt'), compute_dtype=dtype)
    dec_cfg = json.load(open(os.path.join(HP, 'decoder_cfg_d.json')))
    dec_cfg.update(is_training=False)
    return enc_cfg, dec_cfg


def _batch(golden_dir, n):
    """n feature windows with the statistics of real front-end output: the 3 golden windows (speech-like synthetic
    audio through the front-end oracle) at n different gains / time shifts."""
    g = np.load(os.path.join(golden_dir, 'encoder_fwd.npz'))
    rng = np.random.RandomState(21)
    xs = []
    for i in range(n):
        w = g['x'][i % 3]
        xs.append(np.roll(w, int(rng.randint(0, 400)), axis=0) * rng.uniform(0.5, 1.0))
    return np.stack(xs).astype(np.float32), g


_ORACLE = {}


def _oracle_windows(kind, x, idx, we, wd, enc_cfg, dec_cfg):
    """float64 oracle on single windows, cached per (kind, window index): the 64- and the 128-window cases share windows
    (the batch is a fixed sequence), and one window takes ~10 s of CPU.  kind 'e2e': encoder -> decoder; 'dec': the
    decoder on the oracle's posteriors rounded to bf16 (the bf16 decoder's input format).  Inference is per window."""
    out = []
    for i in idx:
        key = (kind, int(i))
        if key not in _ORACLE:
            with torch.no_grad():
                _, pr, cls, _ = mo.encoder_forward(torch.from_numpy(x[i:i + 1]).double(), we, enc_cfg)
                ppg = pr if kind == 'e2e' else pr.float().bfloat16().double()
                ym, ys = mo.decoder_forward(ppg, wd, dec_cfg)
            _ORACLE[key] = (pr[0].numpy(), cls[0].numpy(), ym[0].numpy(), ys[0].numpy())
        out.append(_ORACLE[key])
    return tuple(np.stack([o[j] for o in out]) for j in range(4))


def _stats(dev, ref):
    e = np.abs(dev.astype(np.float64) - ref)
    return dict(max=float(e.max()), rms=float(np.sqrt((e ** 2).mean())), ref_max=float(np.abs(ref).max()),
                ref_rms=float(np.sqrt((ref ** 2).mean())))


@pytest.fixture(scope='module')
def models(golden_dir):
    from encoder import encoder_spec_phn
    from decoder import decoder_specs
    import contextlib
    import io
    enc_cfg, dec_cfg = _cfgs(golden_dir, 'bfloat16')
    with contextlib.redirect_stdout(io.StringIO()):
        enc = encoder_spec_phn(enc_cfg, None)
        dec = decoder_specs(dec_cfg, None, enc)
    wd = mo.init_weights(dec_cfg, 'decoder', seed=2, perturb_bn=True)
    dec.store.load_dict(dict(wd), strict=False)
    import tf_bundle
    we = tf_bundle.read_bundle(os.path.join(golden_dir, 'enc_14_ckpt', 'encoder-136512'))
    we = {k: v for k, v in we.items() if k.startswith('encoder/')}
    return enc, dec, enc_cfg, dec_cfg, mo.to_torch(we, torch.float64), mo.to_torch(wd, torch.float64)


@pytest.mark.parametrize('W', [64, 128])
def test_bf16_full_path_at_bench_batch_vs_oracle(models, golden_dir, W):
    """encode -> decode of W windows in ONE launch sequence (the kernels a W-window batch selects: MFMA recurrences,
    paired bank tiles with the XCD-aware block map, the projection on bank tiles, 256-row conv256 blocks, fused
    prenets / highway chains / encoder front), bf16, shipped sizes, vs the float64 oracle on 3 of the W windows."""
    enc, dec, enc_cfg, dec_cfg, we, wd = models
    x, _ = _batch(golden_dir, W)
    r = dec.predict(x, batch_size=W, n_streams=1)
    assert r.y_mel.shape == (W, 400, 80) and r.y_stft.shape == (W, 400, 201) and r.y_phn.shape == (W, 400, 61)
    sub = [0, 63, W - 1] if W > 64 else [0, 31, 63]       # (windows 0 and 63 serve both batch sizes)
    pr, cls, ym, ys = _oracle_windows('e2e', x, sub, we, wd, enc_cfg, dec_cfg)
    # posteriors: the bound test_model_gpu.py states for the bf16 encoder
    ep = np.abs(r.y_phn[sub] - pr)
    flips = np.argmax(r.y_phn[sub], -1) != cls
    assert ep.max() < 0.15 and ep.mean() < 2e-3 and flips.mean() < 0.03, (ep.max(), ep.mean(), flips.mean())
    # frames farther than 16 frames (the widest filter's reach) from any argmax flip
    far = np.ones_like(flips)
    for n_, t_ in zip(*np.nonzero(flips)):
        far[n_, max(0, t_ - 16):t_ + 17] = False
    for name, dev, ref in (('y_mel', r.y_mel[sub], ym), ('y_stft', r.y_stft[sub], ys)):
        s_all = _stats(dev, ref)
        s_far = _stats(dev[far], ref[far]) if far.any() else dict(max=0.0, rms=0.0)
        print('W=%d %s end-to-end: max %.4f rms %.5f (ref max %.3f rms %.3f); away from argmax flips (%.1f %% of frames): '
              'max %.4f rms %.5f' % (W, name, s_all['max'], s_all['rms'], s_all['ref_max'], s_all['ref_rms'],
                                     100 * far.mean(), s_far['max'], s_far['rms']))
        assert s_all['max'] <= 0.25 * max(1.0, s_all['ref_max']), (name, s_all)              # (B)
        assert s_all['rms'] <= 5e-2 * max(1.0, s_all['ref_rms']), (name, s_all)
        assert s_all['max'] <= REG_MAX and s_all['rms'] <= REG_RMS, (name, s_all)             # (R)


@pytest.mark.parametrize('W', [64, 128])
def test_bf16_decoder_at_bench_batch_given_oracle_posteriors(models, golden_dir, W):
    """The decoder alone at the bench's batch: both stages in bf16 on W windows fed with the ORACLE's posteriors
    (rounded to bf16, the decoder's input format) -- isolates the decoder's accumulated bf16 error, bound (A)."""
    import modules
    enc, dec, enc_cfg, dec_cfg, we, wd = models
    x, _ = _batch(golden_dir, W)
    sub = [0, 63, W - 1] if W > 64 else [0, 31, 63]
    pr_np, _, ym_np, ys_np = _oracle_windows('dec', x, sub, we, wd, enc_cfg, dec_cfg)
    pr_sub = torch.from_numpy(pr_np)
    # the W-window batch: the HIP encoder's own float32 posteriors everywhere, the oracle's at the compared windows
    ppg = torch.zeros((W, 400, 64), dtype=torch.float32, device='cuda')
    ppg[:, :, :61] = torch.from_numpy(enc.predict(x, batch_size=W)).cuda()
    ppg[sub, :, :61] = pr_sub.float().cuda()
    y_mel, y_stft = dec.forward_from_ppg(modules.convert(ppg, torch.bfloat16))
    torch.cuda.synchronize()
    for name, dev, ref in (('y_mel', y_mel[sub].cpu().numpy(), ym_np), ('y_stft', y_stft[sub].cpu().numpy(), ys_np)):
        s = _stats(dev, ref)
        print('W=%d %s decoder-only bf16: max %.4f rms %.5f (ref max %.3f rms %.3f)' % (W, name, s['max'], s['rms'], s['ref_max'], s['ref_rms']))
        assert s['max'] <= 0.11 * max(1.0, s['ref_max']), (name, s)                            # (A)
        assert s['rms'] <= 2.5e-2 * max(1.0, s['ref_rms']), (name, s)
        assert s['max'] <= REG_MAX and s['rms'] <= REG_RMS, (name, s)                          # (R)


def test_f32_full_path_at_64_windows_vs_oracle(golden_dir):
    """The reference's own arithmetic type at the launch size bench.py's "f32" figure times: encode -> decode of 64
    windows in ONE launch sequence, float32, shipped sizes (/root/reference/decoder.py:75-182, 447-465) -- tile and
    kernel selection depend on M = 25,600 rows (128-row f32 tiles over 200 row blocks, 64-window recurrences) --
    against the float64 oracle on 3 of the 64 windows.  Tolerances as test_model_gpu.py states them for 2 windows:
    posteriors 2e-5, y_mel / y_stft 1e-3 absolute."""
    from encoder import encoder_spec_phn
    from decoder import decoder_specs
    import contextlib
    import io
    import tf_bundle
    enc_cfg, dec_cfg = _cfgs(golden_dir, 'float32')
    with contextlib.redirect_stdout(io.StringIO()):
        enc = encoder_spec_phn(enc_cfg, None)
        dec = decoder_specs(dec_cfg, None, enc)
    wd = mo.init_weights(dec_cfg, 'decoder', seed=2, perturb_bn=True)
    dec.store.load_dict(dict(wd), strict=False)
    we = tf_bundle.read_bundle(os.path.join(golden_dir, 'enc_14_ckpt', 'encoder-136512'))
    we = mo.to_torch({k: v for k, v in we.items() if k.startswith('encoder/')}, torch.float64)
    W = 64
    x, _ = _batch(golden_dir, W)
    r = dec.predict(x, batch_size=W, n_streams=1)
    assert r.y_mel.shape == (W, 400, 80) and r.y_stft.shape == (W, 400, 201) and r.y_phn.shape == (W, 400, 61)
    sub = [0, 31, 63]
    pr, cls, ym, ys = _oracle_windows('e2e', x, sub, we, mo.to_torch(wd, torch.float64), enc_cfg, dec_cfg)   # (same weights
    ep = np.abs(r.y_phn[sub] - pr).max()                                            # and windows as the bf16 tests: cached)
    assert ep < 2e-5, ep
    for name, dev, ref in (('y_mel', r.y_mel[sub], ym), ('y_stft', r.y_stft[sub], ys)):
        s = _stats(dev, ref)
        print('f32 W=64 %s: max %.3e rms %.3e (ref max %.3f)' % (name, s['max'], s['rms'], s['ref_max']))
        assert s['max'] < 1e-3, (name, s)
    # the other windows of the batch are the same three signals at other gains / shifts: finite, and no window is a
    # copy of another (a tile-mapping fault would show as repeated or zero rows)
    assert np.isfinite(r.y_stft).all() and np.abs(r.y_stft).reshape(W, -1).max(1).min() > 0
    assert len({r.y_mel[i].tobytes() for i in range(W)}) == W


def test_f32_encoder_at_batch_64_vs_golden(golden_dir):
    """BASELINE configs[2] (C3): encoder_spec_phn forward, float32, 64 windows in one launch (tile selection depends
    on M) against the committed float64 golden vectors (real enc_14 weights), same tolerances as the 3-window test:
    logits 1e-4, posteriors 2e-5, argmax exact where the oracle's top-2 margin exceeds 1e-4."""
    from encoder import encoder_spec_phn
    enc_cfg, _ = _cfgs(golden_dir, 'float32')
    enc = encoder_spec_phn(enc_cfg, None)
    enc.restore()
    g = np.load(os.path.join(golden_dir, 'encoder_fwd.npz'))
    idx = np.arange(64) % 3
    x = g['x'][idx]
    o = enc.run([enc.y_logits, enc.y_pred, enc.y_pred_class, enc.CBHG_out], {enc.inputs: x})
    y_logits, y_pred, y_cls, cbhg = o
    assert y_logits.shape == (64, 400, 61) and y_cls.shape == (64, 400) and y_cls.dtype == np.int32
    assert np.abs(cbhg - g['CBHG_out'][idx]).max() < 1e-4
    assert np.abs(y_logits - g['y_logits'][idx]).max() < 1e-4 * max(1.0, np.abs(g['y_logits']).max())
    assert np.abs(y_pred - g['y_pred'][idx]).max() < 2e-5
    srt = np.sort(g['y_logits'][idx], -1)
    safe = (srt[..., -1] - srt[..., -2]) > 1e-4
    assert safe.mean() > 0.99 and np.array_equal(y_cls[safe], g['y_pred_class'][idx][safe])
    # predict() at batch_size 64 is the same launch
    assert np.array_equal(enc.predict(x, batch_size=64), y_pred)
