"""Encoder / decoder model objects on the MI355X vs the committed golden vectors (float64 oracle
outputs with the reference's real enc_14 weights, tools/make_golden.py) and vs the oracle run
live on seeded inputs."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import model_oracle as mo

pytestmark = pytest.mark.gpu

HP = os.path.join(ROOT, 'speech-cloner_amd', 'hp')


def _enc_cfg(golden_dir, dtype='float32'):
    cfg = json.load(open(os.path.join(HP, 'encoder_cfg_d.json')))
    cfg['is_training'] = False
    cfg['model_path'] = os.path.join(golden_dir, 'enc_14_ckpt')
    cfg['compute_dtype'] = dtype
    return cfg


def test_encoder_restore_and_golden_forward(golden_dir, capsys):
    from encoder import encoder_spec_phn
    enc = encoder_spec_phn(_enc_cfg(golden_dir), None)
    enc.restore()
    assert 'Restored:' in capsys.readouterr().out
    assert enc.i_global_step == 136512 and enc.i_epoch == 947
    g = np.load(os.path.join(golden_dir, 'encoder_fwd.npz'))
    o = enc.run([enc.y_logits, enc.y_pred, enc.y_pred_class, enc.CBHG_out], {enc.inputs: g['x']})
    y_logits, y_pred, y_cls, cbhg = o
    assert y_logits.shape == (3, 400, 61) and y_cls.shape == (3, 400) and y_cls.dtype == np.int32
    # float32 tolerances (SURVEY.md section 8c suggests logits 1e-4, softmax 1e-5; the f32 MFMA
    # path sums K = 2304 products per output in a different order than the oracle, which moves
    # probabilities near 1 by up to ~1.2e-5, so the softmax bound is stated as 2e-5)
    assert np.abs(cbhg - g['CBHG_out']).max() < 1e-4
    assert np.abs(y_logits - g['y_logits']).max() < 1e-4 * max(1.0, np.abs(g['y_logits']).max())
    assert np.abs(y_pred - g['y_pred']).max() < 2e-5
    # argmax: exact wherever the oracle's top-2 margin exceeds 1e-4
    srt = np.sort(g['y_logits'], -1)
    safe = (srt[..., -1] - srt[..., -2]) > 1e-4
    assert safe.mean() > 0.99 and np.array_equal(y_cls[safe], g['y_pred_class'][safe])
    # predict() chunks like the reference (batch_size windows per launch) and is chunk-invariant
    p1 = enc.predict(g['x'], batch_size=2)
    assert p1.shape == (3, 400, 61) and np.abs(p1 - y_pred).max() < 2e-6   # tile shape may differ with M


def test_encoder_bf16_within_tolerance(golden_dir):
    from encoder import encoder_spec_phn
    enc = encoder_spec_phn(_enc_cfg(golden_dir, 'bfloat16'), None)
    enc.restore()
    g = np.load(os.path.join(golden_dir, 'encoder_fwd.npz'))
    y_pred = enc.predict(g['x'])
    # bf16 activations/weights with f32 accumulation through 10 layers + a 400-step recurrence:
    # posteriors of ambiguous frames move by up to ~0.1; stated tolerance: max 0.15, mean 2e-3,
    # >= 97 % identical argmax
    err = np.abs(y_pred - g['y_pred'])
    assert err.max() < 0.15 and err.mean() < 2e-3, (err.max(), err.mean())
    assert (np.argmax(y_pred, -1) == g['y_pred_class']).mean() > 0.97


def test_restore_missing_checkpoint_exits(golden_dir, tmp_path, capsys):
    from encoder import encoder_spec_phn
    cfg = _enc_cfg(golden_dir)
    cfg['model_path'] = str(tmp_path)
    enc = encoder_spec_phn(cfg, None)
    with pytest.raises(SystemExit) as e:
        enc.restore()
    assert e.value.code == 1 and 'Model not found' in capsys.readouterr().err


def test_save_restore_roundtrip(golden_dir, tmp_path):
    from encoder import encoder_spec_phn
    import tf_bundle
    enc = encoder_spec_phn(_enc_cfg(golden_dir), None)
    enc.restore()
    enc.cfg_d['model_path'] = str(tmp_path / 'ck')
    enc.save(verbose=False)
    assert tf_bundle.latest_checkpoint(str(tmp_path / 'ck')).endswith('encoder-136512')
    a = tf_bundle.read_bundle(os.path.join(golden_dir, 'enc_14_ckpt', 'encoder-136512'))
    b = tf_bundle.read_bundle(str(tmp_path / 'ck' / 'encoder-136512'))
    assert set(a) == set(b) and all(np.array_equal(a[k], b[k]) for k in a)


def _load_small_decoder(golden_dir, dtype):
    from decoder import decoder_specs
    g = np.load(os.path.join(golden_dir, 'decoder_fwd_small.npz'))
    cfg = json.loads(str(g['cfg']))
    cfg['compute_dtype'] = dtype
    dec = decoder_specs(cfg, None, None)
    w = {k[2:]: g[k] for k in g.files if k.startswith('w:')}
    assert set(w) == set(dec.store.vars), set(w) ^ set(dec.store.vars)      # TF variable names line up
    dec.store.load_dict(w)
    return dec, g


def test_decoder_small_golden_f32(golden_dir):
    dec, g = _load_small_decoder(golden_dir, 'float32')
    r = dec.predict(g['ppg'])
    assert r._fields == ('y_mel', 'y_stft', 'y_phn')
    assert np.abs(r.y_mel - g['y_mel']).max() < 1e-4 and np.abs(r.y_stft - g['y_stft']).max() < 1e-4
    assert np.array_equal(r.y_phn, g['ppg'])
    assert dec.get_input_shape() == (40, 61)
    l = dec.exec_calc_metrics(g['ppg'], g['y_mel'], g['y_stft'])
    assert all(v < 1e-4 for v in l)                          # losses vs its own golden outputs ~ 0


def test_decoder_small_golden_bf16(golden_dir):
    dec, g = _load_small_decoder(golden_dir, 'bfloat16')
    r = dec.predict(g['ppg'])
    assert np.abs(r.y_mel - g['y_mel']).max() < 3e-2 and np.abs(r.y_stft - g['y_stft']).max() < 3e-2


def test_full_size_encode_decode_vs_oracle(golden_dir):
    """Shipped hyper-parameters (hp/*.json: E=256/512, K=32, 400 frames), 2 windows: the whole
    encode->decode chain on the device against the float64 oracle with identical weights."""
    from encoder import encoder_spec_phn
    from decoder import decoder_specs
    g = np.load(os.path.join(golden_dir, 'encoder_fwd.npz'))
    dec_cfg = json.load(open(os.path.join(HP, 'decoder_cfg_d.json')))
    dec_cfg['is_training'] = False
    enc = encoder_spec_phn(_enc_cfg(golden_dir), None)
    dec = decoder_specs(dec_cfg, None, enc)                   # restores the encoder (decoder.py:57)
    wd = mo.init_weights(dec_cfg, 'decoder', seed=2, perturb_bn=True)
    dec.store.load_dict(dict(wd), strict=False)
    x = g['x'][:2]
    r = dec.predict(x)
    assert r.y_mel.shape == (2, 400, 80) and r.y_stft.shape == (2, 400, 201) and r.y_phn.shape == (2, 400, 61)
    assert np.abs(r.y_phn - g['y_pred'][:2]).max() < 2e-5
    ym, ys = mo.decoder_forward(torch.from_numpy(g['y_pred'][:2]).double(), mo.to_torch(wd, torch.float64), dec_cfg)
    for dev, ref, nm in ((r.y_mel, ym, 'y_mel'), (r.y_stft, ys, 'y_stft')):
        err = np.abs(dev - ref.numpy()).max()
        assert err < 1e-3, '%s err %.3e' % (nm, err)          # mel/stft abs 1e-3 f32


def test_predict_stream_overlap_is_bit_identical(golden_dir):
    """decoder.predict issues independent window chunks on several HIP streams; the outputs must
    be exactly those of the sequential order."""
    from encoder import encoder_spec_phn
    from decoder import decoder_specs
    g = np.load(os.path.join(golden_dir, 'encoder_fwd.npz'))
    dec_cfg = json.load(open(os.path.join(HP, 'decoder_cfg_d.json')))
    dec_cfg['is_training'] = False
    enc = encoder_spec_phn(_enc_cfg(golden_dir, 'bfloat16'), None)
    dec = decoder_specs(dec_cfg, None, enc)
    x = np.concatenate([g['x'], g['x'][::-1] * 0.5, g['x'] * 0.25], 0)         # 9 windows -> 5 chunks of 2
    from conftest import poison_gpu_state
    a = dec.predict(x, batch_size=2, n_streams=1)
    for n_streams in (3, 1, 2):
        poison_gpu_state()                      # stale LDS / recycled allocations must not matter
        b = dec.predict(x, batch_size=2, n_streams=n_streams)
        for u_, v_ in zip(a, b):
            assert not np.isnan(v_).any()
            assert np.array_equal(u_, v_)


def test_full_batches_in_flight_are_bit_identical(golden_dir):
    """The bench's configuration: 64-window batches (MFMA recurrences, paired bank tiles, the projection on bank
    tiles, the fused encoder front and prenets) issued on several HIP streams at once must give exactly the
    sequential results; eight streams over single-window chunks likewise (the small-batch kernels)."""
    from encoder import encoder_spec_phn
    from decoder import decoder_specs
    from conftest import poison_gpu_state
    g = np.load(os.path.join(golden_dir, 'encoder_fwd.npz'))
    dec_cfg = json.load(open(os.path.join(HP, 'decoder_cfg_d.json')))
    dec_cfg['is_training'] = False
    enc = encoder_spec_phn(_enc_cfg(golden_dir, 'bfloat16'), None)
    dec = decoder_specs(dec_cfg, None, enc)
    rng = np.random.RandomState(9)
    big = np.concatenate([g['x'] * s for s in rng.uniform(0.2, 1.0, 86)], 0)[:256]      # 256 windows = 4 batches of 64
    a = dec.predict(big, batch_size=64, n_streams=1)
    for n_streams in (4, 2):
        poison_gpu_state()
        b = dec.predict(big, batch_size=64, n_streams=n_streams)
        for u_, v_ in zip(a, b):
            assert not np.isnan(v_).any() and np.array_equal(u_, v_)
    small = big[:17]
    a = dec.predict(small, batch_size=1, n_streams=1)
    poison_gpu_state()
    b = dec.predict(small, batch_size=1, n_streams=8)
    for u_, v_ in zip(a, b):
        assert np.array_equal(u_, v_)


def test_first_use_on_side_streams_builds_weight_copies_safely(golden_dir):
    """The kernel-layout weight copies are built on first use, on whatever stream the first chunk runs on; a
    second chunk on another stream must not read them half-built: the very first predict after a restore, with
    streams, already equals the sequential result."""
    from encoder import encoder_spec_phn
    from decoder import decoder_specs
    g = np.load(os.path.join(golden_dir, 'encoder_fwd.npz'))
    dec_cfg = json.load(open(os.path.join(HP, 'decoder_cfg_d.json')))
    dec_cfg['is_training'] = False
    x = np.concatenate([g['x'] * s for s in (1.0, 0.7, 0.4, 0.2)], 0)           # 12 windows -> 6 chunks of 2
    outs = []
    for n_streams in (3, 1):
        enc = encoder_spec_phn(_enc_cfg(golden_dir, 'bfloat16'), None)
        dec = decoder_specs(dec_cfg, None, enc)                                   # fresh stores: every cache empty
        torch.cuda.synchronize()
        outs.append(dec.predict(x, batch_size=2, n_streams=n_streams))
    for u_, v_ in zip(*outs):
        assert not np.isnan(u_).any() and np.array_equal(u_, v_)


def test_forced_mfma_recurrence_on_small_chunks_under_stream_overlap(golden_dir):
    """Regression case for the one wrong result on record (round 1, gpurun_out/gru_mfma_tests.log: y_stft of a
    3-stream predict differed from the sequential pass once, with the MFMA recurrence forced onto 2-window chunks).
    DESIGN.md section 9 audits every launch of that configuration; this test IS that configuration -- MFMA
    recurrence forced for H = 128 / 256 with 2 of the 16 sequence slots used, chunks of two windows and a ragged
    last chunk, NaN-poisoned LDS and allocator -- once on three streams, once on four with an unrelated stream
    keeping the chip busy with LDS-filling filter-bank launches (timing perturbation)."""
    import _vc
    import modules
    from encoder import encoder_spec_phn
    from decoder import decoder_specs
    from conftest import poison_gpu_state
    g = np.load(os.path.join(golden_dir, 'encoder_fwd.npz'))
    dec_cfg = json.load(open(os.path.join(HP, 'decoder_cfg_d.json')))
    dec_cfg['is_training'] = False
    enc = encoder_spec_phn(_enc_cfg(golden_dir, 'bfloat16'), None)
    dec = decoder_specs(dec_cfg, None, enc)
    x = np.concatenate([g['x'], g['x'][::-1] * 0.5, g['x'] * 0.25], 0)         # 9 windows -> 5 chunks of 2
    noise_stream = torch.cuda.Stream()
    noise_store = modules.VariableStore('bfloat16')
    noise_x = torch.randn(16, 400, 256, device='cuda').to(torch.bfloat16)
    try:
        _vc.set_option('gru_mfma', 1)
        a = dec.predict(x, batch_size=2, n_streams=1)
        assert not any(np.isnan(v).any() for v in a)
        # the two distinct cases (a regression case, not a fault hunt: repeating it would be looking for luck)
        for noise, n_streams in ((False, 3), (True, 4)):
            poison_gpu_state()
            if noise:
                with torch.cuda.stream(noise_stream), modules.variable_store(noise_store), modules.variable_scope('noise'):
                    for _ in range(6):
                        modules.conv1d_banks(noise_x, K=32, is_training=False)
            b = dec.predict(x, batch_size=2, n_streams=n_streams)
            for name, u_, v_ in zip(a._fields, a, b):
                assert np.array_equal(u_, v_), 'noise %s, %d streams: %s differs from the sequential pass (max %g)' % (
                    noise, n_streams, name, np.abs(u_ - v_).max())
        torch.cuda.synchronize()
    finally:
        _vc.set_option('gru_mfma', -1)


def test_forward_rejects_non_float32_features(golden_dir):
    """forward() takes float32 device features (predict() / run() convert host arrays); a float64 tensor would be read
    as the wrong type by the kernels, so it is refused loudly."""
    from encoder import encoder_spec_phn
    enc = encoder_spec_phn(_enc_cfg(golden_dir, 'bfloat16'), None)
    enc.restore()
    g = np.load(os.path.join(golden_dir, 'encoder_fwd.npz'))
    with pytest.raises(ValueError, match='float32'):
        enc.forward(torch.from_numpy(g['x'].astype(np.float64)).cuda())
    with pytest.raises(ValueError, match='float32'):
        enc.forward(torch.from_numpy(g['x']))                      # host tensor
