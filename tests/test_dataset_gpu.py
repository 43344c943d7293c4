"""GPU parity of the on-device feature cache + window samplers (speech-cloner_amd/sound_ds.py,
vc_gather_rows) against oracle/dataset_oracle.py, which restates
/root/reference/sound_ds.py:116-350, ARCTIC_reader.py:109-175/277-362, TIMIT_reader.py:144-210/474-523.

Which utterances / offsets a seeded run draws is integer logic and must match exactly; window
contents are compared (a) bit-exactly with the device cache they are cut from and (b) with the
oracle's numpy features at the front-end tolerance (1e-4 abs, power dB 2e-4)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import dataset_oracle as do
from oracle import frontend_oracle as fo

pytestmark = pytest.mark.gpu


def _cfg():
    cfg = json.load(open(os.path.join(ROOT, 'speech-cloner_amd', 'hp', 'ds_dec_cfg_d.json')))
    cfg.update(n_timesteps=100, random_seed=3, verbose=False, ds_norm=(0.0, 1.0))
    if 'hop_length' not in cfg:
        cfg['hop_length'] = int(cfg['hop_length_ms'] * cfg['sample_rate'] / 1000.0)
        cfg['win_length'] = int(cfg['win_length_ms'] * cfg['sample_rate'] / 1000.0)
    return cfg


def _corpus(n=23, seed=0, phonemes=None):
    rng = np.random.RandomState(seed)
    lens = rng.randint(3000, 30000, n)
    lens[2] = 5000                       # 63 frames  < n_timesteps -> padded / skipped
    lens[7] = 7920                       # 100 frames == n_timesteps -> still "short" (<=)
    wav = [fo.synth_speech(1, int(L), seed=100 + i)[0] for i, L in enumerate(lens)]
    ds = {'wav': wav, 'spk_id': np.array(['bdl', 'slt', 'rms'])[rng.randint(0, 3, n)],
          'ds_type': np.array(['TRAIN', 'TEST'])[(rng.rand(n) < 0.3).astype(int)]}
    if phonemes is not None:
        phn_v = []
        for L in lens:
            cuts = np.sort(rng.choice(np.arange(200, L - 200), size=6, replace=False))
            edges = [0] + list(cuts) + [int(L)]
            phn_v.append([(int(edges[k]), int(edges[k + 1]), phonemes[rng.randint(len(phonemes))]) for k in range(7)])
        ds['phn_v'] = phn_v
    return ds


def test_gather_rows_kernel():
    import sound_ds
    rng = np.random.RandomState(0)
    src = torch.from_numpy(rng.standard_normal((1000, 201)).astype(np.float32)).cuda()
    idx = rng.randint(-1, 1000, 5000).astype(np.int64)
    pad = torch.arange(201, dtype=torch.float32)
    out = sound_ds.gather_rows(src, idx, pad).cpu().numpy()
    ref = np.where(idx[:, None] >= 0, src.cpu().numpy()[np.maximum(idx, 0)], pad.numpy()[None, :])
    assert np.array_equal(out, ref)
    out0 = sound_ds.gather_rows(src, idx).cpu().numpy()
    assert np.array_equal(out0, np.where(idx[:, None] >= 0, ref, 0.0))
    assert sound_ds.gather_rows(src, np.zeros(0, np.int64)).shape == (0, 201)
    with pytest.raises(IndexError):
        sound_ds.gather_rows(src, np.array([1000]))


def test_cache_matches_per_utterance_front_end():
    import sound_ds
    cfg, ds = _cfg(), _corpus()
    d = sound_ds.Sound_DS(cfg, ds, cache_batch=5)
    ref = do.build_cache(ds['wav'], cfg)
    assert int(d.nframes.sum()) == d.cache['mfcc'].shape[0] == sum(r.shape[0] for r in ref['mfcc'])
    for i in range(len(ds['wav'])):
        got = d.get_spec(i)
        assert got._fields == ('mfcc', 'mel_dB', 'power_dB')
        for nm, tol in (('mfcc', 1e-4), ('mel_dB', 1e-4), ('power_dB', 2e-4)):
            g, r = getattr(got, nm), ref[nm][i]
            assert g.shape == r.shape == (1 + len(ds['wav'][i]) // cfg['hop_length'], r.shape[1])
            assert np.abs(g - r).max() < tol, (i, nm)
    # a different batching of the ragged batches gives the same cache, bit for bit
    d2 = sound_ds.Sound_DS(cfg, ds, cache_batch=64)
    for i in (0, 2, 11, 22):
        assert np.array_equal(d.get_spec(i).power_dB, d2.get_spec(i).power_dB)


def test_spec_window_sampler_draws_the_reference_windows():
    import sound_ds
    cfg, ds = _cfg(), _corpus()
    d = sound_ds.Sound_DS(cfg, ds)
    ref_cache = do.build_cache(ds['wav'], cfg)
    flt = {'spk_id': ['bdl', 'slt']}
    for sample_trn in (True, False):
        got = list(d.spec_window_sampler(batch_size=4, n_epochs=3, sample_trn=sample_trn, prop_val=0.3,
                                         ds_filter_d=flt, yield_idxs=True))
        ref = list(do.spec_window_sampler(ds, ref_cache, 100, cfg['random_seed'], batch_size=4, n_epochs=3,
                                          sample_trn=sample_trn, prop_val=0.3, ds_filter_d=flt))
        assert len(got) == len(ref) > 0
        for g, r in zip(got, ref):
            assert np.array_equal(g[3], r[3])                                # [i_s, i_e, i_sample] exact
            for k, tol in ((0, 1e-4), (1, 1e-4), (2, 2e-4)):
                assert torch.is_tensor(g[k]) and g[k].is_cuda and g[k].dtype == torch.float32
                assert tuple(g[k].shape) == r[k].shape and np.abs(g[k].cpu().numpy() - r[k]).max() < tol
            for b, (i_s, i_e, i) in enumerate(g[3]):                         # bit-exact cut of the device cache
                sp = d.get_spec(i).mel_dB
                want = np.zeros((100, sp.shape[1]), np.float32)
                want[:min(100, len(sp) - i_s)] = sp[i_s:i_e]
                assert np.array_equal(g[1][b].cpu().numpy(), want)
    assert d.get_n_windows(0.3, flt) == do.get_n_windows(ds, cfg, 0.3, flt)


def test_phoneme_window_samplers():
    import sound_ds
    cfg = _cfg()
    ds = _corpus(phonemes=sound_ds.ARCTIC_PHONEMES_43)
    a = sound_ds.ARCTIC(cfg, ds)
    ref_cache = do.build_cache(ds['wav'], cfg, ds['phn_v'], a.phn2ohv)
    assert np.array_equal(a.get_spec(5).phn, ref_cache['phn'][5]) and a.get_spec(5).phn.dtype == np.int32
    got = list(a.window_sampler(batch_size=5, n_epochs=2, prop_val=0.2, ds_filter_d={'spk_id': ['bdl', 'rms', 'slt']},
                                yield_idxs=True, output='numpy'))
    ref = list(do.arctic_window_sampler(ds, ref_cache, 100, cfg['random_seed'], a.phn2idx['pau'], batch_size=5,
                                        n_epochs=2, prop_val=0.2, ds_filter_d={'spk_id': ['bdl', 'rms', 'slt']}))
    assert len(got) == len(ref) > 0
    padded = 0
    for g, r in zip(got, ref):
        assert np.array_equal(g[2], r[2])
        assert np.abs(g[0] - r[0]).max() < 1e-4
        assert np.array_equal(g[1], r[1])                                    # one-hot targets incl. 'pau' padding
        padded += int((g[2][:, 0] == 0).sum())
    assert padded > 0
    # TIMIT flavour: no split, short utterances skipped
    t = sound_ds.TIMIT(cfg, _corpus(phonemes=sound_ds.TIMIT_PHONEMES_61))
    ds_t = t.ds
    ref_cache = do.build_cache(ds_t['wav'], cfg, ds_t['phn_v'], t.phn2ohv)
    np.random.seed(11)
    got = list(t.window_sampler(batch_size=4, n_epochs=2, ds_filter_d={'ds_type': 'TRAIN'}, yield_idxs=True))
    np.random.seed(11)
    ref = list(do.timit_window_sampler(ds_t, ref_cache, 100, batch_size=4, n_epochs=2, ds_filter_d={'ds_type': 'TRAIN'}))
    assert len(got) == len(ref) > 0
    for g, r in zip(got, ref):
        assert np.array_equal(g[2], r[2]) and np.array_equal(g[1].cpu().numpy(), r[1])
        assert not np.isin(g[2][:, 2], [2, 7]).any()                         # the two short utterances never appear
        assert g[1].shape == (4, 100, 61)


def test_encoder_trains_from_the_sampler():
    """encoder.train's inner loop (encoder.py:332-334) fed by the device sampler."""
    import sound_ds
    from encoder import encoder_spec_phn
    cfg = _cfg()
    t = sound_ds.TIMIT(cfg, _corpus(phonemes=sound_ds.TIMIT_PHONEMES_61))
    ecfg = {'model_name': 'encoder', 'input_shape': [100, 80], 'n_output': 61, 'embed_size': 32, 'num_conv_banks': 3,
            'num_highwaynet_blocks': 2, 'dropout_rate': 0.1, 'is_training': True, 'use_Cudnn': False, 'use_lstm': False,
            'learning_rate': 2e-3, 'decay': 0.0, 'beta1': 0.9, 'beta2': 0.999, 'epsilon': 1e-8}
    enc = encoder_spec_phn(ecfg, t)
    losses = []
    for mfcc, phn in t.window_sampler(batch_size=8, n_epochs=12, ds_filter_d=None):
        losses.append(enc.exec_train_step(mfcc, phn)[0])
    assert len(losses) >= 20 and np.isfinite(losses).all()
    assert np.mean(losses[-5:]) < 0.95 * np.mean(losses[:5])      # random labels: only the priors are learnable
