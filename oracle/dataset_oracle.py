"""CPU ORACLE (test infrastructure, NOT product code) for the dataset cache and window samplers.

Restates with plain numpy/python loops the logic of
  /root/reference/sound_ds.py:116-211    Sound_DS.get_ds_filter
  /root/reference/sound_ds.py:214-222    Sound_DS.get_n_windows
  /root/reference/sound_ds.py:262-350    Sound_DS.spec_window_sampler
  /root/reference/ARCTIC_reader.py:109-175, 277-362   create_spec_cache, window_sampler
  /root/reference/TIMIT_reader.py:144-210, 474-523    create_phn_mfcc_cache, window_sampler
with the h5py file replaced by a dict  name -> list of per-utterance arrays (the same data model:
``ds_h5py[name][str(i)]``).  The np.random call sequence (global generator: seed(0) + shuffle for the
train/validation split, seed(random_seed), one shuffle per epoch, one randint per long utterance)
is kept literally, because which windows a seeded run draws is the observable behaviour.

PARITY STATUS: integer / index logic, no third-party arithmetic involved; the feature values come
from oracle/frontend_oracle.py (parity unpinned at the librosa boundary, see there).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
"""
import numpy as np

from oracle import frontend_oracle as fo


def build_cache(wavs, cfg_d, phn_v=None, phn2ohv=None):
    """ARCTIC_reader.py:109-175: per utterance calc_MFCC_input (+ calc_PHN_target)."""
    keys = ('pre_emphasis', 'hop_length', 'win_length', 'n_mels', 'n_mfcc', 'n_fft', 'window',
            'mfcc_normaleze_first_mfcc', 'mfcc_norm_factor', 'calc_mfcc_derivate', 'M_dB_norm_factor',
            'P_dB_norm_factor', 'mean_abs_amp_norm', 'clip_output')
    kw = {k: cfg_d[k] for k in keys}
    cache = {'mfcc': [], 'mel_dB': [], 'power_dB': []}
    if phn_v is not None:
        cache['phn'] = []
    for i, y in enumerate(wavs):
        mfcc, mel, pdb = fo.calc_MFCC_input(y, sr=cfg_d['sample_rate'], **kw)
        cache['mfcc'].append(mfcc)
        cache['mel_dB'].append(mel)
        cache['power_dB'].append(pdb)
        if phn_v is not None:
            phn = fo.calc_PHN_target(len(y), phn_v[i], phn2ohv, hop_length=cfg_d['hop_length'],
                                     win_length=cfg_d['win_length'])
            assert mfcc.shape[0] == phn.shape[0]
            cache['phn'].append(phn)
    return cache


def get_ds_filter(ds, ds_filter_d):
    """sound_ds.py:116-211, element by element."""
    n = len(ds['wav'])
    f = [True] * n
    if ds_filter_d is None:
        return np.array(f)
    for c, v in ds_filter_d.items():
        if c == 'split_d' or v is None:
            continue
        if c not in ds:
            raise Exception('field not found')
        vals = list(v) if isinstance(v, (list, tuple)) else [v]
        for i in range(n):
            f[i] = f[i] and any(ds[c][i] == x for x in vals)
    sd = ds_filter_d.get('split_d')
    if sd is not None:
        key, typ, (p0, p1) = sd['split_key'], sd['split_type'], sd['split_props_v']
        for k in sorted(set(ds[key][i] for i in range(n) if f[i])):
            members = [i for i in range(n) if f[i] and ds[key][i] == k]
            n_trn, n_val = int(len(members) * p0), int(len(members) * p1)
            for pos, i in enumerate(members):
                part = 'trn' if pos < n_trn else ('val' if pos < n_val else 'tst')
                if part != typ:
                    f[i] = False
    return np.array(f)


def get_n_windows(ds, cfg_d, prop_val, ds_filter_d):
    f = get_ds_filter(ds, ds_filter_d)
    n = sum(len(w) // (cfg_d['hop_length'] * cfg_d['n_timesteps']) for w, k in zip(ds['wav'], f) if k)
    n_trn = int((1 - prop_val) * n)
    return n_trn, n - n_trn


def _split(samples_v, prop_val, sample_trn, random_seed):
    if prop_val > 0.0:
        np.random.seed(0)
        idx_v = np.arange(samples_v.shape[0])
        np.random.shuffle(idx_v)
        n_val = int(prop_val * samples_v.shape[0])
        samples_v = samples_v[idx_v[:-n_val]] if sample_trn else samples_v[idx_v[-n_val:]]
        np.random.seed(random_seed)
    return samples_v


def _loop(cache, names, samples_v, n_timesteps, batch_size, n_epochs, randomize_samples, short, pad_fix=None):
    batch, idxs = [[] for _ in names], []
    for _ in range(n_epochs):
        if randomize_samples:
            np.random.shuffle(samples_v)
        for i_sample in samples_v:
            i = int(i_sample)
            spec_len = cache[names[0]][i].shape[0]
            if spec_len <= n_timesteps:
                if short == 'skip':
                    continue
                i_s, i_e = 0, n_timesteps
                pad_len = n_timesteps - spec_len
                cut = [np.concatenate([cache[nm][i][:], np.zeros((pad_len, cache[nm][i].shape[1]))], axis=0)
                       for nm in names]
                if pad_fix is not None and pad_len > 0:
                    pad_fix(cut, pad_len)
            else:
                i_s = np.random.randint(0, spec_len - n_timesteps)
                i_e = i_s + n_timesteps
                cut = [cache[nm][i][i_s:i_e] for nm in names]
            for b, c in zip(batch, cut):
                b.append(c)
            idxs.append([i_s, i_e, i])
            if len(idxs) == batch_size:
                yield tuple(np.array(b) for b in batch) + (np.array(idxs),)
                batch, idxs = [[] for _ in names], []


def spec_window_sampler(ds, cache, n_timesteps, random_seed, batch_size=32, n_epochs=1, randomize_samples=True,
                        sample_trn=True, prop_val=0.3, ds_filter_d={}):
    """sound_ds.py:262-350 (always yields idxs last)."""
    f = get_ds_filter(ds, ds_filter_d)
    samples_v = _split(np.array([str(i) for i in np.arange(len(f))[f]]), prop_val, sample_trn, random_seed)
    return _loop(cache, ('mfcc', 'mel_dB', 'power_dB'), samples_v, n_timesteps, batch_size, n_epochs,
                 randomize_samples, 'pad')


def arctic_window_sampler(ds, cache, n_timesteps, random_seed, pau_idx, batch_size=32, n_epochs=1,
                          randomize_samples=True, sample_trn=True, prop_val=0.3, ds_filter_d={}):
    """ARCTIC_reader.py:277-362."""
    f = get_ds_filter(ds, ds_filter_d)
    samples_v = _split(np.array([str(i) for i in np.arange(len(f))[f]]), prop_val, sample_trn, random_seed)

    def fix(cut, pad_len):
        cut[1][-pad_len:, pau_idx] = 1.0
    return _loop(cache, ('mfcc', 'phn'), samples_v, n_timesteps, batch_size, n_epochs, randomize_samples, 'pad', fix)


def timit_window_sampler(ds, cache, n_timesteps, batch_size=32, n_epochs=1, randomize_samples=True, ds_filter_d={}):
    """TIMIT_reader.py:474-523."""
    f = get_ds_filter(ds, ds_filter_d)
    samples_v = [str(i) for i in np.arange(len(f))[f]]
    return _loop(cache, ('mfcc', 'phn'), samples_v, n_timesteps, batch_size, n_epochs, randomize_samples, 'skip')
