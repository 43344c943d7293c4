"""Feature cache + window samplers with the surface of the reference's dataset objects.

Mirrors what the training loops consume from /root/reference/sound_ds.py (``Sound_DS``:
``get_ds_filter`` :116-211, ``get_n_windows`` :214-222, ``get_spec`` :225-248, ``_zero_pad`` :252-259,
``spec_window_sampler`` :262-350), ARCTIC_reader.py (``create_spec_cache`` :109-175,
``window_sampler`` :277-362) and TIMIT_reader.py (``create_phn_mfcc_cache`` :144-210,
``window_sampler`` :474-523) -- i.e. what ``encoder.train`` (encoder.py:300-356) and
``decoder.train`` (decoder.py:379-444) call on their ``ds`` argument.

MI355X-first differences:
  * the reference computes features one utterance at a time on the CPU and stores them in an h5py
    file; here ``create_spec_cache`` runs the batched HIP front-end over length-bucketed ragged
    batches and keeps the result in HBM as ragged arenas ``[total_frames, C]`` (+ host row tables);
  * the samplers draw the same windows with the same ``np.random`` call sequence as the reference
    (so a seeded run picks identical utterances and offsets), but cut them out of the arena with one
    gather launch and yield float32 ``torch.cuda`` tensors ``[batch, n_timesteps, C]`` (pass
    ``output='numpy'`` for host arrays; the models' entry points accept both).
Parsing of the TIMIT / CMU-ARCTIC directory trees, the pickle cache of raw audio, plotting and
playback are out of scope (corpora are not available): the dataset is
constructed from in-memory arrays in the layout of the reference's ``self.ds`` dict.
"""
import sys
from collections import namedtuple

import numpy as np

import _vc
import audio_lib

_FE_KEYS = ('pre_emphasis', 'hop_length', 'win_length', 'n_mels', 'n_mfcc', 'n_fft', 'window',
            'mfcc_normaleze_first_mfcc', 'mfcc_norm_factor', 'calc_mfcc_derivate', 'M_dB_norm_factor',
            'P_dB_norm_factor', 'mean_abs_amp_norm', 'clip_output')

TIMIT_PHONEMES_61 = ['b', 'd', 'g', 'p', 't', 'k', 'dx', 'q', 'bcl', 'dcl', 'gcl', 'pcl', 'tcl', 'kcl', 'jh', 'ch',
                     's', 'sh', 'z', 'zh', 'f', 'th', 'v', 'dh', 'm', 'n', 'ng', 'em', 'en', 'eng', 'nx',
                     'l', 'r', 'w', 'y', 'hh', 'hv', 'el', 'iy', 'ih', 'eh', 'ey', 'ae', 'aa', 'aw', 'ay', 'ah', 'ao',
                     'oy', 'ow', 'uh', 'uw', 'ux', 'er', 'ax', 'ix', 'axr', 'ax-h', 'pau', 'epi', 'h#']   # TIMIT_reader.py:54-61
ARCTIC_PHONEMES_43 = ['b', 'd', 'g', 'p', 't', 'k', 'jh', 'ch', 's', 'sh', 'z', 'zh', 'f', 'th', 'v', 'dh', 'm', 'n',
                      'ng', 'l', 'r', 'w', 'y', 'hh', 'aa', 'ae', 'ah', 'ao', 'aw', 'ax', 'ay', 'eh', 'er', 'ey', 'ih',
                      'iy', 'ow', 'oy', 'uh', 'uw', 'H#', 'pau', 'ssil']                                    # ARCTIC_reader.py:44-51


def gather_rows(src, index, pad_row=None, out=None):
    """dst[r] = src[index[r]] (index < 0 -> pad_row or zeros) on the GPU.  src [R, C] float32 cuda,
    index int64 host array or cuda tensor [n]; returns [n, C] (written into ``out`` when given)."""
    import torch
    if not torch.is_tensor(index):
        index = torch.from_numpy(np.ascontiguousarray(index, dtype=np.int64))
    index = index.to(device=src.device, dtype=torch.int64).contiguous()
    if src.dim() != 2 or not src.is_contiguous():
        raise ValueError(' - ERROR, gather_rows: src must be a contiguous [rows, C] tensor')
    n, Cw = index.numel(), src.shape[1]
    if n and int(index.max()) >= src.shape[0]:
        raise IndexError(' - ERROR, gather_rows: index out of range')
    if out is None:
        dst = torch.empty((n, Cw), dtype=src.dtype, device=src.device)
    else:
        dst = out
        if tuple(dst.shape) != (n, Cw) or not dst.is_contiguous() or dst.dtype != src.dtype:
            raise ValueError(' - ERROR, gather_rows: out must be a contiguous [n, C] tensor with the dtype of src')
    if pad_row is not None:
        pad_row = pad_row.to(device=src.device, dtype=src.dtype).contiguous()
        if pad_row.numel() != Cw:
            raise ValueError(' - ERROR, gather_rows: pad_row width')
    if n:
        _vc.check(_vc.lib().vc_gather_rows(_vc.ptr(src), _vc.ptr(index), _vc.ptr(pad_row), n, Cw * src.element_size(),
                                           _vc.ptr(dst), _vc.current_stream()))
    return dst


class Sound_DS():
    """In-memory dataset + device feature cache.

    cfg_d : the reference's dataset config dict (hp/ds_*_cfg_d.json keys: sample_rate, hop_length or
            hop_length_ms, win_length or win_length_ms, n_timesteps, random_seed, verbose, ds_norm,
            and the calc_MFCC_input arguments).
    ds    : dict in the layout of the reference's ``self.ds``: 'wav' -> sequence of 1-D float arrays,
            optional 'phn_v' -> per utterance sequence of (start, end, phoneme) records, and any
            number of per-utterance label arrays ('spk_id', 'ds_type', ...) that filters refer to.
    phonemes : phoneme inventory for one-hot targets (TIMIT_PHONEMES_61 / ARCTIC_PHONEMES_43)."""

    def __init__(self, cfg_d, ds, phonemes=None, build_cache=True, cache_batch=64):
        self.cfg_d = cfg_d
        if 'hop_length' not in self.cfg_d.keys():
            self.cfg_d['hop_length'] = int(self.cfg_d['hop_length_ms'] * self.cfg_d['sample_rate'] / 1000.0)
            print(" - cfg_d['hop_length'] = {:d}".format(self.cfg_d['hop_length']))
        if 'win_length' not in self.cfg_d.keys():
            self.cfg_d['win_length'] = int(self.cfg_d['win_length_ms'] * self.cfg_d['sample_rate'] / 1000.0)
            print(" - cfg_d['win_length'] = {:d}".format(self.cfg_d['win_length']))
        self.random_seed = cfg_d.get('random_seed', None)
        self.verbose = cfg_d.get('verbose', False)
        self.ds_norm = cfg_d.get('ds_norm', (0.0, 1.0))
        self.n_mfcc = cfg_d['n_mfcc']
        self.n_timesteps = cfg_d['n_timesteps']
        self.sample_rate = cfg_d['sample_rate']
        if self.random_seed is not None:
            np.random.seed(self.random_seed)

        self.ds = dict(ds)
        wav = np.empty(len(ds['wav']), dtype=object)
        for i, w in enumerate(ds['wav']):
            wav[i] = np.asarray(w)
        self.ds['wav'] = wav
        for k, v in list(self.ds.items()):
            if k not in ('wav', 'phn_v'):
                self.ds[k] = np.asarray(v)
        self._normalize_ds()
        self.ds_phoneme_v = None if phonemes is None else np.array(phonemes)
        if phonemes is not None:
            self.make_phoneme_convertion_dicts()
        self.cache = None
        self._cache_batch = int(cache_batch)
        if build_cache:
            self.create_spec_cache()

    # --------------------------------------------------------------------------- reference helpers
    def _normalize_ds(self):
        """sound_ds.py:56-63."""
        if self.verbose:
            print(' - normalize_ds: Normalizando ondas con: add={:0.02f}  mult={:0.02f}'.format(*self.ds_norm))
        for i in range(len(self.ds['wav'])):
            self.ds['wav'][i] = self.ds_norm[1] * (self.ds['wav'][i] + self.ds_norm[0])
        return None

    def make_phoneme_convertion_dicts(self):
        """ARCTIC_reader.py:252-270 / TIMIT_reader.py:339-360."""
        self.phn2ohv, self.phn2idx, self.idx2phn = {}, {}, {}
        for idx, phn in enumerate(self.ds_phoneme_v):
            ohv = np.zeros(len(self.ds_phoneme_v))
            ohv[idx] = 1.0
            self.phn2ohv[phn] = ohv
            self.phn2idx[phn] = idx
            self.idx2phn[idx] = phn
        self.n_phn = len(self.ds_phoneme_v)
        return None

    def get_ds_filter(self, ds_filter_d={}):
        """sound_ds.py:116-211: boolean utterance mask.  Every key of ``ds_filter_d`` names a label
        array of the dataset and keeps the utterances whose label equals the value (or one of the
        values of a list); key 'split_d' = {'split_key', 'split_type' in trn|val|tst,
        'split_props_v': (p_trn, p_val)} then keeps, per distinct value of ds[split_key], the first
        p_trn share (trn), the next up to p_val (val) or the rest (tst) of the surviving utterances.
        Same exceptions and warnings as the reference."""
        n = self.ds['wav'].shape[0]
        keep = np.ones(n, dtype=bool)
        if ds_filter_d is None:
            return keep
        split_d = ds_filter_d.get('split_d', None)
        for field, wanted in ds_filter_d.items():
            if field == 'split_d':
                continue
            if field not in self.ds.keys():
                raise Exception(' - ERROR, get_ds_fillter: campo "{}" no encontrado en el ds'.format(field))
            if wanted is None:
                continue
            values = wanted if type(wanted) in (list, tuple) else [wanted]
            hit = np.zeros(n, dtype=bool)
            for v in values:
                hit |= np.asarray(self.ds[field] == v, dtype=bool)
            keep &= hit
        if split_d is not None:
            if type(split_d) is not dict:
                raise Exception(' - ERROR, get_ds_fillter: split_d debe ser class dict')
            split_key, split_type, props = split_d['split_key'], split_d['split_type'], split_d['split_props_v']
            if split_key not in self.ds.keys():
                raise Exception(' - ERROR, get_ds_fillter: campo para split "{}" no encontrado en el ds'.format(split_key))
            if split_type not in ['trn', 'val', 'tst']:
                raise Exception(' - ERROR, get_ds_fillter: tipo de split no reconocido "{}"'.format(split_type))
            if type(props) is not tuple or len(props) != 2:
                raise Exception(' - ERROR, get_ds_fillter: split_props_v="{}", deberia ser un tupla de len 2'.format(props))
            if props[0] > props[1]:
                raise Exception(' - ERROR, get_ds_fillter: split_props_v="{}", el segundo elemento no puede set superior al primero.'.format(props))
            for k in np.unique(self.ds[split_key][keep]):
                members = np.flatnonzero(keep & (self.ds[split_key] == k))
                n_trn, n_val = int(len(members) * props[0]), int(len(members) * props[1])
                part = {'trn': members[:n_trn], 'val': members[n_trn:n_val], 'tst': members[n_val:]}[split_type]
                keep[members] = False
                keep[part] = True
                if len(part) == 0:
                    print('WARNING, no se selecciona ningun dato para k="{}" con split_key="{}". revisar valores de filtrado.'.format(k, split_key), file=sys.stderr)
        if keep.sum() == 0:
            print('WARNING, no se selecciona ningun dato. Revisar campos de filtrado', file=sys.stderr)
        return keep

    def get_n_windows(self, prop_val=0.3, ds_filter_d={}):
        """sound_ds.py:214-222."""
        f_s = self.get_ds_filter(ds_filter_d)
        n_windows = sum([s.shape[0] // (self.cfg_d['hop_length'] * self.cfg_d['n_timesteps']) for s in self.ds['wav'][f_s]])
        n_windows_trn = int((1 - prop_val) * n_windows)
        n_windows_val = n_windows - n_windows_trn
        return n_windows_trn, n_windows_val

    def _zero_pad(self, *to_pad, pad_len=10):
        """sound_ds.py:252-259 (host helper, kept for callers that use it directly)."""
        return [np.concatenate([spec, np.zeros((pad_len, spec.shape[1]))], axis=0) for spec in to_pad]

    # --------------------------------------------------------------------------- device cache
    def create_spec_cache(self, cfg_d=None):
        """ARCTIC_reader.py:109-175 / TIMIT_reader.py:144-210 on the GPU: features of every utterance
        into ragged arenas ``self.cache[name] = [total_frames, C]``; utterance i owns rows
        ``starts[i] : starts[i] + nframes[i]``."""
        import torch
        if cfg_d is None:
            cfg_d = self.cfg_d
        kw = {k: cfg_d[k] for k in _FE_KEYS}
        kw['sr'] = cfg_d['sample_rate']
        n = len(self.ds['wav'])
        hop = cfg_d['hop_length']
        lens = np.array([len(w) for w in self.ds['wav']], dtype=np.int64)
        self.nframes = (1 + lens // hop).astype(np.int64)
        order = np.argsort(-lens, kind='stable')                  # length buckets: little padding per batch
        # arena order = processing order, so every batch fills one contiguous block of rows
        self.starts = np.zeros(n, dtype=np.int64)
        self.starts[order] = np.concatenate([[0], np.cumsum(self.nframes[order])[:-1]])
        total = int(self.nframes.sum())
        arenas = None
        for b0 in range(0, n, self._cache_batch):
            ids = order[b0:b0 + self._cache_batch]
            L = int(lens[ids].max())
            host = np.zeros((len(ids), L), dtype=np.float32)
            for j, i in enumerate(ids):
                host[j, :lens[i]] = self.ds['wav'][i]
            outs = audio_lib.calc_MFCC_input_batch(torch.from_numpy(host).cuda(), lens[ids].astype(np.int32), **kw)
            if arenas is None:
                arenas = [torch.empty((total, o.shape[2]), dtype=torch.float32, device=o.device) for o in outs]
            Fmax = outs[0].shape[1]
            rows = np.concatenate([j * Fmax + np.arange(self.nframes[i], dtype=np.int64) for j, i in enumerate(ids)])
            r0 = int(self.starts[ids[0]])
            d_rows = torch.from_numpy(rows).to(outs[0].device)
            for a, o in zip(arenas, outs):
                gather_rows(o.view(-1, o.shape[2]), d_rows, out=a[r0:r0 + len(rows)])
            if self.verbose:
                print(' - Saved: {} of {} samples'.format(min(b0 + self._cache_batch, n), n))
        self.cache = {'mfcc': arenas[0], 'mel_dB': arenas[1], 'power_dB': arenas[2]}
        if 'phn_v' in self.ds and self.ds_phoneme_v is not None:
            phn = np.zeros((total, self.n_phn), dtype=np.float32)
            for i in range(n):
                t = audio_lib.calc_PHN_target(self.ds['wav'][i], self.ds['phn_v'][i], self.phn2ohv,
                                              hop_length=hop, win_length=cfg_d['win_length'])
                assert t.shape[0] == self.nframes[i], '- ERROR, create_spec_cache: para la muestra {}, mfcc.shape[0] != phn.shape[0]'.format(i)
                phn[self.starts[i]:self.starts[i] + self.nframes[i]] = t
            self.cache['phn'] = torch.from_numpy(phn).to(arenas[0].device)
        return None

    def spec_len(self, i_sample):
        return int(self.nframes[int(i_sample)])

    def get_spec(self, i_sample):
        """sound_ds.py:225-248: host copies of one utterance's cached features."""
        names = [k for k in ('mfcc', 'mel_dB', 'power_dB', 'phn') if k in self.cache]
        s = int(self.starts[int(i_sample)])
        e = s + int(self.nframes[int(i_sample)])
        vals = [self.cache[k][s:e].cpu().numpy() for k in names]
        if 'phn' in names:
            vals[names.index('phn')] = vals[names.index('phn')].astype(np.int32)
        return namedtuple('ret', ' '.join(names))(*vals)

    # --------------------------------------------------------------------------- samplers
    def _split_trn_val(self, samples_v, prop_val, sample_trn):
        """The seeded split of sound_ds.py:268-283 / ARCTIC_reader.py:283-297."""
        if prop_val > 0.0:
            np.random.seed(0)
            idx_v = np.arange(samples_v.shape[0])
            np.random.shuffle(idx_v)
            n_val = int(prop_val * samples_v.shape[0])
            idx_trn = idx_v[:-n_val]
            idx_val = idx_v[-n_val:]
            samples_v = samples_v[idx_trn] if sample_trn else samples_v[idx_val]
            np.random.seed(self.random_seed)
        return samples_v

    def _emit(self, names, rows, pads, output):
        """rows: int64 [batch * n_timesteps] arena row per output frame (-1 = padding)."""
        outs = []
        for nm in names:
            t = gather_rows(self.cache[nm], rows, pads.get(nm))
            t = t.view(-1, self.n_timesteps, t.shape[1])
            outs.append(t.cpu().numpy() if output == 'numpy' else t)
        return outs

    def _window_loop(self, names, samples_v, batch_size, n_epochs, randomize_samples, yield_idxs, short, pads, output):
        """Common body of the three samplers.  short = 'pad' (sound_ds.py:302-316,
        ARCTIC_reader.py:318-336) or 'skip' (TIMIT_reader.py:496-497)."""
        T = self.n_timesteps
        ar = np.arange(T, dtype=np.int64)
        rows, idxs_v = [], []
        n_warning = 0
        for i_epoch in range(n_epochs):
            if randomize_samples:
                np.random.shuffle(samples_v)
            for i_sample in samples_v:
                spec_len = self.spec_len(i_sample)
                base = int(self.starts[int(i_sample)])
                if spec_len <= T:
                    if short == 'skip':
                        continue
                    i_s, i_e = 0, T
                    r = np.where(ar < spec_len, base + ar, -1)
                    if n_warning < 5:
                        print('WARNING: padding!!!'.format(i_sample))
                        n_warning += 1
                else:
                    i_s = np.random.randint(0, spec_len - T)
                    i_e = i_s + T
                    r = base + i_s + ar
                rows.append(r)
                idxs_v.append([i_s, i_e, int(i_sample)])
                if len(rows) == batch_size:
                    outs = self._emit(names, np.concatenate(rows), pads, output)
                    if yield_idxs:
                        yield tuple(outs) + (np.array(idxs_v),)
                    else:
                        yield tuple(outs)
                    rows, idxs_v = [], []

    def _samples(self, ds_filter_d):
        f_s = self.get_ds_filter(ds_filter_d)
        return np.array([str(i) for i in np.arange(f_s.shape[0])[f_s]])

    def spec_window_sampler(self, batch_size=32, n_epochs=1, randomize_samples=True, sample_trn=True, prop_val=0.3,
                            ds_filter_d={}, yield_idxs=False, output='device'):
        """sound_ds.py:262-350: yields (mfcc, mel_dB, power_dB[, idxs]) windows of n_timesteps frames,
        one random window per utterance and epoch, short utterances zero padded."""
        samples_v = self._split_trn_val(self._samples(ds_filter_d), prop_val, sample_trn)
        return self._window_loop(('mfcc', 'mel_dB', 'power_dB'), samples_v, batch_size, n_epochs, randomize_samples,
                                 yield_idxs, 'pad', {}, output)


class ARCTIC(Sound_DS):
    """ARCTIC_reader.ARCTIC's training-facing surface over in-memory arrays."""

    def __init__(self, cfg_d, ds, phonemes=ARCTIC_PHONEMES_43, **kw):
        Sound_DS.__init__(self, cfg_d, ds, phonemes, **kw)

    def window_sampler(self, batch_size=32, n_epochs=1, randomize_samples=True, sample_trn=True, prop_val=0.3,
                       ds_filter_d={'spk_id': ['bdl', 'rms', 'slt', 'clb']}, yield_idxs=False, output='device'):
        """ARCTIC_reader.py:277-362: (mfcc, phn[, idxs]); padding frames of short utterances are
        labelled 'pau' (the reference reaches that index through a module-level ``arctic`` object,
        ARCTIC_reader.py:331; here it is the dataset's own table)."""
        import torch
        samples_v = self._split_trn_val(self._samples(ds_filter_d), prop_val, sample_trn)
        pau = torch.zeros(self.n_phn, dtype=torch.float32)
        pau[self.phn2idx['pau']] = 1.0
        return self._window_loop(('mfcc', 'phn'), samples_v, batch_size, n_epochs, randomize_samples, yield_idxs, 'pad',
                                 {'phn': pau}, output)


class TIMIT(Sound_DS):
    """TIMIT_reader.TIMIT's training-facing surface over in-memory arrays."""

    def __init__(self, cfg_d, ds, phonemes=TIMIT_PHONEMES_61, **kw):
        Sound_DS.__init__(self, cfg_d, ds, phonemes, **kw)

    def window_sampler(self, batch_size=32, n_epochs=1, randomize_samples=True, ds_filter_d={'ds_type': 'TRAIN'},
                       yield_idxs=False, output='device'):
        """TIMIT_reader.py:474-523: (mfcc, phn[, idxs]); utterances of <= n_timesteps frames are skipped."""
        samples_v = list(self._samples(ds_filter_d))
        return self._window_loop(('mfcc', 'phn'), samples_v, batch_size, n_epochs, randomize_samples, yield_idxs, 'skip',
                                 {}, output)
