"""Conversion driver helpers with the surface of the reference's ``test.py``.

Mirrors /root/reference/test.py:46-306: ``compound`` (stitching of half-overlapped window
predictions), ``conversion`` (non-overlapping windows) and ``conversion2`` (two passes shifted by
half a window, stitched) -- the pure integer framing that turns an utterance's features into
[N, n_timesteps, C] batches for ``decoder.predict`` and back, followed by the Griffin-Lim vocoder
(audio_lib.from_power_to_wav on the GPU, test.py:146-168).  Plotting, audio playback and wav
writing (test.py:28-43, 171-188) are UI side effects and out of scope.  ``vocoder``: 'default' =
audio_lib.from_power_to_wav, any callable with that signature, or None to skip audio synthesis
(``y_wav_true`` / ``y_wav_pred`` are then None).
"""
from collections import namedtuple

import numpy as np


def compound_index(N, T):
    """Source of every output frame of ``compound``: int arrays (which, window, frame) where
    which = 0 selects y0 and 1 selects y1.  Closed form of the loop at test.py:58-80:
    y0[0][:T-q], then alternately y1[i][q:T-q], y0[i+1][q:T-q], finally y0[N-1][q:], q = T//4."""
    q = T // 4
    fr = np.arange(T)
    head, mid, tail = fr[:-q], fr[q:-q], fr[q:]
    which, win, frame = [np.zeros(len(head), np.int64)], [np.zeros(len(head), np.int64)], [head]
    i_0, i_1 = 1, 0
    while i_1 < N - 1 or i_0 < N - 1:
        if i_1 < N - 1:
            which.append(np.ones(len(mid), np.int64)); win.append(np.full(len(mid), i_1, np.int64)); frame.append(mid)
            i_1 += 1
        if i_0 < N - 1:
            which.append(np.zeros(len(mid), np.int64)); win.append(np.full(len(mid), i_0, np.int64)); frame.append(mid)
            i_0 += 1
    which.append(np.zeros(len(tail), np.int64)); win.append(np.full(len(tail), N - 1, np.int64)); frame.append(tail)
    return np.concatenate(which), np.concatenate(win), np.concatenate(frame)


def compound(y0, y1):
    """test.py:46-84.  y0 [N, T, X], y1 [N-1, T, X] -> [N*T, X]: the centre half of every window
    (three quarters at both ends), alternating between the two passes."""
    N, T = y0.shape[0], y0.shape[1]
    which, win, frame = compound_index(N, T)
    if y1.shape[0] < N - 1:
        raise IndexError(' - ERROR, compound: y1 must hold N-1 windows')
    both = np.concatenate([y0, y1[:max(N - 1, 0)]], axis=0)
    return both[win + which * N, frame]


def window_plan(n_frames, cfg_d, t_s, t_e):
    """Integer window arithmetic of test.py:92-119 / 211-238.  Returns (pad_len, n_s, n_e) for an
    utterance of ``n_frames`` frames; raises like the reference when the span is empty."""
    hop = cfg_d['hop_length']
    n_times = cfg_d['n_timesteps']
    pad_len = 0
    if n_frames % n_times != 0:
        pad_len = n_times - (n_frames % n_times)
    n_hop_s = t_s * cfg_d['sample_rate'] // hop
    n_hop_e = t_e * cfg_d['sample_rate'] // hop
    n_hop_e = min(n_hop_e, n_frames + pad_len)
    n_delta = n_times * ((n_hop_e - n_hop_s) // n_times)
    n_s = n_hop_s
    n_e = n_hop_s + n_delta
    if n_e <= n_s:
        raise Exception(' - ERROR, translate: n_e <= n_s.')
    return pad_len, n_s, n_e


def _pad_all(mfcc, mel, stft, pad_len):
    if pad_len == 0:
        return mfcc, mel, stft
    print('Padding!!')
    out = []
    for a in (mfcc, mel, stft):
        out.append(np.concatenate([a, np.zeros((pad_len, a.shape[1]))], axis=0))
    print(out[0].shape, out[1].shape, out[2].shape)
    return out


def _vocode(vocoder, stft_true, stft_pred, cfg_d, n_iter, realse, giffin_lim_input):
    if vocoder is None:
        return None, None
    if vocoder == 'default':
        import audio_lib
        vocoder = audio_lib.from_power_to_wav
    kw = dict(P_dB_norm_factor=cfg_d['P_dB_norm_factor'], pre_emphasis=cfg_d['pre_emphasis'],
              hop_length=cfg_d['hop_length'], win_length=cfg_d['win_length'],
              mean_abs_amp_norm=15 * cfg_d['mean_abs_amp_norm'], n_iter=n_iter, n_fft=cfg_d['n_fft'])
    y_true = vocoder(stft_true, realse=1.0, **kw) if giffin_lim_input else None
    return y_true, vocoder(stft_pred, realse=realse, **kw)


def conversion2(decoder, mfcc, mel, stft, cfg_d, t_s=5, t_e=60, n_iter=200, output_path='./output',
                file_name='y_wav', realse=1.0, save_output=False, giffin_lim_input=True, play_conversion=False,
                vocoder='default'):
    """test.py:87-201: half-overlapped double pass + ``compound``."""
    n_times = cfg_d['n_timesteps']
    pad_len, n_s, n_e = window_plan(mfcc.shape[0], cfg_d, t_s, t_e)
    mfcc, mel, stft = _pad_all(mfcc, mel, stft, pad_len)

    mfcc_input0 = mfcc[n_s:n_e].reshape((-1, n_times, mfcc.shape[-1]))
    y_pred0 = decoder.predict(mfcc_input0)
    if n_e - n_s > n_times:
        half = n_times // 2
        mfcc_input1 = mfcc[(n_s + half):(n_e - half)].reshape((-1, n_times, mfcc.shape[-1]))
        y_pred1 = decoder.predict(mfcc_input1)
        mel_pred = compound(y_pred0.y_mel, y_pred1.y_mel)
        stft_pred = compound(y_pred0.y_stft, y_pred1.y_stft)
        phn_pred = compound(y_pred0.y_phn, y_pred1.y_phn)
    else:
        mel_pred = y_pred0.y_mel.reshape((-1, y_pred0.y_mel.shape[-1]))
        stft_pred = y_pred0.y_stft.reshape((-1, y_pred0.y_stft.shape[-1]))
        phn_pred = y_pred0.y_phn.reshape((-1, y_pred0.y_phn.shape[-1]))

    mel_true = mel[n_s:n_e]
    stft_true = stft[n_s:n_e]
    y_wav_true, y_wav_pred = _vocode(vocoder, stft_true, stft_pred, cfg_d, n_iter, realse, giffin_lim_input)
    ret_tuple = namedtuple('conversion', 'y_wav_true y_wav_pred mel_true mel_pred stft_true stft_pred phn_pred')
    return ret_tuple(y_wav_true, y_wav_pred, mel_true, mel_pred, stft_true, stft_pred, phn_pred)


def conversion(decoder, mfcc, mel, stft, cfg_d, t_s=5, t_e=60, n_iter=200, output_path='./output',
               file_name='y_wav', realse=1.0, save_output=False, giffin_lim_input=True, play_conversion=False,
               vocoder='default'):
    """test.py:206-306: single pass over non-overlapping windows."""
    n_times = cfg_d['n_timesteps']
    pad_len, n_s, n_e = window_plan(mfcc.shape[0], cfg_d, t_s, t_e)
    mfcc, mel, stft = _pad_all(mfcc, mel, stft, pad_len)
    y_pred = decoder.predict(mfcc[n_s:n_e].reshape((-1, n_times, mfcc.shape[-1])))
    mel_true = mel[n_s:n_e]
    mel_pred = y_pred.y_mel.reshape((-1, y_pred.y_mel.shape[-1]))
    stft_true = stft[n_s:n_e]
    stft_pred = y_pred.y_stft.reshape((-1, y_pred.y_stft.shape[-1]))
    y_wav_true, y_wav_pred = _vocode(vocoder, stft_true, stft_pred, cfg_d, n_iter, realse, giffin_lim_input)
    ret_tuple = namedtuple('conversion', 'y_wav_true y_wav_pred mel_true mel_pred stft_true stft_pred')
    return ret_tuple(y_wav_true, y_wav_pred, mel_true, mel_pred, stft_true, stft_pred)
