#!/usr/bin/env python
"""Benchmark of the hot path on MI355X.  Contract: prints ONE JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload full|frontend|train|vocoder]

A "step" is one pass of the hot path over one synthetic batch that is already resident in
HBM.  Multi-GPU: one process per GPU (torch.distributed.run), the utterance batch is sharded
one full batch per rank with NO data-path collective (inference shards by utterance: SURVEY.md
section 8e) => weak scaling; the timed region is bracketed by barrier + synchronize and the MAX
over ranks is reported.

Workloads
  full      (default) the metric's path: STFT+mel front-end on batch 32 x 4 s @ 16 kHz (the input
            of BASELINE.json configs[1]) -> 64 windows of 400 frames -> encoder + decoder forward
            (bf16 MFMA, f32 accumulate; --dtype float32 for the exact-f32 path); 25,600 frames
            per step and rank.
  frontend  BASELINE.json configs[1] alone: STFT+mel front-end, fp32 (25,632 frames).
Extra objects on the line: "roofline" (dominant kernel, HIP-event timed on the launch stream),
"cpu_baseline" (oracle timed on the host, rank 0, N=1 only), "stages".
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# Independent steps are pipelined over HIP streams (see bench_full); the HIP runtime multiplexes streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams that share a queue serialise.  Must be set before
# the runtime initialises.  Measured (tools/ab_step.py, interleaved on one box): 3 streams / 4 queues 2.09 ms per
# step, 4 streams / 4 queues 2.52, 10 streams / 16 queues 1.92.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')

import numpy as np
import torch

FE_KW = dict(sr=16000, pre_emphasis=0.97, hop_length=80, win_length=400, n_mels=80, n_mfcc=40,
             n_fft=None, window='hann', mfcc_normaleze_first_mfcc=True, mfcc_norm_factor=0.01,
             calc_mfcc_derivate=True, M_dB_norm_factor=0.01, P_dB_norm_factor=0.01,
             mean_abs_amp_norm=0.003, clip_output=True)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FE_BYTES_PER_FRAME = 1764        # 320 B in + 1,444 B out per frame


def synth_audio(B, L, seed):
    """Speech-like synthetic audio, generated with torch on the host so
    bench.py does not depend on the oracle for its inputs."""
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(L, dtype=torch.float64) / 16000.0
    f0 = 80 + 170 * torch.rand(B, 1, generator=g, dtype=torch.float64)
    x = torch.zeros(B, L, dtype=torch.float64)
    for h in range(1, 6):
        ph = 2 * np.pi * torch.rand(B, 1, generator=g, dtype=torch.float64)
        x += (1.0 / h) * torch.sin(2 * np.pi * f0 * h * t + ph)
    am_f = 2 + 6 * torch.rand(B, 1, generator=g, dtype=torch.float64)
    am_p = 2 * np.pi * torch.rand(B, 1, generator=g, dtype=torch.float64)
    x *= 0.55 + 0.45 * torch.sin(2 * np.pi * am_f * t + am_p)
    x += 0.05 * torch.randn(B, L, generator=g, dtype=torch.float64)
    x *= 0.5 / x.abs().amax(dim=1, keepdim=True)
    return x.float()


def _pmc_traffic(kernel):
    """HBM-side bytes per launch from the committed rocprofv3 --pmc passes (profiles/r01/pmc_summary.json)."""
    try:
        return json.load(open(os.path.join(ROOT, 'profiles', 'r01', 'pmc_summary.json')))[kernel]['traffic_bytes_per_launch']
    except Exception:
        return None


def time_events(fn, iters):
    """Average device time of fn() in ms, HIP events on the current (= launch) stream."""
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def bench_frontend(args, rank, world):
    import audio_lib
    B, L = 32, 64000
    wav = synth_audio(B, L, seed=rank).cuda()
    Fmax = 1 + L // 80
    frames = B * Fmax
    out = None

    def step():
        nonlocal out
        out = audio_lib.calc_MFCC_input_batch(wav, None, out=out, **FE_KW)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0

    extra = {}
    if rank == 0:
        iters = max(20, args.steps)
        k = {}
        for name, mask in (('power400', 2), ('finalize', 4), ('all', 7)):       # (sum|x| rides in the STFT kernel)
            k[name] = time_events(lambda m=mask: audio_lib.calc_MFCC_input_batch(
                wav, None, out=out, stage_mask=m, **FE_KW), iters)
        alg = FE_BYTES_PER_FRAME * frames
        ach = alg / (k['power400'] * 1e-3) / 1e9
        extra['roofline'] = {'kernel': 'fe_power400_kernel', 'bound': 'hbm', 'achieved': round(ach, 1),
                             'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4),
                             'traffic': _pmc_traffic('fe_power400_kernel'), 'algorithmic_bytes_per_launch': alg,
                             'avg_kernel_ms': round(k['power400'], 5)}
        extra['stages'] = {'kernel_ms': {n: round(v, 5) for n, v in k.items()},
                           'frontend_pipeline_GBps': round(alg / (k['all'] * 1e-3) / 1e9, 1)}
    return frames, dt, extra, {'workload': 'frontend: STFT+mel+MFCC, batch 32 x 4 s @ 16 kHz (BASELINE configs[1])',
                               'batch': B, 'samples': L, 'frames_per_step_per_gpu': frames}


def load_models(dtype, rank):
    """encoder (real enc_14 weights from the committed trimmed checkpoint) + decoder (random-init
    weights of the shipped architecture: Glorot kernels, TF default biases -- no decoder checkpoint
    exists, .gitignore:3 of the reference) at hp/*.json sizes."""
    import contextlib
    import io
    from encoder import encoder_spec_phn
    from decoder import decoder_specs
    from aux_func import load_cfg_d
    hp = os.path.join(ROOT, 'speech-cloner_amd', 'hp')
    with contextlib.redirect_stdout(io.StringIO()):
        enc_cfg = load_cfg_d(os.path.join(hp, 'encoder_cfg_d.json'))
        dec_cfg = load_cfg_d(os.path.join(hp, 'decoder_cfg_d.json'))
        enc_cfg.update(is_training=False, compute_dtype=dtype,
                       model_path=os.path.join(ROOT, 'tests', 'golden', 'enc_14_ckpt'))
        dec_cfg.update(is_training=False)
        enc = encoder_spec_phn(enc_cfg, None)
        dec = decoder_specs(dec_cfg, None, enc)
    return enc, dec


ENC_FLOP_PER_FRAME = 482720
DEC_FLOP_PER_FRAME = 66321920
MFMA_BF16_PEAK_TF = 2500.0           # MI355X_MICROARCH.md: dense bf16
MFMA_F32_PEAK_TF = 157.3


def bench_full(args, rank, world):
    """front-end on 32 x 4 s (configs[1] input) -> first 800 frames of every utterance as two
    400-frame windows -> encode + decode, `--window-batch` windows per launch."""
    import audio_lib
    import modules
    B, L, T = 32, 64000, 400
    wav = synth_audio(B, L, seed=rank).cuda()
    enc, dec = load_models(args.dtype, rank)
    nwin = B * 2
    frames = nwin * T
    fe_out = None
    res = {}

    streams = [torch.cuda.Stream() for _ in range(args.streams)] if args.streams > 1 else None
    chunk_no = [0]

    def step():
        nonlocal fe_out
        fe_out = audio_lib.calc_MFCC_input_batch(wav, None, out=fe_out, **FE_KW)
        x = fe_out[0][:, :2 * T, :].reshape(nwin, T, 80)
        if streams is None:
            for i in range(0, nwin, args.window_batch):
                o = dec.forward(x[i:i + args.window_batch].contiguous())
                res[i] = (o['y_mel'], o['y_stft'], o['y_phn'])
            return
        # independent window chunks on separate HIP streams: one chunk's latency-bound recurrences
        # (<= 128 workgroups) overlap with another chunk's GEMMs
        # (consecutive steps are independent batches, so they also alternate streams: step i+1's
        # front-end and GEMMs run under step i's recurrences; everything is joined by the
        # synchronize() that closes the timed region)
        main = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(main)
        for i in range(0, nwin, args.window_batch):
            st_ = streams[chunk_no[0] % len(streams)]
            chunk_no[0] += 1
            st_.wait_event(ready)
            with torch.cuda.stream(st_):
                xi = x[i:i + args.window_batch].contiguous()
                xi.record_stream(st_)
                copied = torch.cuda.Event()
                copied.record(st_)
                main.wait_event(copied)                # the next step's front-end overwrites fe_out
                o = dec.forward(xi)
                res[(i, chunk_no[0] % (2 * len(streams)))] = (o['y_mel'], o['y_stft'], o['y_phn'])

    # setup, like loading the model: every stream's allocator pool and every weight-layout cache is built once
    for _ in range(2 * args.streams if streams else 1):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0

    extra = {}
    if rank == 0:
        # dominant-kernel timings with HIP events on the launch stream
        x = fe_out[0][:, :2 * T, :].reshape(nwin, T, 80)[:args.window_batch].contiguous()
        st = dec.store
        W = args.window_batch
        peak = MFMA_BF16_PEAK_TF if args.dtype == 'bfloat16' else MFMA_F32_PEAK_TF
        with modules.variable_store(st), modules.variable_scope('decoder'), modules.variable_scope('step2'), \
                modules.variable_scope('CBHG'):
            pre = torch.randn(W, T, 256, device='cuda').to(st.dtype)
            ms_bank = time_events(lambda: modules.conv1d_banks(pre, K=32, is_training=False), 20)
            bank = modules.conv1d_banks(pre, K=32, is_training=False)
            ms_p1 = time_events(lambda: modules.conv1d(bank, filters=256, size=3, scope="conv1d_1", bn_scope="conv1d_1",
                                                       activation_fn='relu', pool_input=2), 20)
            ms_gru = time_events(lambda: modules.gru(pre, num_units=256, bidirection=True), 5)
        fl_bank = 2.0 * 256 * 128 * 528 * W * T
        fl_p1 = 2.0 * 3 * 4096 * 256 * W * T
        ms_fe = time_events(lambda: audio_lib.calc_MFCC_input_batch(wav, None, out=fe_out, **FE_KW), 20)
        ms_enc = time_events(lambda: enc.forward(x), 5)
        ms_all = time_events(lambda: dec.forward(x), 5)
        ach = fl_bank / (ms_bank * 1e-3) / 1e12
        # HBM-side bytes per launch of this kernel from the committed rocprofv3 --pmc passes
        # (profiles/r01/pmc_summary.json: (2*FETCH_SIZE + WRITE_SIZE) * 1024, gfx950 correction)
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, 'profiles', 'r01', 'pmc_summary.json')))
            if args.dtype == 'bfloat16' and W == 64:
                traffic = pm['bank256_kernel_bf16_step2']['traffic_bytes_per_launch']
        except Exception:
            traffic = None
        kname = 'bank256_kernel' if args.dtype == 'bfloat16' else 'conv_kernel<float32>'
        extra['roofline'] = {'kernel': '%s (decoder step2 conv1d_banks, 32 filter widths)' % kname,
                             'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
                             'frac': round(ach / peak, 4), 'traffic': traffic,
                             'algorithmic_flop_per_launch': fl_bank, 'avg_kernel_ms': round(ms_bank, 4)}
        extra['stages'] = {
            'frontend_ms': round(ms_fe, 4), 'frontend_frames_per_s': round(B * (1 + L // 80) / (ms_fe * 1e-3), 1),
            'frontend_GBps_vs_8TBps': round(FE_BYTES_PER_FRAME * B * (1 + L // 80) / (ms_fe * 1e-3) / 1e9, 1),
            'encoder_ms_per_%d_windows' % W: round(ms_enc, 4),
            'encode_decode_ms_per_%d_windows' % W: round(ms_all, 4),
            'dec2_bank_ms': round(ms_bank, 4), 'dec2_bank_TFLOPs': round(ach, 2),
            'dec2_proj1_ms': round(ms_p1, 4), 'dec2_proj1_TFLOPs': round(fl_p1 / (ms_p1 * 1e-3) / 1e12, 2),
            'dec2_gru_ms': round(ms_gru, 4), 'dec2_gru_us_per_step': round(ms_gru * 1e3 / T, 3),
            'model_TFLOPs_end_to_end': round((ENC_FLOP_PER_FRAME + DEC_FLOP_PER_FRAME) * W * T / (ms_all * 1e-3) / 1e12, 2)}
    cfg = {'workload': 'full: STFT+mel front-end on batch 32 x 4 s @ 16 kHz (configs[1] input) -> 64 windows of 400 '
                       'frames -> encoder (enc_14 weights) + decoder (hp/decoder_cfg_d.json sizes, random init), '
                       '%d windows per launch, independent steps pipelined over %d HIP streams' % (args.window_batch, args.streams),
           'batch': B, 'samples': L, 'windows': nwin, 'frames_per_step_per_gpu': frames, 'model_dtype': args.dtype,
           'streams': args.streams}
    return frames, dt, extra, cfg


def bench_train(args, rank, world):
    """BASELINE configs[4]: decoder training step (fwd + bwd + Adam, float32) on synthetic
    ARCTIC-slt-shaped targets, 32 windows per GPU, gradients all-reduced over RCCL."""
    import contextlib
    import io
    B, T = 32, 400
    enc, dec = None, None
    with contextlib.redirect_stdout(io.StringIO()):
        from aux_func import load_cfg_d
        from encoder import encoder_spec_phn
        from decoder import decoder_specs
        hp = os.path.join(ROOT, 'speech-cloner_amd', 'hp')
        enc_cfg = load_cfg_d(os.path.join(hp, 'encoder_cfg_d.json'))
        dec_cfg = load_cfg_d(os.path.join(hp, 'decoder_cfg_d.json'))
        enc_cfg.update(is_training=False, model_path=os.path.join(ROOT, 'tests', 'golden', 'enc_14_ckpt'))
        dec_cfg.update(is_training=True)
        enc = encoder_spec_phn(enc_cfg, None)
        dec = decoder_specs(dec_cfg, None, enc)
    g = torch.Generator().manual_seed(100 + rank)
    mfcc = (torch.rand(B, T, 80, generator=g) * 0.4 - 0.2).cuda()
    mel = (torch.rand(B, T, 80, generator=g) * 0.8).cuda()
    stft = (torch.rand(B, T, 201, generator=g) * 0.8).cuda()
    last = None
    for _ in range(args.warmup):
        last = dec.exec_train_step(mfcc, mel, stft)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = dec.exec_train_step(mfcc, mel, stft)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    extra = {'stages': {'last_loss': float(last[2]), 'global_step': int(last[3]),
                        'params': int(dec._trainer.total), 'allreduce_MB': round(dec._trainer.total * 4 / 1e6, 1)}}
    cfg = {'workload': 'train: decoder fwd+bwd+Adam (float32) on 32 windows x 400 frames per GPU, encoder frozen '
                       '(BASELINE configs[4])', 'global_batch': B * world, 'frames_per_step_per_gpu': B * T}
    return B * T, dt, extra, cfg


def bench_vocoder(args, rank, world):
    """Griffin-Lim vocoder: one step = from_power_to_wav on a batch of
    predicted power spectrograms, 200 iterations as test.py:87 uses."""
    import audio_lib
    B, F, n_iter = 16, 1000, 200
    g = torch.Generator().manual_seed(300 + rank)
    wav = synth_audio(B, 80 * (F - 1), seed=300 + rank).cuda()
    _, _, P = audio_lib.calc_MFCC_input_batch(wav, None, **FE_KW)
    P = P[:, :F].contiguous()
    ph = (torch.rand(B, F, 201, generator=g) * math.pi).cuda()
    kw = dict(P_dB_norm_factor=0.01, pre_emphasis=0.97, hop_length=80, win_length=400, mean_abs_amp_norm=0.045,
              n_iter=n_iter, n_fft=None, realse=1.0, phase0=ph)
    for _ in range(args.warmup):
        audio_lib.from_power_to_wav_batch(P, None, **kw)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = audio_lib.from_power_to_wav_batch(P, None, **kw)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = audio_lib.from_power_to_wav_batch(P, None, **kw)
    e1.record()
    torch.cuda.synchronize()
    us_iter = e0.elapsed_time(e1) * 1e3 / n_iter
    # per launch: every frame is gathered once (5 overlapping reads hit L2) and written once
    alg_bytes = B * F * (2 * 400 + 201) * 4
    extra = {'stages': {'us_per_iteration': round(us_iter, 2), 'iterations': n_iter,
                        'audio_seconds_per_step': round(B * 80 * (F - 1) / 16000.0, 1)},
             'roofline': {'bound': 'hbm', 'achieved': round(alg_bytes / (us_iter * 1e-6) / 1e9, 1), 'peak': 8000.0,
                          'unit': 'GB/s', 'frac': round(alg_bytes / (us_iter * 1e-6) / 8e12, 4), 'traffic': None,
                          'kernel': 'gl_iter400_kernel<false> (time per iteration from events around the 200-launch '
                                    'chain; transform arithmetic and LDS traffic, not HBM, set its duration)'}}
    cfg = {'workload': 'vocoder: from_power_to_wav, %d utterances x %d frames (5 s each), %d Griffin-Lim iterations, '
                       'n_fft 400 hop 80' % (B, F, n_iter), 'frames_per_step_per_gpu': B * F}
    return B * F, dt, extra, cfg


def cpu_baseline_vocoder():
    """oracle/vocoder_oracle.py (numpy float64 restatement of the librosa loop) on one utterance of
    the same shape with 20 of the 200 iterations, scaled to 200."""
    from oracle import frontend_oracle as fo
    from oracle import vocoder_oracle as vo
    F = 1000
    wav = synth_audio(1, 80 * (F - 1), seed=300).numpy()[0]
    P = fo.calc_MFCC_input(wav, **FE_KW)[2][:F]
    t0 = time.perf_counter()
    vo.from_power_to_wav(P, 0.01, 0.97, 80, 400, 0.045, n_iter=20, n_fft=None, seed=0)
    dt = (time.perf_counter() - t0) * 10.0
    return {'value': round(F / dt, 1), 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
            'sample': '1 of the 16 utterances (1000 frames), 20 of the 200 iterations timed and scaled x10; '
                      'oracle/vocoder_oracle.py (numpy restatement of librosa.istft/stft; single thread)'}


def cpu_baseline_full():
    """Oracle timed on the host: front-end on 2 utterances (numpy) + encode/decode of 2 windows with
    torch-CPU float32 ops at the shipped sizes (all host threads)."""
    from oracle import frontend_oracle as fo
    from oracle import model_oracle as mo
    import contextlib
    import io
    import tf_bundle
    from aux_func import load_cfg_d
    hp = os.path.join(ROOT, 'speech-cloner_amd', 'hp')
    with contextlib.redirect_stdout(io.StringIO()):
        enc_cfg = load_cfg_d(os.path.join(hp, 'encoder_cfg_d.json'))
        dec_cfg = load_cfg_d(os.path.join(hp, 'decoder_cfg_d.json'))
    w = tf_bundle.read_bundle(os.path.join(ROOT, 'tests', 'golden', 'enc_14_ckpt', 'encoder-136512'), verify_crc=False)
    we = mo.to_torch({k: v for k, v in w.items() if k.startswith('encoder/')})
    wd = mo.to_torch(mo.init_weights(dec_cfg, 'decoder', seed=2))
    wav = synth_audio(2, 64000, seed=0).numpy()
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, 16))                 # the GPU box gives one GPU a 16-core share
    torch.set_num_threads(ncores)
    t0 = time.perf_counter()
    feats = [fo.calc_MFCC_input(wav[b], **FE_KW)[0][:400] for b in range(2)]
    x = torch.from_numpy(np.stack(feats))
    with torch.no_grad():
        _, pr, _, _ = mo.encoder_forward(x, we, enc_cfg)
        mo.decoder_forward(pr, wd, dec_cfg)
    dt = time.perf_counter() - t0
    return {'value': round(800 / dt, 1), 'unit': 'frames/s', 'cores': ncores, 'kind': 'port',
            'sample': '2 utterances -> 2 windows (800 frames) through oracle/frontend_oracle.py (numpy) and '
                      'oracle/model_oracle.py (torch-CPU float32, %d threads); one pass, no warm-up' % ncores}


def cpu_baseline_frontend():
    """Oracle (numpy/scipy restatement of librosa's path) timed on the host: 8 utterances of the
    same workload (~2-4 s of CPU work per pass, one warm-up + 2 timed passes)."""
    from oracle import frontend_oracle as fo
    wav = synth_audio(8, 64000, seed=0).numpy()
    fo.calc_MFCC_input(wav[0], **FE_KW)
    t0 = time.perf_counter()
    n = 0
    for _ in range(2):
        for b in range(wav.shape[0]):
            n += fo.calc_MFCC_input(wav[b], **FE_KW)[0].shape[0]
    dt = time.perf_counter() - t0
    return {'value': round(n / dt, 1), 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
            'sample': '2 passes over 8 of the 32 utterances (4 s each) through oracle/frontend_oracle.py '
                      '(numpy/scipy restatement of librosa 0.6; single thread)'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=6)
    ap.add_argument('--workload', default='full', choices=['full', 'frontend', 'train', 'vocoder'])
    ap.add_argument('--dtype', default='bfloat16', choices=['bfloat16', 'float32'])
    ap.add_argument('--window-batch', type=int, default=64)
    ap.add_argument('--streams', type=int, default=10,
                    help='HIP streams the independent window chunks / consecutive steps are pipelined over')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    import dist_util
    rank, local, world = dist_util.env_world()
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU')
    torch.cuda.set_device(local)
    dist_util.init('nccl')

    if args.workload == 'frontend':
        frames, dt, extra, cfg = bench_frontend(args, rank, world)
    elif args.workload == 'train':
        frames, dt, extra, cfg = bench_train(args, rank, world)
        args.no_cpu_baseline = True
    elif args.workload == 'vocoder':
        frames, dt, extra, cfg = bench_vocoder(args, rank, world)
    else:
        frames, dt, extra, cfg = bench_full(args, rank, world)

    dt = dist_util.max_over_ranks(dt, device='cuda')
    if rank == 0:
        line = {'metric': 'mel frames/sec', 'value': round(frames * world * args.steps / dt, 1), 'unit': 'frames/s',
                'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': round(dt / args.steps * 1e3, 5), 'higher_is_better': True, 'scaling': 'weak',
                'vs_baseline': None, 'dtype': 'f32' if args.workload in ('frontend', 'train', 'vocoder') else
                ('bf16' if args.dtype == 'bfloat16' else 'f32'), 'data': 'synthetic',
                'config': dict(cfg, parallelism='utterance-sharded x%d, no collective' % world)}
        line.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = {'frontend': cpu_baseline_frontend, 'vocoder': cpu_baseline_vocoder}.get(
                args.workload, cpu_baseline_full)()
        print(json.dumps(line))
    dist_util.finalize()


if __name__ == '__main__':
    main()
