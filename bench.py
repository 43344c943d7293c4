#!/usr/bin/env python
"""Benchmark of the hot path on MI355X.  Contract: prints ONE JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload full|frontend|train|vocoder]

Launch forms.  (1) Under torch.distributed.run (RANK / WORLD_SIZE in the environment): this process is one rank.
(2) Plain `python bench.py --gpus N` with N > 1: this process makes NO GPU call; it starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child, relays rank 0's JSON line and exits
with the child's status.  Either way n_gpus is the number of ranks that actually joined the process group, and the
run fails if that differs from --gpus.

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
    """HBM-side bytes per launch from the committed rocprofv3 --pmc passes (profiles/rNN/pmc_summary.json, newest round first)."""
    for rnd in ('r03', 'r02', 'r01'):
        try:
            return json.load(open(os.path.join(ROOT, 'profiles', rnd, 'pmc_summary.json')))[kernel]['traffic_bytes_per_launch']
        except Exception:
            continue
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


def _pmc_frontend_traffic():
    one = _pmc_traffic('fe400_one_launch')
    if one is not None:
        return one
    tr = [_pmc_traffic(n) for n in ('fe400_stats_pass', 'fe400_feature_pass')]
    return (tr[0] + tr[1]) if all(t is not None for t in tr) else None


def frontend_kernel_figures(wav, out, iters):
    """BASELINE configs[1] judged on bytes: the algorithmic 1,764 B/frame of the WHOLE front-end against the time of ALL
    its launches (the statistics pass exists only because the normalisations need the utterance's extremes first;
    crediting the bytes to the feature pass alone would flatter it).  HIP events around back-to-back launch sets."""
    import audio_lib
    B, L = wav.shape
    frames = B * (1 + L // 80)
    k = {}
    for name, mask in (('stats_pass', 2), ('feature_pass', 4), ('all', 6)):
        k[name] = time_events(lambda m=mask: audio_lib.calc_MFCC_input_batch(wav, None, out=out, stage_mask=m, **FE_KW), iters)
    alg = FE_BYTES_PER_FRAME * frames
    ach = alg / (k['all'] * 1e-3) / 1e9
    return k, alg, ach, frames


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
        iters = max(50, args.steps)
        k, alg, ach, _ = frontend_kernel_figures(wav, out, iters)
        extra['roofline'] = {'kernel': 'fe400_fused_kernel (the whole front-end in one launch; stats_pass / feature_pass: the two-launch form)',
                             'bound': 'hbm', 'achieved': round(ach, 1),
                             'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4),
                             'traffic': _pmc_frontend_traffic(),
                             'algorithmic_bytes_per_launch': alg,
                             'avg_kernel_ms': round(k['all'], 5),
                             'timing': 'HIP events on the launch stream, average of %d back-to-back launches' % iters,
                             'two_launch_form_ms': round(k['stats_pass'] + k['feature_pass'], 5),
                             'valu_floor_note': 'the launch issues 12.8 M vector wave-instructions (PMC, profiles/r03; the two-launch '
                                                'form 17.0 M): at one per 2 cycles per SIMD that alone is ~11 us = 50 % of the 8 TB/s '
                                                'line; the f32 FFT and the wait for the utterance, not HBM, bound this path '
                                                '(DESIGN.md section 6)'}
        extra['stages'] = {'kernel_ms': {n: round(v, 5) for n, v in k.items()},
                           'frontend_pipeline_GBps': round(alg / (k['all'] * 1e-3) / 1e9, 1)}
    return frames, dt, extra, {'workload': 'frontend: STFT+mel+MFCC, batch 32 x 4 s @ 16 kHz (BASELINE configs[1])',
                               'batch': B, 'samples': L, 'frames_per_step_per_gpu': frames}


def frontend_side_measurement(wav):
    """BASELINE configs[1] inside the default line (rank 0): the front-end alone on the same resident 32 x 4 s batch,
    every frame stored (25,632 frames), HIP events over 50 back-to-back launch sets."""
    import audio_lib
    out = audio_lib.calc_MFCC_input_batch(wav, None, **FE_KW)
    k, alg, ach, frames = frontend_kernel_figures(wav, out, 50)
    return {'workload': 'BASELINE configs[1]: STFT+mel+MFCC on %d x 4 s @ 16 kHz, float32, %d frames' % (wav.shape[0], frames),
            'ms': round(k['all'], 5), 'kernel': 'fe400_fused_kernel (one launch)',
            'two_launch_form': {'stats_pass_ms': round(k['stats_pass'], 5), 'feature_pass_ms': round(k['feature_pass'], 5)},
            'frames_per_s': round(frames / (k['all'] * 1e-3), 1),
            'GBps': round(ach, 1), 'frac': round(ach / HBM_PEAK_GBS, 4), 'peak_GBps': HBM_PEAK_GBS,
            'algorithmic_bytes': alg, 'traffic': _pmc_frontend_traffic(),
            'timing': 'HIP events on the launch stream, 50 back-to-back launches'}


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


class _Pipeline:
    """The bench's step: front-end on the resident batch -> 64 windows -> encode + decode, consecutive steps rotating
    over `n_streams` HIP streams (independent batches: one step's latency-bound recurrences run under another step's
    MFMA-bound filter banks; everything is joined by the synchronize() that closes a timed region)."""

    def __init__(self, wav, dec, window_batch, n_streams):
        import audio_lib
        self.audio_lib, self.wav, self.dec, self.wb = audio_lib, wav, dec, window_batch
        self.streams = [torch.cuda.Stream() for _ in range(n_streams)] if n_streams > 1 else None
        self.n_streams = n_streams
        self.chunk_no = 0
        self.step_no = 0
        # The front-end stores the first 800 frames of every 4 s utterance (801 exist; all of them count for the
        # utterance's statistics), so [B, 800, 80] IS the window batch [2B, 400, 80]: no copy, no torch kernel in the
        # step.  One set of feature buffers per step in flight; a set is rewritten only after its consumers finished.
        self.fe_ring = [None] * (n_streams + 1 if n_streams > 1 else 1)
        self.fe_done = [[] for _ in self.fe_ring]
        self.res = {}
        self.nwin = wav.shape[0] * 2

    def step(self):
        k = self.step_no % len(self.fe_ring)
        self.step_no += 1
        main = torch.cuda.current_stream()
        for ev in self.fe_done[k]:
            main.wait_event(ev)
        self.fe_done[k] = []
        self.fe_ring[k] = self.audio_lib.calc_MFCC_input_batch(self.wav, None, out=self.fe_ring[k], out_frames=800, **FE_KW)
        x = self.fe_ring[k][0].view(self.nwin, 400, 80)
        if self.streams is None:
            for i in range(0, self.nwin, self.wb):
                o = self.dec.forward(x[i:i + self.wb])
                self.res[i] = (o['y_mel'], o['y_stft'], o['y_phn'])
            return
        ready = torch.cuda.Event()
        ready.record(main)
        for i in range(0, self.nwin, self.wb):
            st_ = self.streams[self.chunk_no % len(self.streams)]
            self.chunk_no += 1
            st_.wait_event(ready)
            with torch.cuda.stream(st_):
                o = self.dec.forward(x[i:i + self.wb])
                self.res[(i, self.chunk_no % (2 * len(self.streams)))] = (o['y_mel'], o['y_stft'], o['y_phn'])
                ev = torch.cuda.Event()
                ev.record(st_)
                self.fe_done[k].append(ev)

    def setup(self):
        # like loading the model: every stream's allocator pool and every weight-layout cache is built once
        for _ in range(2 * self.n_streams if self.streams else 1):
            self.step()
        torch.cuda.synchronize()

    def timed(self, steps, warmup, world):
        """Throughput mode (several batches in flight): a launch costs its CU time, not its makespan.  The two kernels that
        fill an idle chip at the price of extra workgroup time -- the k = 3 projection's K split (+0.8 % ms/step here) and
        the one-launch front-end, whose blocks wait for their utterance (+5.5 %) -- are switched off for this loop
        (`_vc.throughput_mode()`; profiles/r03/ab_step_latency_kernels.log); the single-stream figures of the same line
        run with the library's defaults."""
        import _vc
        if self.streams is not None:
            with _vc.throughput_mode():
                return self._timed(steps, warmup, world)
        return self._timed(steps, warmup, world)

    def _timed(self, steps, warmup, world):
        for _ in range(warmup):
            self.step()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        return time.perf_counter() - t0


def _rocprof_bank_avg():
    """Average duration (ms) of the step-2 filter-bank launch in the committed rocprofv3 --kernel-trace --stats run of
    tools/prof_kernels.py bank (>= 50 calls), profiles/rNN/bank_step2_kernel_stats.csv -- reported beside the live
    HIP-event figure so the line can be checked against profiles/."""
    import csv
    for rnd in ('r03', 'r02', 'r01'):
        f = os.path.join(ROOT, 'profiles', rnd, 'bank_step2_kernel_stats.csv')
        try:
            for r in csv.DictReader(open(f)):
                if 'bank256_kernel' in r['Name']:
                    return {'file': 'profiles/%s/bank_step2_kernel_stats.csv' % rnd, 'calls': int(r['Calls']),
                            'avg_ms': round(float(r['AverageNs']) / 1e6, 4)}
        except Exception:
            continue
    return None


def bench_full(args, rank, world):
    """front-end on 32 x 4 s (configs[1] input) -> first 800 frames of every utterance as two
    400-frame windows -> encode + decode, `--window-batch` windows per launch."""
    import audio_lib
    import modules
    B, L, T = args.batch, 64000, 400
    wav = synth_audio(B, L, seed=rank).cuda()
    enc, dec = load_models(args.dtype, rank)
    nwin = B * 2
    frames = nwin * T
    pipe = _Pipeline(wav, dec, args.window_batch, args.streams)
    pipe.setup()
    dt = pipe.timed(args.steps, args.warmup, world)
    fe_out = pipe.fe_ring[0]

    extra = {}
    if rank == 0:
        # dominant-kernel timings with HIP events on the launch stream
        x = fe_out[0].view(nwin, T, 80)[:args.window_batch]
        st = dec.store
        W = args.window_batch
        peak = MFMA_BF16_PEAK_TF if args.dtype == 'bfloat16' else MFMA_F32_PEAK_TF
        NL = 50                                          # launches averaged for the roofline line
        with modules.variable_store(st), modules.variable_scope('decoder'), modules.variable_scope('step2'), \
                modules.variable_scope('CBHG'):
            pre = torch.randn(W, T, 256, device='cuda').to(st.dtype)
            # the two launches exactly as modules.CBHG issues them: the bank launch stores the max-pooled result where
            # its kernel can (pool_output='auto'), and the projection then reads a plain operand
            ms_bank = time_events(lambda: modules.conv1d_banks(pre, K=32, is_training=False, pool_output='auto'), NL)
            bank, pooled = modules.conv1d_banks(pre, K=32, is_training=False, pool_output='auto')
            ms_p1 = time_events(lambda: modules.conv1d(bank, filters=256, size=3, scope="conv1d_1", bn_scope="conv1d_1",
                                                       activation_fn='relu', pool_input=0 if pooled else 2), 50)
            ms_gru = time_events(lambda: modules.gru(pre, num_units=256, bidirection=True), 5)
        fl_bank = 2.0 * 256 * 128 * 528 * W * T
        fl_p1 = 2.0 * 3 * 4096 * 256 * W * T
        ms_fe = time_events(lambda: audio_lib.calc_MFCC_input_batch(wav, None, out=fe_out, out_frames=800, **FE_KW), 20)
        ms_enc = time_events(lambda: enc.forward(x), 5)
        ms_all = time_events(lambda: dec.forward(x), 5)
        ach = fl_bank / (ms_bank * 1e-3) / 1e12
        # HBM-side bytes per launch of this kernel from the committed rocprofv3 --pmc passes
        # ((2*FETCH_SIZE + WRITE_SIZE) * 1024, gfx950 correction of MI355X_MICROARCH.md)
        traffic = None
        if args.dtype == 'bfloat16' and W == 64:
            for rnd in ('r03', 'r02', 'r01'):
                try:
                    pm = json.load(open(os.path.join(ROOT, 'profiles', rnd, 'pmc_summary.json')))
                    traffic = pm['bank256_kernel_bf16_step2']['traffic_bytes_per_launch']
                    break
                except Exception:
                    continue
        kname = 'bank256_kernel' if args.dtype == 'bfloat16' else 'conv_kernel<float32>'
        extra['roofline'] = {'kernel': '%s (decoder step2 conv1d_banks, 32 filter widths)' % kname,
                             'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
                             'frac': round(ach / peak, 4), 'traffic': traffic,
                             'algorithmic_flop_per_launch': fl_bank, 'avg_kernel_ms': round(ms_bank, 4),
                             'timing': 'HIP events on the launch stream, average of %d back-to-back launches on this box '
                                       '(the quantity frac is computed from); rocprof_committed = the same launch under '
                                       'rocprofv3 --kernel-trace --stats on the box that produced profiles/' % NL,
                             'rocprof_committed': _rocprof_bank_avg() if (args.dtype == 'bfloat16' and W == 64) else None}
        one = (ms_fe + ms_all * (nwin / W)) * 1e-3
        extra['single_stream'] = {'value': round(frames / one, 1), 'unit': 'frames/s', 'ms_per_step': round(one * 1e3, 4),
                                  'what': 'latency-bound caller: one batch at a time on one stream (front-end + encode + '
                                          'decode of %d windows, HIP events, no overlap between batches)' % nwin}
        extra['stages'] = {
            'frontend_ms': round(ms_fe, 4), 'frontend_frames_per_s': round(B * (1 + L // 80) / (ms_fe * 1e-3), 1),
            'frontend_GBps_vs_8TBps': round(FE_BYTES_PER_FRAME * B * (1 + L // 80) / (ms_fe * 1e-3) / 1e9, 1),
            'encoder_ms_per_%d_windows' % W: round(ms_enc, 4),
            'encode_decode_ms_per_%d_windows' % W: round(ms_all, 4),
            'dec2_bank_ms': round(ms_bank, 4), 'dec2_bank_TFLOPs': round(ach, 2),
            'dec2_proj1_ms': round(ms_p1, 4), 'dec2_proj1_TFLOPs': round(fl_p1 / (ms_p1 * 1e-3) / 1e12, 2),
            'dec2_gru_ms': round(ms_gru, 4), 'dec2_gru_us_per_step': round(ms_gru * 1e3 / T, 3),
            'model_TFLOPs_end_to_end': round((ENC_FLOP_PER_FRAME + DEC_FLOP_PER_FRAME) * W * T / (ms_all * 1e-3) / 1e12, 2)}
    # the reference's own precision (float32 arithmetic end to end, exact-f32 MFMA), same workload and pipeline, in the
    # same driver-observed line; every rank runs it so that ranks stay in step, rank 0 reports its own figure
    if args.dtype == 'bfloat16' and not args.no_f32:
        pipe = None
        torch.cuda.empty_cache()
        enc32, dec32 = load_models('float32', rank)
        pipe32 = _Pipeline(wav, dec32, args.window_batch, min(args.streams, 4))
        pipe32.setup()
        k32 = max(3, min(args.steps, 10))
        dt32 = pipe32.timed(k32, 1, 1)
        if rank == 0:
            extra['f32'] = {'value': round(frames / (dt32 / k32), 1), 'unit': 'frames/s per GPU', 'ms_per_step': round(dt32 / k32 * 1e3, 4),
                            'steps': k32, 'what': 'same step with float32 weights / activations / accumulation (the '
                                                  "reference's arithmetic type), %d streams" % min(args.streams, 4)}
        del pipe32, enc32, dec32
        torch.cuda.empty_cache()
    # BASELINE configs[1] (front-end alone) and configs[4] (training step) in the same driver-observed line
    if not args.no_side:
        if rank == 0:
            extra['frontend'] = frontend_side_measurement(wav)
        pipe = enc = dec = None
        torch.cuda.empty_cache()
        tr_line = train_side_measurement(rank, world, 'cpu' if args.backend == 'gloo' else 'cuda')
        if rank == 0:
            extra['train'] = tr_line
    cfg = {'workload': 'full: STFT+mel front-end on batch %d x 4 s @ 16 kHz (configs[1] input) -> %d windows of 400 '
                       'frames -> encoder (enc_14 weights) + decoder (hp/decoder_cfg_d.json sizes, random init), '
                       '%d windows per launch, independent steps pipelined over %d HIP streams' % (B, nwin, args.window_batch, args.streams),
           'batch': B, 'samples': L, 'windows': nwin, 'frames_per_step_per_gpu': frames, 'model_dtype': args.dtype,
           'streams': args.streams}
    return frames, dt, extra, cfg


def rccl_info(backend):
    """What the collective library is, for the day the line is produced on more than one GPU."""
    info = {'backend': backend, 'ranks': torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1}
    try:
        info['version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())        # RCCL reports through torch's nccl module
    except Exception as e:                                                          # pragma: no cover
        info['version'] = 'unavailable: %s' % e
    return info


def train_measure(rank, world, steps, warmup):
    """BASELINE configs[4] per-GPU shape: `steps` decoder training steps (fwd + bwd + Adam, float32, 32 windows x 400
    frames; under data parallelism the two gradient buckets are all-reduced, overlapped with the backward pass) timed
    wall-clock between barriers + synchronize; then the exchange alone (whole arena and bucket by bucket, HIP events)
    and, on rank 0, the dominant kernel.  Returns (frames per step, seconds, stages dict, roofline dict or None)."""
    import contextlib
    import io
    B, T = 32, 400
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
    for _ in range(warmup):
        last = dec.exec_train_step(mfcc, mel, stft)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = dec.exec_train_step(mfcc, mel, stft)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    tr = dec._trainer
    # the one exchange step of the path, alone: all-reduce of the flat gradient arena (every rank takes part), and the
    # two buckets forward_backward puts on the wire (stage 2 first, under stage 1's backward)
    ar_ms, buckets = None, None
    if world > 1:
        def timed_allreduce(t):
            torch.distributed.barrier()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 5
        ar_ms = timed_allreduce(tr.grad)
        buckets = []
        for st_ in ('step2', 'step1'):
            lo, hi = tr._slice_of('%s/%s/' % (dec._scope, st_))
            ms = timed_allreduce(tr.grad[lo:hi])
            buckets.append({'bucket': st_, 'MB': round((hi - lo) * 4 / 1e6, 1), 'allreduce_ms': round(ms, 4),
                            'busbw_GBps': round(2.0 * (world - 1) / world * (hi - lo) * 4 / (ms * 1e-3) / 1e9, 1)})
    stages = {'last_loss': float(last[2]), 'global_step': int(last[3]),
              'params': int(tr.total), 'allreduce_MB': round(tr.total * 4 / 1e6, 1),
              'allreduce_ms': None if ar_ms is None else round(ar_ms, 4),
              'allreduce_busbw_GBps': None if ar_ms is None else
              round(2.0 * (world - 1) / world * tr.total * 4 / (ar_ms * 1e-3) / 1e9, 1),
              'allreduce_buckets': buckets}
    roof = None
    if rank == 0:
        import gemm16
        import modules
        # dominant kernel of the step: gemm16_kernel on the step-2 filter bank (forward; the data- and filter-gradient
        # launches of the same layer do the same FLOPs), HIP events over 20 launches.  `achieved` is the ALGORITHMIC
        # rate -- the float32 convolution's 2 * MACs -- against the f32-input MFMA peak, the hardware's own float32
        # matrix rate; the kernel EXECUTES three float16 products per algorithmic one (`executed`).
        H, K, M = 256, 32, B * T
        pre = torch.randn(M, H, device='cuda')
        kern = [tr.w('decoder/step2/CBHG/conv1d_banks' + ('/conv1d' if k == 1 else '/num_%d/conv1d' % k) + '/conv1d/kernel')
                for k in range(1, K + 1)]
        w16 = gemm16.Weights16(pre.device)
        pairs, cs = gemm16.bank_forward_operands(w16, kern, H)
        w16.refresh()
        x16, rs = gemm16.split16(pre, M, H, H, T)
        zb = torch.empty((M, 128 * K), device='cuda')
        ms_bank = time_events(lambda: gemm16.gemm16(x16, rs, M, T, H, pairs, zb, 128 * K, col_scale=cs), 20)
        ms_split = time_events(lambda: gemm16.split16(pre, M, H, H, T), 20)
        import _vc
        with _vc.options(f32_f16x3=0), modules.variable_store(dec.store), modules.variable_scope('decoder'), \
                modules.variable_scope('step2'), modules.variable_scope('CBHG'):         # the f32-input MFMA kernel, for comparison
            ms_f32 = time_events(lambda: modules.conv1d_banks(pre.view(B, T, H), K=32, is_training=False), 20)
        fl_bank = 2.0 * 256 * 128 * 528 * B * T
        ach = fl_bank / (ms_bank * 1e-3) / 1e12
        roof = {'kernel': 'gemm16_kernel (decoder step2 conv1d_banks forward: float32 convolution as 3 float16 MFMA products '
                          'of exactly split operands)', 'bound': 'mfma',
                # ALGORITHMIC rate (the float32 convolution's 2 * MACs per launch / duration) against the dense peak of the
                # matrix type the kernel executes on (float16 products, float32 accumulation); the three products per
                # algorithmic FLOP are the price of float32 accuracy and count against the fraction
                'achieved': round(ach, 2), 'peak': MFMA_BF16_PEAK_TF, 'unit': 'TFLOP/s',
                'frac': round(ach / MFMA_BF16_PEAK_TF, 4), 'traffic': _pmc_traffic('gemm16_bank_step2_train'),
                'algorithmic_bytes_per_launch': M * H * 4 + 256 * 128 * 528 * 4 + M * 128 * K * 4,
                'algorithmic_flop_per_launch': fl_bank, 'avg_kernel_ms': round(ms_bank, 4),
                'executed': {'TFLOP/s': round(3 * ach, 1), 'peak': MFMA_BF16_PEAK_TF, 'frac': round(3 * ach / MFMA_BF16_PEAK_TF, 4),
                             'products_per_algorithmic_flop': 3},
                'vs_f32_mfma': {'peak': MFMA_F32_PEAK_TF, 'achieved_over_peak': round(ach / MFMA_F32_PEAK_TF, 4),
                                'f32_mfma_kernel_ms': round(ms_f32, 4), 'f32_mfma_kernel_TFLOPs': round(fl_bank / (ms_f32 * 1e-3) / 1e12, 1)},
                'operand_split_ms': round(ms_split, 4),
                'timing': 'HIP events on the launch stream, average of 20 back-to-back launches'}
        stages['step_TFLOPs_at_3x_forward'] = round(3 * DEC_FLOP_PER_FRAME * B * T / (dt / steps) / 1e12, 1)
        stages['step_frac_of_f32_mfma_peak'] = round(stages['step_TFLOPs_at_3x_forward'] / MFMA_F32_PEAK_TF, 4)
    del dec, enc, tr
    torch.cuda.empty_cache()
    return B * T, dt, stages, roof


def bench_train(args, rank, world):
    """BASELINE configs[4]: decoder training step (fwd + bwd + Adam, float32) on synthetic
    ARCTIC-slt-shaped targets, 32 windows per GPU, gradients all-reduced over RCCL."""
    frames, dt, stages, roof = train_measure(rank, world, args.steps, args.warmup)
    extra = {'stages': stages}
    if roof is not None:
        extra['roofline'] = roof
    cfg = {'workload': 'train: decoder fwd+bwd+Adam (float32) on 32 windows x 400 frames per GPU, encoder frozen '
                       '(BASELINE configs[4])', 'global_batch': 32 * world, 'frames_per_step_per_gpu': frames}
    return frames, dt, extra, cfg


def train_side_measurement(rank, world, reduce_device='cuda'):
    """BASELINE configs[4] inside the default line: 10 training steps after 3 warm-ups on every rank (the gradient
    exchange is a collective), max over ranks; rank 0 reports."""
    import dist_util
    steps, warmup = 10, 3
    frames, dt, stages, roof = train_measure(rank, world, steps, warmup)
    dt = dist_util.max_over_ranks(dt, device=reduce_device) if world > 1 else dt
    return {'workload': 'BASELINE configs[4] per-GPU shape: decoder fwd+bwd+Adam, float32, 32 windows x 400 frames per GPU, '
                        'encoder frozen; data parallel x%d' % world,
            'ms_per_step': round(dt / steps * 1e3, 4), 'steps': steps, 'warmup': warmup,
            'frames_per_s': round(frames * world * steps / dt, 1), 'step_TFLOPs': stages.get('step_TFLOPs_at_3x_forward'),
            'step_frac_of_f32_mfma_peak': stages.get('step_frac_of_f32_mfma_peak'),
            'roofline': None if roof is None else {k: roof[k] for k in ('kernel', 'achieved', 'peak', 'unit', 'frac', 'avg_kernel_ms',
                                                                         'executed', 'vs_f32_mfma')},
            'allreduce_ms': stages['allreduce_ms'], 'allreduce_busbw_GBps': stages['allreduce_busbw_GBps'],
            'allreduce_buckets': stages['allreduce_buckets'], 'last_loss': stages['last_loss']}


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


def _host_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))                         # the GPU box gives one GPU a 16-core share


def _median_time(fn, warmups=2, passes=5):
    """median of >= 5 passes after 2 warm-ups."""
    for _ in range(warmups):
        fn()
    ts = []
    for _ in range(passes):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], ts


def cpu_baseline_full():
    """Oracle timed on the host: front-end on 2 utterances (numpy) + encode/decode of 2 windows with
    torch-CPU float32 ops at the shipped sizes (all host threads); 2 warm-ups, median of 5 passes."""
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
    ncores = _host_threads()
    torch.set_num_threads(ncores)

    def one_pass():
        feats = [fo.calc_MFCC_input(wav[b], **FE_KW)[0][:400] for b in range(2)]
        x = torch.from_numpy(np.stack(feats))
        with torch.no_grad():
            _, pr, _, _ = mo.encoder_forward(x, we, enc_cfg)
            mo.decoder_forward(pr, wd, dec_cfg)

    dt, all_t = _median_time(one_pass)
    return {'value': round(800 / dt, 1), 'unit': 'frames/s', 'cores': ncores, 'kind': 'port',
            'sample': '2 utterances -> 2 windows (800 frames) through oracle/frontend_oracle.py (numpy) and '
                      'oracle/model_oracle.py (torch-CPU float32, %d threads): CPU restatement (TF/librosa-equivalent), '
                      'not TensorFlow; 2 warm-ups, median of 5 passes (min %.3f s, max %.3f s)' % (ncores, all_t[0], all_t[-1])}


def cpu_baseline_train():
    """Oracle training step timed on the host: forward (train mode) + autograd backward of the decoder at the shipped
    sizes on ONE window of 400 frames, torch-CPU float32, all host threads; 2 warm-ups, median of 5 passes."""
    from oracle import model_oracle as mo
    import contextlib
    import io
    from aux_func import load_cfg_d
    hp = os.path.join(ROOT, 'speech-cloner_amd', 'hp')
    with contextlib.redirect_stdout(io.StringIO()):
        dec_cfg = load_cfg_d(os.path.join(hp, 'decoder_cfg_d.json'))
    dec_cfg['dropout_rate'] = 0.0                      # (masks are an input of the oracle; their cost is nil)
    ncores = _host_threads()
    torch.set_num_threads(ncores)
    wd = mo.to_torch(mo.init_weights(dec_cfg, 'decoder', seed=2), torch.float32, requires_grad=True)
    g = torch.Generator().manual_seed(5)
    ppg = torch.softmax(torch.randn(1, 400, 61, generator=g), -1)
    tm, ts = torch.rand(1, 400, 80, generator=g) * 0.8, torch.rand(1, 400, 201, generator=g) * 0.8

    def one_pass():
        for v in wd.values():
            v.grad = None
        ym, ys = mo.decoder_forward(ppg, wd, dec_cfg, is_training=True, masks=None, stats_out={})
        mo.decoder_loss(ym, ys, tm, ts, dec_cfg)[2].backward()

    dt, all_t = _median_time(one_pass)
    return {'value': round(400 / dt, 1), 'unit': 'frames/s', 'cores': ncores, 'kind': 'port',
            'sample': '1 window (400 frames) of the 32: train-mode forward + autograd backward through oracle/model_oracle.py '
                      '(torch-CPU float32, %d threads; no Adam, no dropout masks); 2 warm-ups, median of 5 passes '
                      '(min %.2f s, max %.2f s)' % (ncores, all_t[0], all_t[-1])}


def cpu_baseline_frontend():
    """Oracle (numpy/scipy restatement of librosa's path) timed on the host: 4 utterances of the same workload per
    pass; 2 warm-ups, median of 5 passes, single thread."""
    from oracle import frontend_oracle as fo
    wav = synth_audio(4, 64000, seed=0).numpy()
    n = [0]

    def one_pass():
        n[0] = sum(fo.calc_MFCC_input(wav[b], **FE_KW)[0].shape[0] for b in range(wav.shape[0]))

    dt, all_t = _median_time(one_pass)
    return {'value': round(n[0] / dt, 1), 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
            'sample': '4 of the 32 utterances (4 s each) per pass through oracle/frontend_oracle.py (numpy/scipy '
                      'restatement of librosa 0.6; single thread); 2 warm-ups, median of 5 passes'}


def bench_stub(args, rank, world):
    """Launcher self-test (tests/test_bench_launcher_cpu.py): no GPU, no kernels -- a fixed amount of host arithmetic
    per step so that the launch / rendezvous / barrier / max-over-ranks / JSON plumbing of --gpus N can run on gloo."""
    if os.environ.get('BENCH_STUB_FAIL_RANK') == str(rank):       # test hook: a failing rank must fail the launcher
        raise SystemExit(7)
    x = torch.ones(256, 256)
    for _ in range(args.warmup):
        x = (x @ x) / 256.0
    dist_util_barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = (x @ x) / 256.0
    dist_util_barrier()
    dt = time.perf_counter() - t0
    assert float(x[0, 0]) == 1.0
    return 25600, dt, {}, {'workload': 'stub: launcher self-test on the CPU (no kernels, not a measurement)',
                           'frames_per_step_per_gpu': 25600}


def dist_util_barrier():
    import dist_util
    dist_util.barrier()


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a child process tree.  This
    process has made no GPU call (importing torch does not initialise HIP) and makes none: it only relays."""
    import subprocess
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('MASTER_ADDR', '127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + argv
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        out = out.rstrip('\n')
        if out.startswith('{') and '"metric"' in out:
            line = out                                  # rank 0's result line: relayed last, exactly once
        elif out:
            print(out, file=sys.stderr)
    rc = proc.wait()
    if rc != 0:
        print('bench.py: a rank failed (torch.distributed.run exit status %d)' % rc, file=sys.stderr)
        return rc if 0 < rc < 256 else 1
    if line is None:
        print('bench.py: the ranks exited without a result line', file=sys.stderr)
        return 1
    print(line)
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=6)
    ap.add_argument('--workload', default='full', choices=['full', 'frontend', 'train', 'vocoder', 'stub'])
    ap.add_argument('--dtype', default='bfloat16', choices=['bfloat16', 'float32'])
    ap.add_argument('--batch', type=int, default=32,
                    help='utterances per step of the full workload (BASELINE metric: 32; configs[3] reads as 64 -> 128 windows)')
    ap.add_argument('--window-batch', type=int, default=64)
    ap.add_argument('--streams', type=int, default=10,
                    help='HIP streams the independent window chunks / consecutive steps are pipelined over')
    ap.add_argument('--backend', default=None, choices=['nccl', 'gloo'],
                    help='process-group backend (default: nccl = RCCL; the stub workload uses gloo)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-f32', action='store_true', help='skip the float32 side measurement of the full workload')
    ap.add_argument('--no-side', action='store_true',
                    help='skip the front-end-alone and training-step side measurements of the full workload')
    args = ap.parse_args(argv)
    if args.gpus < 1:
        raise SystemExit('bench.py: --gpus must be >= 1')

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, argv))              # before ANY GPU call of this process

    import dist_util
    rank, local, world = dist_util.env_world()
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)' % (args.gpus, world))
    stub = args.workload == 'stub'
    backend_arg = args.backend
    if not stub:
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs a GPU')
        ndev = torch.cuda.device_count()
        if local >= ndev and backend_arg != 'gloo':
            raise SystemExit('bench.py: rank %d has no GPU of its own (%d visible); RCCL needs one device per rank' % (local, ndev))
        torch.cuda.set_device(local % ndev)             # (ranks share a card only in gloo rehearsals on a one-GPU box)
    backend = args.backend or ('gloo' if stub else 'nccl')
    dist_util.init(backend)
    joined = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    if joined != args.gpus:
        raise SystemExit('bench.py: %d ranks joined the process group, --gpus %d' % (joined, args.gpus))

    if args.workload == 'frontend':
        frames, dt, extra, cfg = bench_frontend(args, rank, world)
    elif args.workload == 'train':
        frames, dt, extra, cfg = bench_train(args, rank, world)
    elif args.workload == 'vocoder':
        frames, dt, extra, cfg = bench_vocoder(args, rank, world)
    elif stub:
        frames, dt, extra, cfg = bench_stub(args, rank, world)
    else:
        frames, dt, extra, cfg = bench_full(args, rank, world)

    dt = dist_util.max_over_ranks(dt, device='cpu' if backend == 'gloo' else 'cuda')
    if rank == 0:
        par = {'train': 'data parallel x%d: one batch per rank, one all-reduce of the flat gradient arena per step' % joined,
               }.get(args.workload, 'utterance-sharded x%d, no collective' % joined)
        line = {'metric': 'mel frames/sec', 'value': round(frames * joined * args.steps / dt, 1), 'unit': 'frames/s',
                'n_gpus': joined, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': round(dt / args.steps * 1e3, 5), 'higher_is_better': True, 'scaling': 'weak',
                'vs_baseline': None, 'dtype': 'f32' if args.workload in ('frontend', 'train', 'vocoder', 'stub') else
                ('bf16' if args.dtype == 'bfloat16' else 'f32'), 'data': 'synthetic',
                'config': dict(cfg, parallelism=par)}
        line.update(extra)
        if not stub:
            line['rccl'] = rccl_info(backend)
        if joined == 1 and not args.no_cpu_baseline and not stub:
            line['cpu_baseline'] = {'frontend': cpu_baseline_frontend, 'vocoder': cpu_baseline_vocoder,
                                    'train': cpu_baseline_train}.get(args.workload, cpu_baseline_full)()
        print(json.dumps(line), flush=True)
    dist_util.finalize()


if __name__ == '__main__':
    main()
