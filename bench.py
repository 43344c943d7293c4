#!/usr/bin/env python
"""Benchmark of the hot path on MI355X.  Contract: prints ONE JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload frontend|full]

A "step" is one pass of the hot path over one synthetic batch that is already resident in
HBM.  Multi-GPU: one process per GPU (torch.distributed.run), the utterance batch is sharded
one full batch per rank with NO data-path collective (inference shards by utterance: SURVEY.md
section 8e) => weak scaling; the timed region is bracketed by barrier + synchronize and the MAX
over ranks is reported.

Workloads
  frontend  BASELINE.json configs[1]: STFT+mel front-end, batch 32 x 4 s @ 16 kHz, fp32
            (25,632 frames per step and rank).
Extra objects on the line: "roofline" (dominant kernel, HIP-event timed on the launch stream),
"cpu_baseline" (oracle timed on the host, rank 0, N=1 only), "stages".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

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
        for name, mask in (('abssum', 1), ('power400', 2), ('finalize', 4), ('all', 7)):
            k[name] = time_events(lambda m=mask: audio_lib.calc_MFCC_input_batch(
                wav, None, out=out, stage_mask=m, **FE_KW), iters)
        alg = FE_BYTES_PER_FRAME * frames
        ach = alg / (k['power400'] * 1e-3) / 1e9
        extra['roofline'] = {'kernel': 'fe_power400_kernel', 'bound': 'hbm', 'achieved': round(ach, 1),
                             'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4),
                             'traffic': None, 'algorithmic_bytes_per_launch': alg,
                             'avg_kernel_ms': round(k['power400'], 5)}
        extra['stages'] = {'kernel_ms': {n: round(v, 5) for n, v in k.items()},
                           'frontend_pipeline_GBps': round(alg / (k['all'] * 1e-3) / 1e9, 1)}
    return frames, dt, extra, {'workload': 'frontend: STFT+mel+MFCC, batch 32 x 4 s @ 16 kHz (BASELINE configs[1])',
                               'batch': B, 'samples': L, 'frames_per_step_per_gpu': frames}


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
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='frontend', choices=['frontend'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU')
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.distributed.init_process_group('nccl', device_id=torch.device('cuda', local))

    frames, dt, extra, cfg = bench_frontend(args, rank, world)

    t = torch.tensor([dt], dtype=torch.float64, device='cuda')
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())
    if rank == 0:
        line = {'metric': 'mel frames/sec', 'value': round(frames * world * args.steps / dt, 1), 'unit': 'frames/s',
                'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': round(dt / args.steps * 1e3, 5), 'higher_is_better': True, 'scaling': 'weak',
                'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                'config': dict(cfg, parallelism='utterance-sharded x%d, no collective' % world)}
        line.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline_frontend()
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
