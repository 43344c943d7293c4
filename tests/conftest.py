"""pytest configuration: registers the `gpu` marker and puts the package (a flat directory of
modules, like the reference: `from audio_lib import ...`) and the repo root on sys.path.

  python -m pytest tests -q -m "not gpu"   CPU suite (oracle vs golden vectors, host logic, ABI)
  python -m pytest tests -q -m gpu         parity tests proper: HIP path vs oracle, on an MI355X
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'speech-cloner_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
REFERENCE = '/root/reference'


def poison_gpu_state():
    """Leaves NaN where stale bytes could be picked up by a kernel that reads something it never
    wrote: in LDS (a reduction over NaN on every CU) and in the free blocks of torch's caching
    allocator (so the next torch.empty() hands out NaN-filled memory)."""
    import torch
    import training
    nan = torch.full((4096, 2048), float('nan'), device='cuda')
    out = torch.empty(2048, device='cuda')
    training._Ops.col_sum(nan, 4096, 2048, 2048, out)
    junk = [torch.full((n,), float('nan'), device='cuda') for n in (1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18)]
    torch.cuda.synchronize()
    del junk, nan, out


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def reference_dir():
    if not os.path.isdir(REFERENCE):
        pytest.skip('/root/reference not present on this machine')
    return REFERENCE


FE_KW = dict(sr=16000, pre_emphasis=0.97, hop_length=80, win_length=400, n_mels=80, n_mfcc=40,
             n_fft=None, window='hann', mfcc_normaleze_first_mfcc=True, mfcc_norm_factor=0.01,
             calc_mfcc_derivate=True, M_dB_norm_factor=0.01, P_dB_norm_factor=0.01,
             mean_abs_amp_norm=0.003, clip_output=True)

FE_KW_GENERIC = dict(FE_KW, hop_length=40, win_length=400, n_fft=512, n_mels=64, n_mfcc=20,
                     window='hamming', calc_mfcc_derivate=False, mfcc_norm_factor=1.0,
                     M_dB_norm_factor=1.0, P_dB_norm_factor=1.0, clip_output=False,
                     mfcc_normaleze_first_mfcc=False, pre_emphasis=0.0, mean_abs_amp_norm=1.0)
