"""Host side of one training step: how long Python takes to ENQUEUE the step (no synchronisation until the end) against
the step's wall time, and where that host time goes (cProfile, top 25 by cumulative time).
python tools/train_cpu_profile.py"""
import cProfile, io, os, pstats, sys, time, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch
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
B, T = 32, 400
g = torch.Generator().manual_seed(100)
mfcc = (torch.rand(B, T, 80, generator=g) * 0.4 - 0.2).cuda()
mel = (torch.rand(B, T, 80, generator=g) * 0.8).cuda()
stft = (torch.rand(B, T, 201, generator=g) * 0.8).cuda()
for _ in range(3):
    dec.exec_train_step(mfcc, mel, stft)
tr = dec._trainer
torch.cuda.synchronize()
enq, wall = [], []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.forward_backward(mfcc, mel, stft)
    tr.apply_gradients(1)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3)
    wall.append((t2 - t0) * 1e3)
print('host enqueue time per step: %s ms; wall until the device is idle: %s ms' % (
    ' '.join('%.1f' % v for v in enq), ' '.join('%.1f' % v for v in wall)))
pr = cProfile.Profile()
pr.enable()
tr.forward_backward(mfcc, mel, stft)
tr.apply_gradients(1)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28)
print(s.getvalue()[:6000])
# device allocations per step: a steady-state step should be served from the caching allocator (hipMalloc / hipFree block
# the host and synchronise the device)
for i in range(3):
    a = torch.cuda.memory_stats()
    tr.forward_backward(mfcc, mel, stft)
    tr.apply_gradients(1)
    torch.cuda.synchronize()
    b = torch.cuda.memory_stats()
    print('step %d: hipMalloc calls %d, hipFree calls %d, allocator retries %d, reserved %.2f GB, peak allocated %.2f GB' % (
        i, b['num_device_alloc'] - a['num_device_alloc'], b['num_device_free'] - a['num_device_free'],
        b['num_alloc_retries'] - a['num_alloc_retries'], b['reserved_bytes.all.current'] / 1e9, b['allocated_bytes.all.peak'] / 1e9))
