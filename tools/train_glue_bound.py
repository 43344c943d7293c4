"""What parts of the training step cost in WALL time, by leaving them out (WRONG numerics, timing only): every weight-gradient
launch (side stream) -> what all of them together cost; the kernel-layout caches kept stale -> upper bound of what a
one-launch refresh of the per-step weight re-layouts could gain.  Round 3, MI355X: 39.1 / 31.2 / 38.2 ms per step.
(Also measured and not kept: the transposed copies of the forward activations built under the forward pass instead of
in front of each weight-gradient launch: 39.25 vs 39.17 ms.)
python tools/train_glue_bound.py"""
import os, sys, time, contextlib, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch
with contextlib.redirect_stdout(io.StringIO()):
    from aux_func import load_cfg_d
    from encoder import encoder_spec_phn
    from decoder import decoder_specs
    hp = os.path.join(ROOT, 'speech-cloner_amd', 'hp')
    ec = load_cfg_d(os.path.join(hp, 'encoder_cfg_d.json')); dc = load_cfg_d(os.path.join(hp, 'decoder_cfg_d.json'))
    ec.update(is_training=False, model_path=os.path.join(ROOT, 'tests', 'golden', 'enc_14_ckpt')); dc.update(is_training=True)
    enc = encoder_spec_phn(ec, None); dec = decoder_specs(dc, None, enc)
g = torch.Generator().manual_seed(100)
mfcc = (torch.rand(32, 400, 80, generator=g) * 0.4 - 0.2).cuda()
mel = (torch.rand(32, 400, 80, generator=g) * 0.8).cuda()
stft = (torch.rand(32, 400, 201, generator=g) * 0.8).cuda()
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        dec.exec_train_step(mfcc, mel, stft)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for _ in range(3):
    dec.exec_train_step(mfcc, mel, stft)
a = [run(5) for _ in range(3)]
inv = dec.store.invalidate
dec.store.invalidate = lambda: None
b = [run(5) for _ in range(3)]
dec.store.invalidate = inv
c = [run(5) for _ in range(3)]
import training
wg = training._Ops.wgrad
training._Ops.wgrad = staticmethod(lambda *a_, **k_: None)
d = [run(5) for _ in range(3)]
training._Ops.wgrad = wg
print('shipped           ms/step', ['%.2f' % v for v in a])
print('no weight-gradient launches (wrong numerics: what ALL of them cost the step)', ['%.2f' % v for v in d])
print('caches kept stale ms/step', ['%.2f' % v for v in b], '(wrong numerics: upper bound of the gain)')
print('shipped again     ms/step', ['%.2f' % v for v in c])
