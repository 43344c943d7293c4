"""What does the encoder cost INSIDE the pipelined step (three streams)?  bench.py's full step with the
encoder's forward replaced by its cached result (A/B on the same box).  The kernel sum says 0.21 ms of
launches + 0.30 ms of recurrence; this prints what the step time actually loses."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch, bench, audio_lib
B, L, T, W, NS = 32, 64000, 400, 64, 3
wav = bench.synth_audio(B, L, seed=0).cuda()
enc, dec = bench.load_models('bfloat16', 0)
streams = [torch.cuda.Stream() for _ in range(NS)]
fe_out = None
cnt = [0]
def step(front_end=True):
    global fe_out
    if front_end or fe_out is None:
        fe_out = audio_lib.calc_MFCC_input_batch(wav, None, out=fe_out, **bench.FE_KW)
    x = fe_out[0][:, :2 * T, :].reshape(B * 2, T, 80)
    ready = torch.cuda.Event(); ready.record(torch.cuda.current_stream())
    st_ = streams[cnt[0] % NS]; cnt[0] += 1
    st_.wait_event(ready)
    with torch.cuda.stream(st_):
        xi = x[:W].contiguous(); xi.record_stream(st_)
        return dec.forward(xi)
def run(n=30, **kw):
    for _ in range(3): step(**kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step(**kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
a = run()
real = enc.forward
cached = real(fe_out[0][:, :2 * T, :].reshape(B * 2, T, 80)[:W].contiguous(), ppg_pad_to=64, ppg_dtype=torch.bfloat16)
enc.forward = lambda x, **kw: cached
b = run()
c = run(front_end=False)
enc.forward = real
d = run()
print('full step %.3f ms | encoder cached %.3f ms | + front-end skipped %.3f ms | full again %.3f ms' % (a, b, c, d))
