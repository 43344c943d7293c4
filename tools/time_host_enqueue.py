"""Host time to ENQUEUE one bench step (Python + ctypes + allocator) vs the device time per step:
if the two are close the pipelined step is host-bound and kernel work hides behind launch overhead."""
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
def step():
    global fe_out
    fe_out = audio_lib.calc_MFCC_input_batch(wav, None, out=fe_out, **bench.FE_KW)
    x = fe_out[0][:, :2 * T, :].reshape(B * 2, T, 80)
    ready = torch.cuda.Event(); ready.record(torch.cuda.current_stream())
    st_ = streams[cnt[0] % NS]; cnt[0] += 1
    st_.wait_event(ready)
    with torch.cuda.stream(st_):
        xi = x[:W].contiguous(); xi.record_stream(st_)
        return dec.forward(xi)
for _ in range(5): step()
torch.cuda.synchronize()
# (a) device-bound rate
t0 = time.perf_counter()
for _ in range(30): step()
t_enq = (time.perf_counter() - t0) / 30 * 1e3
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 30 * 1e3
# (b) pure host cost: enqueue while the device is idle-ish (small batches would still run; measure call cost alone)
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18)
print('enqueue %.3f ms/step, enqueue+drain %.3f ms/step' % (t_enq, t_all))
print(s.getvalue()[:3500])
