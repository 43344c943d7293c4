"""A/B of run-time switches INSIDE the three-stream step, on one box, interleaved (ABAB...) so that box and
clock drift cancel: every configuration runs `rounds` times `steps` steps; mean and spread of ms/step.
  python tools/ab_step.py "prenet_chain=0" "bank256_xcd=1" ...      (the default is always included)
Switch names: modules.OPTIONS keys (prenet_chain, highway_chain, cbhg_front) and vc_set_option names (include/vc_hip.h)."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch, bench, audio_lib, modules, _vc
DEFAULTS = dict(modules.OPTIONS)
cfgs = [''] + sys.argv[1:]
keys = sorted({kv.split('=')[0] for c in cfgs for kv in c.split(',') if kv})
NS, steps, rounds = 10, 60, 6
wav = bench.synth_audio(32, 64000, seed=0).cuda()
enc, dec = bench.load_models('bfloat16', 0)
streams = [torch.cuda.Stream() for _ in range(16)]
ns = [NS]
chunk = [64]
fe_out, cnt = None, [0]
def step():
    global fe_out
    fe_out = audio_lib.calc_MFCC_input_batch(wav, None, out=fe_out, **bench.FE_KW)
    x = fe_out[0][:, :800, :].reshape(64, 400, 80)
    main = torch.cuda.current_stream()
    ready = torch.cuda.Event(); ready.record(main)
    for i in range(0, 64, chunk[0]):
        st_ = streams[cnt[0] % ns[0]]; cnt[0] += 1
        st_.wait_event(ready)
        with torch.cuda.stream(st_):
            xi = x[i:i + chunk[0]].contiguous(); xi.record_stream(st_)
            ev = torch.cuda.Event(); ev.record(st_); main.wait_event(ev)
            dec.forward(xi)
def apply(c):
    modules.OPTIONS.update(DEFAULTS)
    for k in keys:
        if k not in DEFAULTS and k not in ('STREAMS', 'CHUNK'): _vc.set_option(k, -1)
    ns[0] = NS; chunk[0] = 64
    for kv in c.split(','):
        if kv:
            k, v = kv.split('=')
            if k == 'STREAMS': ns[0] = int(v)              # pseudo-switch: number of streams the steps rotate over
            elif k == 'CHUNK': chunk[0] = int(v)           # pseudo-switch: windows per launch (64 = one chunk per step)
            elif k in DEFAULTS: modules.OPTIONS[k] = v != '0'
            else: _vc.set_option(k, int(v))
res = {c: [] for c in cfgs}
for c in cfgs:                                   # build every cache first
    apply(c)
    for _ in range(4): step()
torch.cuda.synchronize()
for r in range(rounds):
    for c in cfgs:
        apply(c)
        for _ in range(6): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): step()
        torch.cuda.synchronize()
        res[c].append((time.perf_counter() - t0) / steps * 1e3)
base = statistics.mean(res[''])
for c in cfgs:
    v = res[c]
    print('%-28s %.4f ms/step  (min %.4f max %.4f)  %+5.2f %% vs default' % (c or 'default', statistics.mean(v), min(v), max(v), 100 * (statistics.mean(v) / base - 1)))
