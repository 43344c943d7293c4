"""How well do the three in-flight steps overlap?  From a rocprofv3 --kernel-trace CSV of bench.py:
wall time split by which classes of kernels are running (MFMA-bound tiles / recurrences / other).
  rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/overlap_analysis.py --run [n_streams]
  python3 tools/overlap_analysis.py OUT/*/*kernel_trace.csv"""
import csv, sys, collections, os
if sys.argv[1] == '--run':          # the bench's pipelined loop and nothing else (clean trace)
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
        sys.path.insert(0, p)
    import torch, bench, audio_lib
    NS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    wav = bench.synth_audio(32, 64000, seed=0).cuda()
    enc, dec = bench.load_models('bfloat16', 0)
    streams = [torch.cuda.Stream() for _ in range(NS)]
    fe_out = None
    for i in range(24):
        fe_out = audio_lib.calc_MFCC_input_batch(wav, None, out=fe_out, **bench.FE_KW)
        x = fe_out[0][:, :800, :].reshape(64, 400, 80)
        ready = torch.cuda.Event(); ready.record(torch.cuda.current_stream())
        st_ = streams[i % NS]; st_.wait_event(ready)
        with torch.cuda.stream(st_):
            xi = x.contiguous(); xi.record_stream(st_)
            dec.forward(xi)
    torch.cuda.synchronize()
    sys.exit(0)
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
def cls(n):
    if 'bank256' in n or 'conv256' in n or 'highway_chain' in n or 'conv_kernel' in n or 'gemm_kernel' in n or 'cbhg_small' in n: return 'mfma'
    if 'gru_' in n and 'pack' not in n: return 'gru'
    return 'other'
fe = sorted(int(r['Start_Timestamp']) for r in rows if 'fe_power400' in r['Kernel_Name'])
lo, hi = fe[8], fe[20]                                             # steady state: steps 8..20 of 24
print('ms per step in the window: %.3f' % ((hi - lo) / 12e6))
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if e < lo or s > hi: continue
    c = cls(r['Kernel_Name'])
    ev.append((max(s, lo), 1, c)); ev.append((min(e, hi), -1, c))
ev.sort()
act = collections.Counter(); last = lo; acc = collections.Counter()
for t, d, c in ev:
    key = ('mfma' if act['mfma'] else '') + ('+gru' if act['gru'] else '') + ('+other' if act['other'] else '') or 'idle'
    acc[key] += t - last
    acc['n_mfma=%d' % min(act['mfma'], 3)] += t - last
    last = t; act[c] += d
tot = hi - lo
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print('%-18s %6.1f %%' % (k, 100.0 * v / tot))
busy = collections.Counter()
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s >= lo and e <= hi: busy[cls(r['Kernel_Name'])] += e - s
print('kernel-time sums over the window / wall:', {k: round(v / tot, 2) for k, v in busy.items()})

# idle gaps: what ran before / after
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows if lo <= int(r['Start_Timestamp']) < hi)
end, prev, gaps = iv[0][1], iv[0][2], collections.Counter()
for s_, e_, n_ in iv[1:]:
    if s_ > end: gaps[(prev[:34], n_[:34])] += s_ - end
    if e_ > end: end, prev = e_, n_
print('idle by (kernel before, kernel after), us per step:')
for k, v in gaps.most_common(12): print('  %7.1f  %s -> %s' % (v / 12e3, k[0], k[1]))
