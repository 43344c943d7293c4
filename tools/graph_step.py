"""One encode+decode of 64 windows captured into a HIP graph (torch.cuda.CUDAGraph over the library's launches) against
the same step enqueued launch by launch: single-stream latency with and without the inter-launch gaps."""
import os, sys, contextlib, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch, bench
with contextlib.redirect_stdout(io.StringIO()):
    enc, dec = bench.load_models('bfloat16', 0)
x = (torch.rand(64, 400, 80, device='cuda') * 0.4 - 0.2)
for _ in range(3):
    ref = dec.forward(x)
torch.cuda.synchronize()
ms_eager = bench.time_events(lambda: dec.forward(x), 10)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        dec.forward(x)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    out = dec.forward(x)
torch.cuda.synchronize()
g.replay()
torch.cuda.synchronize()
same = all(torch.equal(out[k], ref[k]) for k in ('y_mel', 'y_stft', 'y_phn'))
ms_graph = bench.time_events(lambda: g.replay(), 10)
print('eager %.4f ms   graph replay %.4f ms   (%+.1f %%)   outputs equal: %s' % (ms_eager, ms_graph, 100 * (ms_graph / ms_eager - 1), same))
