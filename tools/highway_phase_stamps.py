"""Where highway_chain_kernel spends its cycles (-DVC_ABLATE build: s_memtime sums per phase, thread 0 of one workgroup).
  bash tools/build_ablate.sh && VC_LIB_PATH=build/libvc_hip_ablate.so python tools/highway_phase_stamps.py [H] [layers]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch, modules, _vc, bench
assert _vc.lib().vc_ablate_build()
H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else (6 if H == 256 else 4)
N, T = 64, 400
st = modules.VariableStore('bfloat16')
x = (torch.randn(N, T, H, device='cuda') * 0.5).to(st.dtype)
with modules.variable_store(st), modules.variable_scope('h'):
    def run():
        # the launch alone: the recurrence that follows in highway_chain(gru_scope=...) is not wanted here
        return modules.highway_chain(x, H, L)
    run()
    ms_plain = bench.time_events(run, 20)
    import modules as m
    orig = m._gru_recurrence
    m._gru_recurrence = lambda xproj, *a: xproj
    def run2():
        return modules.highway_chain(x, H, L, gru_scope='gru')
    run2()
    ms_tail = bench.time_events(run2, 20)
    m._gru_recurrence = orig
torch.cuda.synchronize()
buf = (C.c_ulonglong * 8)()
fn = _vc.lib().vc_ablate_read_highway_stamps
fn.restype = C.c_int
assert fn(buf) == 0
names = ['activation tile + biases -> LDS, first weights, barrier', 'matrix phase (all layers)', 'gate phase (all layers)',
         'barrier (all layers)', 'GRU input projection tail']
tot = sum(buf[i] for i in range(5))
print('H = %d, %d layers, %d x %d frames: %.1f us without the projection tail, %.1f us with it (events); stamped workgroup %d cycles' % (
    H, L, N, T, ms_plain * 1e3, ms_tail * 1e3, tot))
for i, nm in enumerate(names):
    print('   %-58s %8d cycles  (%4.1f %%)%s' % (nm, buf[i], 100.0 * buf[i] / max(tot, 1), '   = %d per layer' % (buf[i] // L) if 1 <= i <= 3 else ''))
