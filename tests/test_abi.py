"""The C-ABI library loads and exports every symbol include/vc_hip.h declares (no compute
calls, no GPU needed), and the ctypes table in _vc.py covers exactly those symbols."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'vc_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(vc_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_exported():
    import _vc
    if not os.path.exists(_vc.LIB_PATH):
        pytest.fail('libvc_hip.so not built (run __graft_entry__.build())')
    names = _declared_symbols()
    assert len(names) >= 10
    h = ctypes.CDLL(_vc.LIB_PATH)
    missing = [n for n in names if not hasattr(h, n)]
    assert not missing, missing
    assert set(_vc._SIGS) == set(names), set(_vc._SIGS) ^ set(names)


def test_ctypes_signatures_follow_the_header_prototypes():
    """Argument count and kind (pointer / int32 / size_t / float) of every entry of _vc._SIGS against the prototype in
    include/vc_hip.h: an argument added, dropped or moved on one side only fails here."""
    import _vc
    from conftest import ROOT
    src = open(os.path.join(ROOT, 'include', 'vc_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    protos = dict((m.group(2), (m.group(1), m.group(3))) for m in
                  re.finditer(r'\n([A-Za-z_][\w \*]*?)\b(vc_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;', src))
    assert set(protos) == set(_vc._SIGS)

    def kind_c(decl):
        d = ' '.join(decl.split())
        if '*' in d:
            return 'ptr'
        if d.startswith('size_t'):
            return 'size'
        if d.startswith('float'):
            return 'float'
        if d.startswith('int32_t') or d.startswith('int '):
            return 'int'
        if d.startswith('int64_t') or d.startswith('unsigned long long') or d.startswith('uint64_t'):
            return 'i64'
        raise AssertionError('unhandled C parameter: ' + d)

    def kind_py(t):
        if t is ctypes.c_void_p or t is ctypes.c_char_p or (isinstance(t, type) and issubclass(t, ctypes._Pointer)):
            return 'ptr'
        return {ctypes.c_float: 'float', ctypes.c_int32: 'int', ctypes.c_int64: 'i64', ctypes.c_uint64: 'i64', ctypes.c_size_t: 'size'}[t]

    for name, (ret, args) in protos.items():
        c_args = [] if args.strip() in ('', 'void') else [a for a in args.split(',')]
        py_ret, py_args = _vc._SIGS[name]
        assert len(c_args) == len(py_args), (name, len(c_args), len(py_args))
        for i, (ca, pa) in enumerate(zip(c_args, py_args)):
            assert kind_c(ca) == kind_py(pa), (name, i, ca.strip(), pa)


def test_version_and_errors_without_gpu():
    import _vc
    lib = _vc.lib()
    hdr = open(os.path.join(ROOT, 'include', 'vc_hip.h')).read()
    assert int(re.search(r'#define\s+VC_ABI_VERSION\s+(\d+)', hdr).group(1)) == _vc.VC_ABI_VERSION == lib.vc_version()
    assert lib.vc_target_arch() == b'gfx950'
    # argument validation happens before any HIP call
    cfg = _vc.FrontendCfg(16000, 80, 400, 200, 80, 40, 0.97, 0.003, 0.01, 0.01, 0.01, 1, 1, 1)
    rc = lib.vc_frontend_host_tables(ctypes.byref(cfg), None, None)
    assert rc == 1 and b'n_fft' in lib.vc_last_error()
    with pytest.raises(_vc.VCError):
        _vc.check(rc)


def test_struct_layout_matches_header():
    import _vc
    assert ctypes.sizeof(_vc.FrontendCfg) == 14 * 4


def test_every_struct_field_sits_where_the_header_puts_it(tmp_path):
    """The ctypes mirrors of the header's structs, checked against the C compiler: a small program that includes
    include/vc_hip.h prints sizeof and every field's offsetof; each must equal ctypes' (same field names, same order).
    A field added on one side only -- or in a different place -- fails here, not as a wrong launch on the GPU."""
    import os
    import subprocess
    import _vc
    from conftest import ROOT
    pairs = [('vc_frontend_cfg', _vc.FrontendCfg), ('vc_gemm_group', _vc.GemmGroup), ('vc_gemm_desc', _vc.GemmDesc),
             ('vc_wgrad_group', _vc.WgradGroup), ('vc_wgrad_desc', _vc.WgradDesc), ('vc_layout_item', _vc.LayoutItem),
             ('vc_cbhg_front_desc', _vc.CbhgFrontDesc), ('vc_w16_item', _vc.W16Item), ('vc_gemm16_pair', _vc.Gemm16Pair),
             ('vc_gemm16_desc', _vc.Gemm16Desc)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "vc_hip.h"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append('  printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    got = {}
    for ln in subprocess.check_output([str(exe)], text=True).splitlines():
        a, b, c = ln.split()
        got[(a, b)] = int(c)
    for cname, cls in pairs:
        assert got[(cname, 'sizeof')] == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)


def test_fused_launch_shape_queries_and_validation_without_gpu():
    """The shape predicates of the fused launches are pure host code, and argument validation of the fused entry
    points fails (loudly, with a message) before any HIP call."""
    import _vc
    lib = _vc.lib()
    # the shipped encoder shape (hp/encoder_cfg_d.json) takes the one-launch front; nothing else does
    assert lib.vc_cbhg_front_supported(80, 80, 40, 6, 128, 1, 40, 400) == 1
    for bad in ((80, 80, 40, 16, 128, 1, 40, 400), (61, 256, 128, 32, 128, 4, 128, 400), (80, 80, 40, 6, 128, 5, 40, 400),
                (80, 80, 40, 6, 128, 1, 40, 4)):
        assert lib.vc_cbhg_front_supported(*bad) == 0, bad
    assert lib.vc_cbhg_front_coef_floats() == 3232                  # layout documented in include/vc_hip.h
    assert lib.vc_prenet_chain_supported(64, 256, 128) == 1 and lib.vc_prenet_chain_supported(80, 512, 256) == 1
    assert lib.vc_prenet_chain_supported(80, 80, 40) == 0 and lib.vc_prenet_chain_supported(64, 512, 256) == 0
    d = _vc.CbhgFrontDesc()
    d.n_features, d.prenet_units, d.width, d.n_banks, d.bank_filters, d.n_highway, d.gru_units, d.T = 80, 80, 40, 6, 128, 1, 40, 400
    rc = lib.vc_cbhg_front(ctypes.byref(d), None)                    # every pointer NULL
    assert rc != 0 and b'vc_cbhg_front' in lib.vc_last_error()
    d.n_banks = 16
    rc = lib.vc_cbhg_front(ctypes.byref(d), None)
    assert rc != 0 and b'unsupported shape' in lib.vc_last_error()
    rc = lib.vc_prenet_chain(None, 0, 128, 80, 80, 512, 256, None, None, None, None, None, 256, None)
    assert rc != 0 and b'vc_prenet_chain' in lib.vc_last_error()
    rc = lib.vc_mfma_pack(None, 32, 16, 16, 0, None, None)
    assert rc != 0 and b'vc_mfma_pack' in lib.vc_last_error()
    assert ctypes.sizeof(_vc.CbhgFrontDesc) % 8 == 0


def test_kernel_options_are_explicit_and_ablations_are_not_shipped():
    """Kernel selection goes through vc_set_option only: the library does not import getenv, unknown names are
    errors, and the result-corrupting ablation switches are rejected by the shipped (non -DVC_ABLATE) build."""
    import subprocess
    import _vc
    lib = _vc.lib()
    assert lib.vc_ablate_build() == 0
    assert _vc.get_option('gru_mfma') == -1
    _vc.set_option('gru_mfma', 1)
    assert _vc.get_option('gru_mfma') == 1
    with _vc.options(gru_mfma=0, bank256=0):
        assert _vc.get_option('gru_mfma') == 0 and _vc.get_option('bank256') == 0
    assert _vc.get_option('gru_mfma') == 1 and _vc.get_option('bank256') == -1
    _vc.set_option('gru_mfma', -1)
    with pytest.raises(_vc.VCError, match='unknown option'):
        _vc.set_option('no_such_switch', 1)
    for name in ('ablate_bank256', 'ablate_bank256_only', 'ablate_cbhg_front'):
        with pytest.raises(_vc.VCError, match='VC_ABLATE'):
            _vc.set_option(name, 1)
    und = subprocess.run(['nm', '-D', '--undefined-only', _vc.LIB_PATH], capture_output=True, text=True).stdout
    assert 'getenv' not in und
    import modules
    src = open(os.path.join(ROOT, 'speech-cloner_amd', 'modules.py')).read() + open(os.path.join(ROOT, 'speech-cloner_amd', '_vc.py')).read()
    assert 'os.environ[' not in src and 'environ.setdefault' not in src      # the package never writes the environment
    assert set(modules.OPTIONS) == {'prenet_chain', 'highway_chain', 'cbhg_front'}


def test_a_library_of_another_abi_version_is_refused(tmp_path):
    """include/vc_hip.h VC_ABI_VERSION: exported signatures changed between versions (arguments inserted in the middle),
    so a stale build -- e.g. an old A/B library named by VC_LIB_PATH -- must be refused at load time, not called with
    shifted pointers.  A stub library reporting version 1 stands in for the stale build (a separate interpreter: the
    binding caches its handle)."""
    import subprocess
    import sys
    src = tmp_path / 'stale.c'
    src.write_text('int vc_version(void) { return 1; }\n')
    so = tmp_path / 'libvc_stale.so'
    subprocess.check_call(['gcc', '-shared', '-fPIC', str(src), '-o', str(so)])
    code = ('import sys; sys.path.insert(0, %r); import _vc\n'
            'try:\n    _vc.lib()\nexcept _vc.VCError as e:\n    print("REFUSED", e)\n' % os.path.join(ROOT, 'speech-cloner_amd'))
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, VC_LIB_PATH=str(so)), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and 'REFUSED' in r.stdout and 'ABI version 1' in r.stdout, r.stdout + r.stderr


def test_workspace_queries_are_host_only():
    """vc_conv_gemm_workspace_bytes / vc_gru_workspace_bytes make no GPU call: the split-K scratch of the 256-channel long-K
    projection (include/vc_hip.h, vc_gemm_desc.d_workspace) is asked for exactly where the launch would split -- <= 128
    row tiles of 256 frames, K = taps x Cin >= 4096, bf16 -- and not at all with proj256_split = 0 or for other launches."""
    import _vc
    lib = _vc.lib()

    def desc(M, cin, taps, N=256, dtype=_vc.VC_BF16):
        d = _vc.GemmDesc()
        d.dtype, d.mode, d.d_X = dtype, _vc.GEMM_PLAIN, 4096            # (any non-NULL, 16-byte aligned address: never dereferenced)
        d.M, d.T, d.Cin, d.ldx, d.N, d.n_groups = M, 400, cin, cin, N, 1
        g = d.groups[0]
        g.d_Bt, g.K, g.taps, g.pad_l, g.c_off = 4096, taps * cin, taps, (taps - 1) // 2, 0
        d.act, d.d_C, d.ldc = _vc.ACT_RELU, 4096, N
        return d

    q = lambda d: lib.vc_conv_gemm_workspace_bytes(ctypes.byref(d))
    ntm = 100                                                            # 64 windows x 400 frames / 256
    assert q(desc(25600, 4096, 3)) == ((ntm * 8 + 255) // 256) * 256 + ntm * 262144
    assert q(desc(51200, 4096, 3)) == 0                                  # 200 row tiles fill the chip by themselves
    assert q(desc(25600, 4096, 3, N=128)) == 0 and q(desc(25600, 256, 3)) == 0 and q(desc(25600, 4096, 3, dtype=_vc.VC_F32)) == 0
    with _vc.options(proj256_split=0):
        assert q(desc(25600, 4096, 3)) == 0
    with _vc.options(proj256=0):
        assert q(desc(25600, 4096, 3)) == 0
    assert lib.vc_gru_workspace_bytes(256, _vc.VC_BF16) == 2 * 3 * 256 * 256 * 2
    assert lib.vc_gru_workspace_bytes(40, _vc.VC_BF16) >= 2 * 3 * 40 * 40 * 2
    assert _vc.THROUGHPUT_OPTIONS == {'proj256_split': 0, 'fe_fused': 0}
    with _vc.throughput_mode():
        assert _vc.get_option('fe_fused') == 0 and _vc.get_option('proj256_split') == 0
    assert _vc.get_option('fe_fused') == -1


def test_split_float16_entry_points_validate_on_the_host():
    """vc_gemm16 / vc_split16 / vc_transpose_split16 refuse bad shapes before anything is launched (no GPU needed), and the
    workspace query is pure host arithmetic."""
    import _vc
    h = _vc.lib()
    assert h.vc_gemm16_workspace_bytes(12800, 4096, 1) == 256 + 50 * 8 * 262144     # 50 row tiles, up to 8 K ranges each
    assert h.vc_gemm16_workspace_bytes(12800, 256, 16) == 0                          # many pairs: never split
    assert h.vc_gemm16_workspace_bytes(12800, 100, 1) == 0                           # not a multiple of 64
    assert h.vc_split16(None, 400, 128, 128, 400, None, None, 0, 0, None, None, None) == 1      # VC_ERR_INVALID
    d = _vc.Gemm16Desc()
    assert h.vc_gemm16(ctypes.byref(d), None) == 1      # VC_ERR_INVALID
    assert b'vc_gemm16' in h.vc_last_error()


def test_the_tile_kernels_keep_their_accumulators_in_registers():
    """bank256_kernel and gemm16_kernel hold 128 accumulator registers per lane at 2 waves per SIMD: a source change that
    tips the allocation over 256 registers turns into scratch spills and a several-fold slowdown with correct results
    (it happened once: a per-wave epilogue predicate).  The compiler's own resource report must say 0 spills."""
    import os
    import re
    import subprocess
    from conftest import ROOT
    csrc = os.path.join(ROOT, 'speech-cloner_amd', 'csrc')
    for src, kern in (('vc_gemm16.hip', 'gemm16_kernel'), ('vc_bank256.hip', 'bank256_kernel')):
        out = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-fno-slp-vectorize', '-std=c++17', '--offload-arch=gfx950',
                              '--cuda-device-only', '-I', os.path.join(ROOT, 'include'), '-I', csrc, '-c', os.path.join(csrc, src),
                              '-o', os.devnull, '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr[-2000:]
        blocks = out.stderr.split('Function Name:')
        mine = [b for b in blocks if kern in b.splitlines()[0]]
        assert mine, 'no resource report for %s' % kern
        spill = re.search(r'VGPRs Spill: (\d+)', mine[0])
        scratch = re.search(r'ScratchSize \[bytes/lane\]: (\d+)', mine[0])
        assert spill and int(spill.group(1)) == 0, (kern, mine[0])
        assert scratch and int(scratch.group(1)) == 0, (kern, mine[0])
