"""Training convolutions at the benchmark's per-GPU shape (32 windows x 400 frames): the float32-MFMA kernels against the
split-float16 path (vc_split16 + vc_gemm16), HIP events, 20 back-to-back launches each.  python tools/ab_gemm16.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch, gemm16, modules, training, _vc

dev = torch.device('cuda')
N, T = 32, 400
M = N * T


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def report(name, flop, t32, t16, tsplit):
    print('%-44s f32 MFMA %.3f ms (%5.1f TFLOP/s)   f16x3 %.3f ms (+ split %.3f) = %5.1f TFLOP/s float32-equivalent   x%.2f' % (
        name, t32, flop / t32 / 1e9, t16, tsplit, flop / (t16 + tsplit) / 1e9, t32 / (t16 + tsplit)))


def main():
    for H in (256, 128):
        K = 32
        CB = 128 * K
        g = torch.Generator().manual_seed(1)
        Ws = [(torch.randn(k, H, 128, generator=g) * 0.05).to(dev) for k in range(1, K + 1)]
        W1 = (torch.randn(3, CB, H, generator=g) * 0.02).to(dev)
        x = torch.randn(M, H, generator=g).to(dev)
        zb = torch.randn(M, CB, generator=g).to(dev)
        dq = torch.randn(M, H, generator=g).to(dev)
        w16 = gemm16.Weights16(dev)
        fp, fcs = gemm16.bank_forward_operands(w16, Ws, H)
        dp, dcs = gemm16.conv_dgrad_operands(w16, W1)
        if H == 256:
            pp, pcs = gemm16.conv_forward_operands(w16, W1)
            bp, bcs = gemm16.bank_dgrad_operands(w16, Ws, H)
        w16.refresh()
        print('H = %d: vc_weights16 over %d items: %.3f ms' % (H, len(w16.items), timed(w16.refresh)))
        # bank forward
        groups = [(W.permute(2, 0, 1).reshape(128, -1).contiguous(), k * H, k, (k - 1) // 2, 128 * (k - 1)) for k, W in enumerate(Ws, 1)]
        out = torch.empty((M, CB), device=dev)
        flop = 2.0 * M * 528 * H * 128
        t32 = timed(lambda: modules.gemm_launch(x, M, T, H, H, 128, groups, out, CB, _vc.VC_F32, out_f32=True))
        ts = timed(lambda: gemm16.split16(x, M, H, H, T))
        x16, rs = gemm16.split16(x, M, H, H, T)
        t16 = timed(lambda: gemm16.gemm16(x16, rs, M, T, H, fp, out, CB, col_scale=fcs))
        report('H=%d bank forward' % H, flop, t32, t16, ts)
        # projection data gradient (dQ1 [M, H] -> dP [M, CB])
        bt = W1.flip(0).permute(1, 0, 2).reshape(CB, 3 * H).contiguous()
        flop = 2.0 * M * 3 * H * CB
        t32 = timed(lambda: modules.gemm_launch(dq, M, T, H, H, CB, [(bt, 3 * H, 3, 1, 0)], out, CB, _vc.VC_F32, out_f32=True))
        ts = timed(lambda: gemm16.split16(dq, M, H, H, T))
        d16, drs = gemm16.split16(dq, M, H, H, T)
        t16 = timed(lambda: gemm16.gemm16(d16, drs, M, T, H, dp, out, CB, col_scale=dcs))
        report('H=%d projection data gradient' % H, flop, t32, t16, ts)
        # bank filter gradients (12,800-frame contraction): transposes + wgrad_kernel vs transposed splits + gemm16 (atomics)
        grads = [torch.zeros_like(W) for W in Ws]
        arena = torch.zeros(sum(W.numel() for W in Ws), device=dev)
        views, off = [], 0
        for W in Ws:
            views.append(arena[off:off + W.numel()].view(W.shape))
            off += W.numel()

        def f32_wgrad():
            xt, ldx = training._Ops.transpose(x, M, H, H, T)
            zt, ldz = training._Ops.transpose(zb, M, CB, CB, T)
            grp = [(128 * (k - 1), 128, k, -((k - 1) // 2), grads[k - 1], 128) for k in range(1, K + 1)]
            training._Ops.wgrad(xt, ldx, H, M, T, zt, ldz, grp)

        def f16_prep():
            return (gemm16.transpose_split16(x, M, H, H, T, shift0=-(K // 2 - 1), n_shifts=K),
                    gemm16.transpose_split16(zb, M, CB, CB, T))
        (XT, rsX), (ZT, rsZ) = f16_prep()
        flop = 2.0 * M * 528 * H * 128
        t32 = timed(f32_wgrad)
        ts = timed(f16_prep)
        t16 = timed(lambda: gemm16.bank_wgrad(XT, rsX, ZT, rsZ, H, K, M, views, arena, splits=6 if H >= 256 else 8))
        report('H=%d bank filter gradients (incl. operand transposes)' % H, flop, t32, t16, ts)
        print('      K ranges per tile (float atomics): ' + '   '.join('%d: %.3f ms' % (sp, timed(lambda: gemm16.bank_wgrad(
            XT, rsX, ZT, rsZ, H, K, M, views, arena, splits=sp))) for sp in (1, 2, 3, 4, 6, 8)))
        del XT, ZT
        if H != 256:
            continue
        # projection forward on pool(relu(bn(Zb)))
        sc, sh = torch.rand(CB, device=dev) + 0.5, torch.randn(CB, device=dev)
        q1 = torch.empty((M, H), device=dev)
        btf = W1.permute(2, 0, 1).reshape(H, 3 * CB).contiguous()
        flop = 2.0 * M * 3 * CB * H
        t32 = timed(lambda: modules.gemm_launch(zb, M, T, CB, CB, H, [(btf, 3 * CB, 3, 1, 0)], q1, H, _vc.VC_F32, pro_scale=sc,
                                                pro_shift=sh, pro_relu=1, pro_pool=1, out_f32=True))
        ts = timed(lambda: gemm16.split16(zb, M, CB, CB, T, scale=sc, shift=sh, relu=1, pool=1))
        z16, zrs = gemm16.split16(zb, M, CB, CB, T, scale=sc, shift=sh, relu=1, pool=1)
        ws = torch.empty(_vc.lib().vc_gemm16_workspace_bytes(M, CB, 1), dtype=torch.uint8, device=dev)
        t16 = timed(lambda: gemm16.gemm16(z16, zrs, M, T, CB, pp, q1, H, col_scale=pcs, workspace=ws))
        report('H=256 projection forward (BN+relu+pool operand)', flop, t32, t16, ts)
        # bank data gradient
        grp = [(W.flip(0).permute(1, 0, 2).reshape(H, -1).contiguous(), k * 128, k, k - 1 - (k - 1) // 2, 128 * (k - 1)) for k, W in enumerate(Ws, 1)]
        dd = torch.zeros((M, H), device=dev)
        flop = 2.0 * M * 528 * H * 128
        t32 = timed(lambda: modules.gemm_launch(zb, M, T, 128, CB, H, grp, dd, H, _vc.VC_F32, out_f32=True, sum_groups=16))
        ts = timed(lambda: gemm16.split16(zb, M, CB, CB, T))
        z16, zrs = gemm16.split16(zb, M, CB, CB, T)
        t16 = timed(lambda: gemm16.gemm16(z16, zrs, M, T, CB, bp, dd, H, col_scale=bcs, ragged=True, accumulate=True, workspace=ws))
        report('H=256 bank data gradient (16-way atomics vs ragged walk)', flop, t32, t16, ts)


if __name__ == '__main__':
    main()
