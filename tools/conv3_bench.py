#!/usr/bin/env python
"""The halo-resident 3x3 convolution (csrc/rih_conv3.hip, ops.HALO3) against the tap-by-tap implicit GEMM (rih_gemm engine 2) on
the 3x3 shapes of the ResNet50 training step at B = 64 -- forward (with / without the BatchNorm statistics epilogue) and data
gradient: HIP-event time per launch, TF/s of algorithmic FLOPs, interleaved rounds (A, B, A, B ...) in ONE process, and the
maximum error of both against an fp64 product on an 8-image slice.  Operand magnitudes as in a step: BatchNorm + ReLU
activations, He-scaled weights, gradients of order 1e-4.  Run on the GPU box:  python tools/conv3_bench.py"""
import os
import sys
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderih_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
B = int(os.environ.get('CONV3_B', '32' if os.environ.get('CONV3_SET') == 'hrnet' else '64'))
# (H = W, Cin, Cout, launches per step forward): trunk layer1 / aux decoders @64, layer2 + aux @32, layer3 + aux @16 (16 x 16
# patches); CONV3_SET=hrnet: the BasicBlock convolutions of HRNet-W32's branches at B = 32 (8 modules x 4 blocks x 2 per branch)
SHAPES = [(64, 64, 64, 3), (64, 128, 128, 2), (32, 128, 128, 5), (16, 256, 256, 5), (16, 128, 128, 2)]
if os.environ.get('CONV3_SET') == 'hrnet':
    SHAPES = [(64, 32, 32, 64), (32, 64, 64, 56), (16, 128, 128, 32)]
ROUNDS, ITERS = 5, 10


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS * 1000.0


def main():
    assert ops.ENGINE == 2
    print('device', torch.cuda.get_device_name(0), 'B', B, flush=True)
    total = {True: 0.0, False: 0.0}
    for H, Cin, Cout, per_step in SHAPES:
        torch.manual_seed(H + Cin + Cout)
        x = torch.relu(torch.randn(B, H, H, Cin, device=dev) * 1.3 + 0.2)
        w = (torch.randn(Cout, Cin, 3, 3, device=dev) * (2.0 / (9 * Cin)) ** 0.5)
        dy = torch.randn(B, H, H, Cout, device=dev) * 1e-4 * torch.exp(torch.randn(B, H, H, 1, device=dev))
        flops = 2.0 * B * H * H * Cout * 9 * Cin
        y = {h: torch.empty(B, H, H, Cout, device=dev) for h in (True, False)}
        dx = {h: torch.empty(B, H, H, Cin, device=dev) for h in (True, False)}
        bx, bw, bdy = ops.bound_of(x), ops.bound_of(w), ops.bound_of(dy)
        planes_f = ops._h2_weight(w, Cin, False)
        planes_d = ops._h2_weight(w, Cin, True)
        wp_f = ops._packed_weight(w, Cin, False)
        wp_d = ops._packed_weight(w, Cin, True)
        M, K = B * H * H, 9 * Cin
        geom_f = (H, H, Cin, H, H, 3, 3, 1, 1, 1, 1)
        geom_d = (H, H, Cout, H, H, 3, 3, 1, 1, 1, 1)

        def halo_fwd(stats):
            from renderih_amd._lib import Conv3Desc
            import ctypes as C
            d = Conv3Desc()
            d.x, d.w_h2, d.y, d.amax_x, d.amax_w = x.data_ptr(), planes_f[0].data_ptr(), y[True].data_ptr(), bx.data_ptr(), bw.data_ptr()
            d.imgs, d.H, d.W, d.C, d.N, d.ldx, d.ldy, d.Kpad, d.relu = B, H, H, Cin, Cout, Cin, Cout, planes_f[1], 0
            st = torch.empty(M // 64, 2, Cout, device=dev) if stats else None
            if stats:
                d.stats = st.data_ptr()
            return lambda: ops.check(ops._L().rih_conv3x3(C.byref(d), ops._stream()), 'rih_conv3x3'), st

        def halo_dgrad():
            from renderih_amd._lib import Conv3Desc
            import ctypes as C
            d = Conv3Desc()
            d.x, d.w_h2, d.y, d.amax_x, d.amax_w = dy.data_ptr(), planes_d[0].data_ptr(), dx[True].data_ptr(), bdy.data_ptr(), bw.data_ptr()
            d.imgs, d.H, d.W, d.C, d.N, d.ldx, d.ldy, d.Kpad, d.relu = B, H, H, Cout, Cin, Cout, Cin, planes_d[1], 0
            return lambda: ops.check(ops._L().rih_conv3x3(C.byref(d), ops._stream()), 'rih_conv3x3')

        def gemm_fwd(stats):
            holder = [None]

            def fn():
                holder[0] = ops.StatsHolder() if stats else None
                ops.gemm(x, wp_f, y[False], M, Cout, K, Cin, Cout, Cout, a_mode=0, b_mode=0, geom=geom_f, stats=holder[0],
                         amax_a=bx, amax_b=bw)
            return fn

        def gemm_dgrad():
            return lambda: ops.gemm(dy, wp_d, dx[False], M, Cin, 9 * Cout, Cout, Cin, Cin, a_mode=0, b_mode=0, geom=geom_d,
                                    amax_a=bdy, amax_b=bw)
        cases = [('fwd', halo_fwd(False)[0], gemm_fwd(False)), ('fwd+stats', halo_fwd(True)[0], gemm_fwd(True)),
                 ('dgrad', halo_dgrad(), gemm_dgrad())]
        for name, fh, fg in cases:
            for f in (fh, fg):
                for _ in range(3):
                    f()
            torch.cuda.synchronize()
            th, tg = [], []
            for _ in range(ROUNDS):
                th.append(timed(fh))
                tg.append(timed(fg))
            mh, mg = sorted(th)[ROUNDS // 2], sorted(tg)[ROUNDS // 2]
            print('%3dx%-3d %4d->%-4d %-9s  halo %7.1f us %6.1f TF/s   implicit GEMM %7.1f us %6.1f TF/s   x%.2f   (min %.1f / %.1f)'
                  % (H, H, Cin, Cout, name, mh, flops / mh / 1e6, mg, flops / mg / 1e6, mg / mh, min(th), min(tg)), flush=True)
            if name != 'fwd':
                total[True] += mh * per_step
                total[False] += mg * per_step
        # accuracy on 8 images against fp64
        n = 8
        ref = F.conv2d(x[:n].permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
        sc = float(ref.abs().max())
        e = {h: float((y[h][:n].double() - ref).abs().max()) / sc for h in (True, False)}
        f32 = float((F.conv2d(x[:n].permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1).double() - ref).abs().max()) / sc
        refd = F.conv_transpose2d(dy[:n].permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
        scd = float(refd.abs().max())
        ed = {h: float((dx[h][:n].double() - refd).abs().max()) / scd for h in (True, False)}
        print('        error vs fp64 (max |d| / max |ref|): forward halo %.2e implicit %.2e torch-fp32 %.2e; dgrad halo %.2e implicit %.2e'
              % (e[True], e[False], f32, ed[True], ed[False]), flush=True)
    print('per training step (forward with statistics + data gradient, launches as in ResNet50 B = 64): halo %.0f us, implicit GEMM %.0f us'
          % (total[True], total[False]))


if __name__ == '__main__':
    main()
