"""Per-shape timing of the fp16-storage backbone's convolutions at BASELINE configs[4] (B = 256): one eager forward with an event
pair around every rih_hconv launch (ops.PROFILE), grouped by shape; each shape against ITS roofline -- the larger of
flop / 2500 TF/s (dense f16 MFMA) and algorithmic bytes / 8 TB/s (HBM).  Usage: python tools/hconv_sweep.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from renderih_amd import ops                                    # noqa: E402
from renderih_amd.model import build_model                      # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = build_model(dropout=0.05).to(dev).eval()
    model.use_fp16_backbone()
    img = torch.randn(B, 3, 256, 256, device=dev)
    with torch.no_grad():
        for _ in range(2):
            model(img)
        torch.cuda.synchronize()
        groups = {}
        for _ in range(3):
            ops.PROFILE = []
            model(img)
            torch.cuda.synchronize()
            recs, ops.PROFILE = ops.PROFILE, None
            for f, e0, e1, tag in recs:
                if len(tag) < 11 or tag[8] != 'f16':
                    continue
                g = groups.setdefault((tag[0], tag[1], tag[2]) + tuple(tag[10]), [0, 0.0, f, tag[9]])
                g[0] += 1
                g[1] += e0.elapsed_time(e1)
    pairs = []
    for _ in range(64):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        b.record()
        pairs.append((a, b))
    torch.cuda.synchronize()
    empty = sorted(a.elapsed_time(b) for a, b in pairs)[32]
    rows = []
    for key, (n, ms, f, nb) in groups.items():
        us = 1000.0 * (ms / n - empty)
        ideal = max(f / 2500e12, nb / 8e12) * 1e6
        rows.append((us * n / 3, key, n // 3, us, f / us / 1e6, nb / us / 1e3, ideal, ideal / us,
                     'mfma' if f / 2500e12 > nb / 8e12 else 'hbm'))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    tot_ideal = sum(r[6] * r[2] for r in rows)
    print('B = %d; %d shapes; per forward %.2f ms in rih_hconv, roofline sum %.2f ms (frac %.3f); event-pair overhead %.1f us'
          % (B, len(rows), tot / 1000, tot_ideal / 1000, tot_ideal / tot, 1000 * empty))
    print('%8s %6s %6s | %4s %4s %2s %2s %3s %3s | %3s %9s %8s %8s %9s %6s %5s' % ('M', 'Cout', 'K', 'H', 'W', 'k', 's', 'res', 'f32',
                                                                             'n', 'us', 'TF/s', 'GB/s', 'ideal us', 'frac', 'bound'))
    for t, key, n, us, tf, gb, ideal, frac, bound in rows:
        print('%8d %6d %6d | %4d %4d %2d %2d %3d %3d | %3d %9.1f %8.1f %8.1f %9.1f %6.3f %5s'
              % (key + (n, us, tf, gb, ideal, frac, bound)))


if __name__ == '__main__':
    main()
