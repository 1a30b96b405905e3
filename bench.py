#!/usr/bin/env python
"""bench.py -- images/sec of the pose-network hot path (forward + loss + backward + optimizer step) on MI355X.

Workload = BASELINE.json configs[1]: batch 64 per GPU, 256x256 RGB, ResNet50 encoder + GCN/attention mesh decoder
(the reference's `HandNET_GCN`), fp32, training mode with the reference's dropout 0.05, synthetic inputs/labels,
random-init weights.  N>1: one process per GPU (torch.distributed, backend "nccl" = RCCL), plain data parallel with
the reference's DDP semantics (gradient averaging, grad-less parameters tolerated, per-GPU BatchNorm) through one
bucketed RCCL all-reduce per step (renderih_amd/dp.py; `--ddp` switches to torch DDP itself); weak scaling (batch 64
per GPU).

The step is `renderih_amd.train.TrainStep`: forward + fused mesh loss + backward in three stages captured as hipGraphs, each
stage's gradient bucket all-reduced on a side stream while the next stage runs, fused Adam; `comm_ms_exposed` is the time the
compute stream waited for the last bucket.  `eager_reference_loop` is what the reference's UNMODIFIED loop gets on this
package (eager launches, torch mirror of core/Loss.py, foreach Adam) -- measured next to the headline so that the cost of not
adopting the three-line TrainStep patch is visible.

Prints ONE JSON line (rank 0).  `roofline`: the GEMM/implicit-conv kernel family (rih_gemm), timed live with HIP events
on the launch stream; achieved = SURVEY 8d algorithmic FLOPs (53.2 GFLOP/img fwd+bwd) x images per step / GEMM-family
time per step (`achieved_launched` = the FLOPs actually launched: dead branches are skipped).  `roofline_hbm`: the BatchNorm
kernel family (largest HBM-bound group) the same way in GB/s against 8 TB/s; `roofline_mano`: the MANO layer micro-benchmark
(4096 hands, PCA-45) in GB/s of its algorithmic bytes.  The default engine (2) computes every fp32 product as THREE fp16 MFMA products (scaled two-term
operand split, fp32 accumulate): ceiling 2500 / 3 = 833 TFLOP/s of fp32-equivalent work; engine 1 (six bf16 products, the
decoder's Linears) 416.7; engine 0 / flash attention the native f32 MFMA peak 157.3.  `roofline.peak` is the ceiling of the engine that
produced most of the family's time, `by_engine` carries every engine's own fraction.  `cpu_baseline`: the CPU oracle (a port of the
reference's PyTorch-CPU path) timed on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_IMG_FWD_BWD = 53.2      # 3 x 17.72 GFLOP/img forward (conv + matmul, 2*MAC)
PEAK_FP32_MFMA_TF = 157.3         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TF = 2500.0        # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16)


def measured_traffic():
    """HBM traffic of the rih_gemm family per step from the newest committed PMC collection (profiles/*/traffic_*.json;
    bench.py cannot run rocprofv3 on itself).  None when no collection is committed."""
    import glob
    import re

    def version(path):      # profiles/rNN/traffic_vMM.json -> (NN, MM): newest round, newest collection
        m = re.search(r'r(\d+)[/\\]traffic_v(\d+)\.json$', path)
        return (int(m.group(1)), int(m.group(2))) if m else (-1, -1)
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*', 'traffic_*.json')), key=version)
    if not files:
        return None
    with open(files[-1]) as fh:
        d = json.load(fh)
    n = max(int(d.get('launches_per_step', 0)), 1)
    return {'hbm_read_GB_per_step': d['fetch_GB_per_step_corrected'], 'hbm_write_GB_per_step': d['write_GB_per_step'],
            'launches_per_step': n,
            'hbm_read_MB_per_launch': round(1000.0 * d['fetch_GB_per_step_corrected'] / n, 2),
            'hbm_write_MB_per_launch': round(1000.0 * d['write_GB_per_step'] / n, 2),
            'from': os.path.relpath(files[-1], ROOT), 'note': 'ResNet50 B=64 step; ' + d['correction']}


def synth_batch(B, device, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, 3, 256, 256, generator=g)
    lab = {'v3d_l': 0.05 * torch.randn(B, 778, 3, generator=g), 'v3d_r': 0.05 * torch.randn(B, 778, 3, generator=g),
           'v2d_l': 256 * torch.rand(B, 778, 2, generator=g), 'v2d_r': 256 * torch.rand(B, 778, 2, generator=g),
           'root_rel': 0.05 * torch.randn(B, 3, generator=g)}
    return img.to(device), {k: v.to(device) for k, v in lab.items()}


def cpu_baseline(seconds=40.0, batch=16, family='a', threads=(16, 32, 64), inference=False):
    """The CPU baseline of SURVEY 8d -- the reference's PyTorch-CPU path, forward + backward at B = 16 -- on THIS box's host
    cores.  kind = "port": /root/reference does not exist on the GPU box, so the timed code is oracle/net_oracle.py, the
    functional restatement of the reference modules (same torch CPU operators in the same order, pinned to the real modules by
    tests/golden).  torch's intra-op thread count is swept (one iteration each after one warm-up: 128 threads on a 128-core
    host oversubscribe the small decoder operators), then >= 3 iterations are timed at the best setting; bounded to about
    `seconds` of CPU work.  inference=True (configs[4]): eval-mode forward under no_grad only.
    `reference_ratio`: port images/s divided by the REAL reference modules' images/s, measured where both can run (the build
    container, same weights / image / threads: tools/cpu_port_vs_reference.py -> profiles/r05/cpu_port_vs_reference.json)."""
    from oracle import net_oracle
    from renderih_amd import assets
    from renderih_amd.model import build_model
    torch.manual_seed(0)
    if family == 'b':
        from renderih_amd.lijun import build_graph_model
        m = build_graph_model(0.0)
    else:
        m = build_model(0.0)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    for k, v in sd.items():
        if v.is_floating_point() and 'running' not in k and 'dense_coor' not in k:
            v.requires_grad_(True)
    graph = net_oracle.graph_from_dicts(assets.load_graph_dict('left'), assets.load_graph_dict('right'))
    img = torch.randn(batch, 3, 256, 256)

    def one():
        t0 = time.time()
        if inference:
            with torch.no_grad():
                net_oracle.handnet_forward(sd, graph, img, training=False)
            return time.time() - t0
        out = net_oracle.handnet_forward(sd, graph, img, training=True)
        net_oracle.scalar_loss(out).backward()
        for v in sd.values():
            v.grad = None
        return time.time() - t0
    ncpu = os.cpu_count() or 1
    before = torch.get_num_threads()
    cand = sorted({min(t, ncpu) for t in threads})
    t_start = time.time()
    torch.set_num_threads(cand[0])
    one()                                   # warm-up (allocator, oneDNN primitive caches)
    sweep = {}
    for t in cand:
        torch.set_num_threads(t)
        sweep[t] = one()
        if time.time() - t_start > 0.6 * seconds:
            break
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    times = []
    while len(times) < 3 or (time.time() - t_start < seconds and len(times) < 10):
        times.append(one())
    torch.set_num_threads(before)
    t_total = sum(times)
    ratio = None
    try:
        with open(os.path.join(ROOT, 'profiles', 'r05', 'cpu_port_vs_reference.json')) as fh:
            rr = json.load(fh)
        ratio = {'reference_ratio': rr['reference_ratio'],
                 'reference_ratio_source': 'profiles/r05/cpu_port_vs_reference.json: fwd+bwd B = %d on %d threads of the build '
                                           'container, port %.2f vs reference modules %.2f images/s'
                                           % (rr['batch'], rr['threads'], rr['port_images_per_sec'], rr['reference_images_per_sec'])}
    except (OSError, KeyError, ValueError):
        pass
    out = {'value': round(batch * len(times) / t_total, 3), 'unit': 'images/sec', 'cores': best, 'host_cores': ncpu,
           'kind': 'port', 'thread_sweep_s_per_iter': {str(k): round(v, 2) for k, v in sweep.items()},
           'sample': 'oracle (CPU restatement of the reference path; the reference modules cannot travel to the GPU box) '
                     '%s, batch %d, %d timed iterations (%.0f s) at %d threads after 1 warm-up + a %d-point thread sweep'
                     % ('eval-mode forward' if inference else 'fwd+bwd', batch, len(times), t_total, best, len(sweep))}
    if ratio is not None:
        out.update(ratio)
    return out


def mano_roofline(device, hands=4096, iters=20):
    """MANO layer micro-benchmark.  The survey books the layer as HBM-bound (1.46 MB basis + 9.9 KB per hand), but
    its arithmetic intensity is 118 FLOP/B (1.17 MFLOP per hand against a ridge of ~20 on the exact-fp32 MFMA): both fractions
    are reported -- `frac` against HBM as the survey asks, `mfma_f32` against the 157.3 TF/s f32 matrix peak that actually
    bounds it."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import mano_bench
    r = mano_bench.measure(hands, iters, device)
    flop = 1.17e6 * hands                       # blend 0.69 + skinning 0.34 + joints / chain 0.14 MFLOP per hand
    tf = flop / (r['fwd_us'] * 1e-6) / 1e12
    return {'bound': 'hbm', 'achieved': r['fwd_GBps'], 'peak': 8000.0, 'unit': 'GB/s', 'frac': r['fwd_frac_of_8TBps'],
            'traffic': None,
            'mfma_f32': {'bound': 'mfma-f32', 'achieved': round(tf, 2), 'peak': PEAK_FP32_MFMA_TF, 'unit': 'TFLOP/s',
                         'frac': round(tf / PEAK_FP32_MFMA_TF, 4)},
            'kernel': 'rih_mano_fwd (ManoLayer.forward, %d hands, PCA-45): %.1f us, %.0f hands/s; '
            'algorithmic bytes = 1.46 MB basis + 9.9 KB per hand, 1.17 MFLOP per hand' % (hands, r['fwd_us'], r['fwd_hands_per_s']),
            'fwdbwd_us': r['fwdbwd_us'], 'fwdbwd_hands_per_s': r['fwdbwd_hands_per_s']}


_REAL_STDOUT = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  RCCL prints a version banner through C stdio when a communicator comes up
    (buffered: it lands at process exit, i.e. BEHIND the JSON line, and every rank prints one), so file descriptor 1 is pointed
    at stderr for the whole process -- Python's and every library's writes -- and the JSON line alone goes to the saved
    descriptor (`emit`)."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    data = (json.dumps(obj) + '\n').encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


def config5(args):
    """BASELINE configs[4]: inference only, batch 256, fp16-storage backbone (BatchNorm folded, f16 MFMA, fp32 accumulate; the
    mesh decoder and the MANO layer stay fp32), encoder + attention decoder + MANO layer on 2 x 256 hands captured in ONE
    hipGraph; W untimed + K timed replays between synchronizations.  MPJPE against a pretrained checkpoint needs licence-gated
    assets (deferred); the deviation of the predicted vertices from the fp32 path on the same inputs is reported."""
    import torch
    from renderih_amd import assets
    from renderih_amd.model import build_model
    from renderih_amd.manolayer import ManoLayer, rodrigues_batch
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    B = args.batch if args.batch else 256
    model = build_model(dropout=0.05).to(dev).eval()
    mano = {s_: ManoLayer(assets.synthetic_mano_dict(s_)).to(dev) for s_ in ('left', 'right')}
    img = torch.randn(B, 3, 256, 256, device=dev)
    with torch.no_grad():
        ref32 = model(img[:8])[0]['verts3d']
        model.use_fp16_backbone()
        got16 = model(img[:8])[0]['verts3d']
    dev_err = max(float((got16[s_] - ref32[s_]).abs().max() / ref32[s_].abs().max()) for s_ in ('left', 'right'))
    root = rodrigues_batch(torch.randn(B, 3)).to(dev)
    pose, shape = (0.5 * torch.randn(B, 45)).to(dev), torch.randn(B, 10).to(dev)

    def forward():
        with torch.no_grad():
            out = model(img)
            hands = [mano[s_](root, pose, shape) for s_ in ('left', 'right')]
        return out, hands

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        forward()
    torch.cuda.current_stream().wait_stream(side)
    roof = None
    if not args.no_roofline:
        # one EAGER forward with an event pair around every matrix-pipe launch (rih_hconv of the fp16-storage backbone, rih_gemm
        # and flash attention of the fp32 decoder); the empty-pair cost is calibrated and taken off as in the training bench
        from renderih_amd import ops
        torch.cuda.synchronize()
        ops.PROFILE = []
        forward()
        torch.cuda.synchronize()
        recs, ops.PROFILE = ops.PROFILE, None
        pairs = []
        for _ in range(64):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            b.record()
            pairs.append((a, b))
        torch.cuda.synchronize()
        empty = sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2]
        fam = {}
        for f, e0, e1, tag in recs:
            eng = tag[8] if len(tag) > 8 else 'other'
            if tag[6] == 30:
                eng = 'flash'
            k = ('backbone f16 (rih_hconv)' if eng == 'f16' else 'decoder attention (rih_flash_fwd, split-bf16 products)'
                 if eng == 'flash' else 'decoder f32 (rih_gemm engine %s)' % eng)
            a = fam.setdefault(k, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += max(e0.elapsed_time(e1) - empty, 0.0)
            a[2] += f
        # ceilings: one f16 product per MAC on the fp16-storage backbone (2500 TF/s dense); the fp32 decoder runs the split
        # engines: six bf16 products (2500 / 6) or, where operand bounds exist, three fp16 products (2500 / 3)
        peaks = {'f16': PEAK_BF16_MFMA_TF, '1': PEAK_BF16_MFMA_TF / 6.0, '2': PEAK_BF16_MFMA_TF / 3.0, '0': PEAK_FP32_MFMA_TF}
        fams = {}
        for k, (n_, ms_, fl) in sorted(fam.items()):
            key = 'f16' if 'rih_hconv' in k else '1' if 'flash' in k else k.split('engine ')[-1].rstrip(')')
            pk = peaks.get(key, PEAK_BF16_MFMA_TF / 6.0)
            fams[k] = {'launches': n_, 'ms': round(ms_, 3), 'achieved': round(fl / max(ms_, 1e-9) / 1e9, 1), 'peak': round(pk, 1),
                       'frac': round(fl / max(ms_, 1e-9) / 1e9 / pk, 4)}
        top = max(fams.items(), key=lambda kv: kv[1]['ms'])
        roof = {'bound': 'mfma', 'achieved': top[1]['achieved'], 'peak': top[1]['peak'], 'unit': 'TFLOP/s', 'frac': top[1]['frac'],
                'traffic': None, 'kernel': top[0] + ': the matrix-pipe family with the most time in one eager forward',
                'families': fams, 'event_pair_overhead_us': round(1000.0 * empty, 2)}
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = forward()
    for _ in range(args.warmup):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g.replay()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    del keep
    emit({'metric': 'images/sec inference (encoder + attention decoder + MANO layer), batch 256 fp16, hipGraph',
                      'value': round(B * args.steps / el, 2), 'unit': 'images/sec', 'n_gpus': 1, 'steps': args.steps,
                      'warmup': args.warmup, 'ms_per_step': round(1000.0 * el / args.steps, 3), 'higher_is_better': True,
                      'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16 storage / f32 accumulate (backbone); f32 (decoder, MANO)',
                      'data': 'synthetic',
                      'config': {'workload': 'BASELINE configs[4]: inference-only, batch=%d, hipGraph-captured encoder + attention '
                                             'decoder + MANO layer (2 x %d hands) on 1 x MI355X' % (B, B),
                                 'vertices_rel_deviation_from_fp32_path': dev_err,
                                 'mpjpe': 'deferred: needs the pretrained checkpoint and InterHand2.6M (licence-gated)'},
          'roofline': roof,
          'cpu_baseline': None if args.no_cpu_baseline else cpu_baseline(seconds=25.0, inference=True)})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--encoder', default='resnet50', choices=['resnet50', 'hrnet32'],
                    help='resnet50 = BASELINE configs[1]/[2] (the headline metric); hrnet32 = configs[3]')
    ap.add_argument('--family', default='a', choices=['a', 'b', 'b-mano'],
                    help="a = models/model.py (the headline configuration); b = the reference's second family, "
                         "common/myhand/lijun_model_graph.load_graph_model (SURVEY 8f rank 1; resnet50 only); "
                         "b-mano = lijun_model_newgraph.load_new_model (MANO layer inside the forward)")
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default 64, hrnet32: 32)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--torch-adam', action='store_true', help='torch.optim.Adam(fused=True) instead of renderih_amd.optim.Adam')
    ap.add_argument('--no-graph', action='store_true',
                    help='launch every kernel from Python each step instead of replaying a captured hipGraph')
    ap.add_argument('--ddp', action='store_true',
                    help='N>1: use torch DistributedDataParallel(find_unused_parameters=True) like the reference trainer '
                         'instead of renderih_amd.dp.GradAllReducer (same gradients, ~25 ms/step more host work)')
    ap.add_argument('--force-dist', action='store_true',
                    help='initialise RCCL and wrap the model in DDP even at world size 1 (exercises the N>1 code path)')
    ap.add_argument('--no-stages', action='store_true', help='one backward stage / one gradient bucket instead of three')
    ap.add_argument('--stages', action='store_true',
                    help='three backward stages even on one rank (default: three when gradients are exchanged -- the stages exist to '
                         'overlap the per-stage all-reduce with the next stage --, one on a single rank, where the whole '
                         'backward then groups its weight gradients into four launches instead of twelve: +1.2 %% same-box)')
    ap.add_argument('--no-overlap', action='store_true', help='all-reduce on the compute stream (no overlap with backward)')
    ap.add_argument('--no-reference-loop', action='store_true', help="skip the 5-step measurement of the reference's unmodified loop")
    ap.add_argument('--dump-gemm', default=None, help='write the per-launch GEMM profile of one step to this JSON file')
    ap.add_argument('--config5', action='store_true',
                    help='BASELINE configs[4] instead of the training step: batch-256 fp16-storage inference in one hipGraph')
    args = ap.parse_args()
    if args.config5:
        claim_stdout()
        return config5(args)
    if args.batch is None:
        args.batch = 32 if args.encoder == 'hrnet32' else 64

    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher -- one rank per GPU under torch.distributed.run
        # (what the driver's own command line does), rendezvous on 127.0.0.1
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)
    claim_stdout()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and not (args.gpus == 1 and world == 1):
        raise SystemExit('bench.py --gpus %d was launched with WORLD_SIZE=%d: the two must agree (launch with '
                         '`--nproc-per-node %d`, or run `python bench.py --gpus %d` and let it spawn the ranks)'
                         % (args.gpus, world, args.gpus, args.gpus))
    staged = (not args.no_stages) and (args.stages or world > 1 or args.force_dist)
    if os.environ.get('RIH_BENCH_SPAWN_PROBE') == '1':
        # CPU test of the launch plumbing (tests/test_bench_spawn.py): every rank joins a gloo group, rank 0 reports the world
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if world > 1:
            dist.init_process_group('gloo', rank=rank, world_size=world)
            t = torch.ones(1)
            dist.all_reduce(t)
            ranks = int(t.item())
            dist.destroy_process_group()
        else:
            ranks = 1
        if rank == 0:
            emit({'spawn_probe': True, 'n_gpus': world, 'ranks_seen': ranks, 'gpus_arg': args.gpus})
        return
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (HIP kernels only, no CPU fallback)')
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)

    from renderih_amd import ops, assets
    from renderih_amd.model import build_model
    from renderih_amd.loss import GraphLoss, FusedMeshLoss, calc_loss_GCN_fused
    from renderih_amd.manolayer import ManoLayer

    torch.manual_seed(0)
    if args.family != 'a':
        if args.encoder != 'resnet50':
            raise SystemExit('--family b / b-mano are resnet50 configurations')
        from renderih_amd.lijun import build_graph_model, build_new_model
        model = (build_graph_model if args.family == 'b' else build_new_model)(dropout=0.05).to(device).train()
    else:
        model = build_model(dropout=0.05, encoder_type=args.encoder).to(device).train()
    model.decoder.unsample_layer.weight.requires_grad_(False)          # core/gcn_trainer.py:102-103
    net = model
    if dist_on and args.ddp:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local], find_unused_parameters=True)
    # the reference's optimizer (utils/defaults.yaml: Adam, lr 3e-4, weight decay 1e-2) as ONE launch over all 843 tensors
    # (renderih_amd.optim.Adam = rih_adam_multi, torch.optim.Adam semantics and state_dict layout; --torch-adam = torch's
    # fused multi-tensor Adam: 24 launches, 0.89 ms per step)
    from renderih_amd import optim as rih_optim
    adam = torch.optim.Adam if args.torch_adam else rih_optim.Adam
    opt = adam([p for p in model.parameters() if p.requires_grad], lr=3e-4, weight_decay=1e-2,
               **({'fused': True} if args.torch_adam else {}))

    mano = {s: ManoLayer(assets.synthetic_mano_dict(s)) for s in ('left', 'right')}
    gl = {s: GraphLoss(mano[s].J_regressor, mano[s].get_faces(), level=4, device=device) for s in ('left', 'right')}
    conv = model.decoder.converter
    fused_loss = FusedMeshLoss(gl['left'], gl['right'], conv['left'], conv['right'])      # core/Loss.py on one HIP kernel
    B = args.batch
    img, lab = synth_batch(B, device, seed=rank)

    def loss_fn(out, labels):
        return calc_loss_GCN_fused(fused_loss, None, *out, labels['v2d_l'], labels['v2d_r'], labels['v3d_l'], labels['v3d_r'],
                                   labels['root_rel'])[0]

    def fwd_bwd(module=None):
        loss = loss_fn((net if module is None else module)(img), lab)
        loss.backward()
        return loss

    def barrier():
        if dist_on:
            torch.distributed.barrier()

    eager_ref = None
    if rank == 0 and world == 1 and not args.no_reference_loop and not args.ddp:
        # the reference's unmodified loop (core/gcn_trainer.py:219-251) on this package: eager launches, the torch mirror of
        # core/Loss.py, optimizer.zero_grad / backward / step with torch's default (foreach) Adam
        from renderih_amd.loss import calc_loss_GCN
        opt_ref = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=3e-4, weight_decay=1e-2)
        state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}

        def ref_step():
            opt_ref.zero_grad()
            out = model(img)
            loss, _ = calc_loss_GCN(None, 0, gl['left'], gl['right'], conv['left'], conv['right'], *out, lab['v2d_l'],
                                    lab['v2d_r'], lab['v3d_l'], lab['v3d_r'], lab['root_rel'])
            loss.backward()
            opt_ref.step()
        for _ in range(2):
            ref_step()
        torch.cuda.synchronize()
        if os.environ.get('RIH_PROFILE_REF_LOOP'):          # host-side profile of the eager loop (it is launch-bound)
            import cProfile
            import pstats
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(3):
                ref_step()
            torch.cuda.synchronize()
            pr.disable()
            pstats.Stats(pr, stream=sys.stderr).sort_stats('tottime').print_stats(45)
        t0 = time.perf_counter()
        for _ in range(5):
            ref_step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        eager_ref = {'images_per_sec': round(B / dt, 1), 'ms_per_step': round(1e3 * dt, 2),
                     'what': 'unmodified reference loop on this package: eager kernel launches, core/Loss.py torch mirror, '
                             'foreach Adam (no TrainStep, no hipGraph, no fused loss)'}
        model.load_state_dict(state0)
        del opt_ref, state0
        for p_ in model.parameters():
            p_.grad = None

    use_graph = not args.no_graph and not args.ddp
    trainer = None
    if args.ddp:
        def step():
            opt.zero_grad(set_to_none=True)
            loss = fwd_bwd()
            opt.step()
            return loss
    else:
        from renderih_amd.train import TrainStep
        try:
            trainer = TrainStep(model, opt, loss_fn, (img, lab), use_graph=use_graph, stages=staged,
                                overlap=not args.no_overlap, force_exchange=args.force_dist)
        except Exception as exc:                    # keep the benchmark alive if capture fails: same step, launched eagerly
            print('[bench] hipGraph capture failed on rank %d (%s: %s); launching eagerly' % (rank, type(exc).__name__, exc),
                  file=sys.stderr, flush=True)
            use_graph = False
            ops.DROPOUT_SEED_TENSOR = None
            torch.cuda.synchronize()
            trainer = TrainStep(model, opt, loss_fn, (img, lab), use_graph=False, stages=staged,
                                overlap=not args.no_overlap, force_exchange=args.force_dist)
        use_graph = trainer.use_graph
        step = trainer

    for _ in range(args.warmup):
        step()
    barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    ms_median = per_step[len(per_step) // 2]
    if dist_on:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.item())
    comm_exposed = trainer.comm_ms_exposed() if trainer is not None else None

    roof = roof_hbm = roof_mano = None
    if not args.no_roofline and rank == 0:
        ops.PROFILE = []
        ops.PROFILE_ELEM = []
        for p_ in model.parameters():
            p_.grad = None
        if trainer is not None:
            # one EAGER step of the same TrainStep configuration (staged backward, grouped weight-gradient launches, deferred
            # reductions) with an event pair around every GEMM-family launch.  process_group=False: the other ranks are not
            # in this block, so no collective may be issued here
            from renderih_amd.train import TrainStep as _TS
            prof = _TS(model, opt, loss_fn, (img, lab), use_graph=False, stages=staged, process_group=False)
            prof()
            ops.PROFILE = []            # (the first eager step also walks the autograd graph once: profile the second)
            ops.PROFILE_ELEM = []
            prof()
            del prof
        else:
            fwd_bwd(model)  # --ddp: local forward + loss + backward on the bare module (no reducer, no DDP wrapper)
        torch.cuda.synchronize()
        recs = ops.PROFILE
        erecs = ops.PROFILE_ELEM
        ops.PROFILE = None
        ops.PROFILE_ELEM = None
        # an event pair around NOTHING still measures ~2-4 us (record/timestamp cost); calibrate it and take it off every
        # launch, otherwise the ~700 decoder-sized launches of 10-20 us are over-counted against the rocprofv3 durations
        pairs = []
        for _ in range(64):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            b.record()
            pairs.append((a, b))
        torch.cuda.synchronize()
        empty = sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2]
        ms = sum(max(e0.elapsed_time(e1) - empty, 0.0) for _, e0, e1, _ in recs)
        launched = sum(f for f, _, _, _ in recs)
        by = {}
        for f, e0, e1, tag in recs:
            k = 'tile%d a%d b%d' % (tag[6], tag[4], tag[5])
            a = by.setdefault(k, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1)
            a[2] += f
        if args.dump_gemm:
            rows = sorted(([round(e0.elapsed_time(e1) * 1000, 1), round(f / max(e0.elapsed_time(e1), 1e-6) / 1e9, 2)] + list(tag)
                           for f, e0, e1, tag in recs), reverse=True)
            with open(args.dump_gemm, 'w') as fh:
                json.dump({'columns': ['us', 'launched_tflops', 'M', 'N', 'K', 'batch', 'a_mode', 'b_mode', 'tile', 'splitk'],
                           'rows': rows, 'by_variant': by}, fh)
        top = max(by.items(), key=lambda kv: kv[1][1])
        gflop_img = GFLOP_PER_IMG_FWD_BWD if args.encoder == 'resnet50' else 88.5     # 3 x 29.49 (HRNet-W32)
        if args.family != 'a':
            gflop_img = 39.8        # 3 x 13.27 GFLOP/img forward (torch flop counter on the oracle; the
            #                         MANO head of b-mano adds 0.01)
        achieved = gflop_img * B / ms            # GFLOP / ms = TFLOP/s
        split = (ops.ENGINE >= 1)
        by_engine = {}
        for f, e0, e1, tag in recs:
            # flash attention (tag[6] 30) and the row-chain kernel (40) run exact-fp32 MFMA but are their own families; the fp16
            # convolution of configs[4] carries tag[8] 'f16'
            key = 'flash_attention' if tag[6] == 30 else 'engine%s' % (tag[8],)
            a = by_engine.setdefault(key, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += max(e0.elapsed_time(e1) - empty, 0.0)
            a[2] += f
        # `peak` = the ceiling of the arithmetic that produced MOST of the family's time (round-5: the line used to divide by the
        # six-product ceiling 2500 / 6 whichever engine ran, which overstated engine 2's fraction twofold); the six-product
        # fraction of rounds 2-4 stays beside it for comparison across rounds
        ceilings = {'engine2': PEAK_BF16_MFMA_TF / 3.0, 'engine1': PEAK_BF16_MFMA_TF / 6.0, 'engine0': PEAK_FP32_MFMA_TF,
                    'flash_attention': PEAK_FP32_MFMA_TF}
        dominant = max(by_engine.items(), key=lambda kv: kv[1][1])[0] if by_engine else ('engine1' if split else 'engine0')
        peak = ceilings.get(dominant, PEAK_BF16_MFMA_TF / 6.0)
        ems = sum(max(e0.elapsed_time(e1) - empty, 0.0) for _, e0, e1, _ in erecs)
        ebytes = sum(b_ for b_, _, _, _ in erecs)
        roof_hbm = {'bound': 'hbm', 'achieved': round(ebytes / max(ems, 1e-9) / 1e6, 1), 'peak': 8000.0, 'unit': 'GB/s',
                    'frac': round(ebytes / max(ems, 1e-9) / 1e6 / 8000.0, 4), 'traffic': None,
                    'kernel': 'BatchNorm family (rih_bn_stats + rih_bn_apply, rih_bn_bwd: %d calls, %.2f ms and %.1f GB of '
                              'algorithmic traffic per step; each call = 2 launches)' % (len(erecs), ems, ebytes / 1e9)}
        roof = {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                'frac': round(achieved / peak, 4),
                'achieved_launched': round(launched / ms / 1e9, 2), 'frac_launched': round(launched / ms / 1e9 / peak, 4),
                'traffic': measured_traffic() if args.encoder == 'resnet50' and B == 64 and args.family == 'a' else None,
                'kernel': ('rih_gemm engine 2 where operand bounds exist (gemm_split_kernel<..., ENG 2>: fp32 = 3 x '
                           'v_mfma_f32_32x32x16_f16 on a scaled two-term fp16 split, ceiling 2500 / 3 = 833 TF/s), engine 1 elsewhere; '
                           '`peak` / `frac` = the ceiling of the engine with the most time; `frac_of_six_product_peak` is the figure '
                           'rounds 2-4 quoted'
                           if ops.ENGINE == 2 else
                           'rih_gemm engine 1 (gemm_split_kernel / gemm_split256_kernel: fp32 = 6 x '
                           'v_mfma_f32_32x32x16_bf16 on a 3-term bf16 split; peak = 2500 TF/s bf16 dense / 6)'
                           if split else 'rih_gemm engine 0 (gemm_kernel, v_mfma_f32_32x32x2_f32)'),
                'peak_is': 'ceiling of %s, the arithmetic with the most GEMM-family time of the step' % dominant,
                'frac_of_six_product_peak': round(achieved / (PEAK_BF16_MFMA_TF / 6.0), 4),
                'frac_of_three_product_peak': round(achieved / (PEAK_BF16_MFMA_TF / 3.0), 4),
                'by_engine': {k: {'launches': v[0], 'ms': round(v[1], 3), 'launched_tflops': round(v[2] / max(v[1], 1e-9) / 1e9, 1),
                                  'ceiling': round(ceilings.get(k, 0.0), 1),
                                  'frac_of_ceiling': (round(v[2] / max(v[1], 1e-9) / 1e9 / ceilings[k], 4) if k in ceilings else None)}
                              for k, v in sorted(by_engine.items())},
                'native_f32_mfma_peak': PEAK_FP32_MFMA_TF,
                'frac_of_native_f32_mfma_peak': round(achieved / PEAK_FP32_MFMA_TF, 4),
                'launches_per_step': len(recs), 'gemm_ms_per_step': round(ms, 3),
                'event_pair_overhead_us': round(1000.0 * empty, 2),
                'launched_tflops': round(launched / ms / 1e9, 2),
                'avg_launch_us': round(1000.0 * ms / max(len(recs), 1), 2),
                'top_variant': {'name': top[0], 'launches': top[1][0], 'ms': round(top[1][1], 3),
                                'tflops': round(top[1][2] / top[1][1] / 1e9, 2)}}

    if not args.no_roofline and rank == 0 and world == 1:
        try:
            roof_mano = mano_roofline(device)
        except Exception as exc:  # noqa: BLE001
            print('[bench] MANO micro-benchmark failed: %s' % exc, file=sys.stderr, flush=True)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(family=args.family) if args.family != 'b-mano' else None

    if rank == 0:
        n_img = B * world * args.steps
        line = {'metric': 'images/sec fwd+bwd @256x256 two-hand', 'value': round(n_img / elapsed, 2),
                'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': round(1000.0 * elapsed / args.steps, 3), 'ms_per_step_median_hip_events': round(ms_median, 3),
                'higher_is_better': True, 'scaling': 'weak',
                'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                'config': {'workload': ('second model family (common/myhand/lijun_model_graph.load_graph_model): batch=%d/GPU '
                                        '256x256 ResNet50 trunk + MLP-block dual-graph decoder, fwd + loss + bwd + Adam step, '
                                        'dropout 0.05, fp32' % B) if args.family == 'b' else
                           ('second model family with the MANO layer in the forward (lijun_model_newgraph.load_new_model): '
                            'batch=%d/GPU 256x256, fwd + loss + bwd + Adam step, dropout 0.05, fp32' % B)
                           if args.family == 'b-mano' else
                           ('BASELINE configs[1]: batch=%d/GPU 256x256 ResNet50 + cross-hand attention decoder, '
                                        'fwd + loss + bwd + Adam step, dropout 0.05, fp32' % B) if args.encoder == 'resnet50'
                           else ('BASELINE configs[3] model: batch=%d/GPU 256x256 HRNet-W32 + cross-hand attention decoder, '
                                 'fwd + loss + bwd + Adam step, dropout 0.05, fp32' % B),
                           'global_batch': B * world, 'parallelism': 'dp%d' % world, 'loss': round(final_loss, 4),
                           'hipgraph': bool(use_graph),
                           'backward_stages': (trainer.nstage if trainer is not None else 1),
                           'weight_gradients_on_side_stream': bool(trainer is not None and trainer.side_wgrad),
                           'gradient_buckets_MB': ([round(x / 1e6, 1) for x in trainer.bucket_bytes()] if trainer is not None else None),
                           'grad_allreduce': ('none (1 rank)' if not dist_on else 'torch DDP' if args.ddp else
                                              'per-stage buckets on a side stream, overlapped with the next backward stage'
                                              if not args.no_overlap else 'per-stage buckets on the compute stream'),
                           'optimizer': 'torch.optim.Adam(fused=True)' if args.torch_adam else 'renderih_amd.optim.Adam (rih_adam_multi, one launch)',
                           'flash_attention': bool(ops.FLASH_ATTN), 'one_kernel_attention_rih_fused_attn': bool(ops.FUSED_ATTN),
                           'gemm_engine': ('engine 2: fp32 via a scaled two-term fp16 split, 3 MFMA products, fp32 accumulate, on the '
                                           'convolutions (operand bounds from the BatchNorm kernels); engine 1 elsewhere'
                                           if ops.ENGINE == 2 else
                                           'fp32 via 3-term bf16 split, 6 MFMA products, fp32 accumulate (fp32-grade error)'
                                           if ops.ENGINE == 1 else 'native f32 MFMA')},
                'comm_ms_exposed': (None if comm_exposed is None else round(comm_exposed, 3)),
                'eager_reference_loop': eager_ref,
                'roofline': roof, 'roofline_hbm': roof_hbm, 'roofline_mano': roof_mano, 'cpu_baseline': cpu}
        emit(line)
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
