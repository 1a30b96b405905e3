"""Minimal reproduction attempts for the crash of a captured step with more than one side stream (renderih_amd/streams.py).

Each case runs in its own process (`python -X faulthandler tools/capture_fork_min.py CASE`); with no argument all cases are run
one after the other and a one-line verdict per case is printed.  Cases, from the least to the most of the real step:
  torch2 / torch3   plain torch.mm chains on 2 / 3 side streams, forward only, forked and joined inside torch.cuda.graph
  torchbw2 / 3      the same through renderih_amd.streams.fork_join with autograd: forward AND backward inside the capture
  rih2 / rih3       renderih_amd.ops convolutions + BatchNorm on the side streams, forward and backward inside the capture
  rihstage2 / 3     as rih*, but the backward captured in a SECOND graph (what train.TrainStep does: forward graph, then one
                    graph per backward stage over the autograd graph the first capture built)
  a2a2 / a2a3       TWO fork_joins, the second one's thunks read EVERY output of the first (HRNet's fuse rows): in the backward
                    a gradient produced on one side stream is consumed on another (side -> side event waits)
  a2afwd2 / 3       the same, forward only
  hrnet2 / hrnet3   the real HRNet-W32 encoder (renderih_amd.hrnet), forward + backward in one capture"""
import os
import subprocess
import sys

CASES = ['torchbw2', 'rih3', 'rihstage3', 'a2afwd2', 'a2afwd3', 'a2a2', 'a2a3', 'hrnet2', 'hrnet3']


def run(case):
    import torch
    n = int(case[-1])
    os.environ['RIH_SIDE_STREAMS'] = str(n)
    os.environ['RIH_SIDE_CAPTURE_MAX'] = str(n)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    from renderih_amd import streams
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    kind = case[:-1]
    branches = n + 1
    if kind == 'torch':
        xs = [torch.randn(512, 512, device=dev) for _ in range(branches)]
        w = torch.randn(512, 512, device=dev) * 0.04
        side = [torch.cuda.Stream() for _ in range(n)]
        out = [None] * branches
        g = torch.cuda.CUDAGraph()
        s0 = torch.cuda.Stream()
        s0.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s0):
            with torch.cuda.graph(g, stream=s0):
                for s in side:
                    s.wait_stream(s0)
                out[0] = (xs[0] @ w) @ w
                for k, s in enumerate(side):
                    with torch.cuda.stream(s):
                        out[k + 1] = (xs[k + 1] @ w) @ w
                for s in side:
                    s0.wait_stream(s)
                y = sum(out)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        print('ok', float(y.abs().sum()))
        return
    if kind == 'hrnet':
        from renderih_amd.hrnet import HighResolutionNet
        net = HighResolutionNet('w32', in_channels=4).to(dev).train()
        ximg = torch.randn(4, 256, 256, 4, device=dev)
        ws, gam, bet = list(net.parameters()), [], []

        def step():
            loss = sum(y.square().mean() for y in net(ximg))
            loss.backward()
            return loss
    elif kind in ('a2a', 'a2afwd'):
        ws = [torch.randn(256, 256, device=dev, requires_grad=True) for _ in range(2 * branches)]
        gam, bet = [], []
        x = torch.randn(256, 256, device=dev)

        def forward():
            ys = streams.fork_join([(lambda k=k: torch.relu(x @ ws[k])) for k in range(branches)],
                                   reads=[[x, ws[k]] for k in range(branches)])

            def row(i):
                acc = ys[i] @ ws[branches + i]
                for j in range(branches):
                    if j != i:
                        acc = acc + ys[j] * 0.5
                return acc
            rows = streams.fork_join([(lambda i=i: row(i)) for i in range(branches)], reads=[ys] * branches)
            return sum(r.square().mean() for r in rows)

        def step():
            loss = forward()
            if kind == 'a2a':
                loss.backward()
            return loss
    elif kind == 'torchbw':
        ws = [torch.randn(512, 512, device=dev, requires_grad=True) for _ in range(branches)]
        x = torch.randn(512, 512, device=dev)

        def step():
            outs = streams.fork_join([(lambda k=k: torch.relu(x @ ws[k]) @ ws[k]) for k in range(branches)],
                                     reads=[[x, ws[k]] for k in range(branches)])
            loss = sum(o.square().mean() for o in outs)
            loss.backward()
            return loss
    else:
        from renderih_amd import ops
        import torch.nn.functional as F
        C = 32
        ws = [torch.randn(C, C, 3, 3, device=dev, requires_grad=True) for _ in range(branches)]
        gam = [torch.ones(C, device=dev, requires_grad=True) for _ in range(branches)]
        bet = [torch.zeros(C, device=dev, requires_grad=True) for _ in range(branches)]
        x = torch.randn(8, 32 >> 0, 32, C, device=dev)          # channels-last activations as the package keeps them
        xs = [x[:, ::(1 << k), ::(1 << k)].contiguous() for k in range(branches)]

        def branch(k):
            y = ops.conv2d(xs[k], ws[k], stride=1, pad=1)
            rm = torch.zeros(C, device=dev)
            rv = torch.ones(C, device=dev)
            y = ops.batchnorm(y, gam[k], bet[k], rm, rv, training=True, relu=True)
            return ops.conv2d(y, ws[k], stride=1, pad=1)

        def forward():
            outs = streams.fork_join([(lambda k=k: branch(k)) for k in range(branches)], reads=[[xs[k]] for k in range(branches)])
            return sum(o.square().mean() for o in outs)

        def step():
            loss = forward()
            loss.backward()
            return loss
    # warm-up outside capture on a side stream (allocator, lazy inits), as torch's graph recipe asks
    s0 = torch.cuda.Stream()
    s0.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s0):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(s0)
    torch.cuda.synchronize()
    params = [p for p in (ws + (gam + bet if kind.startswith('rih') else []))]
    shown = params[:branches] if kind != 'a2afwd' else []
    for p in params:
        p.grad = None
    if kind == 'rihstage':
        pool = torch.cuda.graph_pool_handle()
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.stream(s0):
            with torch.cuda.graph(g1, pool=pool, stream=s0):
                loss = forward()
            streams.assert_joined()
            with torch.cuda.graph(g2, pool=pool, stream=s0):
                loss.backward()
            streams.assert_joined()
        for _ in range(3):
            g1.replay()
            g2.replay()
    else:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s0):
            with torch.cuda.graph(g, stream=s0):
                loss = step()
            streams.assert_joined()
        for _ in range(3):
            g.replay()
    torch.cuda.synchronize()
    print('ok', float(loss), [float(p.grad.abs().sum()) for p in shown])


if __name__ == '__main__':
    if len(sys.argv) > 1:
        run(sys.argv[1])
        sys.exit(0)
    for case in CASES:
        p = subprocess.run(['timeout', '120', sys.executable, '-X', 'faulthandler', os.path.abspath(__file__), case],
                           capture_output=True, text=True)
        tail = (p.stdout.strip().splitlines() or [''])[-1]
        print('== %-10s rc %4d  %s' % (case, p.returncode, tail[:160]), flush=True)
        if p.returncode != 0:
            print(p.stderr[-3000:], flush=True)
