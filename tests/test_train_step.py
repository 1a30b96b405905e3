"""renderih_amd.train.TrainStep on CPU (C ABI emulated): staged backward == plain backward, bucket order, gloo world 2."""
import os
import socket
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, encoder='resnet50'):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   OMP_NUM_THREADS='2', RIH_TEST_ENCODER=encoder)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'train_step_worker.py')], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=1200)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, 'rank %d failed:\n%s' % (rank, out[-3000:])
        assert ('rank %d ok' % rank) in out, out[-2000:]


def test_staged_backward_and_bucket_order_world2():
    _run(2)


def test_hrnet_staged_backward_and_bucket_order_world2():
    """HRNet-W32 (BASELINE configs[3]): four backward stages -- after the trunk, stage 4, stage 3, stage 2 + stem -- each followed by
    its bucket's all-reduce before the next stage is issued (round-5 verdict item 4 i: one 202 MB bucket until round 6), gradients
    equal to a plain backward, averaged over two gloo ranks."""
    _run(2, 'hrnet32')


def test_shared_parameters_turn_the_batched_reductions_off():
    """ops.deferred_reductions is only valid when nothing reads a weight gradient before the stage ends.  A parameter used
    twice in the forward pass breaks that (autograd sums the two contributions during the backward pass): TrainStep detects
    it from the autograd graph of its first step, warns, falls back to immediate reductions -- and the gradients are right."""
    import sys
    import warnings
    import torch
    sys.path.insert(0, HERE)
    from abi_emulator import emulated_abi
    from renderih_amd import ops
    from renderih_amd.train import TrainStep

    class Shared(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc = torch.nn.Linear(24, 24)
            self.norm = torch.nn.LayerNorm(24, eps=1e-6)
            self.out = torch.nn.Linear(24, 8)

        def forward(self, x):
            h = ops.linear(x, self.fc.weight, self.fc.bias, relu=True)
            h = ops.layernorm(ops.linear(h, self.fc.weight, self.fc.bias), self.norm.weight, self.norm.bias)   # fc twice
            return ops.linear(h, self.out.weight, self.out.bias)

    torch.manual_seed(0)
    with emulated_abi():
        m = Shared()
        x = torch.randn(40, 24)
        (m(x) ** 2).sum().backward()
        want = {k: p.grad.clone() for k, p in m.named_parameters()}
        m.zero_grad(set_to_none=True)
        opt = torch.optim.SGD(m.parameters(), lr=0.0)
        step = TrainStep(m, opt, lambda out, lab: (out ** 2).sum(), (x, {}), process_group=False, use_graph=False)
        assert step.defer_reduce and step.nstage == 1
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            step(x, {})
        assert not step.defer_reduce and any('used more than once' in str(i.message) for i in w)
        for k, p in m.named_parameters():
            assert torch.allclose(p.grad, want[k], rtol=1e-5, atol=1e-6), k


def test_train_step_with_the_one_launch_adam_tracks_torch():
    """TrainStep (deferred reductions on) + renderih_amd.optim.Adam over three steps == plain backward + torch.optim.Adam on a
    copy of the module: same parameters afterwards (fp32 round-off), same optimizer state layout."""
    import copy
    import sys
    import torch
    sys.path.insert(0, HERE)
    from abi_emulator import emulated_abi
    from renderih_amd import ops, optim
    from renderih_amd.train import TrainStep

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(24, 40)
            self.n = torch.nn.LayerNorm(40, eps=1e-6)
            self.b = torch.nn.Linear(40, 8)

        def forward(self, x):
            h = ops.layernorm(ops.linear(x, self.a.weight, self.a.bias, relu=True), self.n.weight, self.n.bias)
            return ops.linear(h, self.b.weight, self.b.bias)

    torch.manual_seed(1)
    with emulated_abi():
        m = Net()
        ref = copy.deepcopy(m)
        x = torch.randn(64, 24)
        kw = dict(lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2)
        o_ref = torch.optim.Adam(ref.parameters(), **kw)
        step = TrainStep(m, optim.Adam(m.parameters(), **kw), lambda out, lab: (out ** 2).mean(), (x, {}),
                         process_group=False, use_graph=False)
        for _ in range(3):
            step(x, {})
            o_ref.zero_grad(set_to_none=True)
            (ref(x) ** 2).mean().backward()
            o_ref.step()
        assert step.defer_reduce
        for (k, p), q in zip(m.named_parameters(), ref.parameters()):
            assert torch.allclose(p, q, rtol=2e-5, atol=2e-6), k
        st = step.opt.state[m.a.weight]
        assert int(st['step']) == 3 and st['exp_avg'].shape == m.a.weight.shape
