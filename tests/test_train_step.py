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


def _run(world):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   OMP_NUM_THREADS='2')
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
