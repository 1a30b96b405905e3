"""One rank of the world_size-2 data-parallel check (launched by tests/test_dp_gloo.py; gloo backend, CPU).

The HIP kernels cannot run here, so the C ABI is emulated on host memory (tests/abi_emulator.py, test infrastructure);
what this pins is the N>1 path of bench.py / the reference trainer (core/gcn_trainer.py:110-115): the module tree under
DistributedDataParallel(find_unused_parameters=True), gradient averaging over ranks, tolerance of the grad-less
parameters, and identical parameters after one optimizer step on every rank."""
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from abi_emulator import emulated_abi
    from oracle.net_oracle import scalar_loss
    from renderih_amd import testing
    from renderih_amd.model import build_model
    with emulated_abi():
        m = build_model(0.0)
        m.load_state_dict(testing.deterministic_state(m.state_dict(), seed=2))
        m.decoder.unsample_layer.weight.requires_grad_(False)          # core/gcn_trainer.py:102-103
        m.train()
        img = testing.seeded_image(1, 20 + rank)                       # each rank its own shard (DistributedSampler)
        # local gradients without the wrapper (BatchNorm statistics are per rank, as in the reference)
        scalar_loss(m(img)).backward()
        local = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
        nograd = sorted(k for k, p in m.named_parameters() if p.grad is None)
        # (a) the package's own reducer (renderih_amd/dp.py, what bench.py uses for N > 1)
        from renderih_amd.dp import GradAllReducer
        red = GradAllReducer(m)
        assert red.reduce() == len(local)
        for k, p in m.named_parameters():
            if k in local:
                want = local[k].clone()
                dist.all_reduce(want)
                want /= world
                assert float((p.grad - want).abs().max()) <= 1e-6 * float(want.abs().max() + 1e-30), k
            else:
                assert p.grad is None, k
        # (a') replayed steps (hipGraph): the graph rewrites fixed gradient buffers in place while `.grad` already points
        # into the bucket -- the reducer must keep reading the buffers (bench.py: use_static_grads after capture)
        m.zero_grad(set_to_none=True)
        red2 = GradAllReducer(m, broadcast_parameters=False)
        static = {}
        for k, p in m.named_parameters():
            if k in local:
                static[k] = local[k].clone()
                p.grad = static[k]
        red2.use_static_grads()
        for scale in (1.0, -3.0):                                      # "replay": new values land in the same buffers
            for k in static:
                static[k].copy_(local[k] * scale)
            assert red2.reduce() == len(local)
            for k, p in m.named_parameters():
                if k in local:
                    want = local[k].clone() * scale
                    dist.all_reduce(want)
                    want /= world
                    assert float((p.grad - want).abs().max()) <= 1e-6 * float(want.abs().max() + 1e-30), (k, scale)
        # (b) torch DDP as the reference trainer wraps the model
        m.zero_grad(set_to_none=True)
        for mod in m.modules():                                        # undo the running-stat update of the first pass
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.num_batches_tracked.zero_()
        m.load_state_dict(testing.deterministic_state(m.state_dict(), seed=2))
        ddp = torch.nn.parallel.DistributedDataParallel(m, find_unused_parameters=True)
        opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=1e-3)
        scalar_loss(ddp(img)).backward()
        worst = 0.0
        for k, p in m.named_parameters():
            if k in local:
                want = local[k].clone()
                dist.all_reduce(want)
                want /= world
                assert p.grad is not None, k
                err = float((p.grad - want).abs().max() / (want.abs().max() + 1e-30))
                worst = max(worst, err)
                assert err < 1e-5, (k, err)
            else:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        opt.step()
        # parameters stay identical on every rank after the step
        flat = torch.cat([p.detach().flatten() for p in m.parameters()])
        ref = flat.clone()
        dist.broadcast(ref, 0)
        assert torch.equal(flat, ref)
    print('rank %d ok: %d tensors reduced (worst rel err %.1e), %d grad-less' % (rank, len(local), worst, len(nograd)))
    # 53 structurally unused under this loss (SURVEY N4 lists 63 for the trainer's loss, which ignores the aux
    # hms/mask/dense heads; scalar_loss touches them) + the frozen unsample_layer.weight
    assert len(nograd) == 54, len(nograd)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
