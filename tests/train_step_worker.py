"""One rank of the world_size-2 check of renderih_amd.train.TrainStep (launched by tests/test_train_step.py; gloo, CPU, the
C ABI emulated on host memory).  Pins: the three backward stages reproduce the gradients of a plain `loss.backward()`;
the buckets are all-reduced in reverse-autograd order, each right after its stage and before the next stage's backward is
issued (the overlap schedule of SURVEY 8e / core/gcn_trainer.py:110-115's DDP); gradients are averaged over the ranks; the
structurally grad-less parameters are in no bucket; parameters stay identical on every rank after the step."""
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
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from abi_emulator import emulated_abi
    from oracle.net_oracle import scalar_loss
    from renderih_amd import testing
    from renderih_amd.model import build_model
    from renderih_amd.train import TrainStep
    enc = os.environ.get('RIH_TEST_ENCODER', 'resnet50')
    nstage = 4 if enc.startswith('hrnet') else 3        # HRNet (round 6): cuts behind stage 2, stage 3 and the trunk
    with emulated_abi():
        m = build_model(0.0, enc)
        m.load_state_dict(testing.deterministic_state(m.state_dict(), seed=2 + rank))      # ranks start DIFFERENT
        m.decoder.unsample_layer.weight.requires_grad_(False)          # core/gcn_trainer.py:102-103
        m.train()
        img = testing.seeded_image(1, 20 + rank)                       # each rank its own shard
        opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=1e-3)
        order = []
        step = TrainStep(m, opt, lambda out, lab: scalar_loss(out), (img, {}), use_graph=False, record_order=order)
        # after construction every rank holds rank 0's parameters (frozen ones included)
        if world > 1:
            flat = torch.cat([p.detach().flatten() for p in m.parameters()])
            ref = flat.clone()
            dist.broadcast(ref, 0)
            assert torch.equal(flat, ref), 'parameters not broadcast'
        # reference: plain backward on the same module state (BatchNorm statistics are per rank, as in the reference)
        m.zero_grad(set_to_none=True)
        scalar_loss(m(img)).backward()
        local = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
        m.zero_grad(set_to_none=True)
        before = {k: p.detach().clone() for k, p in m.named_parameters()}
        loss = step(img, {})
        assert torch.isfinite(loss)
        assert step.defer_reduce, 'the batched reductions were switched off: the shared-parameter guard misfired'
        # first call learns the live set, then reduces in stage order; the second call interleaves
        order.clear()
        for k, p in m.named_parameters():
            p.data.copy_(before[k])
        step(img, {})
        assert order == [(w, i) for i in range(nstage) for w in ('stage', 'reduce')], order
        n_live = 0
        worst = 0.0
        # (a gradient that is mathematically zero -- the key bias of a softmax attention -- is pure round-off, whose value
        # depends on the split-K partition of the weight-gradient GEMMs: errors are measured against the largest gradient too)
        gmax = max(float(v.abs().max()) for v in local.values())
        for k, p in m.named_parameters():
            if k in local:
                want = local[k].clone()
                if world > 1:
                    dist.all_reduce(want)
                    want /= world
                assert p.grad is not None, k
                err = float((p.grad - want).abs().max() / (want.abs().max() + 1e-6 * gmax))
                if not testing.is_null_gradient(k):      # (mathematically zero gradients are round-off only, see above)
                    worst = max(worst, err)
                    assert err < 2e-5, (k, err)
                n_live += 1
            else:
                assert p.grad is None, k
        sizes = step.bucket_bytes()
        assert len(sizes) == nstage and sum(len(v) for v in step.live) == n_live and all(b > 0 for b in sizes)
        if world > 1:
            flat = torch.cat([p.detach().flatten() for p in m.parameters()])
            ref = flat.clone()
            dist.broadcast(ref, 0)
            assert torch.equal(flat, ref), 'parameters diverged after the step'
        # ZeRO-1 as the reference's trainer builds it for distributed runs (core/gcn_trainer.py:121-125: torch's
        # ZeroRedundancyOptimizer shards the optimiser state over the ranks): TrainStep only calls optimizer.step(), so the
        # sharded optimiser drops in -- same parameters on every rank afterwards, and the same update as the plain optimiser
        if world > 1 and nstage == 3:
            from torch.distributed.optim import ZeroRedundancyOptimizer
            after_plain = {k: p.detach().clone() for k, p in m.named_parameters()}
            for k, p in m.named_parameters():
                p.data.copy_(before[k])
            zopt = ZeroRedundancyOptimizer([p for p in m.parameters() if p.requires_grad], optimizer_class=torch.optim.SGD,
                                           lr=1e-3)
            zstep = TrainStep(m, zopt, lambda out, lab: scalar_loss(out), (img, {}), use_graph=False)
            zstep(img, {})      # (the first call of a TrainStep learns the live set, then steps like every later call)
            for k, p in m.named_parameters():
                p.data.copy_(before[k])
            zstep(img, {})      # plain SGD keeps no state: one step from `before` must land on `after_plain`
            flat = torch.cat([p.detach().flatten() for p in m.parameters()])
            ref = flat.clone()
            dist.broadcast(ref, 0)
            assert torch.equal(flat, ref), 'ZeRO: parameters differ between ranks'
            worst_z = max(float((p.detach() - after_plain[k]).abs().max() / (after_plain[k].abs().max() + 1e-30))
                          for k, p in m.named_parameters())
            assert worst_z < 1e-6, worst_z
        # (that the model still works outside the helper -- the trunk hook is inert there -- is checked on the GPU:
        # tests/test_gpu_model.py::test_train_step_staged_graph_replay_matches_plain_backward)
    print('rank %d ok: %d live tensors in buckets of %s bytes, worst rel err %.1e' % (rank, n_live, sizes, worst))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
