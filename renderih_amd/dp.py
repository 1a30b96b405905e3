"""Data-parallel gradient exchange for the pose network: one bucket, one all-reduce per step.

The reference wraps the model in `DistributedDataParallel(find_unused_parameters=True)` (core/gcn_trainer.py:110-115) and
that still works with this package (tests/test_dp_gloo.py).  For throughput, `GradAllReducer` replaces it on the hot
path: DDP's per-iteration graph walk for unused parameters and its per-parameter hooks cost ~25 ms per step here
(measured on MI355X: 92 vs 67 ms at world size 1), while the whole gradient is only 156 MB -- one RCCL all-reduce over
xGMI takes ~1-2 ms on an 8-GPU node and needs no overlap with the 65 ms backward.

Semantics kept from the reference's DDP use: parameters are broadcast from rank 0 at construction; gradients are
averaged over ranks; parameters that received no gradient keep `grad = None` (they must be the same set
on every rank -- they are: the set is structural); BatchNorm statistics stay per rank (no SyncBN).
"""
import torch
import torch.distributed as dist


class GradAllReducer:
    def __init__(self, module, process_group=None, broadcast_parameters=True):
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev, dt = self.params[0].device, self.params[0].dtype
        offs, n = [], 0
        for p in self.params:
            if p.device != dev or p.dtype != dt:
                raise ValueError('GradAllReducer needs all parameters on one device with one dtype')
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4               # keep every slice 16-byte aligned
        self.flat = torch.zeros(n, device=dev, dtype=dt)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for o, p in zip(offs, self.params)]
        self.static = None
        if broadcast_parameters and self.world > 1:
            # the full module state like DDP's _sync_module_states: frozen parameters too (decoder.unsample_layer.weight,
            # core/gcn_trainer.py:102), so ranks that initialised or loaded differently cannot diverge silently
            every = list(module.parameters())
            with torch.no_grad():
                buf = torch.cat([p.detach().reshape(-1) for p in every])
                dist.broadcast(buf, 0, group=process_group)
                o = 0
                for p in every:
                    p.copy_(buf[o:o + p.numel()].view_as(p))
                    o += p.numel()
            for b in module.buffers():
                dist.broadcast(b, 0, group=process_group)

    def use_static_grads(self):
        """Call once after a training step has been captured in a hipGraph (torch.cuda.graph) and before the first
        `reduce()`: every replay rewrites the gradient tensors that exist NOW in place, so `reduce()` must keep
        reading those buffers -- not `.grad`, which it rebinds to views of the bucket."""
        self.static = [p.grad for p in self.params]

    @torch.no_grad()
    def reduce(self):
        """Call after backward() (or after a graph replay): averages every existing gradient over the ranks and leaves
        the result in `.grad` (a view into the bucket)."""
        src = self.static if self.static is not None else [p.grad for p in self.params]
        have = [i for i, g in enumerate(src) if g is not None]
        if len(have) != len(self.params):
            self.flat.zero_()                           # slots of grad-less parameters must contribute zeros
        torch._foreach_copy_([self.views[i] for i in have], [src[i] for i in have])
        if self.world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.mul_(1.0 / self.world)
        for i in have:
            self.params[i].grad = self.views[i]
        return len(have)
