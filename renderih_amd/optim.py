"""Adam / AdamW with the update of ALL parameter tensors in one HIP launch (csrc/rih_elem.hip: adam_multi_kernel).

Drop-in for the optimizer the reference's trainer builds (core/gcn_trainer.py:127 `torch.optim.Adam(optim_params, lr=...)`):
same constructor arguments, same update rule (amsgrad / maximize / capturable / differentiable are not supported and raise),
same `state_dict()` layout -- per parameter its OWN `step` (0-dim float tensor), `exp_avg`, `exp_avg_sq` -- so checkpoints move
between the two in both directions and either optimizer can keep stepping after the load (tests/test_gpu_ops.py,
tests/test_cpu_emulated.py).

Why: torch's fused Adam walks the 843 gradient-carrying tensors of the pose network in 24 multi-tensor launches of ~37 us
(0.89 ms per step on MI355X, profiles/r02); the update needs 4 reads + 3 writes per element = 1.0 GB, i.e. ~0.2 ms of HBM
time.  Here a device-resident table (pointer quadruple + length per tensor, block -> (tensor, chunk) map) drives one launch
per parameter group; the table is rebuilt only when a pointer changes (`.grad` rebound, state loaded).
"""
import torch

from . import _lib, ops


class _Group:
    """Launch plan (device tables) of one parameter group, valid for one list of (parameter, gradient) pointers."""

    def __init__(self):
        self.key = None
        self.plan = None


class Adam(torch.optim.Optimizer):
    _decoupled = False

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, **unsupported):
        for k, v in unsupported.items():
            if k in ('foreach', 'fused'):           # accepted and ignored: there is one implementation
                continue
            if v not in (False, None):
                raise ValueError('renderih_amd.optim.%s does not support %s=%r' % (type(self).__name__, k, v))
        if amsgrad:
            raise ValueError('amsgrad is not supported')
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError('invalid Adam hyper-parameters')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._tables = {}

    # ------------------------------------------------------------------ state
    def _init_state(self, params):
        """exp_avg / exp_avg_sq of the not yet initialised parameters as views of two flat zero buffers (one allocation,
        16-byte aligned views)."""
        new = [p for p in params if len(self.state[p]) == 0]
        if not new:
            return
        offs, n = [], 0
        for p in new:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        dev = new[0].device
        m = torch.zeros(max(n, 4), device=dev, dtype=torch.float32)
        v = torch.zeros(max(n, 4), device=dev, dtype=torch.float32)
        for p, o in zip(new, offs):
            st = self.state[p]
            # a distinct 0-dim counter per parameter, as torch.optim.Adam keeps it: a state_dict() taken here and resumed in
            # torch's own Adam advances every tensor once per parameter, so a shared object would run N times too fast there
            st['step'] = torch.zeros((), dtype=torch.float32)
            st['exp_avg'] = m[o:o + p.numel()].view_as(p)
            st['exp_avg_sq'] = v[o:o + p.numel()].view_as(p)

    def _plan(self, gi, params):
        """Validate, create state, and build the launch plan of a group for the current set of tensors: a list of
        (step counters, device table, block maps, nblocks) -- one entry per distinct step count (one, unless a checkpoint
        was loaded or parameters joined later: a launch has one bias correction)."""
        ops._chk(*params)                   # HIP kernel: GPU tensors only (the CPU test emulation patches this check)
        for p in params:
            if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                raise RuntimeError('renderih_amd.optim: fp32 dense parameters and gradients only')
            if not p.is_contiguous() or not p.grad.is_contiguous():
                raise RuntimeError('renderih_amd.optim: parameters and gradients must be contiguous')
        self._init_state(params)
        for p in params:
            st = self.state[p]
            if not st['exp_avg'].is_contiguous() or not st['exp_avg_sq'].is_contiguous():      # e.g. a loaded checkpoint
                st['exp_avg'], st['exp_avg_sq'] = st['exp_avg'].contiguous(), st['exp_avg_sq'].contiguous()
        by_step, seen = {}, {}
        for p in params:
            st = self.state[p]
            c = st['step']
            if not torch.is_tensor(c):              # checkpoints of old torch versions keep a Python number
                c = st['step'] = torch.tensor(float(c), dtype=torch.float32)
            elif c.device.type != 'cpu' or c.dim() != 0:
                c = st['step'] = c.detach().to('cpu', torch.float32).reshape(())
            if id(c) in seen:                       # a checkpoint written with ONE counter object for many parameters
                c = st['step'] = c.clone()
            seen[id(c)] = True
            e = by_step.setdefault(int(c), ([], []))
            e[0].append(c)
            e[1].append(p)
        chunk = int(_lib.load().rih_adam_chunk())
        dev = params[0].device
        plan = []
        for _, (counters, sub) in sorted(by_step.items()):
            rows, bt, bc = [], [], []
            for i, p in enumerate(sub):
                st = self.state[p]
                rows.append([p.data_ptr(), p.grad.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), p.numel()])
                nc = (p.numel() + chunk - 1) // chunk
                bt.extend([i] * nc)
                bc.extend(range(nc))
            plan.append((counters, torch.tensor(rows, dtype=torch.int64).to(dev), torch.tensor(bt, dtype=torch.int32).to(dev),
                         torch.tensor(bc, dtype=torch.int32).to(dev), len(bt)))
        return plan

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables.clear()            # the state tensors were replaced

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            # signature of the group's tensors: the plan (device tables) is rebuilt only when a pointer changed
            params, sig = [], []
            for p in group['params']:
                g = p.grad
                if g is not None:
                    params.append(p)
                    sig.append(p.data_ptr())
                    sig.append(g.data_ptr())
            if not params:
                continue
            t = self._tables.setdefault(gi, _Group())
            if t.key != sig:
                t.plan = self._plan(gi, params)
                t.key = sig
            b1, b2 = group['betas']
            for counters, table, blk_tensor, blk_chunk, nblocks in t.plan:
                s = int(counters[0])
                ops.check(lib.rih_adam_multi(table.data_ptr(), blk_tensor.data_ptr(), blk_chunk.data_ptr(), nblocks,
                                             float(group['lr']), float(b1), float(b2), float(group['eps']),
                                             float(group['weight_decay']), s + 1, 1 if self._decoupled else 0,
                                             ops._stream()), 'rih_adam_multi')
                torch._foreach_add_(counters, 1)     # every parameter's own counter, one host call
            if len(t.plan) > 1:
                t.key = None            # distinct step counts: regroup next time (they may have converged or been reloaded)
        # the kernel rewrote the parameters through raw pointers: torch's version counters did not move, so every operand bound
        # cached on a parameter (engine 2, ops.bound_of) is stale from here on
        ops.bounds_invalidate()
        return loss


class AdamW(Adam):
    """torch.optim.AdamW semantics (decoupled weight decay: p *= 1 - lr * wd before the update)."""
    _decoupled = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, **kw):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kw)
