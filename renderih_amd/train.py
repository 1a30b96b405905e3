"""One training step of the pose network as the reference's trainer runs it (core/gcn_trainer.py:219-251: forward,
`calc_loss_GCN`, backward, optimizer step; DDP gradient averaging at :110-115) -- packaged so that the reference's loop gets
the measured speed with a three-line change:

    step = TrainStep(model, optimizer, loss_fn, example_batch)      # once, after the model is on the GPU
    loss = step(img, labels)                                        # instead of forward / loss / backward / opt.step

What it does that the plain loop does not:

* **Staged backward.**  The backward pass is cut at the trunk's feature maps into three stages in reverse-autograd order --
  (1) mesh decoder + mid convs + aux decoders, (2) ResNet layer4 + layer3, (3) layer2 + layer1 + stem -- with
  `torch.autograd.grad` on the stage's boundary tensors, so that every stage ends with a COMPLETE set of parameter
  gradients (a bucket).
* **Bucketed gradient all-reduce overlapped with backward** (data parallel, `world > 1`): as soon as a stage's kernels are
  enqueued, its bucket (one flat buffer, gradients copied in by one multi-tensor launch) is all-reduced with RCCL on a
  side stream while the next stage's backward runs on the compute stream; the optimizer waits for the last bucket only.
  Parameters that never receive a gradient (63 tensors incl. every `GCN_ResBlock.norm1`) are excluded from the
  buckets statically -- no per-iteration graph walk (`find_unused_parameters`), no per-parameter hooks.
* **Batched split-K reductions.**  The partial slabs of a stage's weight gradients are summed by one launch per 60
  gradients at the end of the stage (`ops.deferred_reductions`, `rih_splitk_reduce_multi`) instead of 157 small launches per
  step, each of which costs a dependent-kernel slot (>= 4.5 us on this platform) whatever its size.
* **One weight-packing launch.**  The forward and data-gradient operand layouts of every k > 1 convolution are produced by one
  `rih_pack_conv_weight_multi` launch at the start of the step into persistent buffers (`ops.PackCache`) instead of 58 small
  launches in front of their GEMMs.
* **hipGraph replay.**  Forward + loss + stage-1 backward, stage 2 and stage 3 are captured as three graphs sharing one
  memory pool and replayed back to back; the collectives stay eager between the replays (no dependence on RCCL's capture
  support), so the host cost per step is three graph launches + three collectives + one fused optimizer launch instead of
  ~2700 kernel launches.  Dropout masks stay fresh through the device-resident seed word (`ops.DROPOUT_SEED_TENSOR`).

Stages need a trunk that `stage_plan` knows: the ResNet trunk (`model.encoder.resnet`: three stages, cut at its four feature maps and
at layer2's output) or HRNet (`model.encoder.hrnet`, round 6: four stages, cut at the trunk's branch maps and behind its stage 3 and
stage 2 -- the reference's DDP buckets ANY model at 25 MB, core/gcn_trainer.py:110-115; until round 6 HRNet-W32 was one stage and one
202 MB bucket whose all-reduce could overlap nothing); other encoders run as one stage (still bucketed + graphed).
Host-side bookkeeping of the forward that is not a kernel does not happen on a replay: `BatchNorm2d.num_batches_tracked` stays
at its capture-time value (it only matters for `momentum=None`, which the reference never uses); running statistics are
updated by the kernels and are correct.
`comm_ms_exposed()` reports the time the compute stream waited for the last bucket after its own work was done.
"""
import torch

from . import ops, streams


def _trunk(model):
    enc = getattr(model, 'encoder', None)
    return getattr(enc, 'resnet', None)


def _hr_trunk(model):
    enc = getattr(model, 'encoder', None)
    t = getattr(enc, 'hrnet', None)
    return t if (t is not None and all(hasattr(t, n) for n in ('stage2', 'stage3', 'stage4', 'transition3'))) else None


def hrnet_cut_modules(model):
    """HRNet: the modules whose outputs cut the backward pass, in FORWARD order -- the last HighResolutionModule of stage 2 and of
    stage 3 (each returns the list of branch maps that everything downstream is computed from) and the trunk itself."""
    t = _hr_trunk(model)
    return None if t is None else [t.stage2[-1], t.stage3[-1], t]


def stage_parameter_groups(model):
    """Trainable parameters in reverse-autograd order: ResNet [after-trunk, layer4+layer3, layer2+layer1+stem]; HRNet
    [after-trunk, stage4+transition3, stage3+transition2, stage2+transition1+layer1+stem]; anything else one group."""
    trunk = _trunk(model)
    allp = [p for p in model.parameters() if p.requires_grad]
    hr = _hr_trunk(model)
    if hr is not None:
        s4 = [p for m in (hr.stage4, hr.transition3) for p in m.parameters() if p.requires_grad]
        s3 = [p for m in (hr.stage3, hr.transition2) for p in m.parameters() if p.requires_grad]
        ids = {id(p) for p in s4 + s3}
        early = [p for p in hr.parameters() if p.requires_grad and id(p) not in ids]
        ids |= {id(p) for p in early}
        rest = [p for p in allp if id(p) not in ids]
        return [rest, s4, s3, early]
    if trunk is None or not all(hasattr(trunk, n) for n in ('layer1', 'layer2', 'layer3', 'layer4')):
        return [allp]
    late = [p for m in (trunk.layer4, trunk.layer3) for p in m.parameters() if p.requires_grad]
    early = [p for m in (trunk.layer2, trunk.layer1, trunk.conv1, trunk.bn1) for p in m.parameters() if p.requires_grad]
    ids = {id(p) for p in late + early}
    rest = [p for p in allp if id(p) not in ids]
    return [rest, late, early]


class TrainStep:
    def __init__(self, model, optimizer, loss_fn, example_batch, process_group=None, use_graph=True, stages=True,
                 overlap=True, record_order=None, force_exchange=False, side_wgrad=None, defer_reduce=None):
        """model: HandNET_GCN (train mode, on its device).  optimizer: any torch optimizer over model's parameters.
        loss_fn(outputs, labels) -> scalar loss.  example_batch = (img, labels): tensors with the shapes / dtypes of
        every later call (static buffers of the graphs).  process_group: None = default group if torch.distributed is
        initialised, False = no exchange.  record_order: optional list that receives ('stage', i) / ('reduce', i) in
        issue order (tests).  force_exchange: run the bucket copies and collectives even at world size 1 (exercises the
        N > 1 path on one GPU).  stages: True = three backward stages, False = one, 'auto' = three only when gradients are
        exchanged (bench.py's choice: one stage is +1.2 % on a single rank, same-box A/B in profiles/r03/ab/)."""
        import torch.distributed as dist
        self.model, self.opt, self.loss_fn = model, optimizer, loss_fn
        self.order = record_order
        self.group = None
        self.world = 1
        if process_group is not False and dist.is_available() and dist.is_initialized():
            self.group = process_group
            self.world = dist.get_world_size(process_group)
        import os
        self.side_wgrad = (os.environ.get('RIH_SIDE_WGRAD', '0') == '1') if side_wgrad is None else bool(side_wgrad)
        self.defer_reduce = (os.environ.get('RIH_DEFER_REDUCE', '1') == '1') if defer_reduce is None else bool(defer_reduce)
        self.packs = ops.PackCache() if os.environ.get('RIH_PACK_CACHE', '1') == '1' else None
        self.exchange = self.world > 1 or (force_exchange and dist.is_available() and dist.is_initialized())
        if stages == 'auto':                        # the stages exist to overlap the exchange; without one a single stage
            stages = self.exchange                  # groups every weight gradient of the step into 4 launches instead of 12
        self.groups = stage_parameter_groups(model) if stages else [[p for p in model.parameters() if p.requires_grad]]
        self.nstage = len(self.groups)
        self.overlap = overlap and self.exchange
        # upper bound on streams.SIDE inside this object's forward / backward (streams.limit).  None = no extra limit, also in a
        # process that exchanges gradients: rounds 3-4 switched the model's own side streams off there (a captured step with more
        # than one concurrent branch died inside hipStreamEndCapture), which cost HRNet-W32 its three side streams under data
        # parallelism (627 against 774 images/s on one rank).  The crash was root-caused in round 4 -- side streams that wait on each
        # other's events inside a capture -- and is avoided by streams._Hop (every cross-stream gradient travels through the
        # origin stream); the all-reduce itself is never captured (it runs eagerly on `self.side` between the stage replays), so
        # it adds no branch to any capture.  RIH_DP_SIDE_LIMIT=n restores a clamp.
        lim = os.environ.get('RIH_DP_SIDE_LIMIT', '')
        self.side_limit = int(lim) if (lim != '' and self.overlap and self.img_is_cuda(example_batch)) else None
        self.img, self.labels = example_batch
        dev = self.img.device
        self.cuda = dev.type == 'cuda'
        self._bounds = self._cut = None
        self._cutting = False
        self._chain = None          # HRNet: the cut modules in forward order; _att / _det = their attached / detached outputs
        self._att = self._det = None
        if self.nstage == 3:
            self._hook = _trunk(model).register_forward_hook(self._grab)
        elif self.nstage == 4:
            self._chain = hrnet_cut_modules(model)
            self._hooks = [m.register_forward_hook(lambda mod, inp, out, k=k: self._grab_chain(k, out))
                           for k, m in enumerate(self._chain)]
        self.live = None            # per stage: indices of the parameters that receive a gradient
        self.flat = None            # per stage: flat bucket
        self.views = None
        self.static = None          # per stage: gradient tensors rewritten by every replay
        self.graphs = None
        self._e_bwd = self._e_comm = None
        self._exposed = []
        self.side = torch.cuda.Stream(device=dev) if self.cuda else None
        if self.exchange:
            self._broadcast_state()
        self.use_graph = bool(use_graph and self.cuda)
        if self.use_graph:
            self._capture()

    # ------------------------------------------------------------------ pieces
    @staticmethod
    def img_is_cuda(example_batch):
        return bool(getattr(example_batch[0], 'is_cuda', False))

    def _grab(self, module, inputs, output):
        """Forward hook on the trunk, active only inside this helper's own forward: hands DETACHED aliases of the four
        feature maps to the rest of the network, so that stage 1 yields only the gradients that enter the trunk from
        outside (asking autograd for d loss / d x of the attached tensors would run the whole trunk's backward)."""
        if not self._cutting:
            return None
        self._bounds = output       # (x4, x3, x2, x1): layer1..layer4 outputs, NHWC, attached to the trunk's graph
        self._cut = tuple(t.detach().requires_grad_(True) for t in output)
        return self._cut

    def _grab_chain(self, k, output):
        """Forward hook on cut module k of a chain (HRNet), active only inside this helper's own forward: keeps the attached list of
        branch maps and hands detached aliases downstream, so that each backward stage stops at the cut in front of it."""
        if not self._cutting:
            return None
        outs = list(output)
        self._att[k] = outs
        self._det[k] = [t.detach().requires_grad_(True) for t in outs]
        return type(output)(self._det[k]) if isinstance(output, (list, tuple)) else self._det[k]

    def _broadcast_state(self):
        import torch.distributed as dist
        every = list(self.model.parameters())
        with torch.no_grad():
            buf = torch.cat([p.detach().reshape(-1) for p in every])
            dist.broadcast(buf, 0, group=self.group)
            o = 0
            for p in every:
                p.copy_(buf[o:o + p.numel()].view_as(p))
                o += p.numel()
        for b in self.model.buffers():
            dist.broadcast(b, 0, group=self.group)

    def _forward_loss(self):
        self._bounds = self._cut = None
        if self._chain is not None:
            self._att, self._det = [None] * len(self._chain), [None] * len(self._chain)
        ops.begin_step(self.model)      # engine 2: this step's operand bounds (the packed weight planes below are scaled by them)
        if self.packs is not None:
            # every packed weight operand the step needs (forward and data-gradient layouts of the k > 1 convolutions) in one
            # launch, before the first kernel of the forward pass; the per-call packs then find them ready (ops.PackCache)
            ops._PACK = self.packs
            self.packs.refresh()
        self._cutting = True
        try:
            with streams.limit(self.side_limit):
                out = self.model(self.img)
        finally:
            self._cutting = False
            ops._FWD_PREPARED = False       # (a model without ops.begin_forward leaves the flag set)
        return self.loss_fn(out, self.labels)

    def _stage(self, i, loss, carry):
        """Backward stage i with the weight gradients on the side stream (ops.SIDE_WGRAD), joined before returning."""
        if not self.defer_reduce:
            return self._stage_wgrad(i, loss, carry)
        # the split-K partial slabs of the stage's weight gradients are summed by one launch per 60 gradients at the end of
        # the stage (ops.deferred_reductions): torch.autograd.grad returns the gradient tensors without reading them
        with ops.deferred_reductions():
            return self._stage_wgrad(i, loss, carry)

    def _stage_wgrad(self, i, loss, carry):
        if self.side_wgrad and self.cuda:
            ops.side_wgrad_begin(self.img.device)
        try:
            with streams.limit(self.side_limit):
                return self._stage_body(i, loss, carry)
        finally:
            ops.side_wgrad_join()

    def _stage_body(self, i, loss, carry):
        """Backward stage i.  Returns (parameter gradients of the stage, carry for the next stage)."""
        g = self.groups[i]
        if self.nstage == 1:
            return list(torch.autograd.grad([loss], g, allow_unused=True)), None
        if self._chain is not None:
            # chain of cuts c_0 .. c_{n-1} (forward order), n + 1 stages: stage 0 runs from the loss to the last cut, stage i from the
            # attached outputs of cut n - i (seeded with the gradients that arrived at their detached aliases) to cut n - i - 1
            n = len(self._chain)
            roots, seeds = ([loss], None) if i == 0 else (self._att[n - i], list(carry))
            stop = self._det[n - i - 1] if i < n else []
            if i > 0:           # (a branch map that nothing downstream read has no gradient: it contributes nothing)
                keep = [k for k, (r, sd) in enumerate(zip(roots, seeds)) if sd is not None and r.requires_grad]
                roots, seeds = [roots[k] for k in keep], [seeds[k] for k in keep]
            r = torch.autograd.grad(roots, g + stop, grad_outputs=seeds, allow_unused=True)
            return list(r[:len(g)]), (tuple(r[len(g):]) if stop else None)
        x4, x3, x2, x1 = self._bounds
        if i == 0:
            c4, c3, c2, c1 = self._cut
            r = torch.autograd.grad([loss], g + [c1, c2, c3, c4], allow_unused=True)
            return list(r[:len(g)]), tuple(torch.zeros_like(c) if t is None else t
                                           for t, c in zip(r[len(g):], (c1, c2, c3, c4)))
        if i == 1:
            g1, g2, g3, g4 = carry
            r = torch.autograd.grad([x1, x2], g + [x3], grad_outputs=[g1, g2], allow_unused=True)
            return list(r[:len(g)]), (g3 + r[-1], g4)
        g3, g4 = carry
        return list(torch.autograd.grad([x3, x4], g, grad_outputs=[g3, g4], allow_unused=True)), None

    def _setup_buckets(self, grads_per_stage):
        self.live, self.flat, self.views = [], [], []
        for g, grads in zip(self.groups, grads_per_stage):
            live = [j for j, t in enumerate(grads) if t is not None]
            n, offs = 0, []
            for j in live:
                offs.append(n)
                n += (g[j].numel() + 3) // 4 * 4
            flat = torch.zeros(max(n, 4), device=self.img.device, dtype=torch.float32)
            self.live.append(live)
            self.flat.append(flat)
            self.views.append([flat[o:o + g[j].numel()].view_as(g[j]) for o, j in zip(offs, live)])

    def _reduce(self, i, grads):
        """Average stage i's gradients over the ranks; leaves `.grad` of its parameters pointing at the result."""
        import torch.distributed as dist
        g, live = self.groups[i], self.live[i]
        if self.order is not None:
            self.order.append(('reduce', i))
        if not self.exchange:
            # (re-)bind every step: the reference loop's `optimizer.zero_grad()` sets `.grad` to None, and an optimizer
            # silently skips such parameters -- an identity check per tensor costs ~0.1 ms of otherwise idle host time
            for j in live:
                if g[j].grad is not grads[j]:
                    g[j].grad = grads[j]
            return None
        src = [grads[j] for j in live]
        if self.overlap and self.cuda:
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                torch._foreach_copy_(self.views[i], src)
                self.flat[i].mul_(1.0 / self.world)
                dist.all_reduce(self.flat[i], op=dist.ReduceOp.SUM, group=self.group)
        else:
            torch._foreach_copy_(self.views[i], src)
            self.flat[i].mul_(1.0 / self.world)
            dist.all_reduce(self.flat[i], op=dist.ReduceOp.SUM, group=self.group)
        for j, v in zip(live, self.views[i]):
            if g[j].grad is not v:
                g[j].grad = v
        return None

    # ------------------------------------------------------------------ eager step
    def _shared_parameters(self, roots):
        """Parameters that enter the autograd graph more than once (weight sharing).  Their gradient is the SUM of two
        contributions, which autograd forms during the backward pass -- i.e. it would read a deferred reduction early."""
        params = {id(p) for g in self.groups for p in g}      # (the detached trunk maps are leaves too, and used many times)
        stack = [r.grad_fn for r in roots if r is not None and r.grad_fn is not None]
        seen, uses, keep = set(), {}, []        # `keep`: the node wrappers are temporaries -- their id() is only unique alive
        while stack:
            fn = stack.pop()
            if id(fn) in seen:
                continue
            seen.add(id(fn))
            keep.append(fn)
            for nxt, _ in fn.next_functions:
                if nxt is None:
                    continue
                if hasattr(nxt, 'variable'):            # AccumulateGrad: one incoming edge per use of the parameter
                    v = nxt.variable
                    if id(v) in params:
                        uses[id(v)] = uses.get(id(v), 0) + 1
                else:
                    stack.append(nxt)
        return sum(1 for n in uses.values() if n > 1)

    def _step_eager(self):
        with ops.owned_bounds():        # forward + every backward stage: nothing rewrites a weight before _finish()
            loss = self._step_eager_body()
        self._finish()
        return loss

    def _step_eager_body(self):
        loss = self._forward_loss()
        if self.defer_reduce and self.live is None:
            shared = self._shared_parameters([loss] + list(self._bounds or ()) + [t for a in (self._att or ()) for t in (a or ())])
            if shared:
                import warnings
                warnings.warn('TrainStep: %d parameters are used more than once in the forward pass; the batched split-K / '
                              'LayerNorm reductions (defer_reduce) are turned off for this model' % shared)
                self.defer_reduce = False
        carry, per_stage = None, []
        for i in range(self.nstage):
            if self.order is not None:
                self.order.append(('stage', i))
            grads, carry = self._stage(i, loss, carry)
            per_stage.append(grads)
            if self.live is not None:
                self._reduce(i, grads)
        if self.live is None:                       # first step: learn which parameters are live, then reduce in order
            self._setup_buckets(per_stage)
            for i, grads in enumerate(per_stage):
                self._reduce(i, grads)
        return loss.detach()

    def _finish(self):
        if self.packs is not None:      # the optimizer is about to change the weights: nothing may use the packed copies now
            self.packs.stale()
            ops._PACK = None
        if self.cuda and self.exchange and self.overlap:
            self._e_bwd = torch.cuda.Event(enable_timing=True)
            self._e_comm = torch.cuda.Event(enable_timing=True)
            self._e_bwd.record()
            torch.cuda.current_stream().wait_stream(self.side)
            self._e_comm.record()
            self._exposed.append((self._e_bwd, self._e_comm))
            if len(self._exposed) > 64:
                self._exposed = self._exposed[-64:]
        self.opt.step()

    # ------------------------------------------------------------------ hipGraph
    def _capture(self):
        if ops.DROPOUT_SEED_TENSOR is None:
            ops.DROPOUT_SEED_TENSOR = torch.zeros(1, dtype=torch.int64, device=self.img.device)
        seed_word = ops.DROPOUT_SEED_TENSOR
        cap = torch.cuda.Stream(device=self.img.device)
        cap.wait_stream(torch.cuda.current_stream())
        # The two warm-up steps below are REAL steps on `example_batch` (lazy tables, allocator, bucket setup, optimizer
        # state allocation) -- a placeholder batch would otherwise perturb a freshly loaded checkpoint.  Everything they
        # change is put back afterwards: parameters, buffers (BatchNorm running statistics, num_batches_tracked) and the
        # optimizer state (moments, step counts), so construction leaves the training trajectory untouched.
        import copy
        with torch.no_grad():
            saved_p = [p.detach().clone() for p in self.model.parameters()]
            saved_b = [b.detach().clone() for b in self.model.buffers()]
        saved_opt = copy.deepcopy(self.opt.state_dict())
        with torch.cuda.stream(cap):
            for _ in range(2):                      # warm-up on the capture stream
                ops.TABLE_BYTES_STEP = 0
                self._step_eager()
            with torch.no_grad():
                for p, s_ in zip(self.model.parameters(), saved_p):
                    p.copy_(s_)
                for b, s_ in zip(self.model.buffers(), saved_b):
                    if not torch.equal(b, s_):      # (untouched buffers keep their version: constants cached on it stay valid)
                        b.copy_(s_)
        torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize()
        self.opt.load_state_dict(saved_opt)
        del saved_p, saved_b, saved_opt
        if self.exchange:
            import torch.distributed as dist
            dist.barrier(group=self.group)
            torch.cuda.synchronize()                # no collective may be in flight while a stream is capturing
        for p in self.model.parameters():
            p.grad = None
        self.graphs, self.static = [], []
        # pinned staging memory of captured host-to-device copies (launch tables of the grouped weight-gradient GEMMs): the
        # graphs re-read it at every replay, so it lives as long as this object; allocated HERE, before the capture (pinned
        # allocations are illegal while a stream captures), twice the size one warm-up step packed
        ops.TABLE_ARENA = self._table_arena = ops.TableArena(2 * ops.TABLE_BYTES_STEP + (1 << 16))
        # No cyclic garbage collection while a stream captures: a collection that happens to run inside the capture finalises
        # whatever garbage earlier code left behind (another TrainStep's graphs, pinned arenas, streams, tensors with recorded
        # streams) and their destructors call HIP functions that are illegal during a capture -- the process aborts (seen once in
        # round 6: "Fatal Python error: Aborted ... Garbage-collecting" inside BatchNormFn.forward of the first captured stage).
        import gc
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            self._capture_stages(cap, seed_word)
        finally:
            if gc_was_on:
                gc.enable()

    def _capture_stages(self, cap, seed_word):
        pool = None
        with torch.cuda.stream(cap), ops.owned_bounds():
            carry, loss = None, None
            for i in range(self.nstage):
                gph = torch.cuda.CUDAGraph()
                # thread_local: RCCL's watchdog thread polls events of finished collectives; under the default (global) mode
                # such a call from another thread invalidates the capture (hipErrorStreamCaptureUnsupported)
                with torch.cuda.graph(gph, pool=pool, stream=cap, capture_error_mode='thread_local'):
                    if i == 0:
                        seed_word.add_(0x9E3779B1)
                        loss = self._forward_loss()
                    grads, carry = self._stage(i, loss, carry)
                    streams.assert_joined()         # no forked branch may be open when the capture ends
                pool = gph.pool()
                self.graphs.append(gph)
                self.static.append(grads)
            self._static_loss = loss.detach()
        ops.TABLE_ARENA = None
        if self.packs is not None:
            self.packs.stale()
            ops._PACK = None
        torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize()
        for i, grads in enumerate(self.static):
            assert [j for j, t in enumerate(grads) if t is not None] == self.live[i], 'live gradient set changed'

    def _step_graph(self):
        for i, gph in enumerate(self.graphs):
            if self.order is not None:
                self.order.append(('stage', i))
            gph.replay()
            self._reduce(i, self.static[i])
        self._finish()
        return self._static_loss

    # ------------------------------------------------------------------ public
    def __call__(self, img=None, labels=None):
        if img is not None and img is not self.img:
            self.img.copy_(img, non_blocking=True)
        if labels is not None and labels is not self.labels:
            for k, v in labels.items():
                if v is not self.labels[k]:
                    self.labels[k].copy_(v, non_blocking=True)
        try:
            return self._step_graph() if self.use_graph else self._step_eager()
        finally:
            # whatever happened in the step (an exception included): no later forward pass may find this step's packed
            # weights installed as fresh
            if self.packs is not None:
                self.packs.stale()
                if ops._PACK is self.packs:
                    ops._PACK = None

    def comm_ms_exposed(self):
        """Median over the recorded steps of the time the compute stream waited for the gradient exchange after the last
        backward stage (None without an exchange).  Synchronises."""
        if not self._exposed:
            return None
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in self._exposed)
        return t[len(t) // 2]

    def bucket_bytes(self):
        return [int(f.numel()) * 4 for f in (self.flat or [])]
