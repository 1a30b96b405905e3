"""Fork / join of independent launch sequences onto side HIP streams.

HRNet's exchange unit (models/model_zoo/hrnet.py:102-238) is two to four BasicBlock chains that do not see each other until
the fuse layer, and most of their kernels are smaller than the chip (a 64x64-tile GEMM of the 8x8 branch at B = 32 has 128
workgroups for 256 CUs) or sit at the dependent-launch floor.  Run in one stream they serialize; forked onto side streams they
overlap -- eagerly, and as parallel branches of the hipGraph when `train.TrainStep` captures the step (the fork / join events
are captured like the kernels).  The backward needs nothing extra: autograd runs every node on the stream its forward ran on
and orders (and `record_stream`s) the gradients that cross streams.

Allocator safety: a tensor that crosses streams is `record_stream`ed on the stream that reads it, so its block is not handed
out again before that reader has finished (under capture: not before the capture has ended)."""
import os

import torch

# side streams per device; 0 = everything in the calling stream.  HRNet-W32 step, one box (profiles/r04/ab/c10_hr_side*.log):
# 0: 627, 1: 671-673, 2: 737, 3: 774 images/s.  More than one side stream inside a captured step needs the autograd hop of
# fork_join (HOP below): without it hipStreamEndCapture overflows the stack (ROCm 7.0.2, see _Hop).
SIDE = int(os.environ.get('RIH_SIDE_STREAMS', '3'))
# side-stream results re-enter the autograd graph through a node on the calling stream (see _Hop); 0 only to reproduce the crash
HOP = os.environ.get('RIH_FORK_HOP', '1') != '0'
# the clamp on SIDE while a stream captures: none with the hop, 1 without it (tools/capture_fork_min.py raises it to reproduce)
CAPTURE_MAX = int(os.environ.get('RIH_SIDE_CAPTURE_MAX', '64' if HOP else '1'))
_POOL = {}
_LIMIT = None           # `limit(n)`: an upper bound on SIDE for the code inside the context (renderih_amd.train.TrainStep)
_OPEN = []              # side streams forked from the calling stream and not yet joined (assert_joined)
_WARNED = False


class limit:
    """Context: at most n side streams inside (nests; None = no extra limit).  TrainStep uses it instead of rewriting the module
    global, so that its choice dies with its own forward / backward and touches no other model of the process."""

    def __init__(self, n):
        self.n = n

    def __enter__(self):
        global _LIMIT
        self.prev = _LIMIT
        if self.n is not None:
            _LIMIT = self.n if _LIMIT is None else min(_LIMIT, self.n)
        return self

    def __exit__(self, *a):
        global _LIMIT
        _LIMIT = self.prev
        return False


def effective_side(device=None):
    """Side streams fork_join may use right now: RIH_SIDE_STREAMS under the enclosing `limit`.  While a stream captures AND the
    autograd hop is switched off (RIH_FORK_HOP=0), at most ONE: two side streams that wait on each other's events -- the
    backward of an all-to-all exchange does -- make hipStreamEndCapture recurse without end (see _Hop).  A larger request is
    then clamped with one warning, not obeyed."""
    global _WARNED
    n = SIDE if _LIMIT is None else min(SIDE, _LIMIT)
    if n > CAPTURE_MAX and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        if not _WARNED:
            import warnings
            warnings.warn('renderih_amd.streams: RIH_SIDE_STREAMS=%d clamped to %d while a hipGraph is being captured (captured '
                          'steps with more than one side branch and RIH_FORK_HOP=0 crash the HIP runtime)' % (SIDE, CAPTURE_MAX))
            _WARNED = True
        n = CAPTURE_MAX
    return max(n, 0)


def assert_joined():
    """Raise if a side stream forked by fork_join has not been joined back yet.  TrainStep calls it before it ends a stage's
    capture: ending a capture with an un-joined branch is an error the HIP runtime reports badly."""
    if _OPEN:
        raise RuntimeError('renderih_amd.streams: %d side stream(s) forked and not joined' % len(_OPEN))



def side_streams(device, n):
    key = (device.type, device.index)
    pool = _POOL.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


class _Hop(torch.autograd.Function):
    """Identity whose autograd node lives on the CALLING stream.  fork_join passes every tensor a side stream produced through
    it after the join, so that in the backward a gradient never travels from one side stream straight to another: it is
    handed side -> calling stream -> side.  Why that matters: hipStreamWaitEvent on a capturing, non-origin stream appends the
    waiting stream to the EVENT stream's parallel-capture list (every time, not only when it first joins the capture), and
    hipStreamEndCapture walks those lists recursively before clearing them -- two side streams that each waited on the other
    once form a cycle and hip::Stream::EndCapture() recurses until the stack overflows (ROCm 7.0.2; root cause of the SIGSEGV
    of captured steps with more than one side stream: profiles/r04/capture_segv_rocgdb_backtrace_c8.txt).  With the
    hop every event wait of a captured step has the origin stream on one side, and the origin stream is never listed."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g


def _hop(r):
    if torch.is_tensor(r):
        if not r.requires_grad:
            return r
        y = _Hop.apply(r)
        # the hop's output is a fresh view: hand on what the producing kernels left on the tensor -- its operand bound (engine 2:
        # without it every consumer pays an rih_absmax pass)
        b = getattr(r, '_rih_bound', None)
        if b is not None:
            y._rih_bound = (b[0], y._version, b[2])
        return y
    if isinstance(r, (list, tuple)):
        return type(r)(_hop(t) for t in r)
    return r


def _tensors(r):
    if torch.is_tensor(r):
        return [r]
    return [t for t in r if torch.is_tensor(t)] if r is not None else []


def fork_join(thunks, reads=None):
    """[thunk() for thunk in thunks], thunk 0 in the calling stream and thunks 1.. each in a side stream that starts after
    everything enqueued so far and is joined before this returns.  reads[k]: the tensors thunk k reads.  With SIDE == 0, on
    CPU tensors, or with a single thunk: a plain loop."""
    n = len(thunks)
    first = None
    if reads is not None:
        for r in reads:
            for t in _tensors(r):
                first = t
                break
            if first is not None:
                break
    if n < 2 or first is None or not first.is_cuda:
        return [f() for f in thunks]
    side = effective_side()
    if side <= 0:
        return [f() for f in thunks]
    main = torch.cuda.current_stream(first.device)
    streams = side_streams(first.device, min(side, n - 1))
    out = [None] * n
    used = []
    for k in range(1, n):                   # every side stream starts at THIS point of the calling stream ...
        s = streams[(k - 1) % len(streams)]
        if s not in used:
            s.wait_stream(main)
            used.append(s)
            _OPEN.append(s)
    try:
        out[0] = thunks[0]()                # ... and the thunks are issued in list order, so autograd's node order (and with it
        for k in range(1, n):               # the order in which gradients of shared inputs are summed) is the single-stream one
            s = streams[(k - 1) % len(streams)]
            with torch.cuda.stream(s):
                for t in _tensors(reads[k]):
                    t.record_stream(s)
                out[k] = thunks[k]()
    finally:
        for s in used:                      # joined on every path out (an exception inside a thunk included)
            main.wait_stream(s)
            _OPEN.remove(s)
    for k in range(1, n):
        for t in _tensors(out[k]):
            t.record_stream(main)
    if HOP and len(used) > 1 and torch.is_grad_enabled():
        out = [out[0]] + [_hop(o) for o in out[1:]]
    return out
