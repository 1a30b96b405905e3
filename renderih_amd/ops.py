"""Operators of the pose network as `torch.autograd.Function`s over the HIP C ABI.

PyTorch supplies device memory (caching allocator), the current HIP stream and autograd bookkeeping;
all arithmetic runs in librenderih_amd.so.  Activations are NHWC; every tensor is fp32, contiguous,
on the GPU.  There is no CPU or eager fallback: a missing library or a non-GPU tensor raises.
"""
import ctypes as C
import math
import os
import torch

from . import _lib
from ._lib import GemmDesc, check


def _L():
    return _lib.load()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _chk(*ts, dtype=torch.float32):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('renderih_amd ops need GPU tensors (HIP kernels only, no CPU fallback)')
        if dtype is not None and t.dtype != dtype:
            raise RuntimeError('renderih_amd op expects %s; got %s' % (dtype, t.dtype))


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _cdiv(a, b):
    return (a + b - 1) // b


# --------------------------------------------------------------------------------------------- GEMM plumbing
def pick_tile(M, N, batch=1, engine=None, K=0):
    """0: 128x128, 1: 128x64, 2: 64x64, 3: 128x32 -- fill 256 CUs (x 2-3 resident blocks) before growing the tile.
    Returns the tile id; `plan_gemm` adds the forward split-K decision."""
    return plan_gemm(M, N, K, batch, engine)[0]


def plan_gemm(M, N, K, batch=1, engine=None):
    """(tile, splitk) for a forward GEMM.  Measured on MI355X (tools/gemm_bench.py): the split engine wants >= ~1.5
    blocks per CU and long k-loops (its per-block prologue/epilogue is expensive), so deep-K problems with few
    output tiles are split over K instead of shrinking the tile."""
    e = ENGINE if engine is None else engine
    if e == 2 and E2_SHORTK_T1 and K <= 256 and N >= 64 and _cdiv(M, 128) * _cdiv(N, 64) * batch >= 512:
        # engine 2, short reductions on large maps (the 1x1 convolutions of layer1 / layer2 and their data gradients): 128x64
        # tiles beat the 64x64 tiles of engine 1's rule by 5-15 % and the 128x128 ones by 3-8 % (profiles/r04/tile_sweep_engine2_c6.log)
        return 1, 1
    e = min(e, 1)            # otherwise engine 2 shares engine 1's kernels' structure: planned alike
    if N <= 32:
        # 16 < N <= 32 (HRNet's 32-channel branch): a half-empty split-engine tile beats the 128x32 tile of the native-f32
        # engine by 6-9 % (profiles/r02/n32_bench_m31.log); below that the waste is too large
        if e == 1 and N > 16:
            return (1 if (K >= 128 and _cdiv(M, 128) * batch >= 512) else 2), 1
        return 3, 1
    t0 = _cdiv(M, 128) * _cdiv(N, 128) * batch
    t1 = _cdiv(M, 128) * _cdiv(N, 64) * batch
    if e == 1:
        if (K <= 128 or (K <= 256 and N <= 64)) and _cdiv(M, 64) * _cdiv(N, 64) * batch >= 512:
            # short reductions are bound by the output stream: more, smaller tiles in flight win.  (Up to K = 256 before the
            # 16-byte-store epilogue; with it 128x128 wins from K = 256 on: profiles/r02/tile_sweep_m15.log)
            return 2, 1
        if N > 64 and t0 >= 384:
            return 0, 1
        if t1 >= 384:
            return 1, 1
        if batch == 1 and K >= 2048 and N > 64 and t0 >= 32:
            return 0, max(1, min(_cdiv(K, 512), _cdiv(512, t0)))
        tile = 2
    else:
        if N <= 64:
            tile = 1 if t1 >= 384 * 2 else 2      # (historic rule: cdiv(M,128)*batch >= 384)
        elif t0 >= 384:
            return 0, 1
        else:
            tile = 2
    bm, bn = _TILE_MN[tile]
    tiles = _cdiv(M, bm) * _cdiv(N, bn) * batch
    if batch == 1 and K >= 1024 and tiles < 128:
        return tile, max(1, min(_cdiv(K, 256), _cdiv(256, tiles)))
    return tile, 1


_TILE_MN = {0: (128, 128), 1: (128, 64), 2: (64, 64), 3: (128, 32)}
E2_SHORTK_T1 = os.environ.get('RIH_E2_SHORTK_TILE1', '1') == '1'       # engine 2: 128x64 tiles for short reductions on large maps

# MFMA engine of rih_gemm (include/renderih_amd.h): 2 (default since round 4) = fp32 on THREE fp16 MFMA products (scaled two-term
# split, 833 TF ceiling) wherever a call site has operand bounds (the convolutions: bounds from the BatchNorm kernels), engine 1
# elsewhere; 1 = fp32 emulated on the bf16 pipe (three-term split, six products, fp32 accumulate -- fp32-grade accuracy at up to
# 417 TF); 0 = native f32 MFMA (157 TF).  Tile 3 (N <= 32) always runs engine 0.  Measured same-box (profiles/r04/ab/c3_*):
# engine 1 1807 images/s, engine 2 1933-1946 (+7 %); its error against fp64 is half of engine 1's and at or below rocBLAS fp32
# on every bench shape (profiles/r04/e2_bench_c2_production_config.log).
import os as _os
ENGINE = int(_os.environ.get('RIH_GEMM_ENGINE', '2'))

# Engine 2 (RIH_GEMM_ENGINE=2; rih_gemm_desc.engine 2): fp32 on THREE fp16 MFMA products.  Each operand needs a device-resident
# upper bound of its largest magnitude (rih_gemm_desc.amax_a / amax_b) from which the kernel derives its power-of-two scale.
# `bound_of(t)` returns that bound as a one-element tensor: the one the producing kernel left on the tensor (`set_bound`), the
# one cached from an earlier use, or a fresh rih_absmax pass.  A call site that has no bounds for both operands runs engine 1.
class _BoundPool:
    """Zeroed bound blocks (_lib.BOUND_FLOATS floats each: rih_absmax in include/renderih_amd.h) carved out of chunks -- one
    fill launch per 256 blocks; a chunk allocated while a stream captures is re-zeroed by every replay of the graph.
    Chunks are PER STREAM (round 5): a block is handed out on the stream whose next launch writes it (the producing BatchNorm /
    rih_absmax kernel), so its zero fill must be ordered before that launch -- which stream order gives only if the fill ran on
    the SAME stream.  One chunk shared by all streams had its fill on whichever stream happened to exhaust the previous chunk
    (inside a streams.fork_join side branch, say) with nothing ordering it against the other branches: a bound published early
    could be zeroed afterwards, the consuming GEMM then scaled by 1 instead of 2^k -- finite but differently rounded results that
    changed from replay to replay (seen on the captured HRNet-W32 step with three side streams, profiles/r05/ab/c1_pytest_r5.log)
    and an fp16 overflow waiting to happen on large activations."""
    CHUNK = 256

    def __init__(self):
        self.chunk = {}

    def slot(self, device):
        stream = torch.cuda.current_stream(device).cuda_stream if device.type == 'cuda' else 0
        key = (device.type, device.index, stream)
        ent = self.chunk.get(key)
        if ent is None or ent[1] >= self.CHUNK:
            ent = self.chunk[key] = [torch.zeros((self.CHUNK * _lib.BOUND_FLOATS,), device=device, dtype=torch.float32), 0]
        i = ent[1]
        ent[1] = i + 1
        return ent[0][i * _lib.BOUND_FLOATS:(i + 1) * _lib.BOUND_FLOATS]

    def reset(self):
        self.chunk.clear()


_BOUNDS = _BoundPool()
_BOUND_EPOCH = 0            # cached bounds (tensor attribute `_rih_bound`) are valid inside one epoch only
# > 0 inside `owned_bounds()`: a model forward (HandNET_GCN.forward) or a whole training step (TrainStep) owns the current epoch --
# it measured every convolution weight at its top and nothing rewrites a weight before it ends.  Only then is the cached bound
# of a PARAMETER trusted: outside such a scope (ops.conv2d / conv_bn called on their own, a sub-module run standalone, a user
# module) a parameter may have been rewritten by something torch's version counter does not see (rih_adam_multi, `w.data.mul_`)
# since it was measured, and a stale, too small bound overflows engine 2's fp16 planes (round-4 advisor finding) -- there a
# parameter's bound is measured again at every use.
_OWNED = 0


class owned_bounds:
    """Context: the enclosed code owns the bound epoch (see _OWNED).  Nests."""

    def __enter__(self):
        global _OWNED
        _OWNED += 1
        return self

    def __exit__(self, *a):
        global _OWNED
        _OWNED -= 1
        return False


def bounds_invalidate():
    """Every cached bound dies (new epoch; the slot chunks stay).  Called by whatever rewrites parameters behind torch's back:
    renderih_amd.optim.Adam after rih_adam_multi."""
    global _BOUND_EPOCH
    _BOUND_EPOCH += 1


def bound_slot(device):
    return _BOUNDS.slot(device)


def bounds_reset():
    """Forget every cached bound and start new slot chunks.  Called at the top of every model forward (HandNET_GCN.forward,
    TrainStep): parameters are rewritten between steps by kernels torch's version counters do not see (rih_adam_multi), and
    inside a captured step every bound must be (re)computed by a launch of the graph, into slots the graph zeroes."""
    global _BOUND_EPOCH
    _BOUND_EPOCH += 1
    _BOUNDS.reset()


def set_bound(t, slot):
    t._rih_bound = (slot, t._version, _BOUND_EPOCH)
    return t


def cached_bound(t):
    ent = getattr(t, '_rih_bound', None)
    return ent[0] if (ent is not None and ent[1] == t._version and ent[2] == _BOUND_EPOCH) else None


def inherit_bound(y, *xs):
    """y's values are bounded by the largest magnitude among xs (selections / convex combinations / a concatenation of them):
    hand their bound on when every x carries one (engine 2 only; a missing bound leaves y without one -- bound_of then
    measures it)."""
    if ENGINE != 2:
        return y
    slots = [cached_bound(x) for x in xs]
    if any(s_ is None for s_ in slots):
        return y
    b = slots[0]
    for s_ in slots[1:]:
        b = torch.maximum(b, s_)
    return set_bound(y, b)


_FWD_PREPARED = False


def begin_step(model):
    """Top of a training step (TrainStep._forward_loss, before the weight operands are packed): new bound epoch, the bounds of
    every convolution weight in one launch.  The forward pass that follows finds them (begin_forward skips its own reset)."""
    global _FWD_PREPARED
    if ENGINE == 2:
        bounds_reset()
        bound_weights(p for p in model.parameters() if p.dim() == 4)
        _FWD_PREPARED = True


def begin_forward(model):
    """Top of a model's forward pass: operand bounds are per forward pass (see bounds_reset)."""
    global _FWD_PREPARED
    if ENGINE != 2:
        return
    if _FWD_PREPARED:
        _FWD_PREPARED = False
        return
    bounds_reset()
    bound_weights(p for p in model.parameters() if p.dim() == 4)


class LazyBound:
    """bound_of(t) on first call (memoised): call sites hand these to gemm(), which asks for the value only when the launch
    takes engine 2's kernels."""

    def __init__(self, t):
        self.t, self.value = t, None

    def __call__(self):
        if self.value is None:
            self.value = bound_of(self.t)
            self.t = None
        return self.value


def bound_weights(params):
    """Bounds of many tensors (the convolution weights of a model) by ONE rih_absmax_multi launch; afterwards bound_of(p) finds
    them cached.  Called at the top of a forward pass, behind bounds_reset()."""
    ps = [p for p in params if p.is_contiguous()]
    if not ps:
        return
    from ._lib import AbsmaxDesc
    arr = (AbsmaxDesc * len(ps))()
    for d, p in zip(arr, ps):
        slot = bound_slot(p.device)
        d.x, d.out, d.n = p.data_ptr(), slot.data_ptr(), p.numel()
        set_bound(p, slot)
    check(_L().rih_absmax_multi(arr, len(ps), _stream()), 'rih_absmax_multi')


def cat_channels(parts):
    """Channel concatenation of NHWC maps (models/encoder.py:165-173)."""
    return inherit_bound(torch.cat(parts, dim=-1), *parts)


def bound_of(t):
    """Bound block (fp32 tensor of _lib.BOUND_FLOATS floats whose maximum is >= max|t|, see above).  The cache entry dies with
    an in-place modification of t that torch sees, and with the next bounds_reset() / bounds_invalidate(); a parameter's entry is
    used only inside owned_bounds() (see _OWNED)."""
    ent = getattr(t, '_rih_bound', None)
    if (ent is not None and ent[1] == t._version and ent[2] == _BOUND_EPOCH
            and (_OWNED > 0 or not isinstance(t, torch.nn.Parameter))):
        return ent[0]
    slot = bound_slot(t.device)
    tc = _c(t)
    check(_L().rih_absmax(tc.data_ptr(), tc.numel(), slot.data_ptr(), _stream()), 'rih_absmax')
    t._rih_bound = (slot, t._version, _BOUND_EPOCH)
    return slot


# Optional device-resident dropout seed word (a 1-element int64 CUDA tensor): when set, every dropout kernel adds
# *DROPOUT_SEED_TENSOR to its seed on the GPU.  A training step captured in a hipGraph advances the word inside the graph
# (`tensor.add_(1)`) and so draws fresh masks at every replay; forward and backward of one step see the same value.
DROPOUT_SEED_TENSOR = None


def _seed_dev():
    return 0 if DROPOUT_SEED_TENSOR is None else DROPOUT_SEED_TENSOR.data_ptr()


# Run the per-hand decoder layers as paired launches on hands-stacked activations (LinearPairFn & co. below).
PAIR_HANDS = os.environ.get('RIH_PAIR_HANDS', '1') != '0'

# When set to a list, every rih_gemm launch is bracketed by HIP events on the launch stream and
# (flops, start, end, tag) is appended -- bench.py uses this for the live roofline measurement.
PROFILE = None
# Same for the BatchNorm family (the largest HBM-bound kernel group): (algorithmic bytes, start, end, tag) per call.
PROFILE_ELEM = None


def _elem_profile(nbytes, tag, fn):
    if PROFILE_ELEM is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    PROFILE_ELEM.append((float(nbytes), e0, e1, tag))
    return r


# BatchNorm training statistics from the convolution's GEMM epilogue (rih_gemm_desc.stats): `conv_bn` of the encoder installs
# a holder around its convolution; the forward-type GEMM of that convolution fills it when its descriptor takes the split
# engine's fast path (rih_gemm_stats_rows), and the BatchNorm behind it finishes the sums instead of reading the activation.
GEMM_STATS = os.environ.get('RIH_GEMM_STATS', '1') == '1'


class StatsHolder:
    def __init__(self):
        self.part = None        # [T][2][N]: (mean, centred sum of squares) per block of `rows` GEMM rows
        self.T = 0
        self.rows = 0


def gemm(A, B, Cout, M, N, K, lda, ldb, ldc, a_mode=0, b_mode=0, bias=None, R=None, ldr=0,
         nb1=1, nb2=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), splitk=1, kchunk=0, sCsplit=0, alpha=1.0, relu=False,
         geom=None, tile=None, engine=None, cstride=None, ones_row=0, sBias=0, sR=0, collect=None, stats=None, drop=None,
         amax_a=None, amax_b=None, a_seg=None):
    """Enqueue one rih_gemm.  A/B/Cout/bias/R are tensors or raw device pointers.
    a_seg: [(tensor, pitch, first column), ...] -- up to three further pieces of a segmented A operand (rih_gemm_desc.a_seg);
    `lda` / `A` describe the first piece.
    cstride = (s, oh, ow, H, W): store GEMM row (img, i, j) to pixel (img, i*s+oh, j*s+ow) of a [*, H, W] tensor.
    collect: a GroupedGemms -- the problem joins its next grouped launch (rih_gemm_multi) when its kernel variant can ride
    in one, instead of being launched now; the operands are kept alive until then.
    stats: a StatsHolder -- filled with the per-row-block BatchNorm statistics of the output when the descriptor takes the split
    engine's statistics epilogue (left empty otherwise: the BatchNorm then runs its own statistics pass).
    drop = (p, seed): C = dropout(act(alpha A B + bias)) + R in the epilogue when the descriptor takes that path
    (rih_gemm_dropout_ok) -- returns True; otherwise act(alpha A B + bias) is computed WITHOUT R and False is returned: the caller
    finishes with rih_add_dropout(R, C, p, seed), which draws the same mask stream.
    amax_a / amax_b (engine 2): bound blocks (bound_of; or raw device pointers, or thunks that return one -- called only when the
    descriptor takes engine 2's kernels) holding an upper bound of max|A| / max|B| (rih_gemm_desc.amax_a); a call site that
    passes none for either operand runs engine 1."""
    d = GemmDesc()
    if a_seg:
        for i, (t, ld, k0) in enumerate(a_seg):
            d.a_seg[i], d.lda_seg[i], d.k_seg[i] = t.data_ptr(), ld, k0
    if cstride is not None:
        d.cS, d.cOH, d.cOW, d.cH, d.cW = cstride
    d.ones_row = ones_row
    d.sBias1, d.sR1 = sBias, sR
    d.engine = ENGINE if engine is None else engine
    if d.engine == 2 and (amax_a is None or amax_b is None):
        d.engine = 1            # no operand bounds at this call site: the six-product engine needs none
    d.A = A if isinstance(A, int) else A.data_ptr()
    d.B = B if isinstance(B, int) else B.data_ptr()
    d.C = Cout if isinstance(Cout, int) else Cout.data_ptr()
    d.bias = _p(bias)
    d.R = _p(R)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc, d.ldr = lda, ldb, ldc, ldr
    d.a_mode, d.b_mode = a_mode, b_mode
    d.nb1, d.nb2 = nb1, nb2
    d.sA1, d.sA2 = sA
    d.sB1, d.sB2 = sB
    d.sC1, d.sC2 = sC
    d.splitk, d.kchunk, d.sCsplit = splitk, kchunk, sCsplit
    d.alpha = alpha
    d.relu = 1 if relu else 0
    if geom is None:
        cin = K if a_mode != 1 else M
        geom = (1, 1, cin, 1, 1, 1, 1, 1, 1, 0, 0)
    (d.H, d.W, d.Cin, d.Ho, d.Wo, d.KH, d.KW, d.strideA, d.upS, d.padH, d.padW) = geom
    auto_sk = 1
    if tile is None:
        d.tile, auto_sk = plan_gemm(M, N, K, nb1 * nb2 * splitk, d.engine)
    else:
        d.tile = tile
    if d.engine == 2:
        # bounds may be given as thunks (bound_of of a tensor nobody has measured yet costs a pass): resolved only when the
        # descriptor really takes engine 2's kernels (rih_gemm_engine), else the launch runs engine 1 and needs none
        if int(_L().rih_gemm_engine(C.byref(d))) == 2:
            amax_a = amax_a() if callable(amax_a) else amax_a
            amax_b = amax_b() if callable(amax_b) else amax_b
            d.amax_a = amax_a if isinstance(amax_a, int) else _p(amax_a)
            d.amax_b = amax_b if isinstance(amax_b, int) else _p(amax_b)
        else:
            d.engine = 1
            amax_a = amax_b = None
    fused_drop = False
    if drop is not None and drop[0] > 0:
        assert collect is None and stats is None and splitk == 1 and cstride is None
        will_split = auto_sk > 1 and nb1 * nb2 == 1 and not isinstance(Cout, int) and _cdiv(K, _cdiv(_cdiv(K, auto_sk), 32) * 32) > 1
        d.drop_p, d.drop_seed, d.drop_seed_dev = float(drop[0]), int(drop[1]) & 0xFFFFFFFFFFFFFFFF, (_seed_dev() or None)
        if not will_split and not (relu and R is not None) and int(_L().rih_gemm_dropout_ok(C.byref(d))) == 1:
            fused_drop = True
        else:
            d.drop_p, d.drop_seed, d.drop_seed_dev = 0.0, 0, None
            d.R, R = None, None             # the caller adds R behind its own dropout pass
    if auto_sk > 1 and splitk == 1 and nb1 * nb2 == 1 and not isinstance(Cout, int) and cstride is None and not a_seg:
        # few output tiles but a long reduction (e.g. the 8x8 3x3 convs, the 4x4 patch conv): split K over
        # workgroups and finish (bias / residual / ReLU) in a second pass
        kc = _cdiv(_cdiv(K, auto_sk), 32) * 32
        sk = _cdiv(K, kc)
        if sk > 1:
            part = torch.empty((sk, M, N), device=Cout.device, dtype=torch.float32)
            gemm(A, B, part, M, N, K, lda, ldb, N, a_mode=a_mode, b_mode=b_mode, splitk=sk, kchunk=kc,
                 sCsplit=M * N, geom=geom, tile=d.tile, engine=d.engine, amax_a=amax_a, amax_b=amax_b)
            check(_L().rih_splitk_finish(part.data_ptr(), sk, M, N, d.C, ldc, d.bias or 0, d.R or 0, ldr,
                                         alpha, 1 if relu else 0, _stream()), 'rih_splitk_finish')
            return False
    req = stats
    if (req is not None and req.part is None and a_mode == 0 and splitk == 1 and nb1 * nb2 == 1 and cstride is None
            and not isinstance(Cout, int)):
        rows_per = int(_L().rih_gemm_stats_rows(C.byref(d)))
        if rows_per > 0:
            req.T, req.rows = _cdiv(M, rows_per), rows_per
            req.part = torch.empty((req.T, 2, N), device=Cout.device, dtype=torch.float32)
            d.stats = req.part.data_ptr()
    if collect is not None and collect.add(d, (A, B, Cout, amax_a, amax_b), 2.0 * M * N * K * nb1 * nb2):
        return
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_L().rih_gemm(C.byref(d), _stream()), 'rih_gemm')
        e1.record()
        PROFILE.append((2.0 * M * N * K * nb1 * nb2, e0, e1, (M, N, K, nb1 * nb2, a_mode, b_mode, d.tile, splitk,
                                                              int(_L().rih_gemm_engine(C.byref(d))))))
        return fused_drop
    check(_L().rih_gemm(C.byref(d), _stream()), 'rih_gemm')
    return fused_drop


# --------------------------------------------------------------------------------------------- grouped launches
# rih_gemm_multi: the weight-gradient GEMMs of a backward stage -- independent of each other and of everything else until the
# optimizer -- are collected inside `deferred_reductions()` and run as ONE launch per kernel variant at the end of the stage
# (the decoder's 116 nn.Linear gradients per ResNet50 step were 116 launches of ~20 us with a few hundred short workgroups
# each).  The launch table (prepared kernel arguments + block -> problem map) is packed on the host, staged in pinned memory
# and copied to the device in stream order; under hipGraph capture the copy is a node of the graph, so the staging buffer must
# outlive the graph AND must not be allocated while a stream captures (the pinned allocator's event bookkeeping is illegal
# there): the capturing code (renderih_amd.train.TrainStep) installs a `TableArena` -- pinned memory allocated before the
# capture, sized from the table bytes its warm-up steps used (`TABLE_BYTES_STEP`) -- and the tables are carved out of it.
GROUP_WGRAD = int(os.environ.get('RIH_WGRAD_GROUP', '2'))        # 0: off, 1: decoder-sized gradients only, 2: every weight gradient (default:
#                                                                   same-box +2.3 % over 1 on ResNet50, +6 % on HRNet-W32, profiles/r03/ab/h*)
GROUP_KCHUNK = int(os.environ.get('RIH_WGRAD_GROUP_KCHUNK', '1024'))      # pixels per split-K slice in a grouped launch
GROUP_SORT = os.environ.get('RIH_WGRAD_GROUP_SORT', '1') == '1'
# 128x64 tiles for weight gradients with 33..64 output channels and >= WGRAD_T1_MINK pixels (64x64 tiles otherwise): round 6, A/B below
WGRAD_T1 = os.environ.get('RIH_WGRAD_T1', '0') == '1'
WGRAD_T1_MINK = int(os.environ.get('RIH_WGRAD_T1_MINK', '16384'))
TABLE_ARENA = None
TABLE_BYTES_STEP = 0            # table bytes packed since the counter was last reset (TrainStep sizes its arena from it)


class TableArena:
    """Bump allocator over one pinned host buffer (256-byte granules); lives as long as the graphs that re-read it."""

    def __init__(self, nbytes):
        self.buf = torch.empty((max(int(nbytes), 4096),), dtype=torch.uint8, pin_memory=True)
        self.used = 0

    def take(self, nbytes):
        a = (self.used + 255) // 256 * 256
        if a + nbytes > self.buf.numel():
            raise RuntimeError('renderih_amd: launch-table arena exhausted (%d + %d > %d bytes): the captured step packs '
                               'more grouped launches than the warm-up steps did' % (a, nbytes, self.buf.numel()))
        self.used = a + nbytes
        return self.buf[a:a + nbytes]


class GroupedGemms:
    def __init__(self):
        self.items = []         # (variant, descriptor copy, operands kept alive, flops)

    def add(self, d, keep, flops):
        v = int(_L().rih_gemm_multi_variant(C.byref(d)))
        if v < 0:
            return False
        c = GemmDesc()
        C.memmove(C.byref(c), C.byref(d), C.sizeof(GemmDesc))
        self.items.append((v, c, keep, flops))
        return True

    def flush(self):
        global TABLE_BYTES_STEP
        items, self.items = self.items, []
        by = {}
        for v, d, keep, fl in items:
            by.setdefault(v, []).append((d, keep, fl))
        lib = _L()
        for v, group in sorted(by.items()):
            if GROUP_SORT:
                # longest blocks first (a block's run time ~ its reduction length): the launch's tail is then made of short
                # blocks.  The order of the problems does not touch any problem's own summation order.
                group.sort(key=lambda g: -(g[0].kchunk if g[0].splitk > 1 else g[0].K))
            n = len(group)
            arr = (GemmDesc * n)(*[g[0] for g in group])
            nbytes = int(lib.rih_gemm_multi_table_bytes(arr, n))
            if nbytes <= 0:
                raise RuntimeError('renderih_amd: rih_gemm_multi_table_bytes rejected the group')
            ref = next(t for t in group[0][1] if torch.is_tensor(t))
            on_gpu = ref.is_cuda
            if on_gpu:
                # the collected operands (saved activations, gradients, split-K slabs) may have been allocated by backward nodes
                # that ran on a streams.fork_join side stream; they are released right after this launch is enqueued on the
                # CALLING stream, so their blocks must not go back to the side stream's pool before the launch has run
                cur = torch.cuda.current_stream(ref.device)
                for g in group:
                    for t in g[1]:
                        if torch.is_tensor(t) and t.is_cuda:
                            t.record_stream(cur)
            TABLE_BYTES_STEP += nbytes + 256
            if on_gpu and TABLE_ARENA is not None:
                host = TABLE_ARENA.take(nbytes)
            else:
                host = torch.empty((nbytes,), dtype=torch.uint8, pin_memory=on_gpu)
            total = C.c_int32(0)
            got = int(lib.rih_gemm_multi_pack(arr, n, host.data_ptr(), C.byref(total)))
            if got != v:
                raise RuntimeError('renderih_amd: rih_gemm_multi_pack failed (%d)' % got)
            if on_gpu:
                dev = torch.empty((nbytes,), dtype=torch.uint8, device=ref.device)
                dev.copy_(host, non_blocking=True)
            else:
                dev = host
            if PROFILE is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                check(lib.rih_gemm_multi_launch(dev.data_ptr(), v, total.value, _stream()), 'rih_gemm_multi_launch')
                e1.record()
                PROFILE.append((sum(g[2] for g in group), e0, e1,
                                (0, 0, 0, n, (v >> 2) & 1, (v >> 1) & 1, 20 + ((v & 63) >> 3), 0, 2 if v >= 64 else 1)))
            else:
                check(lib.rih_gemm_multi_launch(dev.data_ptr(), v, total.value, _stream()), 'rih_gemm_multi_launch')


# --------------------------------------------------------------------------------------------- weight gradients off the
# critical path (EXPERIMENT, off by default: RIH_SIDE_WGRAD=1).  The backward pass is a dependent chain of data-gradient
# kernels; a layer's weight gradient (split-K GEMM + reduce, ~1/3 of all launches) feeds nothing but the optimizer.  With
# SIDE_WGRAD set (renderih_amd.train.TrainStep does it around each backward stage) `_wgrad` enqueues its launches on a second
# HIP stream that waits for everything issued so far and is joined at the end of the stage.  Measured on MI355X
# (profiles/r02/bench_side_wgrad_m9.log): 43.89 vs 44.02 ms per step -- no gain: the step is bound by the SUM of kernel
# times (every launch, even a 15 us decoder GEMM, fills the CUs with resident workgroups), not by launch gaps or by the
# dependency chain; an eager (no hipGraph) step measures the same 44.0 ms.  Operands are kept alive until the join.
class _SideWork:
    def __init__(self, device):
        self.stream = torch.cuda.Stream(device=device)
        self.keep = []
        self.used = False


SIDE_WGRAD = None


def side_wgrad_begin(device):
    global SIDE_WGRAD
    if SIDE_WGRAD is None or SIDE_WGRAD.stream.device != device:
        SIDE_WGRAD = _SideWork(device)
    return SIDE_WGRAD


def side_wgrad_join(disable=True):
    """Make the current stream wait for the weight gradients issued on the side stream; drop the keep-alive references."""
    global SIDE_WGRAD
    sw = SIDE_WGRAD
    if sw is not None:
        if sw.used:
            torch.cuda.current_stream().wait_stream(sw.stream)
            sw.used = False
        sw.keep.clear()
        if disable:
            SIDE_WGRAD = None


def _pdiff(a, b):
    """Distance in floats between the storage of two fp32 tensors: the nb1 stride that walks from a left-hand parameter
    to the right-hand one, so that both hands' layers run as one batched launch."""
    d = b.data_ptr() - a.data_ptr()
    assert d % 4 == 0
    return d // 4


# Deferred split-K reductions of weight gradients.  Inside `with deferred_reductions():` a weight gradient's partial slabs
# are NOT summed behind its GEMM; the sums of all gradients of the block run in one launch per 60 gradients when the block
# exits.  Valid only where nothing reads a weight gradient before the block ends -- renderih_amd.train.TrainStep wraps each
# backward stage (torch.autograd.grad hands the gradients back untouched); plain loss.backward() must not use it, because
# AccumulateGrad may add a gradient into an existing .grad immediately.
_DEFERRED = None
_DEFERRED_LN = None         # same for the LayerNorm parameter gradients (rih_ln_param_final_multi)
_DEFERRED_GEMM = None       # the weight-gradient GEMMs themselves (GroupedGemms, rih_gemm_multi) when GROUP_WGRAD is on


class deferred_reductions:
    def __enter__(self):
        global _DEFERRED, _DEFERRED_LN, _DEFERRED_GEMM
        assert _DEFERRED is None, 'deferred_reductions does not nest'
        _DEFERRED, _DEFERRED_LN = [], []
        _DEFERRED_GEMM = GroupedGemms() if (GROUP_WGRAD and ENGINE >= 1) else None
        return self

    def __exit__(self, et, ev, tb):
        global _DEFERRED, _DEFERRED_LN, _DEFERRED_GEMM
        pending, _DEFERRED = _DEFERRED, None
        ln, _DEFERRED_LN = _DEFERRED_LN, None
        gg, _DEFERRED_GEMM = _DEFERRED_GEMM, None
        if et is None and gg is not None and gg.items:
            gg.flush()              # the grouped weight-gradient GEMMs first: the reductions below read their slabs
        if et is None and ln:
            from ._lib import LnFinalDesc
            arr = (LnFinalDesc * len(ln))()
            for d, (_, ws, _, dg, _, db, D, nblk) in zip(arr, ln):
                d.ws, d.dg, d.db, d.D, d.nblk = ws, dg, db, D, nblk
            check(_L().rih_ln_param_final_multi(arr, len(ln), _stream()), 'rih_ln_param_final_multi')
        if et is None and pending:
            from ._lib import ReduceDesc
            arr = (ReduceDesc * len(pending))()
            for d, (_, P, _, dst, _, dbp, (S, Mp, M, N, Cin, taps, CinValid, acc, *rest)) in zip(arr, pending):
                d.P, d.dst, d.db = P, dst, dbp
                d.S, d.Mp, d.M, d.N, d.Cin, d.taps, d.CinValid, d.accumulate = S, Mp, M, N, Cin, taps, CinValid, acc
                d.CinPitch = rest[0] if rest else 0
            check(_L().rih_splitk_reduce_multi(arr, len(pending), _stream()), 'rih_splitk_reduce_multi')
        return False


def _wgrad(x, dy, dw, Kpix, Mrows, Ncols, ldx, ldy, geom, Cin_pad, taps, Cin_valid, db=None, nb=1, sx=0, sdy=0, bounds=None,
           dw_slice=None):
    sw = SIDE_WGRAD
    if sw is None or not x.is_cuda:
        return _wgrad_now(x, dy, dw, Kpix, Mrows, Ncols, ldx, ldy, geom, Cin_pad, taps, Cin_valid, db, nb, sx, sdy, bounds,
                          dw_slice)
    sw.stream.wait_stream(torch.cuda.current_stream())      # x, dy (and everything before them) are ready
    with torch.cuda.stream(sw.stream):
        _wgrad_now(x, dy, dw, Kpix, Mrows, Ncols, ldx, ldy, geom, Cin_pad, taps, Cin_valid, db, nb, sx, sdy, bounds, dw_slice)
    sw.keep.extend((x, dy, dw, db))
    sw.used = True


def _wgrad_now(x, dy, dw, Kpix, Mrows, Ncols, ldx, ldy, geom, Cin_pad, taps, Cin_valid, db=None, nb=1, sx=0, sdy=0, bounds=None,
               dw_slice=None):
    """dw (parameter layout) = im2col(x)^T @ dy with split-K over the Kpix pixels.  With `db` (bias gradient, [Ncols])
    the A operand gets an all-ones row behind its Mrows rows, so the same GEMM also produces the column sums of dy.
    nb > 1: that many independent gradients in one GEMM + one reduce launch (x / dy slices sx / sdy floats apart, sx = 0
    for a shared input; dw [nb, ...] and db [nb, Ncols] contiguous)."""
    Mp = Mrows + 4 if db is not None else Mrows
    # small weight matrices (decoder Linears, <= 4x4 tiles of 128): 64x64 tiles give 4x the resident slices per split
    # -- but only for short pixel reductions: from ~16k pixels on, 128x128 tiles with more K-slices win by 15-20%
    # (profiles/r02/tile_sweep_m15.log: 256->128 @64x64, 128<->512 @32x32)
    small = _cdiv(Mp, 128) * _cdiv(Ncols, 128) <= 4 and (Kpix < 16384 or nb > 1)
    tile = 3 if Ncols <= 32 else (2 if (Ncols <= 64 or Mp <= 64 or (ENGINE >= 1 and small)) else 0)
    if ENGINE >= 1 and 16 < Ncols <= 32:
        tile = 2            # weight gradients with 17..32 columns: 64x64 split-engine tile, -26..-29 % (n32_bench_m31.log)
    if nb > 1 and ENGINE >= 1 and Ncols > 32:
        tile = 2            # paired decoder layers (tools/pair_sweep.py): 64x64 tiles win at every measured shape
    if WGRAD_T1 and ENGINE == 2 and nb == 1 and bounds is not None and 32 < Ncols <= 64 and Mp >= 128 and Kpix >= WGRAD_T1_MINK:
        tile = 1            # <= 64 output channels on a large map (layer1's 64-channel convolutions, the stem): 128x64 tiles
    bm, bn = _TILE_MN[tile]
    tiles = _cdiv(Mp, bm) * _cdiv(Ncols, bn) * nb
    if nb > 1 and ENGINE >= 1:
        # measured optimum (profiles/r01/pair_sweep_v15.log): k-chunks of ~320 pixels per slice, at most ~1024 workgroups
        splitk = max(1, min(int(round(Kpix / 320.0)), 1024 // max(tiles, 1), _cdiv(Kpix, 128)))
    else:
        # measured optimum of resident split-K slices (tools/tile_sweep.py): ~512 workgroups, 256 for a single tile
        target = (256 if tiles == 1 else 512) if ENGINE >= 1 else 1024
        splitk = max(1, min(target // max(tiles, 1), _cdiv(Kpix, 128)))
    # grouped launch (rih_gemm_multi at the end of the backward stage): the problems fill the chip TOGETHER, so a problem needs
    # only as many split-K slices as keep its own workgroups reasonably short -- fewer partial slabs to write and to sum
    collect = None
    if _DEFERRED_GEMM is not None and tile in (0, 1, 2) and (GROUP_WGRAD >= 2 or nb > 1 or small):
        collect = _DEFERRED_GEMM
        # (128x128 tiles for the large grouped gradients -- half the operand bytes per product -- measured +0.25 % same-box in round 4,
        # inside the noise: profiles/r04/ab/train_t128.log; the option was removed)
        splitk = max(1, min(splitk, _cdiv(Kpix, GROUP_KCHUNK)))
    kchunk = _cdiv(_cdiv(Kpix, splitk), 32) * 32
    splitk = _cdiv(Kpix, kchunk)
    part = torch.empty((nb, splitk, Mp, Ncols), device=x.device, dtype=torch.float32)
    ones = Mrows if db is not None else 0
    batch = dict(nb1=nb, sA=(sx, 0), sB=(sdy, 0), sC=(splitk * Mp * Ncols, 0)) if nb > 1 else {}
    if bounds is not None:      # (bound of x, bound of dy): engine 2
        batch = dict(batch, amax_a=bounds[0], amax_b=bounds[1])
    if splitk == 1:
        # a single slice still goes through the reduce kernel for the layout change; raw epilogue = alpha 1, no bias
        gemm(x, dy, part, Mp, Ncols, Kpix, ldx, ldy, Ncols, a_mode=1, b_mode=0, geom=geom, tile=tile, ones_row=ones,
             collect=collect, **batch)
    else:
        gemm(x, dy, part, Mp, Ncols, Kpix, ldx, ldy, Ncols, a_mode=1, b_mode=0, splitk=splitk, kchunk=kchunk,
             sCsplit=Mp * Ncols, geom=geom, tile=tile, ones_row=ones, collect=collect, **batch)
    if dw_slice is not None:
        # the gradient lands in a column slice of a wider parameter: dw_slice = (first column, columns of the whole parameter);
        # always through the descriptor form of the reduction (rih_reduce_desc.CinPitch)
        assert nb == 1 and db is None
        entry = (part, part.data_ptr(), dw, dw.data_ptr() + 4 * dw_slice[0] * taps, None, None,
                 (splitk, Mp, Mrows, Ncols, Cin_pad, taps, Cin_valid, 0, dw_slice[1]))
        if _DEFERRED is not None:
            _DEFERRED.append(entry)
        else:
            from ._lib import ReduceDesc
            arr = (ReduceDesc * 1)()
            d = arr[0]
            d.P, d.dst, d.db = entry[1], entry[3], None
            d.S, d.Mp, d.M, d.N, d.Cin, d.taps, d.CinValid, d.accumulate, d.CinPitch = entry[6]
            check(_L().rih_splitk_reduce_multi(arr, 1, _stream()), 'rih_splitk_reduce_multi')
    elif _DEFERRED is not None:
        # summed at the end of the backward stage by ONE launch per 56 gradients (deferred_reductions below)
        per_dw, per_db = dw.numel() // nb, (Ncols if db is not None else 0)
        for b in range(nb):
            _DEFERRED.append((part, part.data_ptr() + 4 * b * splitk * Mp * Ncols, dw, dw.data_ptr() + 4 * b * per_dw, db,
                              (db.data_ptr() + 4 * b * per_db) if db is not None else None,
                              (splitk, Mp, Mrows, Ncols, Cin_pad, taps, Cin_valid, 0)))
    elif nb == 1:
        check(_L().rih_splitk_reduce_bias(part.data_ptr(), splitk, Mp, Mrows, Ncols, dw.data_ptr(), Cin_pad, taps,
                                          Cin_valid, 0, _p(db), _stream()), 'rih_splitk_reduce_bias')
    else:
        check(_L().rih_splitk_reduce_bias_batched(part.data_ptr(), splitk, Mp, Mrows, Ncols, dw.data_ptr(), Cin_pad, taps,
                                                  Cin_valid, 0, _p(db), nb, splitk * Mp * Ncols, dw.numel() // nb,
                                                  Ncols, _stream()), 'rih_splitk_reduce_bias_batched')


def colsum(x2d, rows, Ccols, ldx=None):
    ldx = Ccols if ldx is None else ldx
    out = torch.empty((Ccols,), device=x2d.device, dtype=torch.float32)
    ws = torch.empty((int(_L().rih_colsum_ws_floats(rows, Ccols)),), device=x2d.device, dtype=torch.float32)
    check(_L().rih_colsum(x2d.data_ptr(), rows, Ccols, ldx, out.data_ptr(), 0, ws.data_ptr(), _stream()), 'rih_colsum')
    return out


# --------------------------------------------------------------------------------------------- conv / linear
class PackCache:
    """Packed weight operands of a training step (k > 1 convolutions: forward [(tap, ci)][co] and flipped data-gradient
    operands), kept in persistent buffers and refreshed by ONE rih_pack_conv_weight_multi launch at the start of the step
    (`refresh()`) instead of one ~6 us launch in front of every such GEMM (58 per ResNet50 step).  renderih_amd.train.TrainStep
    installs it around its step (`ops._PACK`); a request the cache has not seen is packed on the spot and joins the next
    refresh.  Valid while the weights do not change between refresh() and the last use -- one optimizer step per refresh."""

    def __init__(self):
        self.entries = {}       # key -> (weight tensor, packed tensor, descriptor fields)
        self.fresh = False

    def refresh(self):
        packs = [e for e in self.entries.values() if e[2][0] != 'h2']
        h2 = [e for e in self.entries.values() if e[2][0] == 'h2']
        if h2:          # H2 weight operands of the halo-resident 3x3 convolutions (rih_conv3x3): one launch per 48
            _h2_launch([(w, dst, f[1:]) for w, dst, f in h2])
        if packs:
            from ._lib import PackDesc
            arr = (PackDesc * len(packs))()
            for d, (w, dst, f) in zip(arr, packs):
                d.w, d.dst = w.data_ptr(), dst.data_ptr()
                (d.Cout, d.Cin, d.KH, d.KW, d.CinPad, d.mode, d.kh0, d.kw0, d.step, d.Th, d.Tw) = f
            check(_L().rih_pack_conv_weight_multi(arr, len(packs), _stream()), 'rih_pack_conv_weight_multi')
        self.fresh = True

    def stale(self):
        self.fresh = False


_PACK = None


def _packed_weight(w, Cx, for_dgrad, sub=None, out=None):
    """Weight operand of an OIHW convolution weight: forward [(tap, ci)][co] (for_dgrad False) or the flipped data-gradient
    operand of the tap subset `sub` = (kh0, kw0, step, Th, Tw) [((th, tw), co)][ci].  Through ops._PACK when one is installed."""
    Cout, Cin, KH, KW = w.shape
    kh0, kw0, step, Th, Tw = sub if sub is not None else (0, 0, 1, KH, KW)
    shape = (KH * KW * Cx, Cout) if not for_dgrad else (Th * Tw * Cout, Cx)
    pc = _PACK if out is None else None
    if pc is not None:
        # the entry holds a reference to the weight's storage, so the address cannot come back as another tensor
        key = (w.data_ptr(), Cout, Cin, KH, KW, Cx, bool(for_dgrad), kh0, kw0, step, Th, Tw)
        e = pc.entries.get(key)
        if e is not None and pc.fresh:
            return e[1]
        if e is None:
            dst = torch.empty(shape, device=w.device, dtype=torch.float32)
            pc.entries[key] = (w.detach(), dst, (Cout, Cin, KH, KW, Cx, 1 if for_dgrad else 0, kh0, kw0, step, Th, Tw))
        else:
            dst = e[1]
    else:
        dst = out if out is not None else torch.empty(shape, device=w.device, dtype=torch.float32)
    if not for_dgrad:
        check(_L().rih_pack_conv_weight(w.data_ptr(), dst.data_ptr(), Cout, Cin, KH, KW, Cx, 0, _stream()),
              'rih_pack_conv_weight')
    else:
        check(_L().rih_pack_conv_weight_sub(w.data_ptr(), dst.data_ptr(), Cout, Cin, KH, KW, Cx, kh0, kw0, step, Th, Tw,
                                            _stream()), 'rih_pack_conv_weight_sub')
    return dst


# --------------------------------------------------------------------------------------------- halo-resident 3x3 convolution
# csrc/rih_conv3.hip (round 5): stride-1 3x3 convolutions -- forward and data gradient -- on engine 2's arithmetic with the input
# halo of an 8 x 32 pixel patch loaded and converted ONCE per 32-channel chunk and the weights as pre-split fp16 planes ("H2",
# one rih_h2_multi launch per step through ops._PACK) staged by LDS-DMA.  RIH_HALO3=0: the tap-by-tap implicit GEMM (rih_gemm).
HALO3 = os.environ.get('RIH_HALO3', '1') == '1'
# the data gradient of a residual block's FIRST 3x3 convolution (BasicBlock.conv1 of HRNet: conv2d_skip) with the skip gradient
# added in the halo kernel's epilogue (rih_conv3_desc.r, ABI 19); RIH_HALO3_RES=0: those launches on rih_gemm's tiled kernel
HALO3_RES = os.environ.get('RIH_HALO3_RES', '1') == '1'


def _h2_launch(items):
    """items: [(weight, planes tensor, (Cout, Cin, KH, KW, CinPad, for_dgrad, Kpad)), ...] -> rih_h2_multi."""
    from ._lib import H2Desc
    arr = (H2Desc * len(items))()
    for d, (w, dst, f, *bound) in zip(arr, items):
        # the planes are scaled with ONE bound block, and that block is what the consuming kernel un-scales with: a caller that has
        # already resolved the weight's bound hands it in (outside owned_bounds() a Parameter is re-measured by every bound_of)
        d.w, d.dst, d.amax = w.data_ptr(), dst.data_ptr(), (bound[0] if bound and bound[0] is not None else bound_of(w)).data_ptr()
        (d.Cout, d.Cin, d.KH, d.KW, d.CinPad, d.for_dgrad, d.Kpad) = f
    check(_L().rih_h2_multi(arr, len(items), _stream()), 'rih_h2_multi')


def _h2_weight(w, Cx, for_dgrad, bound=None):
    """H2 planes ([N][Kpad / 8][2][8] fp16 as a float32 tensor [N][Kpad]) of an OIHW weight: forward operand (N = Cout,
    k = (tap, ci < Cx)) or flipped data-gradient operand (N = Cx, k = (tap, co)); scaled by bound_of(w).  Through ops._PACK when
    one is installed (TrainStep: every H2 operand of the step in one launch at its start)."""
    Cout, Cin, KH, KW = w.shape
    K = KH * KW * (Cx if not for_dgrad else Cout)
    Nn = Cout if not for_dgrad else Cx
    Kp = _cdiv(K, 32) * 32
    f = (Cout, Cin, KH, KW, Cx, 1 if for_dgrad else 0, Kp)
    pc = _PACK
    if pc is not None:
        key = ('h2', w.data_ptr()) + f
        e = pc.entries.get(key)
        if e is not None and pc.fresh:
            return e[1], Kp
        if e is None:
            planes = torch.empty((Nn, Kp), device=w.device, dtype=torch.float32)
            pc.entries[key] = (w, planes, ('h2',) + f)
        else:
            planes = e[1]
    else:
        planes = torch.empty((Nn, Kp), device=w.device, dtype=torch.float32)
    _h2_launch([(w, planes, f, bound)])
    return planes, Kp


def _halo3_ok(x, Cch, Nout, KH, KW, stride, pad, bias=None, residual=None):
    """Shapes rih_conv3x3 takes (csrc/rih_conv3.hip): engine 2, 3x3 / 1 / 1, no bias, 8 x 32 or 16 x 16 pixel patches, whole
    32-channel chunks, 32-column blocks; a residual (the skip gradient of a BasicBlock's first convolution) when HALO3_RES; the
    kernel itself re-checks (rih_conv3x3_ok)."""
    if not (HALO3 and ENGINE == 2) or KH != 3 or KW != 3 or stride != 1 or pad != 1 or bias is not None:
        return False
    if residual is not None and not (HALO3_RES and residual.is_contiguous() and residual.data_ptr() % 16 == 0):
        return False
    N, H, W_, Cx = x.shape
    return (Cx == Cch and Cch % 32 == 0 and Nout % 32 == 0 and ((H % 8 == 0 and W_ % 32 == 0) or (H % 16 == 0 and W_ % 16 == 0))
            and x.is_contiguous() and x.data_ptr() % 16 == 0 and 4 * H * W_ * Cx < (1 << 31))


def conv3x3_halo(x, w, y, for_dgrad, relu=False, stats=None, bx=None, bw=None, R=None):
    """Enqueue rih_conv3x3: y = act(conv3x3(x, w)) (for_dgrad False; x [N,H,W,Cin], y [N,H,W,Cout]) or the data gradient
    y = conv3x3(x = dy, flipped w) (for_dgrad True; x [N,H,W,Cout], y [N,H,W,Cin]).  stats: a StatsHolder, filled.  bx / bw: bound
    thunks (LazyBound) of x and w.  R: a dense [N,H,W,Nout] tensor added before the ReLU (not together with stats).  Returns False
    (nothing enqueued) when the library refuses the descriptor (rih_conv3x3_ok)."""
    from ._lib import Conv3Desc
    N, H, W_, Cch = x.shape
    Nout = y.shape[-1]
    d = Conv3Desc()
    d.x, d.y = x.data_ptr(), y.data_ptr()
    if R is not None:
        assert stats is None and tuple(R.shape) == tuple(y.shape)
        d.r, d.ldr = R.data_ptr(), Nout
    d.imgs, d.H, d.W, d.C, d.N, d.ldx, d.ldy, d.Kpad, d.relu = N, H, W_, Cch, Nout, Cch, Nout, 9 * Cch, 1 if relu else 0
    d.w_h2 = d.amax_x = d.amax_w = x.data_ptr()         # (placeholders for the library's own precondition check)
    if int(_L().rih_conv3x3_ok(C.byref(d))) != 1:       # e.g. an output view that is not 16-byte aligned: the caller takes rih_gemm
        return False
    bwv = bw() if callable(bw) else bw      # ONE bound block of w: the planes are scaled with it and the kernel un-scales with it
    planes, Kp = _h2_weight(w, (Cch if not for_dgrad else Nout), for_dgrad, bound=bwv)
    assert Kp == 9 * Cch
    d.w_h2 = planes.data_ptr()
    d.amax_x = (bx() if callable(bx) else bx).data_ptr()
    d.amax_w = bwv.data_ptr()
    if stats is not None and stats.part is None:
        stats.rows = int(_L().rih_conv3x3_stats_rows(C.byref(d)))
        assert stats.rows > 0
        stats.T = (N * H * W_) // stats.rows
        stats.part = torch.empty((stats.T, 2, Nout), device=x.device, dtype=torch.float32)
        d.stats = stats.part.data_ptr()
    flops = 2.0 * N * H * W_ * Nout * 9 * Cch
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_L().rih_conv3x3(C.byref(d), _stream()), 'rih_conv3x3')
        e1.record()
        PROFILE.append((flops, e0, e1, (N * H * W_, Nout, 9 * Cch, 1, 0, 3, 50, 1, 2)))
        return True
    check(_L().rih_conv3x3(C.byref(d), _stream()), 'rih_conv3x3')
    return True


# --------------------------------------------------------------------------------------------- short-K streaming GEMM
# csrc/rih_conv3.hip panel_kernel (round 5): the 1x1 convolutions with K = 64 / 128 on large maps (layer1 / layer2 conv3 forward,
# conv1 data gradient) as persistent workgroups that keep their slice of the H2 weight planes in LDS and stream the rows.
# RIH_PANEL=0: rih_gemm's tiled kernels.
PANEL = os.environ.get('RIH_PANEL', '1') == '1'


def _panel_ok(a2d_rows, K, N, lda, a, bias=None):
    """Launches rih_panel takes and that are worth it: engine 2, K in {64, 128}, N % 64 (128 at K = 128), whole 128-row tiles, at
    least 256 (row tile, column block) items so that every CU's persistent workgroup has work."""
    if not (PANEL and ENGINE == 2) or bias is not None or K not in (64, 128) or N % 64 != 0 or (K == 128 and N % 128 != 0):
        return False
    if a2d_rows % 128 != 0 or lda % 4 != 0 or a.data_ptr() % 16 != 0 or 4 * a2d_rows * lda >= (1 << 31):
        return False
    cap = 16384 // K
    bn = 256 if (N % 256 == 0 and cap >= 256) else 128 if (N % 128 == 0 and cap >= 128) else 64
    return (a2d_rows // (8192 // K)) * (N // bn) >= 256


def panel_gemm(a, w, c, M, N, K, lda, ldc, for_dgrad, relu=False, stats=None, R=None, ldr=0, ba=None, bw=None):
    """Enqueue rih_panel: c[M][N] = act(a[M][K] W^T (+ R)) with W = the OIHW 1x1 weight `w` as forward (n = co, k = ci) or
    data-gradient (n = ci, k = co) H2 operand.  stats: a StatsHolder, filled.  ba / bw: bound thunks of a and w.  Returns False
    (nothing enqueued) when the library refuses the descriptor (rih_panel_ok)."""
    from ._lib import PanelDesc
    Cout, Cin = w.shape[0], w.shape[1]
    d = PanelDesc()
    d.a, d.c, d.r = a.data_ptr(), c.data_ptr(), _p(R)
    d.M, d.N, d.K, d.lda, d.ldc, d.ldr, d.relu = M, N, K, lda, ldc, ldr, 1 if relu else 0
    d.w_h2 = d.amax_a = d.amax_w = a.data_ptr()         # (placeholders for the library's own precondition check)
    if int(_L().rih_panel_ok(C.byref(d))) != 1:         # e.g. a dskip view that is not 16-byte aligned: the caller takes rih_gemm
        return False
    bwv = bw() if callable(bw) else bw
    planes, Kp = _h2_weight(w, Cin, for_dgrad, bound=bwv)
    assert Kp == K and (N, K) == ((Cin, Cout) if for_dgrad else (Cout, Cin))
    d.w_h2 = planes.data_ptr()
    d.amax_a = (ba() if callable(ba) else ba).data_ptr()
    d.amax_w = bwv.data_ptr()
    if stats is not None and stats.part is None and R is None:
        stats.rows = int(_L().rih_panel_stats_rows(C.byref(d)))
        assert stats.rows > 0
        stats.T = M // stats.rows
        stats.part = torch.empty((stats.T, 2, N), device=a.device, dtype=torch.float32)
        d.stats = stats.part.data_ptr()
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_L().rih_panel(C.byref(d), _stream()), 'rih_panel')
        e1.record()
        PROFILE.append((2.0 * M * N * K, e0, e1, (M, N, K, 1, 0, 3, 51, 1, 2)))
        return True
    check(_L().rih_panel(C.byref(d), _stream()), 'rih_panel')
    return True


# --------------------------------------------------------------------------------------------- long-K plain-row GEMM
# csrc/rih_conv3.hip rows_kernel (round 6): the 1x1 convolutions with K >= ROWS_MINK on plain rows (Bottleneck.conv1 / conv3 of layer1-4
# forward, their data gradients) as 512-thread workgroups on up to 256 x 128 tiles with the H2 weight planes staged by LDS-DMA and three
# A stages.  RIH_ROWS=0: rih_gemm's tiled kernels.  RIH_ROWS_MINK: smallest reduction that goes there (shorter ones: rih_panel / tiled).
ROWS = os.environ.get('RIH_ROWS', '1') == '1'
ROWS_MINK = int(os.environ.get('RIH_ROWS_MINK', '256'))
ROWS_MIN_WGS = int(os.environ.get('RIH_ROWS_MIN_WGS', '128'))
# smallest row count: the H2 planes of a weight cost a conversion pass per step that grows with the WEIGHT (rih_h2_multi: 0.14 ms per
# step for every 1x1 weight of ResNet50 in both forms, 70 % of it layer4's), the kernel's advantage grows with rows x weight -- on the
# 8 x 8 maps (4096 rows at B = 64) it is 3-7 us per launch against ~10 us of plane conversion per weight (profiles/r06/rows/)
ROWS_MIN_M = int(os.environ.get('RIH_ROWS_MIN_M', '8192'))
# narrowest output: on N = 64 (layer1's conv1 forward and conv3 data gradient, 262144 x 256 -> 64: HBM-bound, 335 MB per launch) the tiled
# 128 x 64 kernel is the faster one -- 80 us against 91-95 us on 256 x 64 tiles (88 KB of LDS: one workgroup per CU) and 86-90 us on
# 128 x 64 tiles (two per CU); profiles/r06/rows/c18_rows_bench_n64.log, final/rows_bench.log
ROWS_MIN_N = int(os.environ.get('RIH_ROWS_MIN_N', '128'))


def _rows_ok(a2d_rows, K, N, lda, a, bias=None, c=None, R=None):
    """Launches rih_rows takes and that are worth it: engine 2, K >= ROWS_MINK in whole 32-deep tiles, N % 64, N >= ROWS_MIN_N, whole
    128-row tiles, at least ROWS_MIN_WGS workgroups of the smallest tile; the library re-checks (rih_rows_ok) before the launch."""
    if not (ROWS and ENGINE == 2) or bias is not None or K < ROWS_MINK or K % 32 != 0 or N % 64 != 0 or a2d_rows % 128 != 0:
        return False
    if a2d_rows < ROWS_MIN_M or N < ROWS_MIN_N:
        return False
    if lda % 4 != 0 or a.data_ptr() % 16 != 0 or 4 * a2d_rows * lda >= (1 << 31):
        return False
    if (c is not None and c.data_ptr() % 16 != 0) or (R is not None and R.data_ptr() % 16 != 0):
        return False
    return (a2d_rows // 128) * (N // 64) >= ROWS_MIN_WGS


def rows_gemm(a, w, c, M, N, K, lda, ldc, for_dgrad, relu=False, stats=None, R=None, ldr=0, ba=None, bw=None):
    """Enqueue rih_rows: c[M][N] = act(a[M][K] W^T (+ R)) with W = the OIHW 1x1 weight `w` as forward (n = co, k = ci) or
    data-gradient (n = ci, k = co) H2 operand.  stats: a StatsHolder, filled.  ba / bw: bound thunks of a and w.  Returns False
    (nothing enqueued) when the library refuses the descriptor: the caller then takes rih_gemm."""
    from ._lib import PanelDesc
    Cout, Cin = w.shape[0], w.shape[1]
    d = PanelDesc()
    d.a, d.c, d.r = a.data_ptr(), c.data_ptr(), _p(R)
    d.M, d.N, d.K, d.lda, d.ldc, d.ldr, d.relu = M, N, K, lda, ldc, ldr, 1 if relu else 0
    d.w_h2 = d.amax_a = d.amax_w = a.data_ptr()         # (placeholders for the library's own precondition check)
    if int(_L().rih_rows_ok(C.byref(d))) != 1:
        return False
    bwv = bw() if callable(bw) else bw                  # ONE bound block of w: the planes are scaled with it and the kernel unscales with it
    planes, Kp = _h2_weight(w, Cin, for_dgrad, bound=bwv)
    assert Kp == K and (N, K) == ((Cin, Cout) if for_dgrad else (Cout, Cin))
    d.w_h2 = planes.data_ptr()
    d.amax_a = (ba() if callable(ba) else ba).data_ptr()
    d.amax_w = bwv.data_ptr()
    if stats is not None and stats.part is None and R is None:
        stats.rows = int(_L().rih_rows_stats_rows(C.byref(d)))
        assert stats.rows > 0
        stats.T = M // stats.rows
        stats.part = torch.empty((stats.T, 2, N), device=a.device, dtype=torch.float32)
        d.stats = stats.part.data_ptr()
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_L().rih_rows(C.byref(d), _stream()), 'rih_rows')
        e1.record()
        PROFILE.append((2.0 * M * N * K, e0, e1, (M, N, K, 1, 0, 3, 52, 1, 2)))
        return True
    check(_L().rih_rows(C.byref(d), _stream()), 'rih_rows')
    return True


# --------------------------------------------------------------------------------------------- stem convolution
# csrc/rih_conv3.hip rows_kernel<STEM> (round 6): encoder.resnet.conv1 -- 7 x 7 / 2 / 3 on the 4-channel padded image -- on engine 2 with an
# im2col loader.  Before: rih_gemm's GENERAL kernel on engine 1 (Cin = 4 is not a multiple of 32: no fast path), 270-280 us per step at
# 96 TF/s.  RIH_STEM=0 restores that.
STEM = os.environ.get('RIH_STEM', '1') == '1'


def stem_conv(x, w, y, relu=False, stats=None, bx=None, bw=None):
    """Enqueue rih_stem: y [N, H/2, W/2, 64] = act(conv7x7/2/3(x [N, H, W, 4], w [64, <= 4, 7, 7])).  Returns False (nothing enqueued)
    when the shape is not the kernel's (rih_stem_ok): the caller takes rih_gemm."""
    from ._lib import Conv3Desc
    N, H, W_, Cx = x.shape
    Cout = w.shape[0]
    d = Conv3Desc()
    d.x, d.y = x.data_ptr(), y.data_ptr()
    d.imgs, d.H, d.W, d.C, d.N, d.ldx, d.ldy, d.Kpad, d.relu = N, H, W_, Cx, Cout, Cx, Cout, 224, 1 if relu else 0
    d.w_h2 = d.amax_x = d.amax_w = x.data_ptr()         # (placeholders for the library's own precondition check)
    if int(_L().rih_stem_ok(C.byref(d))) != 1:
        return False
    bwv = bw() if callable(bw) else bw
    planes, Kp = _h2_weight(w, Cx, False, bound=bwv)
    assert Kp == 224
    d.w_h2 = planes.data_ptr()
    d.amax_x = (bx() if callable(bx) else bx).data_ptr()
    d.amax_w = bwv.data_ptr()
    M = N * (H // 2) * (W_ // 2)
    if stats is not None and stats.part is None:
        stats.rows, stats.T = 64, M // 64
        stats.part = torch.empty((stats.T, 2, Cout), device=x.device, dtype=torch.float32)
        d.stats = stats.part.data_ptr()
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_L().rih_stem(C.byref(d), _stream()), 'rih_stem')
        e1.record()
        PROFILE.append((2.0 * M * Cout * 49 * Cx, e0, e1, (M, Cout, 49 * Cx, 1, 0, 3, 53, 1, 2)))
        return True
    check(_L().rih_stem(C.byref(d), _stream()), 'rih_stem')
    return True


class Conv2dFn(torch.autograd.Function):
    """NHWC conv2d (+bias, +ReLU epilogue) = implicit GEMM on the fp32 MFMA pipe.
    x: [N,H,W,Cx] (Cx >= Cin, extra channels must be zero), w: [Cout,Cin,KH,KW] (the nn.Conv2d parameter)."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, relu, skip=False, grad_masked=False, stats=None):
        """skip=True: also return x itself (an alias).  A residual block feeds its skip path from that second output;
        the gradient arriving there is then added inside the data-gradient GEMM's epilogue (residual operand R)
        instead of by an autograd accumulation pass over the whole activation."""
        _chk(x, w, bias)
        x = _c(x)
        w = _c(w)
        N, H, W_, Cx = x.shape
        Cout, Cin, KH, KW = w.shape
        Ho = (H + 2 * pad - KH) // stride + 1
        Wo = (W_ + 2 * pad - KW) // stride + 1
        y = torch.empty((N, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
        M, K = N * Ho * Wo, KH * KW * Cx
        geom = (H, W_, Cx, Ho, Wo, KH, KW, stride, 1, pad, pad)
        # engine 2: operand bounds (kept for the backward: x is the weight gradient's A operand, w the data gradient's B)
        bx, bw = (LazyBound(x), LazyBound(w)) if ENGINE == 2 else (None, None)
        rows1x1 = KH * KW == 1 and Cx == Cin and stride == 1 and pad == 0       # a plain-row GEMM on the stored [Cout][Cin] weight
        if (Cx == Cin and _halo3_ok(x, Cx, Cout, KH, KW, stride, pad, bias)
                and conv3x3_halo(x, w, y, False, relu=relu, stats=stats, bx=bx, bw=bw)):
            pass
        elif (STEM and ENGINE == 2 and KH == 7 and KW == 7 and stride == 2 and pad == 3 and Cx == 4 and Cout == 64 and bias is None
              and x.is_contiguous() and stem_conv(x, w, y, relu=relu, stats=stats, bx=bx, bw=bw)):
            pass
        elif (rows1x1 and _panel_ok(M, Cin, Cout, Cx, x, bias)
              and panel_gemm(x, w, y, M, Cout, Cin, Cx, Cout, False, relu=relu, stats=stats, ba=bx, bw=bw)):
            pass
        elif (rows1x1 and _rows_ok(M, Cin, Cout, Cx, x, bias)
              and rows_gemm(x, w, y, M, Cout, Cin, Cx, Cout, False, relu=relu, stats=stats, ba=bx, bw=bw)):
            pass
        elif KH * KW == 1 and Cx == Cin:
            gemm(x, w, y, M, Cout, K, Cx, Cin, Cout, a_mode=0, b_mode=1, bias=bias, relu=relu, geom=geom, stats=stats,
                 amax_a=bx, amax_b=bw)
        else:
            wp = _packed_weight(w, Cx, False)
            gemm(x, wp, y, M, Cout, K, Cx, Cout, Cout, a_mode=0, b_mode=0, bias=bias, relu=relu, geom=geom, stats=stats,
                 amax_a=bx, amax_b=bw)
        # grad_masked: the consumer (a BatchNorm with input_relu) hands back a gradient that is already zero where y <= 0
        relu_bwd = relu and not grad_masked
        ctx.save_for_backward(x, w, y if relu_bwd else None)
        ctx.cfg = (stride, pad, relu_bwd, bias is not None)
        ctx.bounds = (bx, bw)
        if skip:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, w, y = ctx.saved_tensors
        stride, pad, relu, has_bias = ctx.cfg
        if dskip is not None:
            dskip = _c(dskip)
        N, H, W_, Cx = x.shape
        Cout, Cin, KH, KW = w.shape
        dy = _c(dy)
        _, Ho, Wo, _ = dy.shape
        M = N * Ho * Wo
        lib = _L()
        bx, bw = ctx.bounds
        bdy = LazyBound(dy) if bx is not None else None     # (also bounds the ReLU-gated gradient below)
        if relu:
            dyr = torch.empty_like(dy)
            check(lib.rih_relu_bwd(dy.data_ptr(), y.data_ptr(), dyr.data_ptr(), dy.numel(), _stream()), 'rih_relu_bwd')
            dy = dyr
        dx = dw = db = None
        if ctx.needs_input_grad[0] and stride > 1:
            # strided conv: s*s dense sub-convolutions, one per parity class (oh, ow) of the input pixel; class rows
            # are written straight into dx (rih_gemm_desc.cS); a zero-stuffed single GEMM would multiply
            # (s*s-1)/(s*s) structural zeros
            classes = []
            for oh in range(stride):
                for ow in range(stride):
                    kh0, kw0 = (oh + pad) % stride, (ow + pad) % stride
                    Th, Tw = len(range(kh0, KH, stride)), len(range(kw0, KW, stride))
                    Hc, Wc = len(range(oh, H, stride)), len(range(ow, W_, stride))
                    if Hc > 0 and Wc > 0:
                        classes.append((oh, ow, kh0, kw0, Th, Tw, Hc, Wc))
            dense = all(c[4] > 0 and c[5] > 0 for c in classes)
            dx = torch.empty_like(x) if dense else torch.zeros_like(x)
            for oh, ow, kh0, kw0, Th, Tw, Hc, Wc in classes:
                if Th == 0 or Tw == 0:
                    continue
                padh, padw = Th - 1 - (oh + pad - kh0) // stride, Tw - 1 - (ow + pad - kw0) // stride
                geom = (Ho, Wo, Cout, Hc, Wc, Th, Tw, 1, 1, padh, padw)
                if KH * KW == 1 and Cx == Cin:
                    wd = w
                else:
                    wd = _packed_weight(w, Cx, True, (kh0, kw0, stride, Th, Tw))
                gemm(dy, wd, dx, N * Hc * Wc, Cx, Th * Tw * Cout, Cout, Cx, Cx, a_mode=0, b_mode=0, geom=geom,
                     cstride=(stride, oh, ow, H, W_), amax_a=bdy, amax_b=bw)
            if dskip is not None:
                dx = dx + dskip
        elif ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            Mx = N * H * W_
            geom = (Ho, Wo, Cout, H, W_, KH, KW, 1, stride, KH - 1 - pad, KW - 1 - pad)
            rows1x1 = KH * KW == 1 and Cx == Cin and stride == 1 and pad == 0
            if (Cx == Cin and _halo3_ok(dy, Cout, Cx, KH, KW, stride, pad, None, dskip)
                    and conv3x3_halo(dy, w, dx, True, bx=bdy, bw=bw, R=dskip)):
                pass
            elif (rows1x1 and _panel_ok(Mx, Cout, Cin, Cout, dy)
                  and panel_gemm(dy, w, dx, Mx, Cin, Cout, Cout, Cx, True, R=dskip, ldr=Cx, ba=bdy, bw=bw)):
                pass
            elif (rows1x1 and _rows_ok(Mx, Cout, Cin, Cout, dy)
                  and rows_gemm(dy, w, dx, Mx, Cin, Cout, Cout, Cx, True, R=dskip, ldr=Cx, ba=bdy, bw=bw)):
                pass
            elif KH * KW == 1 and Cx == Cin:
                gemm(dy, w, dx, Mx, Cin, Cout, Cout, Cin, Cx, a_mode=0, b_mode=0, geom=geom, R=dskip, ldr=Cx,
                     amax_a=bdy, amax_b=bw)
            else:
                wd = _packed_weight(w, Cx, True)
                gemm(dy, wd, dx, Mx, Cx, KH * KW * Cout, Cout, Cx, Cx, a_mode=0, b_mode=0, geom=geom, R=dskip, ldr=Cx,
                     amax_a=bdy, amax_b=bw)
        want_db = has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            if want_db:
                db = torch.empty((Cout,), device=x.device, dtype=torch.float32)
            geom = (H, W_, Cx, Ho, Wo, KH, KW, stride, 1, pad, pad)
            _wgrad(x, dy, dw, M, KH * KW * Cx, Cout, Cx, Cout, geom, Cx, KH * KW, Cin, db=db,
                   bounds=(bx, bdy) if bx is not None else None)
        elif want_db:
            db = colsum(dy, M, Cout)
        if dx is None and dskip is not None:
            dx = dskip
        return dx, dw, db, None, None, None, None, None, None


def conv2d(x, w, bias=None, stride=1, pad=0, relu=False, grad_masked=False, stats=None):
    """stats: optional StatsHolder that receives the BatchNorm statistics of the output from the GEMM epilogue (see gemm)."""
    return Conv2dFn.apply(x, w, bias, stride, pad, relu, False, grad_masked, stats)


def conv2d_skip(x, w, bias=None, stride=1, pad=0, relu=False, grad_masked=False, stats=None):
    """(conv(x), x) -- see Conv2dFn.forward: use the second value for the block's skip path."""
    return Conv2dFn.apply(x, w, bias, stride, pad, relu, True, grad_masked, stats)


class ConvCat1x1Fn(torch.autograd.Function):
    """1x1 convolution (no bias) of the channel concatenation of up to four NHWC maps WITHOUT materialising the concatenation
    (models/encoder.py:165-173: `torch.cat((hms_fmaps[i], dp_fmaps[i], img_fmaps[i]), dim=1)` -> conv1x1 -> ReLU -> BN): the
    forward GEMM reads its A operand piece by piece (rih_gemm_desc.a_seg), the backward writes each part's gradient with its own
    data-gradient GEMM (no slice copies) and each part's weight-gradient columns with its own (grouped) weight-gradient GEMM.
    forward(w, relu, stats, grad_masked, *parts); channel counts must be multiples of 32 (conv1x1_cat checks the preconditions and
    falls back to the concatenation)."""

    @staticmethod
    def forward(ctx, w, relu, stats, grad_masked, *parts):
        """grad_masked: the consumer (a BatchNorm with input_relu) hands back a gradient that is already zero where y <= 0; without
        it the backward gates dy by y > 0 itself (rih_relu_bwd), like Conv2dFn."""
        _chk(w, *parts)
        w = _c(w)
        parts = [_c(p_) for p_ in parts]
        N, H, W_, _ = parts[0].shape
        Cs = [p_.shape[-1] for p_ in parts]
        Cout, Cin = w.shape[0], w.shape[1]
        assert sum(Cs) == Cin and w.shape[2] == w.shape[3] == 1 and 2 <= len(parts) <= 4 and all(c % 32 == 0 for c in Cs)
        M = N * H * W_
        y = torch.empty((N, H, W_, Cout), device=w.device, dtype=torch.float32)
        starts = [sum(Cs[:i]) for i in range(len(Cs))]
        b0 = bw = None
        if ENGINE == 2:         # one bound for the whole operand: the largest of the parts'
            bs = [bound_of(p_) for p_ in parts]
            b0 = bs[0]
            for b_ in bs[1:]:
                b0 = torch.maximum(b0, b_)
            bw = LazyBound(w)
        gemm(parts[0], w, y, M, Cout, Cin, Cs[0], Cin, Cout, a_mode=0, b_mode=1, relu=relu, stats=stats, amax_a=b0, amax_b=bw,
             a_seg=[(parts[i], Cs[i], starts[i]) for i in range(1, len(parts))])
        relu_bwd = relu and not grad_masked
        ctx.save_for_backward(w, y if relu_bwd else None, *parts)
        ctx.cfg = (Cs, starts, relu_bwd)
        ctx.bounds = (bw,)
        return y

    @staticmethod
    def backward(ctx, dy):
        w, y, *parts = ctx.saved_tensors
        Cs, starts, relu_bwd = ctx.cfg
        (bw,) = ctx.bounds
        dy = _c(dy)
        N, H, W_, Cout = dy.shape
        M = N * H * W_
        Cin = w.shape[1]
        if relu_bwd:
            dyr = torch.empty_like(dy)
            check(_L().rih_relu_bwd(dy.data_ptr(), y.data_ptr(), dyr.data_ptr(), dy.numel(), _stream()), 'rih_relu_bwd')
            dy = dyr
        bdy = LazyBound(dy) if bw is not None else None
        wflat = w.view(Cout, Cin)
        dparts = []
        for i, p_ in enumerate(parts):
            if ctx.needs_input_grad[4 + i]:
                dx = torch.empty_like(p_)
                # dx_i = dy W[:, slice]: B(k = co, n = ci) = w[co][start + ci] -- b_mode 0 with pitch Cin from the slice's first column
                gemm(dy, wflat.data_ptr() + 4 * starts[i], dx, M, Cs[i], Cout, Cout, Cin, Cs[i], a_mode=0, b_mode=0,
                     amax_a=bdy, amax_b=bw)
                dparts.append(dx)
            else:
                dparts.append(None)
        dw = None
        if ctx.needs_input_grad[0]:
            dw = torch.empty_like(w)
            for i, p_ in enumerate(parts):
                g = (H, W_, Cs[i], H, W_, 1, 1, 1, 1, 0, 0)
                _wgrad(p_, dy, dw, M, Cs[i], Cout, Cs[i], Cout, g, Cs[i], 1, Cs[i],
                       bounds=(LazyBound(p_), bdy) if bw is not None else None, dw_slice=(starts[i], Cin))
        return (dw, None, None, None) + tuple(dparts)


def _cat_seg_ok(parts, w):
    """Preconditions of rih_gemm's segmented A operand (csrc/rih_gemm.hip gemm_impl: the split engines' fast path only): a split
    engine, 2..4 contiguous 16-byte aligned parts of one pixel grid whose channel counts are multiples of 32, every part and the
    weight below 2 GiB, K and N multiples of 4."""
    if ENGINE < 1 or not (2 <= len(parts) <= 4):
        return False
    shape = parts[0].shape[:-1]
    rows = 1
    for s_ in shape:
        rows *= int(s_)
    for p_ in parts:
        c = p_.shape[-1]
        if (p_.shape[:-1] != shape or c % 32 != 0 or not p_.is_contiguous() or p_.data_ptr() % 16 != 0
                or 4 * rows * c >= (1 << 31)):
            return False
    return (w.dim() == 4 and w.shape[2] == w.shape[3] == 1 and w.shape[0] % 4 == 0 and w.data_ptr() % 16 == 0
            and 4 * w.numel() < (1 << 31))


def conv1x1_cat(parts, w, relu=False, stats=None, grad_masked=None):
    """conv1x1(torch.cat(parts, channel dim), w) without the concatenation (ConvCat1x1Fn) when rih_gemm's segmented operand applies
    (_cat_seg_ok: a split engine, aligned parts with channel counts that are multiples of 32, below 2 GiB each); the concatenation
    followed by conv2d otherwise (engine 0, odd channel counts, giant maps).
    grad_masked: the consumer is a BatchNorm with input_relu, whose backward hands back a gradient already gated by y > 0 (the
    mid convolutions of models/encoder.py:165-173); default = `relu`, the only use inside this package.  With relu=True and
    grad_masked=False the backward applies the ReLU gate itself."""
    if grad_masked is None:
        grad_masked = relu
    if CAT_FREE and _cat_seg_ok(parts, w):
        return ConvCat1x1Fn.apply(w, relu, stats, bool(grad_masked), *parts)
    return conv2d(cat_channels(list(parts)), w, None, stride=1, pad=0, relu=relu, grad_masked=bool(grad_masked and relu), stats=stats)


# the 1x1 convolutions behind channel concatenations (encoder.resnet_mid, HRnet_encoder heads) read their parts in place
CAT_FREE = os.environ.get('RIH_CAT_FREE', '1') == '1'


def pack_folded_conv(w, scale, cin_pad):
    """Inference: OIHW conv weight times a per-output-channel scale (an eval BatchNorm folded in), packed once as the
    [KH*KW*cin_pad, Cout] B operand of the forward implicit GEMM."""
    _chk(w, scale)
    Cout, Cin, KH, KW = w.shape
    ws = _c(w.detach() * scale.view(-1, 1, 1, 1))
    wp = torch.empty((KH * KW * cin_pad, Cout), device=w.device, dtype=torch.float32)
    check(_L().rih_pack_conv_weight(ws.data_ptr(), wp.data_ptr(), Cout, Cin, KH, KW, cin_pad, 0, _stream()),
          'rih_pack_conv_weight')
    return wp


def conv2d_packed(x, wp, KH, KW, bias=None, stride=1, pad=0, relu=False, residual=None):
    """Inference-only convolution on a weight packed by `pack_folded_conv`: y = act(conv(x) + bias + residual) in one GEMM
    launch (no autograd graph).  x [N,H,W,Cx] with Cx = the packing's cin_pad."""
    _chk(x, wp, bias, residual)
    x = _c(x)
    N, H, W_, Cx = x.shape
    K, Cout = wp.shape
    assert K == KH * KW * Cx, (K, KH, KW, Cx)
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W_ + 2 * pad - KW) // stride + 1
    y = torch.empty((N, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
    if residual is not None:
        residual = _c(residual)
        assert residual.shape == y.shape
    gemm(x, wp, y, N * Ho * Wo, Cout, K, Cx, Cout, Cout, a_mode=0, b_mode=0, bias=bias, R=residual, ldr=Cout, relu=relu,
         geom=(H, W_, Cx, Ho, Wo, KH, KW, stride, 1, pad, pad))
    return y


# Dropout behind a Linear inside the GEMM's epilogue (rih_gemm_desc.drop_p, ABI 12) instead of an rih_add_dropout launch behind it
# (57 launches of ~6.5 us per ResNet50 step): same mask stream, so outputs and gradients equal the two-launch form bit for bit
# (tests/test_gpu_ops.py::test_linear_dropout_epilogue_is_bit_identical, green on MI355X in round 4).  Default since round 4:
# same-box 1771.7 -> 1781.9 images/s (+0.6 %, profiles/r04/ab/train_gemm_dropout.log).  RIH_GEMM_DROPOUT=0: the two-launch form.
GEMM_DROPOUT = os.environ.get('RIH_GEMM_DROPOUT', '1') == '1'


def _finish_dropout(fused, y, residual, drop):
    """After gemm(..., drop=drop): nothing to do when the epilogue took the dropout, else residual + dropout(y) by the
    stand-alone kernel (gemm left the residual out in that case)."""
    if fused:
        return y
    out = torch.empty_like(y)
    check(_L().rih_add_dropout(_p(residual), y.data_ptr(), out.data_ptr(), y.numel(), y.shape[-1], 0, drop[0], drop[1],
                               _seed_dev(), _stream()), 'rih_add_dropout')
    return out


def _dropout_grad(dy, drop):
    d = torch.empty_like(dy)
    check(_L().rih_dropout_bwd(dy.data_ptr(), d.data_ptr(), dy.numel(), drop[0], drop[1], _seed_dev(), _stream()), 'rih_dropout_bwd')
    return d


# EXPERIMENT (round 5, verdict item 6: "engine 2 for the decoder's Linears"): RIH_DECODER_E2=1 hands every nn.Linear GEMM of the
# decoder (forward, data gradient, weight gradient; LinearFn and LinearPairFn) operand bounds MEASURED ON DEMAND (bound_of: one
# rih_absmax launch per operand that nobody has measured) so that they take engine 2's kernels.  The extra launches make the
# step slower; what the switch is for is the per-launch GEMM profile (bench.py by_engine): the time those GEMMs would take on
# three fp16 products, i.e. what producer-published bounds (LayerNorm / GEMM / attention epilogues) could buy at most.
DECODER_E2 = os.environ.get('RIH_DECODER_E2', '0') == '1'


def _lin_bounds(x, w, w2=None):
    """(bound thunk of the activation, bound thunk of the weight or of a left / right weight pair) or (None, None)."""
    if not (DECODER_E2 and ENGINE == 2):
        return None, None
    if w2 is None:
        return LazyBound(x), LazyBound(w)
    return LazyBound(x), (lambda: torch.maximum(bound_of(w), bound_of(w2)))


class LinearFn(torch.autograd.Function):
    """y = act(x @ w^T + bias + residual) for x [..., K], w [N, K] (the nn.Linear parameter, read in place).
    drop = (p, seed): y = dropout(act(x @ w^T + bias)) + residual (relu and residual together are not supported)."""

    @staticmethod
    def forward(ctx, x, w, bias, residual, relu, drop=None):
        _chk(x, w, bias, residual)
        x = _c(x)
        w = _c(w)
        Nf, K = w.shape
        M = x.numel() // K
        y = torch.empty(x.shape[:-1] + (Nf,), device=x.device, dtype=torch.float32)
        if residual is not None:
            residual = _c(residual)
        bx, bw = _lin_bounds(x, w)
        if drop is not None and drop[0] > 0:
            assert not (relu and residual is not None)
            fused = gemm(x, w, y, M, Nf, K, K, K, Nf, a_mode=0, b_mode=1, bias=bias, R=residual, ldr=Nf, relu=relu, drop=drop,
                         amax_a=bx, amax_b=bw)
            y = _finish_dropout(fused, y, residual, drop)
        else:
            drop = None
            gemm(x, w, y, M, Nf, K, K, K, Nf, a_mode=0, b_mode=1, bias=bias, R=residual, ldr=Nf, relu=relu, amax_a=bx, amax_b=bw)
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.cfg = (relu, bias is not None, residual is not None, drop)
        ctx.bounds = (bx, bw)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        relu, has_bias, has_res, drop = ctx.cfg
        Nf, K = w.shape
        M = x.numel() // K
        dy = _c(dy)
        lib = _L()
        dres = dy if (has_res and ctx.needs_input_grad[3]) else None     # (with drop: the residual joins behind the dropout)
        if drop is not None:
            dy = _dropout_grad(dy, drop)
        if relu:
            dyr = torch.empty_like(dy)
            check(lib.rih_relu_bwd(dy.data_ptr(), y.data_ptr(), dyr.data_ptr(), dy.numel(), _stream()), 'rih_relu_bwd')
            dy = dyr
        dx = dw = db = None
        bx, bw = ctx.bounds
        bdy = LazyBound(dy) if bx is not None else None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            gemm(dy, w, dx, M, K, Nf, Nf, K, K, a_mode=0, b_mode=0, amax_a=bdy, amax_b=bw)
        want_db = has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            if want_db:
                db = torch.empty((Nf,), device=x.device, dtype=torch.float32)
            _wgrad(x, dy, dw, M, K, Nf, K, Nf, (1, 1, K, 1, 1, 1, 1, 1, 1, 0, 0), K, 1, K, db=db,
                   bounds=(bx, bdy) if bx is not None else None)
        elif want_db:
            db = colsum(dy, M, Nf)
        if drop is None:
            dres = dy if (has_res and ctx.needs_input_grad[3]) else None
        return dx, dw, db, dres, None, None


def linear(x, w, bias=None, residual=None, relu=False, drop=None):
    return LinearFn.apply(x, w, bias, residual, relu, drop)


# ------------------------------------------------------------------------- paired left/right-hand layers
# The decoder runs every block once per hand with separate parameters (DualGraph.py:83-89).  With the two hands'
# activations stacked [2][rows][D] each such pair of layers is ONE launch: the GEMM batch index walks from the left
# parameter tensor to the right one (rih_gemm_desc.sB1 / sBias1 = their distance in memory).
class LinearPairFn(torch.autograd.Function):
    """y[h] = act(x[h] @ w_h^T + b_h + residual[h]), h = left, right;  x [2, ..., K] -> y [2, ..., N].
    wR / bR None: wL [2, N, K] and bL [2, N] hold both hands' parameters (the stacked fused QKV operand)."""

    @staticmethod
    def forward(ctx, x, wL, wR, bL, bR, residual, relu, drop=None):
        _chk(x, wL, wR, bL, bR, residual)
        x, wL = _c(x), _c(wL)
        assert x.shape[0] == 2
        stacked = wR is None
        if stacked:
            assert wL.dim() == 3 and wL.shape[0] == 2 and bR is None
            Nf, K = wL.shape[1:]
            sW, sBias = Nf * K, (0 if bL is None else Nf)
            if bL is not None:
                bL = _c(bL)
        else:
            wR = _c(wR)
            assert wL.shape == wR.shape and (bL is None) == (bR is None)
            Nf, K = wL.shape
            sW, sBias = _pdiff(wL, wR), (0 if bL is None else _pdiff(bL, bR))
        M = x.numel() // (2 * K)
        y = torch.empty(x.shape[:-1] + (Nf,), device=x.device, dtype=torch.float32)
        if residual is not None:
            residual = _c(residual)
        bx, bw = _lin_bounds(x, wL, None if stacked else wR)
        if drop is not None and drop[0] > 0:
            assert not (relu and residual is not None)
            fused = gemm(x, wL, y, M, Nf, K, K, K, Nf, a_mode=0, b_mode=1, bias=bL, R=residual, ldr=Nf, relu=relu, nb1=2,
                         sA=(M * K, 0), sB=(sW, 0), sC=(M * Nf, 0), sBias=sBias, sR=M * Nf, drop=drop, amax_a=bx, amax_b=bw)
            y = _finish_dropout(fused, y, residual, drop)
        else:
            drop = None
            gemm(x, wL, y, M, Nf, K, K, K, Nf, a_mode=0, b_mode=1, bias=bL, R=residual, ldr=Nf, relu=relu, nb1=2,
                 sA=(M * K, 0), sB=(sW, 0), sC=(M * Nf, 0), sBias=sBias, sR=M * Nf, amax_a=bx, amax_b=bw)
        ctx.save_for_backward(x, wL, wR, y if relu else None)
        ctx.cfg = (relu, bL is not None, residual is not None, stacked, Nf, K, sW, drop)
        ctx.bounds = (bx, bw)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wL, wR, y = ctx.saved_tensors
        relu, has_bias, has_res, stacked, Nf, K, sW, drop = ctx.cfg
        M = x.numel() // (2 * K)
        dy = _c(dy)
        dres = dy if (has_res and ctx.needs_input_grad[5]) else None     # (with drop: the residual joins behind the dropout)
        if drop is not None:
            dy = _dropout_grad(dy, drop)
        if relu:
            dyr = torch.empty_like(dy)
            check(_L().rih_relu_bwd(dy.data_ptr(), y.data_ptr(), dyr.data_ptr(), dy.numel(), _stream()), 'rih_relu_bwd')
            dy = dyr
        dx = None
        bx, bw = ctx.bounds
        bdy = LazyBound(dy) if bx is not None else None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            gemm(dy, wL, dx, M, K, Nf, Nf, K, K, a_mode=0, b_mode=0, nb1=2, sA=(M * Nf, 0), sB=(sW, 0), sC=(M * K, 0),
                 amax_a=bdy, amax_b=bw)
        dw = torch.empty((2, Nf, K), device=x.device, dtype=torch.float32)
        db = torch.empty((2, Nf), device=x.device, dtype=torch.float32) if has_bias else None
        _wgrad(x, dy, dw, M, K, Nf, K, Nf, (1, 1, K, 1, 1, 1, 1, 1, 1, 0, 0), K, 1, K, db=db, nb=2, sx=M * K, sdy=M * Nf,
               bounds=(bx, bdy) if bx is not None else None)
        if drop is None:
            dres = dy if (has_res and ctx.needs_input_grad[5]) else None
        if stacked:
            return dx, dw, None, db, None, dres, None, None
        return dx, dw[0], dw[1], (db[0] if has_bias else None), (db[1] if has_bias else None), dres, None, None


def linear_pair(x, mL, mR, residual=None, relu=False, drop=None):
    """Both hands' nn.Linear (modules mL, mR) on the stacked activation x [2, ..., K]."""
    return LinearPairFn.apply(x, mL.weight, mR.weight, mL.bias, mR.bias, residual, relu, drop)


class PatchConvPairFn(torch.autograd.Function):
    """relu(conv(x, w_h) + b_h) for both hands on ONE shared NHWC map x, for the patch convolution of
    img_feat_to_grid (img_attn.py:47-62: kernel = stride = patch, no padding) -> y [2, N, g, g, Cout].
    The non-overlapping patches make the data gradient a plain GEMM dy @ W^T whose rows are whole patches; it is
    un-patchified by one permuted copy instead of stride^2 parity-class sub-convolutions per hand."""

    @staticmethod
    def forward(ctx, x, wL, wR, bL, bR):
        _chk(x, wL, wR, bL, bR)
        x, wL, wR = _c(x), _c(wL), _c(wR)
        N, H, W_, Cx = x.shape
        Cout, Cin, KH, KW = wL.shape
        assert wR.shape == wL.shape and KH == KW and H % KH == 0 and W_ % KW == 0 and Cx == Cin
        g_h, g_w = H // KH, W_ // KW
        M, K = N * g_h * g_w, KH * KW * Cx
        wp = torch.empty((2, K, Cout), device=x.device, dtype=torch.float32)
        for h, w in enumerate((wL, wR)):
            _packed_weight(w, Cx, False, out=wp[h])         # two operands of one strided GEMM: packed in place
        y = torch.empty((2, N, g_h, g_w, Cout), device=x.device, dtype=torch.float32)
        geom = (H, W_, Cx, g_h, g_w, KH, KW, KH, 1, 0, 0)
        tile = plan_gemm(M, Cout, K, 2)[0]
        bm, bn = _TILE_MN[tile]
        tiles = _cdiv(M, bm) * _cdiv(Cout, bn) * 2
        sk = max(1, min(_cdiv(K, 256), _cdiv(512, tiles))) if (K >= 1024 and tiles < 256) else 1
        kc = _cdiv(_cdiv(K, sk), 32) * 32
        sk = _cdiv(K, kc)
        if sk > 1:
            # long reduction, few output tiles (the 4x4 patches of the 32x32 map: K = 4096): split K over workgroups,
            # finish (bias, ReLU) per hand
            part = torch.empty((2, sk, M, Cout), device=x.device, dtype=torch.float32)
            gemm(x, wp, part, M, Cout, K, Cx, Cout, Cout, a_mode=0, b_mode=0, geom=geom, nb1=2, sA=(0, 0),
                 sB=(K * Cout, 0), sC=(sk * M * Cout, 0), splitk=sk, kchunk=kc, sCsplit=M * Cout, tile=tile)
            for h, b in enumerate((bL, bR)):
                check(_L().rih_splitk_finish(part[h].data_ptr(), sk, M, Cout, y[h].data_ptr(), Cout, b.data_ptr(), 0, 0,
                                             1.0, 1, _stream()), 'rih_splitk_finish')
        else:
            gemm(x, wp, y, M, Cout, K, Cx, Cout, Cout, a_mode=0, b_mode=0, bias=bL, relu=True, geom=geom, nb1=2,
                 sA=(0, 0), sB=(K * Cout, 0), sC=(M * Cout, 0), sBias=_pdiff(bL, bR), tile=tile)
        ctx.save_for_backward(x, wp, y)
        ctx.cfg = (tuple(wL.shape), geom)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wp, y = ctx.saved_tensors
        (Cout, Cin, KH, KW), geom = ctx.cfg
        N, H, W_, Cx = x.shape
        g_h, g_w = H // KH, W_ // KW
        M, K = N * g_h * g_w, KH * KW * Cx
        dy = _c(dy)
        dyr = torch.empty_like(dy)
        check(_L().rih_relu_bwd(dy.data_ptr(), y.data_ptr(), dyr.data_ptr(), dy.numel(), _stream()), 'rih_relu_bwd')
        dx = None
        if ctx.needs_input_grad[0]:
            dxp = torch.empty((2, M, K), device=x.device, dtype=torch.float32)       # rows = patches, cols = (kh, kw, ci)
            gemm(dyr, wp, dxp, M, K, Cout, Cout, Cout, K, a_mode=0, b_mode=1, nb1=2, sA=(M * Cout, 0),
                 sB=(K * Cout, 0), sC=(M * K, 0))
            dx = (dxp[0] + dxp[1]).view(N, g_h, g_w, KH, KW, Cx).permute(0, 1, 3, 2, 4, 5).reshape(N, H, W_, Cx)
        dw = torch.empty((2, Cout, Cin, KH, KW), device=x.device, dtype=torch.float32)
        db = torch.empty((2, Cout), device=x.device, dtype=torch.float32)
        _wgrad(x, dyr, dw, M, K, Cout, Cx, Cout, geom, Cx, KH * KW, Cin, db=db, nb=2, sx=0, sdy=M * Cout)
        return dx, dw[0], dw[1], db[0], db[1]


def patch_conv_pair(x, cL, cR):
    return PatchConvPairFn.apply(x, cL.weight, cR.weight, cL.bias, cR.bias)


# --------------------------------------------------------------------------------------------- batch norm
class BatchNormFn(torch.autograd.Function):
    """nn.BatchNorm2d on NHWC rows (+ residual add + ReLU).  Training: batch statistics, running buffers updated
    in place (momentum 0.1, unbiased running_var) exactly like torch; eval: running statistics."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, residual, training, relu, eps, momentum, tile_stats=None, input_relu=False,
                ybound=None):
        _chk(x, gamma, beta, rmean, rvar, residual)
        x = _c(x)
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        lib = _L()
        mean = torch.empty((Cc,), device=x.device, dtype=torch.float32)
        invstd = torch.empty((Cc,), device=x.device, dtype=torch.float32)
        ws = torch.empty((int(lib.rih_bn_ws_floats(rows, Cc)),), device=x.device, dtype=torch.float32)
        y = torch.empty_like(x)
        if residual is not None:
            residual = _c(residual)
        # the backward needs only the sign pattern of a ReLU'd output: one byte per quad (rih_bn_apply relu_mask)
        mask = torch.empty((x.numel() // 4,), device=x.device, dtype=torch.uint8) if relu else None
        # ybound (engine 2): the apply pass leaves max|y| there -- the operand bound of the convolution that reads y (bound_of)

        def run():
            if training and tile_stats is not None and tile_stats[0] == 'blocks':   # rih_gemm's statistics epilogue
                _, part, T, rpb = tile_stats
                assert part.shape[2] == Cc
                check(lib.rih_bn_stats_from_blocks(part.data_ptr(), T, Cc, rows, rpb, eps, momentum, mean.data_ptr(),
                                                   invstd.data_ptr(), _p(rmean), _p(rvar), _stream()), 'rih_bn_stats_from_blocks')
            elif training:
                assert tile_stats is None
                check(lib.rih_bn_stats(x.data_ptr(), rows, Cc, eps, momentum, mean.data_ptr(), invstd.data_ptr(),
                                       _p(rmean), _p(rvar), ws.data_ptr(), _stream()), 'rih_bn_stats')
            else:
                check(lib.rih_bn_eval_stats(rmean.data_ptr(), rvar.data_ptr(), Cc, eps, mean.data_ptr(),
                                            invstd.data_ptr(), _stream()), 'rih_bn_eval_stats')
            check(lib.rih_bn_apply(x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                   _p(residual), y.data_ptr(), rows, Cc, 1 if relu else 0, _p(mask), _p(ybound), _stream()),
                  'rih_bn_apply')
        # algorithmic bytes: statistics read x once (training), apply reads x (+ residual) and writes y
        _elem_profile(x.numel() * (4.0 * ((1 if training and tile_stats is None else 0) + 2 + (1 if residual is not None else 0))
                                   + (0.25 if relu else 0.0)), 'bn_fwd', run)
        ctx.save_for_backward(x, mask, mean, invstd, gamma)
        ctx.cfg = (training, relu, residual is not None, bool(input_relu))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mask, mean, invstd, gamma = ctx.saved_tensors
        training, relu, has_res, input_relu = ctx.cfg
        dy = _c(dy)
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        lib = _L()
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res else None
        dg = torch.empty_like(gamma)
        db = torch.empty_like(gamma)
        ws = torch.empty((int(lib.rih_bn_ws_floats(rows, Cc)),), device=x.device, dtype=torch.float32)
        # algorithmic bytes: reduction pass reads dy and x (+ 1 byte per quad of ReLU pattern), apply reads them again and
        # writes dx (+ dres)
        flags = (0 if training else 1) | (2 if input_relu else 0)
        dxbound = bound_slot(x.device) if ENGINE == 2 else None     # max|dx|: the gradient operand's bound, as ybound above
        run = lambda: check(
            lib.rih_bn_bwd(dy.data_ptr(), x.data_ptr(), 0, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                           dx.data_ptr(), _p(dres), dg.data_ptr(), db.data_ptr(), rows, Cc, 1 if relu else 0,
                           flags, ws.data_ptr(), _p(mask), _p(dxbound), _stream()), 'rih_bn_bwd')
        _elem_profile(x.numel() * (4.0 * (2 * 2 + 1 + (1 if has_res else 0)) + (0.5 if relu else 0.0)), 'bn_bwd', run)
        if dxbound is not None:
            set_bound(dx, dxbound)
        return dx, dg, db, None, None, dres, None, None, None, None, None, None, None


def batchnorm_update_only(x, rmean, rvar, eps=1e-5, momentum=0.1, tile_stats=None):
    """The side effect of a training-mode nn.BatchNorm2d WITHOUT its output: batch statistics of x (from the producing GEMM's
    epilogue blocks when `tile_stats` carries them, by a statistics pass otherwise) folded into the running buffers exactly as
    BatchNormFn.forward does.  For a BatchNorm whose output nobody reads (encoder.resnet_mid, `drop_last`)."""
    _chk(x, rmean, rvar)
    x = _c(x)
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    lib = _L()
    with torch.no_grad():
        mean = torch.empty((Cc,), device=x.device, dtype=torch.float32)
        invstd = torch.empty((Cc,), device=x.device, dtype=torch.float32)
        if tile_stats is not None and tile_stats[0] == 'blocks':
            _, part, T, rpb = tile_stats
            assert part.shape[2] == Cc
            check(lib.rih_bn_stats_from_blocks(part.data_ptr(), T, Cc, rows, rpb, eps, momentum, mean.data_ptr(),
                                               invstd.data_ptr(), _p(rmean), _p(rvar), _stream()), 'rih_bn_stats_from_blocks')
        else:
            assert tile_stats is None
            ws = torch.empty((int(lib.rih_bn_ws_floats(rows, Cc)),), device=x.device, dtype=torch.float32)
            check(lib.rih_bn_stats(x.data_ptr(), rows, Cc, eps, momentum, mean.data_ptr(), invstd.data_ptr(),
                                   _p(rmean), _p(rvar), ws.data_ptr(), _stream()), 'rih_bn_stats')


def batchnorm(x, gamma, beta, rmean, rvar, residual=None, training=True, relu=False, eps=1e-5, momentum=0.1, tile_stats=None,
              input_relu=False):
    """input_relu: x is the output of a ReLU (Conv -> ReLU -> BN); the gradient wrt x then leaves already gated by x > 0."""
    ybound = bound_slot(x.device) if ENGINE == 2 else None
    y = BatchNormFn.apply(x, gamma, beta, rmean, rvar, residual, training, relu, eps, momentum, tile_stats, input_relu, ybound)
    return y if ybound is None else set_bound(y, ybound)


# --------------------------------------------------------------------------------------------- layout / pooling
def nchw_to_nhwc(x, cpad=None):
    """Input-side layout change (no gradient: the image is a leaf input)."""
    _chk(x)
    x = _c(x)
    N, Cc, H, W_ = x.shape
    cpad = Cc if cpad is None else cpad
    y = torch.empty((N, H, W_, cpad), device=x.device, dtype=torch.float32)
    check(_L().rih_nchw_to_nhwc(x.data_ptr(), y.data_ptr(), N, Cc, H, W_, cpad, _stream()), 'rih_nchw_to_nhwc')
    return y


class NhwcToNchwFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c0, c1):
        _chk(x)
        x = _c(x)
        N, H, W_, Cx = x.shape
        y = torch.empty((N, c1 - c0, H, W_), device=x.device, dtype=torch.float32)
        check(_L().rih_nhwc_to_nchw(x.data_ptr() + 4 * c0, y.data_ptr(), N, c1 - c0, H, W_, Cx, _stream()),
              'rih_nhwc_to_nchw')
        ctx.cfg = (c0, c1, Cx)
        return y

    @staticmethod
    def backward(ctx, dy):
        c0, c1, Cx = ctx.cfg
        dy = _c(dy)
        N, Cc, H, W_ = dy.shape
        if c0 == 0 and c1 == Cx:
            dx = torch.empty((N, H, W_, Cx), device=dy.device, dtype=torch.float32)
            check(_L().rih_nchw_to_nhwc(dy.data_ptr(), dx.data_ptr(), N, Cc, H, W_, Cc, _stream()), 'rih_nchw_to_nhwc')
            return dx, None, None
        part = torch.empty((N, H, W_, Cc), device=dy.device, dtype=torch.float32)
        check(_L().rih_nchw_to_nhwc(dy.data_ptr(), part.data_ptr(), N, Cc, H, W_, Cc, _stream()), 'rih_nchw_to_nhwc')
        dx = torch.zeros((N, H, W_, Cx), device=dy.device, dtype=torch.float32)
        dx[..., c0:c1] = part
        return dx, None, None


def nhwc_to_nchw(x, c0=0, c1=None):
    """[N,H,W,C] -> [N,c1-c0,H,W] (user-facing outputs hms / mask / dense)."""
    return NhwcToNchwFn.apply(x, c0, x.shape[-1] if c1 is None else c1)


class MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        x = _c(x)
        N, H, W_, Cc = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W_ - 1) // 2 + 1
        y = torch.empty((N, Ho, Wo, Cc), device=x.device, dtype=torch.float32)
        arg = torch.empty((N, Ho, Wo, Cc), device=x.device, dtype=torch.int8)
        check(_L().rih_maxpool3x3s2_fwd(x.data_ptr(), y.data_ptr(), arg.data_ptr(), N, H, W_, Cc, _stream()),
              'rih_maxpool3x3s2_fwd')
        ctx.save_for_backward(arg)
        ctx.shape = (N, H, W_, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        N, H, W_, Cc = ctx.shape
        dy = _c(dy)
        dx = torch.empty((N, H, W_, Cc), device=dy.device, dtype=torch.float32)
        check(_L().rih_maxpool3x3s2_bwd(dy.data_ptr(), arg.data_ptr(), dx.data_ptr(), N, H, W_, Cc, _stream()),
              'rih_maxpool3x3s2_bwd')
        return dx


def maxpool3x3s2(x):
    return inherit_bound(MaxPoolFn.apply(x), x)         # a maximum of window values


class AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        x = _c(x)
        N, H, W_, Cc = x.shape
        y = torch.empty((N, Cc), device=x.device, dtype=torch.float32)
        check(_L().rih_avgpool_fwd(x.data_ptr(), y.data_ptr(), N, H * W_, Cc, _stream()), 'rih_avgpool_fwd')
        ctx.shape = (N, H, W_, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H, W_, Cc = ctx.shape
        dy = _c(dy)
        dx = torch.empty((N, H, W_, Cc), device=dy.device, dtype=torch.float32)
        check(_L().rih_avgpool_bwd(dy.data_ptr(), dx.data_ptr(), N, H * W_, Cc, _stream()), 'rih_avgpool_bwd')
        return dx


def global_avgpool(x):
    return AvgPoolFn.apply(x)


class UpsampleBilinearFn(torch.autograd.Function):
    """Bilinear x`factor`, align_corners=True (nn.Upsample / F.interpolate of models/encoder.py:31,228-230), NHWC."""

    @staticmethod
    def forward(ctx, x, factor):
        _chk(x)
        x = _c(x)
        N, H, W_, Cc = x.shape
        y = torch.empty((N, factor * H, factor * W_, Cc), device=x.device, dtype=torch.float32)
        check(_L().rih_upsample_bilinear_fwd(x.data_ptr(), y.data_ptr(), N, H, W_, Cc, factor, _stream()),
              'rih_upsample_bilinear_fwd')
        ctx.shape = (N, H, W_, Cc, factor)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H, W_, Cc, factor = ctx.shape
        dy = _c(dy)
        dx = torch.empty((N, H, W_, Cc), device=dy.device, dtype=torch.float32)
        check(_L().rih_upsample_bilinear_bwd(dy.data_ptr(), dx.data_ptr(), N, H, W_, Cc, factor, _stream()),
              'rih_upsample_bilinear_bwd')
        return dx, None


def upsample_bilinear(x, factor):
    return inherit_bound(UpsampleBilinearFn.apply(x, factor), x)        # convex combinations of input values


def upsample_bilinear2x(x):
    return inherit_bound(UpsampleBilinearFn.apply(x, 2), x)


class NearestUpAddFn(torch.autograd.Function):
    """y = acc + nearest_upsample(x, factor) on NHWC tensors (acc may be None): the cross-resolution sum of
    HighResolutionModule's fuse layers (models/model_zoo/hrnet.py:181-183, 229-236) in one pass."""

    @staticmethod
    def forward(ctx, x, acc, factor):
        _chk(x, acc)
        x = _c(x)
        N, H, W_, Cc = x.shape
        if acc is not None:
            acc = _c(acc)
            assert tuple(acc.shape) == (N, factor * H, factor * W_, Cc)
        y = torch.empty((N, factor * H, factor * W_, Cc), device=x.device, dtype=torch.float32)
        check(_L().rih_nearest_up_add_fwd(x.data_ptr(), _p(acc), y.data_ptr(), N, H, W_, Cc, factor, _stream()),
              'rih_nearest_up_add_fwd')
        ctx.shape = (N, H, W_, Cc, factor, acc is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H, W_, Cc, factor, has_acc = ctx.shape
        dy = _c(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((N, H, W_, Cc), device=dy.device, dtype=torch.float32)
            check(_L().rih_nearest_up_bwd(dy.data_ptr(), dx.data_ptr(), N, H, W_, Cc, factor, _stream()),
                  'rih_nearest_up_bwd')
        return dx, (dy if (has_acc and ctx.needs_input_grad[1]) else None), None


def nearest_up_add(x, acc, factor):
    return NearestUpAddFn.apply(x, acc, factor)


# --------------------------------------------------------------------------------------------- row-wise ops
class LayerNormFn(torch.autograd.Function):
    """y = act(LayerNorm(x (+ x2))); the optional second input fuses the residual add in front of the norm.
    skip=True additionally returns x itself (an alias) as a second output: the caller uses it for the skip connection
    around the normalised branch (x + f(LN(x))), and the gradient arriving there is added to dx inside the backward
    kernel instead of by an autograd accumulation pass."""

    @staticmethod
    def forward(ctx, x, x2, g, b, eps, relu, skip):
        _chk(x, x2, g, b)
        x = _c(x)
        if x2 is not None:
            x2 = _c(x2)
        D = x.shape[-1]
        rows = x.numel() // D
        y = torch.empty_like(x)
        mean = torch.empty((rows,), device=x.device, dtype=torch.float32)
        rstd = torch.empty((rows,), device=x.device, dtype=torch.float32)
        check(_L().rih_layernorm_fwd(x.data_ptr(), _p(x2), g.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                     rstd.data_ptr(), rows, D, eps, 1 if relu else 0, _stream()), 'rih_layernorm_fwd')
        ctx.save_for_backward(x, x2, y if relu else None, g, mean, rstd)
        ctx.relu = relu
        ctx.skip = skip
        if skip:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, x2, y, g, mean, rstd = ctx.saved_tensors
        dy = _c(dy)
        if dskip is not None:
            dskip = _c(dskip)
        D = x.shape[-1]
        rows = x.numel() // D
        lib = _L()
        dx = torch.empty_like(x)
        dg = torch.empty_like(g)
        db = torch.empty_like(g)
        nblk = lib.rih_ln_nblk(rows)
        ws = torch.empty((2 * nblk * D,), device=x.device, dtype=torch.float32)
        defer = _DEFERRED is not None
        check(lib.rih_layernorm_bwd(dy.data_ptr(), x.data_ptr(), _p(x2), _p(y), g.data_ptr(), mean.data_ptr(),
                                    rstd.data_ptr(), _p(dskip), dx.data_ptr(), 0 if defer else dg.data_ptr(),
                                    0 if defer else db.data_ptr(), rows, D, 1 if ctx.relu else 0, ws.data_ptr(), _stream()),
              'rih_layernorm_bwd')
        if defer:
            _DEFERRED_LN.append((ws, ws.data_ptr(), dg, dg.data_ptr(), db, db.data_ptr(), D, nblk))
        return dx, (dx if x2 is not None else None), dg, db, None, None, None


def layernorm(x, g, b, eps=1e-6, x2=None, relu=False):
    return LayerNormFn.apply(x, x2, g, b, eps, relu, False)


def layernorm_skip(x, g, b, eps=1e-6):
    """(LayerNorm(x), x) -- see LayerNormFn: use the second value for the skip connection around the branch."""
    return LayerNormFn.apply(x, None, g, b, eps, False, True)


class LayerNormPairFn(torch.autograd.Function):
    """LayerNormFn for both hands in one launch: x (and x2) stacked [2, ..., D], parameters (gL, bL) / (gR, bR)."""

    @staticmethod
    def forward(ctx, x, x2, gL, gR, bL, bR, eps, relu, skip):
        _chk(x, x2, gL, gR, bL, bR)
        x = _c(x)
        if x2 is not None:
            x2 = _c(x2)
        assert x.shape[0] == 2
        D = x.shape[-1]
        rows = x.numel() // (2 * D)
        y = torch.empty_like(x)
        mean = torch.empty((2 * rows,), device=x.device, dtype=torch.float32)
        rstd = torch.empty((2 * rows,), device=x.device, dtype=torch.float32)
        check(_L().rih_layernorm_fwd_grouped(x.data_ptr(), _p(x2), gL.data_ptr(), bL.data_ptr(), y.data_ptr(),
                                             mean.data_ptr(), rstd.data_ptr(), 2, rows, D, _pdiff(gL, gR), _pdiff(bL, bR),
                                             eps, 1 if relu else 0, _stream()), 'rih_layernorm_fwd_grouped')
        ctx.save_for_backward(x, x2, y if relu else None, gL, gR, mean, rstd)
        ctx.relu = relu
        if skip:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, x2, y, gL, gR, mean, rstd = ctx.saved_tensors
        dy = _c(dy)
        if dskip is not None:
            dskip = _c(dskip)
        D = x.shape[-1]
        rows = x.numel() // (2 * D)
        lib = _L()
        dx = torch.empty_like(x)
        dg = torch.empty((2, D), device=x.device, dtype=torch.float32)
        db = torch.empty((2, D), device=x.device, dtype=torch.float32)
        nblk = lib.rih_ln_nblk(rows)
        ws = torch.empty((2 * 2 * nblk * D,), device=x.device, dtype=torch.float32)
        defer = _DEFERRED is not None
        check(lib.rih_layernorm_bwd_grouped(dy.data_ptr(), x.data_ptr(), _p(x2), _p(y), gL.data_ptr(), mean.data_ptr(),
                                            rstd.data_ptr(), _p(dskip), dx.data_ptr(), 0 if defer else dg.data_ptr(),
                                            0 if defer else db.data_ptr(), 2, rows, D, _pdiff(gL, gR),
                                            1 if ctx.relu else 0, ws.data_ptr(), _stream()), 'rih_layernorm_bwd_grouped')
        if defer:
            for h in (0, 1):
                _DEFERRED_LN.append((ws, ws.data_ptr() + 4 * h * nblk * 2 * D, dg, dg.data_ptr() + 4 * h * D, db,
                                     db.data_ptr() + 4 * h * D, D, nblk))
        return dx, (dx if x2 is not None else None), dg[0], dg[1], db[0], db[1], None, None, None


def layernorm_pair(x, mL, mR, x2=None, relu=False):
    return LayerNormPairFn.apply(x, x2, mL.weight, mR.weight, mL.bias, mR.bias, mL.eps, relu, False)


def layernorm_pair_skip(x, mL, mR):
    return LayerNormPairFn.apply(x, None, mL.weight, mR.weight, mL.bias, mR.bias, mL.eps, False, True)


# One-launch attention forward (csrc/rih_attn.hip) instead of QK^T GEMM + softmax + PV GEMM.  OFF by default: measured in
# round 2 at 46.7 ms per step against 45.2 (profiles/r02/bench_m1_fusedattn.log).
FUSED_ATTN = os.environ.get('RIH_FUSED_ATTN', '0') == '1'


# Attention without a score matrix in memory (csrc/rih_flash.hip): forward one launch, backward two; what is kept for the
# backward is the output and one log-sum-exp word per query row instead of two [B, heads, Sq, Sk] probability tensors.  The
# default; RIH_FLASH_ATTN=0 = the three-kernel sequence (batched QK^T GEMM, softmax, batched PV GEMM; five launches backward).
FLASH_ATTN = os.environ.get('RIH_FLASH_ATTN', '1') == '1'


def _flash_ok(d, B, heads):
    return FLASH_ATTN and not FUSED_ATTN and d in (16, 32, 64) and B * heads <= 65535


def _profiled(flops, tag, fn):
    """Run one MFMA-family launch; with ops.PROFILE set, bracket it with events like rih_gemm launches."""
    if PROFILE is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    PROFILE.append((flops, e0, e1, tag))


def _attn_forward(q, q_ld, k, v, kv_ld, B, Sq, Sk, D, heads, drop_p, seed, device, out=None):
    """q/k/v: raw device pointers to the first element of [B,S,*] slices with row pitch q_ld / kv_ld.
    Returns (out, P, Pd) for _attn_backward: the probabilities before / after dropout, or -- on the flash path -- (out, lse, None)
    with lse [B, heads, Sq]; the flash backward also needs `out`."""
    d = D // heads
    ldP = _cdiv(Sk, 4) * 4
    alpha = 1.0 / math.sqrt(d)
    if _flash_ok(d, B, heads):
        if out is None:
            out = torch.empty((B, Sq, D), device=device, dtype=torch.float32)
        lse = torch.empty((B, heads, Sq), device=device, dtype=torch.float32)
        ptr = lambda t: t if isinstance(t, int) else t.data_ptr()
        _profiled(4.0 * B * heads * Sq * Sk * d, (Sq, Sk, d, B * heads, 0, 1, 30, 1, 0), lambda: check(
            _L().rih_flash_attention_fwd(ptr(q), q_ld, ptr(k), ptr(v), kv_ld, B, heads, Sq, Sk, d, alpha, drop_p, seed,
                                         _seed_dev(), out.data_ptr(), D, lse.data_ptr(), _stream()), 'rih_flash_attention_fwd'))
        return out, lse, None
    P = torch.empty((B, heads, Sq, ldP), device=device, dtype=torch.float32)
    if FUSED_ATTN and d in (16, 32, 64) and Sk <= 320 and B * heads <= 65535:
        Pd = torch.empty_like(P) if drop_p > 0 else P
        if out is None:
            out = torch.empty((B, Sq, D), device=device, dtype=torch.float32)
        ptr = lambda t: t if isinstance(t, int) else t.data_ptr()
        check(_L().rih_attention_fwd_fused(ptr(q), q_ld, ptr(k), ptr(v), kv_ld, B, heads, Sq, Sk, d, alpha, drop_p, seed,
                                           _seed_dev(), P.data_ptr(), Pd.data_ptr(), ldP, out.data_ptr(), D, _stream()),
              'rih_attention_fwd_fused')
        return out, P, (Pd if drop_p > 0 else None)
    gemm(q, k, P, Sq, Sk, d, q_ld, kv_ld, ldP, a_mode=0, b_mode=1, nb1=B, nb2=heads, sA=(Sq * q_ld, d),
         sB=(Sk * kv_ld, d), sC=(heads * Sq * ldP, Sq * ldP), alpha=alpha)
    Pd = torch.empty_like(P) if drop_p > 0 else P
    check(_L().rih_softmax_fwd(P.data_ptr(), P.data_ptr(), Pd.data_ptr(), B * heads * Sq, Sk, ldP, drop_p, seed,
                               _seed_dev(), _stream()), 'rih_softmax_fwd')
    if out is None:
        out = torch.empty((B, Sq, D), device=device, dtype=torch.float32)
    gemm(Pd, v, out, Sq, d, Sk, ldP, kv_ld, D, a_mode=0, b_mode=0, nb1=B, nb2=heads,
         sA=(heads * Sq * ldP, Sq * ldP), sB=(Sk * kv_ld, d), sC=(Sq * D, d))
    return out, P, (Pd if drop_p > 0 else None)


def _attn_backward(do, q, q_ld, k, v, kv_ld, dq, dq_ld, dk, dv, dkv_ld, P, Pd, B, Sq, Sk, D, heads, drop_p, seed, out=None):
    """Writes dq / dk / dv (raw pointers, row pitches dq_ld / dkv_ld) given do [B,Sq,D] (contiguous tensor).
    P [B, heads, Sq] (3-d) = the log-sum-exp words of the flash forward, which also needs its output `out` [B,Sq,D]."""
    d = D // heads
    alpha = 1.0 / math.sqrt(d)
    ptr = lambda t: t if isinstance(t, int) else t.data_ptr()
    if P.dim() == 3:
        assert out is not None and out.is_contiguous() and tuple(out.shape) == (B, Sq, D)
        ws = torch.empty_like(P)
        _profiled(14.0 * B * heads * Sq * Sk * d, (Sq, Sk, d, B * heads, 1, 1, 30, 1, 0), lambda: check(
            _L().rih_flash_attention_bwd(do.data_ptr(), D, out.data_ptr(), D, ptr(q), q_ld, ptr(k), ptr(v), kv_ld, B, heads, Sq,
                                         Sk, d, alpha, drop_p, seed, _seed_dev(), P.data_ptr(), ws.data_ptr(), ptr(dq), dq_ld,
                                         ptr(dk), ptr(dv), dkv_ld, _stream()), 'rih_flash_attention_bwd'))
        return
    if Pd is None:
        Pd = P
    ldP = P.shape[-1]
    sP = (heads * Sq * ldP, Sq * ldP)
    if FUSED_ATTN and d in (16, 32, 64) and Sk <= 320 and B * heads <= 65535:
        dS = torch.empty_like(P)
        check(_L().rih_attention_bwd_dq_fused(do.data_ptr(), D, ptr(k), ptr(v), kv_ld, B, heads, Sq, Sk, d, alpha, drop_p,
                                              seed, _seed_dev(), P.data_ptr(), dS.data_ptr(), ldP, ptr(dq), dq_ld,
                                              _stream()), 'rih_attention_bwd_dq_fused')
        check(_L().rih_attention_bwd_dkv_fused(do.data_ptr(), D, ptr(q), q_ld, B, heads, Sq, Sk, d, Pd.data_ptr(),
                                               dS.data_ptr(), ldP, ptr(dk), ptr(dv), dkv_ld, _stream()),
              'rih_attention_bwd_dkv_fused')
        return
    else:
        dS = torch.empty_like(P)
        gemm(do, v, dS, Sq, Sk, d, D, kv_ld, ldP, a_mode=0, b_mode=1, nb1=B, nb2=heads, sA=(Sq * D, d),
             sB=(Sk * kv_ld, d), sC=sP)
        check(_L().rih_softmax_bwd(P.data_ptr(), dS.data_ptr(), B * heads * Sq, Sk, ldP, drop_p, seed, _seed_dev(),
                                   alpha, _stream()), 'rih_softmax_bwd')
        gemm(dS, k, dq, Sq, d, Sk, ldP, kv_ld, dq_ld, a_mode=0, b_mode=0, nb1=B, nb2=heads, sA=sP, sB=(Sk * kv_ld, d),
             sC=(Sq * dq_ld, d))
    gemm(dS, q, dk, Sk, d, Sq, ldP, q_ld, dkv_ld, a_mode=1, b_mode=0, nb1=B, nb2=heads, sA=sP, sB=(Sq * q_ld, d),
         sC=(Sk * dkv_ld, d))
    gemm(Pd, do, dv, Sk, d, Sq, ldP, D, dkv_ld, a_mode=1, b_mode=0, nb1=B, nb2=heads, sA=sP, sB=(Sq * D, d),
         sC=(Sk * dkv_ld, d))


class AttentionFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(d)) [dropout] v over `heads` heads; q [B,Sq,D], k/v [B,Sk,D] (heads split the last dim).
    QK^T / PV and their gradients are batched (batch x head) fp32-MFMA GEMMs reading the head slices in place; the
    row softmax (+ dropout) is one wavefront per row with xor-shuffle reductions."""

    @staticmethod
    def forward(ctx, q, k, v, heads, drop_p, seed):
        _chk(q, k, v)
        q, k, v = _c(q), _c(k), _c(v)
        B, Sq, D = q.shape
        Sk = k.shape[1]
        out, P, Pd = _attn_forward(q.data_ptr(), D, k.data_ptr(), v.data_ptr(), D, B, Sq, Sk, D, heads, drop_p, seed,
                                   q.device)
        ctx.save_for_backward(q, k, v, P, Pd, out if P.dim() == 3 else None)
        ctx.cfg = (heads, drop_p, seed)
        return out

    @staticmethod
    def backward(ctx, do):
        q, k, v, P, Pd, out = ctx.saved_tensors
        heads, drop_p, seed = ctx.cfg
        do = _c(do)
        B, Sq, D = q.shape
        Sk = k.shape[1]
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        _attn_backward(do, q.data_ptr(), D, k.data_ptr(), v.data_ptr(), D, dq.data_ptr(), D, dk.data_ptr(),
                       dv.data_ptr(), D, P, Pd, B, Sq, Sk, D, heads, drop_p, seed, out=out)
        return dq, dk, dv, None, None, None


def attention(q, k, v, heads, drop_p=0.0, seed=0):
    return AttentionFn.apply(q, k, v, heads, drop_p, seed)


class SelfAttentionPackedFn(torch.autograd.Function):
    """Self attention on a packed projection qkv [B,S,3D] (columns [q | k | v], the output of one fused QKV GEMM);
    head slices are read in place with row pitch 3D and the gradient is written straight into one [B,S,3D] buffer."""

    @staticmethod
    def forward(ctx, qkv, heads, drop_p, seed):
        _chk(qkv)
        qkv = _c(qkv)
        B, S, D3 = qkv.shape
        D = D3 // 3
        p0 = qkv.data_ptr()
        out, P, Pd = _attn_forward(p0, D3, p0 + 4 * D, p0 + 8 * D, D3, B, S, S, D, heads, drop_p, seed, qkv.device)
        ctx.save_for_backward(qkv, P, Pd, out if P.dim() == 3 else None)
        ctx.cfg = (heads, drop_p, seed)
        return out

    @staticmethod
    def backward(ctx, do):
        qkv, P, Pd, out = ctx.saved_tensors
        heads, drop_p, seed = ctx.cfg
        do = _c(do)
        B, S, D3 = qkv.shape
        D = D3 // 3
        dqkv = torch.empty_like(qkv)
        p0, g0 = qkv.data_ptr(), dqkv.data_ptr()
        _attn_backward(do, p0, D3, p0 + 4 * D, p0 + 8 * D, D3, g0, D3, g0 + 4 * D, g0 + 8 * D, D3, P, Pd, B, S, S, D,
                       heads, drop_p, seed, out=out)
        return dqkv, None, None, None


def self_attention_packed(qkv, heads, drop_p=0.0, seed=0):
    return SelfAttentionPackedFn.apply(qkv, heads, drop_p, seed)


class CrossAttentionPackedFn(torch.autograd.Function):
    """The cross-hand attention pair of inter_attn.py:93-107 on packed projections Lqkv, Rqkv [B,V,3D]:
    feat_R2L = softmax(Lq Rk^T / sqrt(d)) Rv,  feat_L2R = softmax(Rq Lk^T / sqrt(d)) Lv.
    One autograd node so that each packed gradient buffer is written exactly once (q part by one direction, k/v part
    by the other)."""

    @staticmethod
    def forward(ctx, Lqkv, Rqkv, heads, drop_p, seed_r2l, seed_l2r):
        _chk(Lqkv, Rqkv)
        Lqkv, Rqkv = _c(Lqkv), _c(Rqkv)
        B, V, D3 = Lqkv.shape
        D = D3 // 3
        l0, r0 = Lqkv.data_ptr(), Rqkv.data_ptr()
        o_r2l, P1, Pd1 = _attn_forward(l0, D3, r0 + 4 * D, r0 + 8 * D, D3, B, V, V, D, heads, drop_p, seed_r2l, Lqkv.device)
        o_l2r, P2, Pd2 = _attn_forward(r0, D3, l0 + 4 * D, l0 + 8 * D, D3, B, V, V, D, heads, drop_p, seed_l2r, Lqkv.device)
        fl = P1.dim() == 3
        ctx.save_for_backward(Lqkv, Rqkv, P1, Pd1, P2, Pd2, o_r2l if fl else None, o_l2r if fl else None)
        ctx.cfg = (heads, drop_p, seed_r2l, seed_l2r)
        return o_r2l, o_l2r

    @staticmethod
    def backward(ctx, d_r2l, d_l2r):
        Lqkv, Rqkv, P1, Pd1, P2, Pd2, o1, o2 = ctx.saved_tensors
        heads, drop_p, seed_r2l, seed_l2r = ctx.cfg
        d_r2l, d_l2r = _c(d_r2l), _c(d_l2r)
        B, V, D3 = Lqkv.shape
        D = D3 // 3
        dL, dR = torch.empty_like(Lqkv), torch.empty_like(Rqkv)
        l0, r0, gl, gr = Lqkv.data_ptr(), Rqkv.data_ptr(), dL.data_ptr(), dR.data_ptr()
        # R2L: q from L, k/v from R
        _attn_backward(d_r2l, l0, D3, r0 + 4 * D, r0 + 8 * D, D3, gl, D3, gr + 4 * D, gr + 8 * D, D3, P1, Pd1, B, V, V, D,
                       heads, drop_p, seed_r2l, out=o1)
        # L2R: q from R, k/v from L
        _attn_backward(d_l2r, r0, D3, l0 + 4 * D, l0 + 8 * D, D3, gr, D3, gl + 4 * D, gl + 8 * D, D3, P2, Pd2, B, V, V, D,
                       heads, drop_p, seed_l2r, out=o2)
        return dL, dR, None, None, None, None


def cross_attention_packed(Lqkv, Rqkv, heads, drop_p=0.0, seed_r2l=0, seed_l2r=0):
    return CrossAttentionPackedFn.apply(Lqkv, Rqkv, heads, drop_p, seed_r2l, seed_l2r)


class CrossAttentionStackedFn(torch.autograd.Function):
    """CrossAttentionPackedFn on the hands-stacked projection qkv [2,B,V,3D] -> [2,B,V,D]: slice 0 = feat_R2L (left
    queries over right keys / values), slice 1 = feat_L2R.
    own_keys=True is the second model family's variant (inter_attn_lijun.py:94-112): the scores are each hand's OWN
    q.k^T, only the values come from the other hand: slice 0 = softmax(Lq Lk^T) Rv, slice 1 = softmax(Rq Rk^T) Lv."""

    @staticmethod
    def forward(ctx, qkv, heads, drop_p, seed_r2l, seed_l2r, own_keys):
        _chk(qkv)
        qkv = _c(qkv)
        _, B, V, D3 = qkv.shape
        D = D3 // 3
        l0, r0 = qkv[0].data_ptr(), qkv[1].data_ptr()
        kl, kr = (l0, r0) if own_keys else (r0, l0)         # whose keys the left / the right queries see
        out = torch.empty((2, B, V, D), device=qkv.device, dtype=torch.float32)
        _, P1, Pd1 = _attn_forward(l0, D3, kl + 4 * D, r0 + 8 * D, D3, B, V, V, D, heads, drop_p, seed_r2l, qkv.device,
                                   out=out[0])
        _, P2, Pd2 = _attn_forward(r0, D3, kr + 4 * D, l0 + 8 * D, D3, B, V, V, D, heads, drop_p, seed_l2r, qkv.device,
                                   out=out[1])
        ctx.save_for_backward(qkv, P1, Pd1, P2, Pd2, out if P1.dim() == 3 else None)
        ctx.cfg = (heads, drop_p, seed_r2l, seed_l2r, own_keys)
        return out

    @staticmethod
    def backward(ctx, do):
        qkv, P1, Pd1, P2, Pd2, out = ctx.saved_tensors
        o1, o2 = (out[0], out[1]) if out is not None else (None, None)
        heads, drop_p, seed_r2l, seed_l2r, own_keys = ctx.cfg
        do = _c(do)
        _, B, V, D3 = qkv.shape
        D = D3 // 3
        dqkv = torch.empty_like(qkv)
        l0, r0, gl, gr = qkv[0].data_ptr(), qkv[1].data_ptr(), dqkv[0].data_ptr(), dqkv[1].data_ptr()
        kl, kr, gkl, gkr = (l0, r0, gl, gr) if own_keys else (r0, l0, gr, gl)
        # every slot of dqkv is written exactly once: q by its own direction, k by the direction that read it, v by
        # the other hand's direction
        _attn_backward(do[0], l0, D3, kl + 4 * D, r0 + 8 * D, D3, gl, D3, gkl + 4 * D, gr + 8 * D, D3, P1, Pd1, B, V, V, D,
                       heads, drop_p, seed_r2l, out=o1)
        _attn_backward(do[1], r0, D3, kr + 4 * D, l0 + 8 * D, D3, gr, D3, gkr + 4 * D, gl + 8 * D, D3, P2, Pd2, B, V, V, D,
                       heads, drop_p, seed_l2r, out=o2)
        return dqkv, None, None, None, None, None


def cross_attention_stacked(qkv, heads, drop_p=0.0, seed_r2l=0, seed_l2r=0, own_keys=False):
    return CrossAttentionStackedFn.apply(qkv, heads, drop_p, seed_r2l, seed_l2r, own_keys)


class AddRowsPairFn(torch.autograd.Function):
    """x [2,B,T,D] + e_h [T,D] (the per-hand position embeddings of img_feat_to_grid, img_attn.py:57-64)."""

    @staticmethod
    def forward(ctx, x, eL, eR):
        _chk(x, eL, eR)
        x, eL, eR = _c(x), _c(eL), _c(eR)
        y = torch.empty_like(x)
        T, D = eL.shape
        for h, e in enumerate((eL, eR)):
            check(_L().rih_add_dropout(x[h].data_ptr(), e.data_ptr(), y[h].data_ptr(), y[h].numel(), D, T, 0.0, 0,
                                       _seed_dev(), _stream()), 'rih_add_dropout')
        ctx.T = T
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        T, D = ctx.T, dy.shape[-1]
        Bn = dy[0].numel() // (T * D)
        de = [colsum(dy[h], Bn, T * D).view(T, D) if ctx.needs_input_grad[1 + h] else None for h in (0, 1)]
        return dy, de[0], de[1]


def add_rows_pair(x, eL, eR):
    return AddRowsPairFn.apply(x, eL, eR)


class AddDropoutFn(torch.autograd.Function):
    """y = a + dropout(b); `bcast_rows` > 0 broadcasts b ([rows, D]) over the leading batch dim (position embeddings)."""

    @staticmethod
    def forward(ctx, a, b, drop_p, seed, bcast_rows):
        _chk(a, b)
        if a is not None:
            a = _c(a)
        b = _c(b)
        ref = a if a is not None else b
        y = torch.empty_like(ref)
        D = ref.shape[-1]
        check(_L().rih_add_dropout(_p(a), b.data_ptr(), y.data_ptr(), y.numel(), D, bcast_rows, drop_p, seed,
                                   _seed_dev(), _stream()), 'rih_add_dropout')
        ctx.cfg = (drop_p, seed, bcast_rows, a is not None, tuple(b.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        drop_p, seed, bcast_rows, has_a, bshape = ctx.cfg
        dy = _c(dy)
        da = dy if (has_a and ctx.needs_input_grad[0]) else None
        db = None
        if ctx.needs_input_grad[1]:
            if bcast_rows > 0:
                D = dy.shape[-1]
                Bn = dy.numel() // (bcast_rows * D)
                db = colsum(dy, Bn, bcast_rows * D).view(bshape)
            elif drop_p > 0:
                db = torch.empty_like(dy)
                check(_L().rih_dropout_bwd(dy.data_ptr(), db.data_ptr(), dy.numel(), drop_p, seed, _seed_dev(),
                                           _stream()), 'rih_dropout_bwd')
            else:
                db = dy
        return da, db, None, None, None


def add_dropout(a, b, drop_p=0.0, seed=0):
    return AddDropoutFn.apply(a, b, drop_p, seed, 0)


def add_rows_bcast(a, e):
    """a [B,V,D] + e [V,D] (nn.Embedding of arange(V), DualGraph.py:76-80 / img_attn.py:57-64)."""
    return AddDropoutFn.apply(a, e, 0.0, 0, e.shape[0])


class ReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        x = _c(x)
        y = torch.empty_like(x)
        check(_L().rih_relu_fwd(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), 'rih_relu_fwd')
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(dy)
        check(_L().rih_relu_bwd(dy.data_ptr(), y.data_ptr(), dx.data_ptr(), dy.numel(), _stream()), 'rih_relu_bwd')
        return dx


def relu(x):
    return ReluFn.apply(x)


class GatherRowsFn(torch.autograd.Function):
    """y[b, i, :] = x[b, idx[i], :]; `inv` = (ptr, list) CSR of the inverse map for the deterministic backward."""

    @staticmethod
    def forward(ctx, x, idx, inv_ptr, inv_idx):
        _chk(x)
        x = _c(x)
        B, Vin, D = x.shape
        Vout = idx.numel()
        y = torch.empty((B, Vout, D), device=x.device, dtype=torch.float32)
        check(_L().rih_gather_rows(x.data_ptr(), idx.data_ptr(), y.data_ptr(), B, Vin, Vout, D, _stream()),
              'rih_gather_rows')
        ctx.save_for_backward(inv_ptr, inv_idx)
        ctx.shape = (B, Vin, Vout, D)
        return y

    @staticmethod
    def backward(ctx, dy):
        inv_ptr, inv_idx = ctx.saved_tensors
        B, Vin, Vout, D = ctx.shape
        dy = _c(dy)
        dx = torch.empty((B, Vin, D), device=dy.device, dtype=torch.float32)
        check(_L().rih_scatter_rows_add(dy.data_ptr(), inv_ptr.data_ptr(), inv_idx.data_ptr(), dx.data_ptr(), B, Vin,
                                        Vout, D, _stream()), 'rih_scatter_rows_add')
        return dx, None, None, None


class RowIndex:
    """A fixed row-index map (graph permutation, nearest upsample, token slice) with its inverse lists."""

    def __init__(self, idx, vin, device):
        import numpy as np
        idx = np.asarray(idx, dtype=np.int64)
        order = np.argsort(idx, kind='stable')
        counts = np.bincount(idx, minlength=vin)
        ptr = np.zeros(vin + 1, np.int64)
        ptr[1:] = np.cumsum(counts)
        self.vin = vin
        self.idx = torch.as_tensor(idx, dtype=torch.int32, device=device)
        self.inv_ptr = torch.as_tensor(ptr, dtype=torch.int32, device=device)
        self.inv_idx = torch.as_tensor(order, dtype=torch.int32, device=device)

    def __call__(self, x):
        assert x.shape[1] == self.vin
        return GatherRowsFn.apply(x, self.idx, self.inv_ptr, self.inv_idx)


class ChebyFn(torch.autograd.Function):
    """[x, L x] interleaved on the feature dim (graph_conv_cheby K=2, gcn.py:34-69) with L in CSR."""

    @staticmethod
    def forward(ctx, x, csr, csr_t):
        _chk(x)
        x = _c(x)
        B, V, F_ = x.shape
        y = torch.empty((B, V, 2 * F_), device=x.device, dtype=torch.float32)
        check(_L().rih_cheby_fwd(x.data_ptr(), csr[0].data_ptr(), csr[1].data_ptr(), csr[2].data_ptr(), y.data_ptr(),
                                 B, V, F_, _stream()), 'rih_cheby_fwd')
        ctx.csr_t = csr_t
        ctx.shape = (B, V, F_)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, V, F_ = ctx.shape
        dy = _c(dy)
        t = ctx.csr_t
        dx = torch.empty((B, V, F_), device=dy.device, dtype=torch.float32)
        check(_L().rih_cheby_bwd(dy.data_ptr(), t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), dx.data_ptr(), B, V,
                                 F_, _stream()), 'rih_cheby_bwd')
        return dx, None, None


def cheby_features(x, csr, csr_t):
    return ChebyFn.apply(x, csr, csr_t)


class ProjectFn(torch.autograd.Function):
    """Orthographic projection of utils/manoutils.py:26-44: uv = (scale*img)*xyz[:2] + (trans*img/2 + img/2)."""

    @staticmethod
    def forward(ctx, v, scale, trans, img_size):
        _chk(v, scale, trans)
        v, scale, trans = _c(v), _c(scale), _c(trans)
        B, V, _ = v.shape
        out = torch.empty((B, V, 2), device=v.device, dtype=torch.float32)
        check(_L().rih_project_fwd(v.data_ptr(), scale.data_ptr(), trans.data_ptr(), out.data_ptr(), B, V,
                                   float(img_size), _stream()), 'rih_project_fwd')
        ctx.save_for_backward(v, scale)
        ctx.img = float(img_size)
        return out

    @staticmethod
    def backward(ctx, dout):
        v, scale = ctx.saved_tensors
        dout = _c(dout)
        B, V, _ = v.shape
        dv = torch.empty_like(v)
        ds = torch.empty_like(scale)
        dt = torch.empty((B, 2), device=v.device, dtype=torch.float32)
        check(_L().rih_project_bwd(dout.data_ptr(), v.data_ptr(), scale.data_ptr(), dv.data_ptr(), ds.data_ptr(),
                                   dt.data_ptr(), B, V, ctx.img, _stream()), 'rih_project_bwd')
        return dv, ds, dt, None


def projection_batch(scale, trans2d, v, img_size=256):
    return ProjectFn.apply(v, scale, trans2d, img_size)


def _ln_partials_finish(ws, nblk, D, nh, dg, db):
    """d gamma / d beta [nh, D] from the per-block partials ws [nh][nblk][2][D]: deferred to the end of the backward stage
    (ops.deferred_reductions) or finished now by the same descriptor-list kernel."""
    items = [(ws, ws.data_ptr() + 4 * h * nblk * 2 * D, dg, dg.data_ptr() + 4 * h * D, db, db.data_ptr() + 4 * h * D, D, nblk)
             for h in range(nh)]
    if _DEFERRED_LN is not None:
        _DEFERRED_LN.extend(items)
        return
    from ._lib import LnFinalDesc
    arr = (LnFinalDesc * len(items))()
    for d, (_, wsp, _, dgp, _, dbp, D_, nb) in zip(arr, items):
        d.ws, d.dg, d.db, d.D, d.nblk = wsp, dgp, dbp, D_, nb
    check(_L().rih_ln_param_final_multi(arr, len(items), _stream()), 'rih_ln_param_final_multi')


def _linear_wgrad(x, g, K, Nf, rows, paired, has_bias):
    """Weight / bias gradient of y = x W^T + b from the saved input x [2, rows, K] and the output gradient g [2, rows, Nf]:
    per hand (paired: dw [2, Nf, K]) or summed over both hands (a parameter shared by the hands: dw [Nf, K])."""
    dev = x.device
    geom = (1, 1, K, 1, 1, 1, 1, 1, 1, 0, 0)
    if paired:
        dw = torch.empty((2, Nf, K), device=dev, dtype=torch.float32)
        db = torch.empty((2, Nf), device=dev, dtype=torch.float32) if has_bias else None
        _wgrad(x, g, dw, rows, K, Nf, K, Nf, geom, K, 1, K, db=db, nb=2, sx=rows * K, sdy=rows * Nf)
    else:
        dw = torch.empty((Nf, K), device=dev, dtype=torch.float32)
        db = torch.empty((Nf,), device=dev, dtype=torch.float32) if has_bias else None
        _wgrad(x, g, dw, 2 * rows, K, Nf, K, Nf, geom, K, 1, K, db=db)
    return dw, db
