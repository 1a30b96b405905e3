"""fp16-storage inference backbone (BASELINE configs[4]: "Inference-only: batch=256 fp16, hipGraph-captured
encoder+attention+MANO").  Host side of csrc/rih_half.hip.

`HalfBackbone(encoder, mid_model)` snapshots an eval-mode `encoder.ResNetSimple` + `encoder.resnet_mid` (the reference's
models/encoder.py:67-173) into fp16 weights with every BatchNorm folded, and replays the same dataflow on fp16 NHWC
activations:

    image -> 7x7/2 conv -> 3x3/2 max-pool -> 16 bottlenecks -> x4, x3, x2, x1
    x1 -> [1x1 conv, ReLU, BN] -> 3 x [bilinear x2, 3x3 conv, ReLU, BN] -> 1x1 head          (hms_decoder, dp_decoder)
    concat(hms_fmap_i, dp_fmap_i, x_i) -> [1x1 conv, ReLU, BN]                                 (mid_model), avgpool(x1)

The channel concatenations are never materialised by a copy: the producers write their 256 / 256 / C_i channels straight
into slices of one [B, h, w, 512 + C_i] tensor (every kernel takes a pixel pitch).  What leaves the backbone for the mesh
decoder -- the four mid feature maps, the global feature, the heat-map / mask / dense heads -- is written as fp32 by the
last convolution's epilogue, so the decoder (GCN + attention, a few % of the inference time) runs unchanged in fp32.

Numerics: fp16 storage (11-bit significand) of weights and activations, fp32 accumulation and epilogues.  This is NOT the
1e-4 parity path -- that is the fp32 path, which stays the default; `HandNET_GCN.use_fp16_backbone()` opts in for inference.
The snapshot is taken when it is built: call it again after loading or changing weights.

STATUS: harness-verified (tests/test_half.py), not yet run on a GPU (written after the round's GPU budget was spent).
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib, ops
from .ops import check

F16 = torch.float16


def _cdiv(a, b):
    return (a + b - 1) // b


_ZERO = {}
TALLY = None        # set to {'flop': 0.0, 'bytes': 0.0, 'launches': 0} to count the convolutions' algorithmic work


def zero_page(device):
    """>= 16 bytes of zeros in device memory: the source of every padded operand chunk of rih_hconv."""
    key = (device.type, device.index)
    if key not in _ZERO:
        _ZERO[key] = torch.zeros(64, dtype=F16, device=device)
    return _ZERO[key]


def _geom(t):
    """[N,H,W,C] view with unit channel stride and one pixel pitch -> (N, H, W, C, ld)."""
    N, H, W, Cc = t.shape
    ld = t.stride(2) if W > 1 else (t.stride(1) // W if H > 1 else max(Cc, t.stride(2)))
    assert t.stride(3) == 1 or Cc == 1
    assert W == 1 or t.stride(2) == ld
    assert H == 1 or t.stride(1) == W * ld, (t.shape, t.stride())
    assert N == 1 or t.stride(0) == H * W * ld, (t.shape, t.stride())
    return N, H, W, Cc, ld


def bn_fold(bn, conv_bias=None):
    """Eval-mode BatchNorm as (scale, shift) fp32 device tensors (conv bias, if any, folded into the shift)."""
    Cn = bn.num_features
    dev = bn.running_mean.device
    scale = torch.empty(Cn, device=dev, dtype=torch.float32)
    shift = torch.empty(Cn, device=dev, dtype=torch.float32)
    p = lambda t: 0 if t is None else t.detach().data_ptr()
    check(ops._L().rih_hbn_fold(p(bn.weight), p(bn.bias), p(bn.running_mean), p(bn.running_var), p(conv_bias), float(bn.eps),
                            scale.data_ptr(), shift.data_ptr(), Cn, ops._stream()), 'rih_hbn_fold')
    return scale, shift


class PackedConv:
    """One convolution of the folded network: fp16 weights [Cout][Kpad] + epilogue constants.
    order 'conv-bn'      : y = act(conv(x) * s + t [+ res])      -> s folded into the weights, t = bias
    order 'conv-relu-bn' : y = relu(conv(x) [+ b]) * s + t        -> post scale / shift
    order None           : y = conv(x) [+ b]"""

    def __init__(self, conv, bn=None, order=None, cin_pad=None):
        w = conv.weight.detach()
        ops._chk(w)
        self.Cout, self.Cin_w, self.KH, self.KW = w.shape
        self.Cin = cin_pad or self.Cin_w
        assert self.Cin % 8 == 0 and self.Cin >= self.Cin_w
        self.stride, self.pad = conv.stride[0], conv.padding[0]
        self.Kpad = _cdiv(self.KH * self.KW * self.Cin, 64) * 64
        self.bias = self.post_scale = self.post_shift = None
        scale = None
        if order == 'conv-bn':
            scale, self.bias = bn_fold(bn, conv.bias)
        elif order == 'conv-relu-bn':
            self.post_scale, self.post_shift = bn_fold(bn, None)
            self.bias = None if conv.bias is None else conv.bias.detach().float().contiguous()
        else:
            assert bn is None
            self.bias = None if conv.bias is None else conv.bias.detach().float().contiguous()
        self.w = torch.empty((self.Cout, self.Kpad), device=w.device, dtype=F16)
        wc = w.contiguous()
        check(ops._L().rih_hpack_conv_weight(wc.data_ptr(), 0 if scale is None else scale.data_ptr(), self.w.data_ptr(),
                                         self.Cout, self.Cin_w, self.KH, self.KW, self.Cin, self.Kpad, ops._stream()),
              'rih_hpack_conv_weight')

    def __call__(self, x, relu=False, res=None, out=None, out_f32=False):
        ops._chk(x, res, dtype=F16)
        ops._chk(out, dtype=None)
        N, H, W, Cx, ldx = _geom(x)
        assert Cx == self.Cin and x.dtype == F16, (Cx, self.Cin, x.dtype)
        Ho = (H + 2 * self.pad - self.KH) // self.stride + 1
        Wo = (W + 2 * self.pad - self.KW) // self.stride + 1
        if out is None:
            out = torch.empty((N, Ho, Wo, self.Cout), device=x.device, dtype=torch.float32 if out_f32 else F16)
        No, Hy, Wy, Cy, ldy = _geom(out)
        assert (No, Hy, Wy, Cy) == (N, Ho, Wo, self.Cout) and out.dtype == (torch.float32 if out_f32 else F16)
        d = _lib.HConvDesc()
        d.x, d.w, d.zero, d.y = x.data_ptr(), self.w.data_ptr(), zero_page(x.device).data_ptr(), out.data_ptr()
        d.bias = None if self.bias is None else self.bias.data_ptr()
        d.post_scale = None if self.post_scale is None else self.post_scale.data_ptr()
        d.post_shift = None if self.post_shift is None else self.post_shift.data_ptr()
        d.res, d.ldr = None, 0
        if res is not None:
            Nr, Hr, Wr, Cr, ldr = _geom(res)
            assert (Nr, Hr, Wr, Cr) == (N, Ho, Wo, self.Cout) and res.dtype == F16
            d.res, d.ldr = res.data_ptr(), ldr
        d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW = N, H, W, self.Cin, self.Cout, self.KH, self.KW
        d.stride, d.pad, d.Ho, d.Wo, d.ldx, d.ldy, d.Kpad = self.stride, self.pad, Ho, Wo, ldx, ldy, self.Kpad
        d.relu, d.out_f32 = int(relu), int(out_f32)
        # algorithmic work of this launch (true Cin of the weights, no padding)
        flop = 2.0 * N * Ho * Wo * self.Cout * self.KH * self.KW * self.Cin_w
        nbytes = 2.0 * (N * H * W * self.Cin_w + self.Cout * self.KH * self.KW * self.Cin_w) + \
            (4.0 if out_f32 else 2.0) * N * Ho * Wo * self.Cout + (2.0 * N * Ho * Wo * self.Cout if res is not None else 0.0)
        ops._profiled(flop, (N * Ho * Wo, self.Cout, self.KH * self.KW * self.Cin_w, 1, 1, 0, 0, 1, 'f16', nbytes,
                             (H, W, self.KH, self.stride, int(res is not None), int(out_f32))),
                      lambda: check(ops._L().rih_hconv(C.byref(d), ops._stream()), 'rih_hconv'))
        if TALLY is not None:
            TALLY['flop'] += flop
            TALLY['bytes'] += nbytes
            TALLY['launches'] += 1
        return out


def image_to_nhwc8(img):
    """[B,C<=8,H,W] fp32 NCHW -> [B,H,W,8] fp16."""
    B, Cc, H, W = img.shape
    ops._chk(img, dtype=None)
    img = img.contiguous().float()
    out = torch.empty((B, H, W, 8), device=img.device, dtype=F16)
    check(ops._L().rih_himage_nchw_to_nhwc8(img.data_ptr(), out.data_ptr(), B, Cc, H, W, ops._stream()), 'rih_himage_nchw_to_nhwc8')
    return out


def maxpool3x3s2(x):
    ops._chk(x, dtype=F16)
    N, H, W, Cc, ld = _geom(x)
    y = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc), device=x.device, dtype=F16)
    check(ops._L().rih_hmaxpool3x3s2(x.data_ptr(), y.data_ptr(), N, H, W, Cc, ld, Cc, ops._stream()), 'rih_hmaxpool3x3s2')
    return y


def upsample2x(x):
    ops._chk(x, dtype=F16)
    N, H, W, Cc, ld = _geom(x)
    y = torch.empty((N, 2 * H, 2 * W, Cc), device=x.device, dtype=F16)
    check(ops._L().rih_hupsample2x(x.data_ptr(), y.data_ptr(), N, H, W, Cc, ld, Cc, ops._stream()), 'rih_hupsample2x')
    return y


def global_avgpool(x):
    ops._chk(x, dtype=F16)
    N, H, W, Cc, ld = _geom(x)
    y = torch.empty((N, Cc), device=x.device, dtype=torch.float32)
    check(ops._L().rih_havgpool(x.data_ptr(), y.data_ptr(), N, H * W, Cc, ld, ops._stream()), 'rih_havgpool')
    return y


class _Block:
    def __init__(self, blk):
        self.c1 = PackedConv(blk.conv1, blk.bn1, 'conv-bn')
        self.c2 = PackedConv(blk.conv2, blk.bn2, 'conv-bn')
        self.c3 = PackedConv(blk.conv3, blk.bn3, 'conv-bn')
        self.ds = None if blk.downsample is None else PackedConv(blk.downsample[0], blk.downsample[1], 'conv-bn')

    def __call__(self, x, out=None):
        o = self.c2(self.c1(x, relu=True), relu=True)
        idt = x if self.ds is None else self.ds(x)
        return self.c3(o, relu=True, res=idt, out=out)


class _AuxDecoder:
    """ResNetSimple_decoder (models/encoder.py:21-64): stage i writes its 256 channels into `slots[i]`."""

    def __init__(self, dec):
        self.stages = []
        for seq in dec.models:
            mods = list(seq)
            up = isinstance(mods[0], nn.Upsample)
            if up:
                mods = mods[1:]
            self.stages.append((up, PackedConv(mods[0], mods[2], 'conv-relu-bn')))
        self.head = PackedConv(dec.final_layer)

    def __call__(self, x, slots):
        for (up, pc), slot in zip(self.stages, slots):
            if up:
                x = upsample2x(x)
            x = pc(x, relu=True, out=slot)
        return self.head(x, out_f32=True)


class _Trunk:
    """torchvision ResNet trunk (stem, max-pool, bottleneck stages) -> [x1, x2, x3, x4] (coarsest first).  `slots[i]`, when
    given, is the view the last block of the stage that produces x_i writes into (a slice of a wider tensor)."""

    def __init__(self, r):
        self.stem = PackedConv(r.conv1, r.bn1, 'conv-bn', cin_pad=8)
        self.layers = [[_Block(b) for b in layer] for layer in (r.layer1, r.layer2, r.layer3, r.layer4)]
        self.dims = [self.layers[3 - i][-1].c3.Cout for i in range(4)]            # channels of x1, x2, x3, x4

    def stem_pool(self, img):
        """img: the reference's [B,3,H,W] fp32 NCHW tensor, or the fp16 [B,H,W,8] tensor that
        input_pipeline.BatchPreparer(fp16_nhwc8=True) writes (no fp32 image is then ever materialised)."""
        x = img if (img.dtype == F16 and img.dim() == 4 and img.shape[-1] == 8) else image_to_nhwc8(img)
        if x.shape[1] % 32 or x.shape[2] % 32:
            raise ValueError('fp16 backbone: image sides must be multiples of 32')
        return maxpool3x3s2(self.stem(x, relu=True))

    def stages(self, x, slots=(None, None, None, None)):
        outs = [None] * 4
        for li, layer in enumerate(self.layers):
            lvl = 3 - li
            for bi, blk in enumerate(layer):
                x = blk(x, out=slots[lvl] if bi == len(layer) - 1 else None)
            outs[lvl] = x
        return outs


def _check_eval(*mods):
    if any(m.training for m in mods):
        raise RuntimeError('fp16 backbone folds BatchNorm running statistics: call model.eval() first')


class HalfBackbone:
    """models/encoder.py family: ResNetSimple (trunk + hms / dp aux decoders) + resnet_mid."""

    def __init__(self, encoder, mid_model):
        from .encoder import ResNetSimple, resnet_mid
        if not isinstance(encoder, ResNetSimple) or not isinstance(mid_model, resnet_mid):
            raise NotImplementedError('fp16 backbone: ResNet encoder family only')
        _check_eval(encoder, mid_model)
        with torch.no_grad():
            self.trunk = _Trunk(encoder.resnet)
            self.hms = _AuxDecoder(encoder.hms_decoder)
            self.dp = _AuxDecoder(encoder.dp_decoder)
            self.mid = [PackedConv(seq[0], seq[2], 'conv-relu-bn') for seq in mid_model.convs]
            zero_page(encoder.resnet.conv1.weight.device)
        self.handNum = encoder.handNum
        self.fdim = [st[1].Cout for st in self.hms.stages]                       # 256 x 4
        self.drop_last = False          # set by HandNET_GCN.use_fp16_backbone when its decoder never reads fmaps[-1]

    @torch.no_grad()
    def __call__(self, img):
        B = img.shape[0]
        x = self.trunk.stem_pool(img)
        # concat buffers of the mid model, level i = 0..3 (x1 .. x4 resolution): [hms 256 | dp 256 | x_i (i > 0)]
        h, w = x.shape[1], x.shape[2]
        f = self.fdim
        cats = []
        for i in range(4):
            width = 2 * f[i] + (self.trunk.dims[i] if i > 0 else 0)
            cats.append(torch.empty((B, h >> (3 - i), w >> (3 - i), width), device=img.device, dtype=F16))
        x1 = self.trunk.stages(x, [None] + [cats[i][..., 2 * f[i]:] for i in range(1, 4)])[0]
        hms = self.hms(x1, [cats[i][..., :f[i]] for i in range(4)])
        out = self.dp(x1, [cats[i][..., f[i]:2 * f[i]] for i in range(4)])
        gf = global_avgpool(x1)
        # (model.SKIP_DEAD_MID, opt-in: `decoder.forward` drops the finest map -- 275 GFLOP and a 1 GB fp32 store at B = 256)
        from . import model as _model
        last = len(self.mid) - 1 if (_model.SKIP_DEAD_MID and self.drop_last) else -1
        fmaps = [None if i == last else pc(cats[i], relu=True, out_f32=True) for i, pc in enumerate(self.mid)]
        mask = ops.nhwc_to_nchw(out, 0, self.handNum)
        dp = ops.nhwc_to_nchw(out, self.handNum, out.shape[-1])
        return ops.nhwc_to_nchw(hms), mask, dp, gf, fmaps


class HalfBackboneB:
    """common/myhand/encoder_lijun.py family (what apps/eval_interhand.py instantiates): trunk only, one
    [1x1 conv, ReLU, BN] per scale, global average pool.  Returns (global_feature, fmaps) in fp32."""

    def __init__(self, encoder, mid_model):
        from . import lijun
        if not isinstance(encoder, lijun.ResNetSimple) or not isinstance(mid_model, lijun.resnet_mid):
            raise NotImplementedError('fp16 backbone: ResNet encoder family only')
        _check_eval(encoder, mid_model)
        with torch.no_grad():
            self.trunk = _Trunk(encoder.resnet)
            self.mid = [PackedConv(seq[0], seq[2], 'conv-relu-bn') for seq in mid_model.convs]
            zero_page(encoder.resnet.conv1.weight.device)

    @torch.no_grad()
    def __call__(self, img):
        xs = self.trunk.stages(self.trunk.stem_pool(img))
        return global_avgpool(xs[0]), [pc(x, relu=True, out_f32=True) for pc, x in zip(self.mid, xs)]
