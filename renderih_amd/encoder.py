"""CNN encoder side of the pose network on the HIP ops (drop-in for the reference's models/encoder.py).

Module/parameter names reproduce the reference's `state_dict`: torchvision-style
`resnet.*`, `hms_decoder.models.N.M.*`, `dp_decoder.*`, `mid_model.convs.N.{0,2}.*`.  The nn.Conv2d /
nn.BatchNorm2d objects are parameter containers only -- their torch forward is never called; the forward
below runs NHWC through `renderih_amd.ops` (implicit-GEMM MFMA convs, fused BN(+residual)(+ReLU)).

Reference behaviour kept on purpose: trunk = Conv->BN->ReLU (torchvision Bottleneck v1.5, stride on the
3x3); aux decoders and mid convs = Conv->ReLU->BN (encoder.py:52-54, model_zoo/__init__.py:56-62);
`resnet.fc` exists but is unused; mid.convs[3] is computed although the decoder drops it.
"""
import os

import torch
import torch.nn as nn

from . import ops


def bn_act(bn, x, residual=None, relu=False, tile_stats=None, input_relu=False):
    """nn.BatchNorm2d semantics (train: batch stats + running update; eval: running stats) on NHWC."""
    y = ops.batchnorm(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, residual=residual,
                      training=bn.training, relu=relu, eps=bn.eps, momentum=bn.momentum, tile_stats=tile_stats,
                      input_relu=input_relu)
    if bn.training and bn.num_batches_tracked is not None:
        _PENDING_TRACKED.append(bn.num_batches_tracked)
    return y


_PENDING_TRACKED = []


def flush_batches_tracked():
    """`num_batches_tracked += 1` of every BatchNorm that ran since the last flush, as ONE multi-tensor launch
    (65 one-element kernels per forward otherwise).  Called at the end of every encoder / mid-model forward (and of
    HandNET_GCN.forward), so a sub-module used on its own keeps torch's BatchNorm bookkeeping too."""
    if _PENDING_TRACKED:
        torch._foreach_add_(list(_PENDING_TRACKED), 1)
        _PENDING_TRACKED.clear()


def conv(m, x, relu=False):
    """nn.Conv2d container -> HIP conv (square kernel/stride/padding as everywhere in this network)."""
    return ops.conv2d(x, m.weight, m.bias, stride=m.stride[0], pad=m.padding[0], relu=relu)


def conv_bn(cm, bn, x, residual=None, relu=False, conv_relu=False, skip=False):
    """Conv2d followed by BatchNorm2d (Conv -> BN (-> ReLU) of the trunk, models/encoder.py:107-116, or Conv -> ReLU -> BN of
    the aux decoders / mid convs, :52-54 with conv_relu=True).  skip=True: also returns the alias of x for the block's skip
    path (ops.conv2d_skip).  In training the BatchNorm statistics come out of the convolution's GEMM epilogue when its
    descriptor takes the split engine's fast path (ops.StatsHolder / rih_gemm_desc.stats): (mean, M2) per wave row block,
    merged by rih_bn_stats_from_blocks -- no statistics pass over the activation.  With conv_relu the ReLU's backward gate is
    applied by the BatchNorm's backward kernel (it reads its input anyway): no separate relu_bwd pass."""
    f = ops.conv2d_skip if skip else ops.conv2d
    holder = ops.StatsHolder() if (ops.GEMM_STATS and bn.training) else None
    out = f(x, cm.weight, cm.bias, stride=cm.stride[0], pad=cm.padding[0], relu=conv_relu, grad_masked=conv_relu, stats=holder)
    y, idt = out if skip else (out, None)
    stats = ('blocks', holder.part, holder.T, holder.rows) if (holder is not None and holder.part is not None) else None
    y = bn_act(bn, y, residual=residual, relu=relu, tile_stats=stats, input_relu=conv_relu)
    return (y, idt) if skip else y


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        # the skip path is fed from conv1's second output (an alias of x): its gradient is added inside conv1's
        # data-gradient GEMM instead of by a separate pass over the activation (ops.Conv2dFn)
        out, idt = conv_bn(self.conv1, self.bn1, x, relu=True, skip=True)
        out = conv_bn(self.conv2, self.bn2, out, relu=True)
        if self.downsample is not None:
            idt = conv_bn(self.downsample[0], self.downsample[1], idt)
        return conv_bn(self.conv3, self.bn3, out, residual=idt, relu=True)


class ResNetTrunk(nn.Module):
    """torchvision.models.resnet50/101 layout (the reference pins torchvision 0.13.1, README.md:30)."""

    def __init__(self, layers=(3, 4, 6, 3)):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = self._make(64, layers[0], 1)
        self.layer2 = self._make(128, layers[1], 2)
        self.layer3 = self._make(256, layers[2], 2)
        self.layer4 = self._make(512, layers[3], 2)
        self.fc = nn.Linear(2048, 1000)       # present in reference checkpoints, never used (encoder.py:107-116)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def _make(self, planes, n, stride):
        ds = None
        if stride != 1 or self.inplanes != planes * 4:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False),
                               nn.BatchNorm2d(planes * 4))
        blocks = [Bottleneck(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * 4
        for _ in range(1, n):
            blocks.append(Bottleneck(self.inplanes, planes))
        return nn.Sequential(*blocks)

    def forward(self, x):
        """x: NHWC image padded to 4 channels.  Returns x4,x3,x2,x1 (NHWC)."""
        x = conv_bn(self.conv1, self.bn1, x, relu=True)
        x = ops.maxpool3x3s2(x)
        x4 = self.layer1(x)
        x3 = self.layer2(x4)
        x2 = self.layer3(x3)
        x1 = self.layer4(x2)
        return x4, x3, x2, x1


FOLD_BN = os.environ.get('RIH_FOLD_BN', '0') == '1'


class FoldedTrunk:
    """fp32 inference snapshot of a `ResNetTrunk` in eval mode: every Conv->BN(->ReLU) becomes ONE GEMM launch whose weights
    carry the BatchNorm scale and whose epilogue adds the shift, the residual and the ReLU -- the 53 `bn_apply` passes (a
    read and a write of every activation, about a fifth of the fp32 inference time by traffic) disappear; the arithmetic
    stays fp32, but the folded form computes v*s - m*s where the unfolded one computes (v - m)*s: both terms are rounded
    before they cancel, and the network output moves by a few 1e-5 relative (2.9e-5 on the test fixture) -- inside the 1e-4
    parity bar, yet a third of it, which is why this is not the default.  Opt-in
    (`ResNetSimple.fold_batchnorm()` or RIH_FOLD_BN=1); green on MI355X since round 2 (tests/test_gpu_paths.py).  Snapshot
    semantics: rebuild after changing weights."""

    def __init__(self, trunk):
        if trunk.training:
            raise RuntimeError('folding BatchNorm needs eval mode (running statistics)')
        with torch.no_grad():
            self.stem = self._fold(trunk.conv1, trunk.bn1, 4)
            self.blocks = []
            for layer in (trunk.layer1, trunk.layer2, trunk.layer3, trunk.layer4):
                self.blocks.append([(self._fold(b.conv1, b.bn1), self._fold(b.conv2, b.bn2), self._fold(b.conv3, b.bn3),
                                     None if b.downsample is None else self._fold(b.downsample[0], b.downsample[1]))
                                    for b in layer])

    @staticmethod
    def _fold(conv, bn, cin_pad=None):
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        shift = bn.bias - bn.running_mean * scale
        if conv.bias is not None:
            shift = shift + conv.bias * scale
        Cout, Cin, KH, KW = conv.weight.shape
        return (ops.pack_folded_conv(conv.weight, scale.contiguous(), cin_pad or Cin), KH, KW, shift.contiguous(),
                conv.stride[0], conv.padding[0])

    @staticmethod
    def _run(f, x, relu, residual=None):
        wp, KH, KW, shift, stride, pad = f
        return ops.conv2d_packed(x, wp, KH, KW, shift, stride, pad, relu, residual)

    @torch.no_grad()
    def __call__(self, x):
        x = ops.maxpool3x3s2(self._run(self.stem, x, True))
        outs = []
        for layer in self.blocks:
            for c1, c2, c3, ds in layer:
                o = self._run(c2, self._run(c1, x, True), True)
                idt = x if ds is None else self._run(ds, x, False)
                x = self._run(c3, o, True, residual=idt)
            outs.append(x)
        return outs[0], outs[1], outs[2], outs[3]


class FoldableTrunk:
    """Mixin of the encoders that own a `ResNetTrunk` as `self.resnet`."""
    _folded = None

    def fold_batchnorm(self, enable=True):
        """Inference: run the trunk as a `FoldedTrunk` snapshot whenever the module is in eval mode under no_grad."""
        self._folded = FoldedTrunk(self.resnet) if enable else None
        return self

    def _trunk(self, x):
        if not self.training and not torch.is_grad_enabled():
            if self._folded is None and FOLD_BN:
                self.fold_batchnorm()
            if self._folded is not None:
                return self._folded(x)
        return self.resnet(x)


def _zoo_weights_init(layer):
    """models/model_zoo/__init__.py:35-43 (kaiming_normal_ for Conv2d / Linear)."""
    if isinstance(layer, nn.Conv2d):
        nn.init.kaiming_normal_(layer.weight.data)
    elif isinstance(layer, nn.Linear):
        nn.init.kaiming_normal_(layer.weight.data)
        if layer.bias is not None:
            nn.init.constant_(layer.bias.data, 0.0)


class ResNetSimple_decoder(nn.Module):
    """models/encoder.py:21-64: [1x1 conv, ReLU, BN] @8 then 3x [bilinear x2, 3x3 conv, ReLU, BN], 1x1 head."""

    def __init__(self, expansion=4, fDim=(256, 256, 256, 256), direction=('flat', 'up', 'up', 'up'), out_dim=3):
        super().__init__()
        self.models = nn.ModuleList()
        fDim = [512 * expansion] + list(fDim)
        for i, d in enumerate(direction):
            k = 1 if d == 'flat' else 3
            layers = []
            if d == 'up':
                layers.append(nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True))
            layers += [nn.Conv2d(fDim[i], fDim[i + 1], k, 1, (k - 1) // 2, bias=False), nn.ReLU(inplace=True),
                       nn.BatchNorm2d(fDim[i + 1])]
            self.models.append(nn.Sequential(*layers))
        self.final_layer = nn.Conv2d(fDim[-1], out_dim, 1, 1, 0)

    def forward(self, x):
        fmaps = []
        for seq in self.models:
            mods = list(seq)
            if isinstance(mods[0], nn.Upsample):
                x = ops.upsample_bilinear2x(x)
                mods = mods[1:]
            x = conv_bn(mods[0], mods[2], x, conv_relu=True)
            fmaps.append(x)
        return conv(self.final_layer, x), fmaps


class ResNetSimple(FoldableTrunk, nn.Module):
    """models/encoder.py:67-126."""

    def __init__(self, model_type='resnet50', pretrained=False, fmapDim=(256, 256, 256, 256), handNum=2, heatmapDim=21):
        super().__init__()
        layers = {'resnet50': (3, 4, 6, 3), 'resnet101': (3, 4, 23, 3), 'resnet152': (3, 8, 36, 3)}
        if model_type not in layers:
            raise NotImplementedError('bottleneck ResNets only (the reference path uses resnet50)')
        self.resnet = ResNetTrunk(layers[model_type])
        self.expansion = 4
        self.hms_decoder = ResNetSimple_decoder(self.expansion, fmapDim, out_dim=heatmapDim * handNum)
        for m in self.hms_decoder.modules():
            _zoo_weights_init(m)
        self.dp_decoder = ResNetSimple_decoder(self.expansion, fmapDim, out_dim=handNum + 3 * handNum)
        self.handNum = handNum
        for m in self.dp_decoder.modules():
            _zoo_weights_init(m)

    def forward(self, img):
        """img: [B,3,256,256] NCHW fp32 (the reference's input contract).  Feature maps come back NHWC
        (internal layout); hms/mask/dp are converted to the reference's NCHW."""
        x = ops.nchw_to_nhwc(img, cpad=4)
        x4, x3, x2, x1 = self._trunk(x)
        hms, hms_fmaps = self.hms_decoder(x1)
        out, dp_fmaps = self.dp_decoder(x1)
        mask = ops.nhwc_to_nchw(out, 0, self.handNum)
        dp = ops.nhwc_to_nchw(out, self.handNum, out.shape[-1])
        flush_batches_tracked()
        return ops.nhwc_to_nchw(hms), mask, dp, [x1, x2, x3, x4], hms_fmaps, dp_fmaps


def conv1x1(in_channels, out_channels):
    """models/model_zoo/__init__.py:56-62: Conv(no bias) -> ReLU -> BN."""
    bn = nn.BatchNorm2d(out_channels)
    nn.init.constant_(bn.weight, 1.)
    return nn.Sequential(nn.Conv2d(in_channels, out_channels, 1, 1, 0, bias=False), nn.ReLU(inplace=True), bn)


class resnet_mid(nn.Module):
    """models/encoder.py:129-173."""
    supports_drop_last = True           # forward(..., drop_last=True), see there

    def __init__(self, model_type='resnet50', in_fmapDim=(256, 256, 256, 256), out_fmapDim=(256, 256, 256, 256)):
        super().__init__()
        self.expansion = 4
        self.img_fmaps_dim = [512 * 4, 256 * 4, 128 * 4, 64 * 4]
        self.convs = nn.ModuleList()
        for i in range(len(out_fmapDim)):
            inDim = in_fmapDim[i] * 2 + (self.img_fmaps_dim[i] if i > 0 else 0)
            self.convs.append(conv1x1(inDim, out_fmapDim[i]))
        self.global_feature_dim = 512 * self.expansion
        self.fmaps_dim = list(out_fmapDim)

    def get_info(self):
        return {'global_feature_dim': self.global_feature_dim, 'fmaps_dim': self.fmaps_dim}

    def forward(self, img_fmaps, hms_fmaps, dp_fmaps, drop_last=False):
        """drop_last (not in the reference's signature; HandNET_GCN.forward passes it when RIH_SKIP_DEAD_MID=1): the caller
        will not read fmaps[-1] -- `decoder.forward` drops it (models/decoder.py:130; SURVEY a4: 1.07 of the path's 17.7
        GFLOP per image) and no gradient reaches `convs.3`.  What that branch still owes then is its BatchNorm's
        running statistics in training mode (they are in the state_dict): the convolution and its statistics run, the apply
        pass does not; in eval mode nothing runs.  The slot holds None."""
        gf = ops.global_avgpool(img_fmaps[0])
        fmaps = []
        last = len(self.convs) - 1
        for i, seq in enumerate(self.convs):
            bn = seq[2]
            if drop_last and i == last and not bn.training:
                fmaps.append(None)
                continue
            parts = [hms_fmaps[i], dp_fmaps[i]] + ([img_fmaps[i]] if i > 0 else [])
            # the channel concatenation of models/encoder.py:165-173 is never materialised: the 1x1 convolution reads its parts in
            # place (ops.conv1x1_cat: a segmented GEMM operand; data and weight gradients per part, no slice copies)
            assert seq[0].bias is None
            if drop_last and i == last:
                with torch.no_grad():
                    holder = ops.StatsHolder() if ops.GEMM_STATS else None
                    y = ops.conv1x1_cat(parts, seq[0].weight, relu=True, stats=holder)
                    stats = ('blocks', holder.part, holder.T, holder.rows) if (holder is not None and holder.part is not None) else None
                    ops.batchnorm_update_only(y, bn.running_mean, bn.running_var, eps=bn.eps, momentum=bn.momentum, tile_stats=stats)
                if bn.num_batches_tracked is not None:
                    _PENDING_TRACKED.append(bn.num_batches_tracked)
                fmaps.append(None)
                continue
            holder = ops.StatsHolder() if (ops.GEMM_STATS and bn.training) else None
            y = ops.conv1x1_cat(parts, seq[0].weight, relu=True, stats=holder)
            stats = ('blocks', holder.part, holder.T, holder.rows) if (holder is not None and holder.part is not None) else None
            fmaps.append(bn_act(bn, y, tile_stats=stats, input_relu=True))
        flush_batches_tracked()
        return gf, fmaps


def _hr_name(model_type):
    name = 'w' + model_type[model_type.find('hrnet') + 5:]
    assert name in ('w18', 'w18_small_v1', 'w18_small_v2', 'w30', 'w32', 'w40', 'w44', 'w48', 'w64'), name
    return name


class HRnet_encoder(nn.Module):
    """models/encoder.py:176-240: HRNet trunk; the four branch maps are brought to the finest resolution (bilinear,
    align_corners) and concatenated for the two 1x1 heads (heatmaps 42 ch; mask 1 ch + dense 6 ch)."""

    def __init__(self, model_type, pretrained='', handNum=2, heatmapDim=21):
        super().__init__()
        from .hrnet import HighResolutionNet
        self.hrnet = HighResolutionNet(_hr_name(model_type), in_channels=3)
        self.fmaps_dim = list(self.hrnet.stage_widths[-1])[::-1]           # coarsest first, e.g. [256,128,64,32]
        self.hms_decoder = self.mask_decoder(heatmapDim * handNum)
        self.dp_decoder = self.mask_decoder(1 + 3 * handNum)
        for dec in (self.hms_decoder, self.dp_decoder):
            for m in dec.modules():
                _zoo_weights_init(m)

    def mask_decoder(self, outDim=3):
        c = sum(self.fmaps_dim)
        return nn.Sequential(nn.Conv2d(c, c, 1, 1, 0), nn.BatchNorm2d(c), nn.ReLU(inplace=True), nn.Conv2d(c, outDim, 1, 1, 0))

    @staticmethod
    def _head(seq, x):
        return conv(seq[3], conv_bn(seq[0], seq[1], x, relu=True))

    def forward(self, img):
        ys = self.hrnet(ops.nchw_to_nhwc(img, cpad=4))
        h0 = ys[0].shape[1]
        x = torch.cat([ys[0]] + [ops.upsample_bilinear(y, h0 // y.shape[1]) for y in ys[1:]], dim=-1)
        hms = self._head(self.hms_decoder, x)
        out = self._head(self.dp_decoder, x)
        mask = ops.nhwc_to_nchw(out, 0, 1)[:, 0]                 # `out[:, 0]`: [B, 64, 64]  (encoder.py:235)
        dp = ops.nhwc_to_nchw(out, 1, out.shape[-1])
        flush_batches_tracked()
        return ops.nhwc_to_nchw(hms), mask, dp, ys[::-1], None, None


class hrnet_mid(nn.Module):
    """models/encoder.py:243-352: per-branch 1x1 adapters for the decoder + the HRNet classification-style head
    (Bottleneck per branch, stride-2 3x3 down-sampling chain, 1x1 to 2048, global average) as the global feature."""

    def __init__(self, model_type, in_fmapDim=(256, 256, 256, 256), out_fmapDim=(256, 256, 256, 256)):
        super().__init__()
        _hr_name(model_type)
        in_fmapDim = list(in_fmapDim)
        self.convs = nn.ModuleList([conv1x1(in_fmapDim[i], out_fmapDim[i]) for i in range(len(out_fmapDim))])
        self.global_feature_dim = 2048
        self.fmaps_dim = list(out_fmapDim)
        from .hrnet import make_layer
        pre = in_fmapDim[::-1]                                            # finest first
        head = [32, 64, 128, 256]
        self.incre_modules = nn.ModuleList([make_layer(Bottleneck, c, head[i], 1) for i, c in enumerate(pre)])
        self.downsamp_modules = nn.ModuleList([
            nn.Sequential(nn.Conv2d(head[i] * 4, head[i + 1] * 4, 3, 2, 1), nn.BatchNorm2d(head[i + 1] * 4, momentum=0.1),
                          nn.ReLU(inplace=True)) for i in range(len(pre) - 1)])
        self.final_layer = nn.Sequential(nn.Conv2d(head[3] * 4, 2048, 1, 1, 0), nn.BatchNorm2d(2048, momentum=0.1),
                                         nn.ReLU(inplace=True))

    def get_info(self):
        return {'global_feature_dim': self.global_feature_dim, 'fmaps_dim': self.fmaps_dim}

    def forward(self, img_fmaps, hms_fmaps=None, dp_fmaps=None):
        """img_fmaps: branch maps, coarsest first (NHWC)."""
        fmaps = [conv_bn(seq[0], seq[2], img_fmaps[i], conv_relu=True) for i, seq in enumerate(self.convs)]
        fine_first = img_fmaps[::-1]
        y = self.incre_modules[0](fine_first[0])
        for i, ds in enumerate(self.downsamp_modules):
            down = conv_bn(ds[0], ds[1], y, relu=True)
            y = ops.add_dropout(self.incre_modules[i + 1](fine_first[i + 1]), down)
        y = conv_bn(self.final_layer[0], self.final_layer[1], y, relu=True)
        flush_batches_tracked()
        return ops.global_avgpool(y), fmaps


def load_encoder(cfg):
    """models/encoder.py:355-374 (`pretrained=True` / ENCODER_PRETRAIN_PATH loading there is a download / an optional
    file and is not done here: weights come from `load_state_dict`)."""
    et = cfg.MODEL.ENCODER_TYPE
    if et.find('resnet') != -1:
        encoder = ResNetSimple(model_type=et, pretrained=False, fmapDim=[128, 128, 128, 128], handNum=2, heatmapDim=21)
        mid_model = resnet_mid(model_type=et, in_fmapDim=[128, 128, 128, 128], out_fmapDim=cfg.MODEL.DECONV_DIMS)
        return encoder, mid_model
    if et.find('hrnet') != -1:
        encoder = HRnet_encoder(model_type=et, pretrained='', handNum=2, heatmapDim=21)
        mid_model = hrnet_mid(model_type=et, in_fmapDim=encoder.fmaps_dim, out_fmapDim=cfg.MODEL.DECONV_DIMS)
        return encoder, mid_model
    raise NotImplementedError('encoder type %s' % et)
