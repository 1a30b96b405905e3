"""HRNet backbone (W18..W64) of the pose network on the HIP ops -- the `ENCODER_TYPE: hrnet32` variant of the reference
(models/model_zoo/hrnet.py `HighResolutionNet`, head_type 'none'; BASELINE config 4).

Built from a table of (modules, blocks, widths) per stage rather than the reference's nested config dicts, but the
container nesting (nn.Sequential / nn.ModuleList indices, `None` slots) reproduces the reference `state_dict` keys:
`conv1, bn1, conv2, bn2, layer1.N.*, transitionS.I[.J].{0,1}.*, stageS.M.branches.I.B.*, stageS.M.fuse_layers.I.J.*`.
nn.Conv2d / nn.BatchNorm2d only hold parameters; activations run NHWC through `renderih_amd.ops`.

Semantics kept: every branch convolution is bias-free 3x3 (BasicBlock) with BN momentum 0.1; a fuse layer sums, per
output resolution i, the branch itself, 1x1-conv+BN+nearest-upsample of every lower resolution j>i and a chain of
stride-2 3x3 conv+BN(+ReLU on all but the last) of every higher resolution j<i, then ReLU
(model_zoo/hrnet.py:170-236); a new branch is created from the LAST output of the previous stage (:507-523).
"""
import torch.nn as nn

from . import ops, streams
from .encoder import Bottleneck, conv, conv_bn

# name -> (stage-1 bottlenecks, stage-1 width, [(modules, blocks per branch, widths), ...])   (hrnet.py:611-660)
ARCH = {
    'w18_small_v1': (1, 32, [(1, 2, (16, 32)), (1, 2, (16, 32, 64)), (1, 2, (16, 32, 64, 128))]),
    'w18_small_v2': (2, 64, [(1, 2, (18, 36)), (3, 2, (18, 36, 72)), (2, 2, (18, 36, 72, 144))]),
}
for _n, _w in (('w18', 18), ('w30', 30), ('w32', 32), ('w40', 40), ('w44', 44), ('w48', 48), ('w64', 64)):
    ARCH[_n] = (4, 64, [(1, 4, (_w, 2 * _w)), (4, 4, (_w, 2 * _w, 4 * _w)), (3, 4, (_w, 2 * _w, 4 * _w, 8 * _w))])


def _cbr(cin, cout, k, stride, relu, bias=False):
    """Conv + BN (+ ReLU placeholder so that the Sequential indices match the reference)."""
    mods = [nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, bias=bias), nn.BatchNorm2d(cout, momentum=0.1)]
    if relu:
        mods.append(nn.ReLU(inplace=True))
    return nn.Sequential(*mods)


def run_cbr(seq, x, residual=None, relu=None):
    """Apply a Conv+BN(+ReLU) container; `residual` is added inside the BN kernel."""
    has_relu = len(seq) > 2 and isinstance(seq[2], nn.ReLU)
    return conv_bn(seq[0], seq[1], x, residual=residual, relu=has_relu if relu is None else relu)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes, momentum=0.1)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes, momentum=0.1)
        self.downsample = downsample

    def forward(self, x):
        out, idt = conv_bn(self.conv1, self.bn1, x, relu=True, skip=True)
        if self.downsample is not None:
            idt = conv_bn(self.downsample[0], self.downsample[1], idt)
        return conv_bn(self.conv2, self.bn2, out, residual=idt, relu=True)


def make_layer(block, inplanes, planes, blocks, stride=1):
    ds = None
    if stride != 1 or inplanes != planes * block.expansion:
        ds = nn.Sequential(nn.Conv2d(inplanes, planes * block.expansion, 1, stride, bias=False),
                           nn.BatchNorm2d(planes * block.expansion, momentum=0.1))
    layers = [block(inplanes, planes, stride, ds)]
    for _ in range(1, blocks):
        layers.append(block(planes * block.expansion, planes))
    return nn.Sequential(*layers)


class HighResolutionModule(nn.Module):
    """One exchange unit: parallel BasicBlock branches, then the all-to-all fuse."""

    def __init__(self, widths, blocks):
        super().__init__()
        nb = len(widths)
        self.branches = nn.ModuleList([make_layer(BasicBlock, w, w, blocks) for w in widths])
        fuse = []
        for i in range(nb):
            row = []
            for j in range(nb):
                if j == i:
                    row.append(None)
                elif j > i:       # lower resolution -> 1x1 conv + BN, upsampled by 2^(j-i) at use
                    row.append(nn.Sequential(nn.Conv2d(widths[j], widths[i], 1, 1, 0, bias=False),
                                             nn.BatchNorm2d(widths[i], momentum=0.1),
                                             nn.Upsample(scale_factor=2 ** (j - i), mode='nearest')))
                else:             # higher resolution -> (i-j) stride-2 3x3 convs, the last one changes the width
                    row.append(nn.Sequential(*[_cbr(widths[j], widths[i] if k == i - j - 1 else widths[j], 3, 2,
                                                    relu=(k != i - j - 1)) for k in range(i - j)]))
            fuse.append(nn.ModuleList(row))
        self.fuse_layers = nn.ModuleList(fuse) if nb > 1 else None
        self.relu = nn.ReLU(False)

    def forward(self, xs):
        # the branches are independent until the fuse, and so are the fuse layer's output rows: each set forks onto side
        # streams (streams.fork_join; the finest resolution -- the longest -- stays in the calling stream)
        xs = streams.fork_join([(lambda b=b, x=x: b(x)) for b, x in zip(self.branches, xs)], reads=xs)
        if self.fuse_layers is None:
            return xs
        nb = len(xs)
        return streams.fork_join([(lambda i=i: self._fuse_row(i, xs)) for i in range(nb)], reads=[xs] * nb)

    def _fuse_row(self, i, xs):
        acc = None
        for j in range(len(xs)):
            if j == i:
                acc = xs[j] if acc is None else ops.add_dropout(acc, xs[j])
            elif j > i:
                f = self.fuse_layers[i][j]
                low = conv_bn(f[0], f[1], xs[j])
                acc = ops.nearest_up_add(low, acc, 2 ** (j - i))
            else:
                chain = self.fuse_layers[i][j]
                y = xs[j]
                for k, cb in enumerate(chain):
                    last = (k == len(chain) - 1)
                    y = run_cbr(cb, y, residual=acc if last else None)     # the running sum rides in the last BN
                acc = y
        return ops.relu(acc)


class HighResolutionNet(nn.Module):
    def __init__(self, name='w32', in_channels=3):
        super().__init__()
        n1, w1, stages = ARCH[name]
        self.conv1 = nn.Conv2d(in_channels, 64, 3, 2, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(64, momentum=0.1)
        self.conv2 = nn.Conv2d(64, 64, 3, 2, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(64, momentum=0.1)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = make_layer(Bottleneck, 64, w1, n1)
        pre = [w1 * 4]
        for s, (nmod, nblk, widths) in enumerate(stages, start=2):
            widths = list(widths)
            setattr(self, 'transition%d' % (s - 1), self._transition(pre, widths))
            setattr(self, 'stage%d' % s, nn.Sequential(*[HighResolutionModule(widths, nblk) for _ in range(nmod)]))
            pre = widths
        self.stage_widths = [list(w) for _, _, w in stages]
        for m in self.modules():            # hrnet.py:593-600
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    @staticmethod
    def _transition(pre, cur):
        layers = []
        for i, w in enumerate(cur):
            if i < len(pre):
                layers.append(_cbr(pre[i], w, 3, 1, relu=True) if w != pre[i] else None)
            else:       # new, coarser branch: stride-2 3x3 convs from the last (coarsest) previous branch
                n = i + 1 - len(pre)
                layers.append(nn.Sequential(*[_cbr(pre[-1], w if j == n - 1 else pre[-1], 3, 2, relu=True)
                                              for j in range(n)]))
        return nn.ModuleList(layers)

    def forward(self, x):
        """x: NHWC image (3 channels padded to 4).  Returns the four branch maps, finest first (NHWC)."""
        x = conv_bn(self.conv1, self.bn1, x, relu=True)
        x = conv_bn(self.conv2, self.bn2, x, relu=True)
        ys = [self.layer1(x)]
        for s in (2, 3, 4):
            trans = getattr(self, 'transition%d' % (s - 1))
            xs = []
            for i, t in enumerate(trans):
                if t is None:
                    xs.append(ys[i])
                elif i < len(ys):
                    xs.append(run_cbr(t, ys[i]))
                else:
                    y = ys[-1]
                    for cb in t:
                        y = run_cbr(cb, y)
                    xs.append(y)
            for mod in getattr(self, 'stage%d' % s):
                xs = mod(xs)
            ys = xs
        return ys
