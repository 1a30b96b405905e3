"""Import shims that let the *reference* (/root/reference) modules import in this container.

Used ONLY by the fixture generators under tests/golden/ (run in the build container, where
/root/reference exists).  Nothing here is imported by the product or by the GPU-box tests.

Stubs:
  * yacs.config.CfgNode  -- attribute dict + recursive YAML merge
  * cv2                  -- empty module (only called from data-aug helpers, never the model)
  * torchvision.models   -- resnet{18..152} with torchvision's attribute names (resnet50 used)
"""
import sys
import types
import yaml
import torch
import torch.nn as nn

REF = '/root/reference'


class CfgNode(dict):
    def __init__(self, init=None, new_allowed=True):
        super().__init__()
        if init:
            for k, v in init.items():
                self[k] = CfgNode(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def _merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), CfgNode):
                self[k]._merge(v)
            else:
                self[k] = CfgNode(v) if isinstance(v, dict) else v

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f))

    def set_new_allowed(self, flag):
        pass

    def clone(self):
        import copy
        return copy.deepcopy(self)

    def dump(self):
        return yaml.safe_dump(self)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            idt = self.downsample(x)
        return self.relu(out + idt)


class _ResNet(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(64, layers[0], 1)
        self.layer2 = self._make(128, layers[1], 2)
        self.layer3 = self._make(256, layers[2], 2)
        self.layer4 = self._make(512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(2048, 1000)

    def _make(self, planes, n, stride):
        ds = None
        if stride != 1 or self.inplanes != planes * 4:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False),
                               nn.BatchNorm2d(planes * 4))
        blocks = [_Bottleneck(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * 4
        for _ in range(1, n):
            blocks.append(_Bottleneck(self.inplanes, planes))
        return nn.Sequential(*blocks)


def _resnet50(pretrained=False, **kw):
    return _ResNet([3, 4, 6, 3])


def _resnet101(pretrained=False, **kw):
    return _ResNet([3, 4, 23, 3])


def _unsupported(*a, **k):
    raise NotImplementedError('stub: only bottleneck resnets are provided')


def install():
    if 'yacs' not in sys.modules:
        yacs = types.ModuleType('yacs')
        ycfg = types.ModuleType('yacs.config')
        ycfg.CfgNode = CfgNode
        yacs.config = ycfg
        sys.modules['yacs'] = yacs
        sys.modules['yacs.config'] = ycfg
    if 'cv2' not in sys.modules:
        sys.modules['cv2'] = types.ModuleType('cv2')
    if 'torchvision' not in sys.modules:
        tv = types.ModuleType('torchvision')
        tvm = types.ModuleType('torchvision.models')
        tvm.resnet18 = _unsupported
        tvm.resnet34 = _unsupported
        tvm.resnet50 = _resnet50
        tvm.resnet101 = _resnet101
        tvm.resnet152 = _unsupported
        tv.models = tvm
        sys.modules['torchvision'] = tv
        sys.modules['torchvision.models'] = tvm
    # scipy>=1.14 dropped the scipy.sparse.csr alias the reference asserts against
    import scipy.sparse
    if not hasattr(scipy.sparse, 'csr'):
        csr = types.ModuleType('scipy.sparse.csr')
        csr.csr_matrix = scipy.sparse.csr_matrix
        scipy.sparse.csr = csr
    if REF not in sys.path:
        sys.path.insert(0, REF)


def install_family_b(assets):
    """Extra shims for the reference's second model family (common/myhand/lijun_model_graph.py and friends): the MANO
    wrapper (needs the licence-gated pickles and `manopth`), the trainer's global config (creates directories on import)
    and the mmcv focal loss are replaced by stand-ins; none of them takes part in the network forward that the
    fixtures record (decoder_lijun_graph.py:247-300 never calls them)."""
    import numpy as np
    install()

    cfgm = types.ModuleType('main.config')
    cfgm.cfg = types.SimpleNamespace(mano_flag=True, render=False, normal=True, edge=True, vert2d=True, dice=False,
                                     sdf=False, lambda_sdf=1000000, lambda_render=100, lambda_normal=10,
                                     lambda_edge=100, sdf_thresh=0.01, data_type='interhand_dataaug', mano_path='')
    main = types.ModuleType('main')
    main.config = cfgm
    sys.modules.setdefault('main', main)
    sys.modules.setdefault('main.config', cfgm)

    class _Layer:
        def __init__(self, side):
            d = assets.synthetic_mano_dict(side)
            self.faces = np.asarray(d['f'])
            self.shapedirs = torch.from_numpy(np.asarray(d['shapedirs'], np.float32))
            self.J_regressor = torch.from_numpy(np.asarray(d['J_regressor'].todense(), np.float32))

        def cuda(self):
            return self

    class MANO(nn.Module):
        def __init__(self, hand_type='right'):
            super().__init__()
            self.hand_type = hand_type
            self.layer = _Layer(hand_type)

    class Jr:
        def __init__(self, J_regressor, device='cpu'):
            self.J_regressor = J_regressor

        def __call__(self, v):
            return torch.matmul(self.J_regressor, v)

    mano = types.ModuleType('common.utils.mano')
    mano.MANO, mano.Jr = MANO, Jr
    sys.modules.setdefault('common.utils.mano', mano)
    fl = types.ModuleType('common.utils.focal_loss')
    fl.FocalLoss = type('FocalLoss', (nn.Module,), {})
    sys.modules.setdefault('common.utils.focal_loss', fl)
    bb = types.ModuleType('common.myhand.bbox_decoder')
    bb.load_decoder_cliff = lambda *a, **k: None
    sys.modules.setdefault('common.myhand.bbox_decoder', bb)
