"""Generate tests/golden/*.npz by running the REAL reference modules (from /root/reference) on CPU.

Run in the build container only:  python tests/golden/make_golden.py
The reference's own tests pin no numeric result for this path, so these
reference-generated vectors are what pins the oracle (tests/test_oracle_golden.py) and, through it and
directly, the HIP path (tests/test_gpu_*.py).

Fixtures:
  net_eval.npz / net_train.npz : HandNET_GCN (ResNet50 variant), B=2, seeded image + seeded weights
      (renderih_amd.testing.deterministic_state), eval mode / train mode with dropout=0.
      Outputs of the forward 4-tuple (full for small tensors, signature for big ones), tapped
      intermediates, and for train: scalar loss, parameter-gradient signatures, BN running stats.
  net_lijun_eval.npz / net_lijun_train.npz, state_keys_lijun.json : the second model family
      (common/myhand/lijun_model_graph.HandNET_GCN), same recipe.
  mano_*.npz : reference ManoLayer on a synthetic MANO-shaped pickle, several call conventions,
      outputs and input gradients.
"""
import os
import sys
import tempfile
import warnings
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402

ref_stubs.install()                     # puts /root/reference first on sys.path
warnings.filterwarnings('ignore')

from models.encoder import ResNetSimple, resnet_mid, HRnet_encoder, hrnet_mid  # noqa: E402  (reference)
from models.decoder import decoder as RefDecoder                     # noqa: E402  (reference)
from models.model import HandNET_GCN                                 # noqa: E402  (reference)
from models.manolayer import ManoLayer, rodrigues_batch              # noqa: E402  (reference)

# our package is imported by file path pieces that do not collide with the reference's `models`
import importlib.util  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


assets = _load('rih_assets', os.path.join(ROOT, 'renderih_amd', 'assets.py'))
testing = _load('rih_testing', os.path.join(ROOT, 'renderih_amd', 'testing.py'))
net_oracle = _load('net_oracle', os.path.join(ROOT, 'oracle', 'net_oracle.py'))


def build_reference_model(dropout=0.0, encoder='resnet50'):
    if encoder.startswith('hrnet'):     # models/encoder.py:366-372
        enc = HRnet_encoder(model_type=encoder, pretrained='', handNum=2, heatmapDim=21)
        mid = hrnet_mid(model_type=encoder, in_fmapDim=enc.fmaps_dim, out_fmapDim=[256] * 4)
    else:
        enc = ResNetSimple(model_type='resnet50', pretrained=True, fmapDim=[128] * 4, handNum=2, heatmapDim=21)
        mid = resnet_mid(model_type='resnet50', in_fmapDim=[128] * 4, out_fmapDim=[256] * 4)
    dec = RefDecoder(global_feature_dim=2048, f_in_Dim=[256] * 4, f_out_Dim=[256, 128, 64],
                     gcn_in_dim=[512, 256, 128], gcn_out_dim=[256, 128, 64], graph_k=2, graph_layer_num=4,
                     left_graph_dict=assets.load_graph_dict('left'), right_graph_dict=assets.load_graph_dict('right'),
                     vertex_num=778, dense_coor=assets.synthetic_dense_coor(), num_attn_heads=4,
                     upsample_weight=torch.from_numpy(assets.synthetic_upsample_weight()), dropout=dropout)
    return HandNET_GCN(enc, mid, dec)


def pack(store, name, t):
    t = t.detach()
    if t.numel() <= 20000:
        store[name] = t.cpu().numpy()
    else:
        st, sa = testing.signature(t)
        store[name + '#stats'] = st
        store[name + '#samp'] = sa


def tap_hooks(model, taps):
    hs = []
    enc = model.encoder
    for nm, mod in (('x4', enc.resnet.layer1), ('x3', enc.resnet.layer2), ('x2', enc.resnet.layer3),
                    ('x1', enc.resnet.layer4), ('stem', enc.resnet.maxpool),
                    ('hms_f3', enc.hms_decoder.models[3]), ('dp_f0', enc.dp_decoder.models[0]),
                    ('fmap0', model.mid_model.convs[0]), ('fmap1', model.mid_model.convs[1]),
                    ('fmap2', model.mid_model.convs[2]), ('fmap3', model.mid_model.convs[3]),
                    ('gcn0_left', model.decoder.dual_gcn.layers[0].graph_left),
                    ('imgex0_left', model.decoder.dual_gcn.layers[0].img_ex_left)):
        hs.append(mod.register_forward_hook(lambda m, i, o, nm=nm: taps.__setitem__(nm, o)))

    def dgl_hook(idx):
        def f(m, i, o):
            taps['dgl%d_L' % idx], taps['dgl%d_R' % idx] = o
        return f
    for i in range(3):
        hs.append(model.decoder.dual_gcn.layers[i].register_forward_hook(dgl_hook(i)))
    return hs


def hr_tap_hooks(model, taps):
    hs = []
    for i in range(4):
        hs.append(model.mid_model.convs[i].register_forward_hook(lambda m, a, o, i=i: taps.__setitem__('fmap%d' % i, o)))
    def stage_hook(tag):        # a forward hook must return None (anything else replaces the module output)
        def f(m, a, o):
            for k, t in enumerate(o):
                taps['%s_b%d' % (tag, k)] = t.detach().clone()
        return f
    hs.append(model.encoder.hrnet.stage2.register_forward_hook(stage_hook('s2')))
    hs.append(model.encoder.hrnet.stage4.register_forward_hook(stage_hook('s4')))
    hs.append(model.mid_model.register_forward_hook(lambda m, a, o: taps.__setitem__('gf', o[0])))
    return hs


def net_fixture(mode, encoder='resnet50'):
    torch.manual_seed(0)
    hr = encoder.startswith('hrnet')
    model = build_reference_model(dropout=0.0, encoder=encoder)
    sd = testing.deterministic_state(model.state_dict(), seed=0)
    model.load_state_dict(sd)
    model.train(mode == 'train')
    img = testing.seeded_image(2, seed=0)
    taps = {}
    hs = hr_tap_hooks(model, taps) if hr else tap_hooks(model, taps)
    store = {}
    if mode == 'eval':
        with torch.no_grad():
            out = model(img)
    else:
        out = model(img)
    for h in hs:
        h.remove()
    for k, v in testing.flatten_outputs(out).items():
        pack(store, 'out/' + k, v)
    for k, v in taps.items():
        pack(store, 'tap/' + k, v)
    if mode == 'train':
        loss = net_oracle.scalar_loss(out)
        loss.backward()
        store['loss'] = np.float64(loss.item())
        names = []
        for k, p in model.named_parameters():
            if p.grad is None:
                continue
            names.append(k)
            st, sa = testing.signature(p.grad, nsamp=32)
            store['grad/' + k + '#stats'] = st
            store['grad/' + k + '#samp'] = sa
        store['grad_names'] = np.array(names)
        nsd = model.state_dict()
        bnkeys = ('encoder.hrnet.bn1.running_mean', 'encoder.hrnet.bn1.running_var',
                  'encoder.hrnet.stage4.2.fuse_layers.3.0.2.1.running_var',
                  'encoder.hrnet.stage3.1.branches.2.3.bn2.running_mean', 'mid_model.final_layer.1.running_var',
                  'encoder.hms_decoder.1.running_mean', 'encoder.hrnet.bn1.num_batches_tracked') if hr else \
                 ('encoder.resnet.bn1.running_mean', 'encoder.resnet.bn1.running_var',
                  'encoder.resnet.layer4.2.bn3.running_mean', 'encoder.resnet.layer4.2.bn3.running_var',
                  'mid_model.convs.1.2.running_var', 'encoder.hms_decoder.models.2.3.running_mean',
                  'encoder.resnet.bn1.num_batches_tracked')
        for k in bnkeys:
            store['bnstat/' + k] = nsd[k].numpy()
    path = os.path.join(HERE, 'net_%s%s.npz' % ('hrnet_' if hr else '', mode))
    np.savez_compressed(path, **store)
    print('wrote', path, os.path.getsize(path), 'bytes,', len(store), 'arrays')


def mano_fixture():
    tmp = tempfile.mkdtemp()
    store = {}
    for side in ('right', 'left'):
        pkl = assets.write_synthetic_mano_pkl(os.path.join(tmp, 'MANO_%s.pkl' % side.upper()), side, seed=0)
        cases = [
            # name, B, ncomp or 'rot', center_idx, trans, scale, new_skel
            ('pca45', 5, 45, 9, True, True, False),
            ('pca30', 3, 30, 9, True, False, False),
            ('pca6_nocenter', 2, 6, None, False, False, False),
            ('rot', 4, 'rot', 9, False, True, False),
            ('newskel', 2, 45, 0, True, True, True),
            ('zero_pose', 2, 45, 9, False, False, False),
            ('b1', 1, 45, 9, True, True, False),
            ('b19', 19, 45, 9, True, True, False),
        ]
        for name, B, nc, center, use_t, use_s, new_skel in cases:
            g = torch.Generator().manual_seed(zlibseed(side + name))
            use_pca = nc != 'rot'
            layer = ManoLayer(pkl, center_idx=center, use_pca=use_pca, new_skel=new_skel)
            root = rodrigues_batch(torch.randn(B, 3, generator=g) * 0.7).requires_grad_(True)
            if use_pca:
                pose = (torch.randn(B, nc, generator=g) * (0.0 if name == 'zero_pose' else 0.8)).requires_grad_(True)
                if name == 'zero_pose':     # (near-)zero axis-angle: exercises the +1e-8 epsilon path (N7)
                    with torch.no_grad():
                        pose.copy_(layer.axis2pca(torch.zeros(1, 45)).repeat(B, 1))
            else:
                pose = rodrigues_batch(torch.randn(B * 15, 3, generator=g) * 0.6).view(B, 15, 3, 3).requires_grad_(True)
            shape = (torch.randn(B, 10, generator=g)).requires_grad_(True)
            trans = (torch.randn(B, 3, generator=g) * 0.1).requires_grad_(True) if use_t else None
            scale = (torch.rand(B, generator=g) + 0.5).requires_grad_(True) if use_s else None
            v, j = layer(root, pose, shape, trans=trans, scale=scale)
            wv = torch.randn(v.shape, generator=g)
            wj = torch.randn(j.shape, generator=g)
            (v * wv).sum().add((j * wj).sum()).backward()
            key = 'mano/%s/%s/' % (side, name)
            meta = dict(B=B, ncomp=-1 if nc == 'rot' else nc, center=-1 if center is None else center,
                        use_t=int(use_t), use_s=int(use_s), new_skel=int(new_skel))
            for k2, val in meta.items():
                store[key + 'meta_' + k2] = np.int64(val)
            for nm, t in (('root', root), ('pose', pose), ('shape', shape), ('trans', trans), ('scale', scale),
                          ('wv', wv), ('wj', wj), ('v', v), ('j', j)):
                if t is not None:
                    store[key + nm] = t.detach().numpy()
            for nm, t in (('root', root), ('pose', pose), ('shape', shape), ('trans', trans), ('scale', scale)):
                if t is not None:
                    store[key + 'grad_' + nm] = t.grad.numpy()
    path = os.path.join(HERE, 'mano.npz')
    np.savez_compressed(path, **store)
    print('wrote', path, os.path.getsize(path), 'bytes,', len(store), 'arrays')


def loss_fixture():
    """Reference training loss (core/Loss.py: GraphLoss + calc_loss_GCN) on seeded predictions / labels, with the
    gradients with respect to every prediction tensor, for epoch < NORM_EPOCH (edge term off) and >= (on)."""
    import pickle
    import core.Loss as RefLoss                                     # reference (needs utils.manoutils -> cv2/yacs stubs)
    from utils.config import load_cfg
    tmp = tempfile.mkdtemp()
    up = os.path.join(tmp, 'upsample.pkl')
    with open(up, 'wb') as f:
        pickle.dump(assets.synthetic_upsample_weight(), f)
    RefLoss.get_upsample_path = lambda: up                          # misc/upsample.pkl is not in the checkout
    cfg = load_cfg(os.path.join(ref_stubs.REF, 'utils', 'defaults.yaml'))
    from models.model_zoo import GCN_vert_convert
    store = {}
    B = 3
    g = torch.Generator().manual_seed(123)
    mano = {s: assets.synthetic_mano_dict(s) for s in ('left', 'right')}
    conv, gl = {}, {}
    for s in ('left', 'right'):
        gd = assets.load_graph_dict(s)
        conv[s] = GCN_vert_convert(vertex_num=778, graph_perm_reverse=gd['graph_perm_reverse'], graph_perm=gd['graph_perm'])
        J = torch.from_numpy(np.asarray(mano[s]['J_regressor'].todense(), np.float32))
        gl[s] = RefLoss.GraphLoss(J, np.asarray(mano[s]['f']), level=4, device='cpu')
    t = {}
    for s in ('left', 'right'):
        t['v3d_gt_' + s] = 0.05 * torch.randn(B, 778, 3, generator=g)
        t['v2d_gt_' + s] = 256 * torch.rand(B, 778, 2, generator=g)
        t['v3d_' + s] = (t['v3d_gt_' + s] + 0.02 * torch.randn(B, 778, 3, generator=g)).requires_grad_(True)
        t['v2d_' + s] = (t['v2d_gt_' + s] + 8 * torch.randn(B, 778, 2, generator=g)).requires_grad_(True)
        t['c3d_' + s] = (0.05 * torch.randn(B, 252, 3, generator=g)).requires_grad_(True)
        t['c2d_' + s] = (256 * torch.rand(B, 252, 2, generator=g)).requires_grad_(True)
    # a few large residuals so that both SmoothL1 branches are exercised
    with torch.no_grad():
        t['v3d_left'][0, :5] += 3.0
        t['c3d_right'][1, :3] -= 2.5
    t['root_rel'] = 0.05 * torch.randn(B, 3, generator=g)
    for k, v in t.items():
        store['in/' + k] = v.detach().numpy()
    for epoch in (0, 60):
        for v in t.values():
            if v.requires_grad:
                v.grad = None
        result = {'verts3d': {s: t['v3d_' + s] for s in ('left', 'right')}, 'verts2d': {s: t['v2d_' + s] for s in ('left', 'right')}}
        hd = [{'verts3d': {s: t['c3d_' + s] for s in ('left', 'right')}, 'verts2d': {s: t['c2d_' + s] for s in ('left', 'right')}}]
        total, _, mano_d, coarse_d = RefLoss.calc_loss_GCN(
            cfg, epoch, gl['left'], gl['right'], conv['left'], conv['right'], result, None, hd, None, None, None, None,
            t['v2d_gt_left'], None, t['v2d_gt_right'], None, t['v3d_gt_left'], torch.zeros(B, 21, 3),
            t['v3d_gt_right'], torch.zeros(B, 21, 3), t['root_rel'], 256)
        total.backward()
        key = 'e%d/' % epoch
        store[key + 'total'] = np.float64(total.item())
        for k in ('vert2d_loss', 'vert3d_loss', 'joint_loss', 'norm_loss', 'edge_loss'):
            store[key + k] = np.float64(mano_d[k].item())
        store[key + 'c3d_loss'] = np.float64(coarse_d['v3d_loss'][0].item())
        store[key + 'c2d_loss'] = np.float64(coarse_d['v2d_loss'][0].item())
        for k, v in t.items():
            if v.requires_grad:
                store[key + 'grad_' + k] = v.grad.numpy().copy()
    path = os.path.join(HERE, 'loss.npz')
    np.savez_compressed(path, **store)
    print('wrote', path, os.path.getsize(path), 'bytes,', len(store), 'arrays')


def keys_fixture():
    """Reference state_dict schema (key -> shape) for the ResNet50 and HRNet-W32 variants."""
    import json
    for enc, fn in (('resnet50', 'state_keys.json'), ('hrnet32', 'state_keys_hrnet32.json')):
        torch.manual_seed(0)
        model = build_reference_model(dropout=0.05, encoder=enc)
        sch = {k: list(v.shape) for k, v in model.state_dict().items()}
        path = os.path.join(HERE, fn)
        with open(path, 'w') as f:
            json.dump(sch, f)
        print('wrote', path, len(sch), 'keys')


def build_reference_model_b(dropout=0.0):
    """The reference's second family: common/myhand/lijun_model_graph.HandNET_GCN as `load_graph_model` assembles it
    (lijun_model_graph.py:36-53; mano_flag=True as in main/config.py:80), on the packaged graph assets."""
    import types
    ref_stubs.install_family_b(assets)
    torch.Tensor.cuda = lambda self, *a, **k: self          # the decoder ctor moves MANO tables to the GPU (unused here)
    import common.myhand.lijun_model_graph as lm                         # reference
    from common.myhand.encoder_lijun import ResNetSimple as EncB, resnet_mid as MidB      # reference
    from common.myhand.decoder_lijun_graph import decoder as DecB        # reference
    enc = EncB(model_type='resnet50', pretrained=True, fmapDim=[128] * 4, handNum=2, heatmapDim=21)
    mid = MidB(model_type='resnet50', in_fmapDim=[2048, 1024, 512, 256], out_fmapDim=[256] * 4)
    cfg = types.SimpleNamespace(render=False, edge=True, normal=True, vert2d=True, dice=False)
    dec = DecB(cfg, global_feature_dim=2048, f_in_Dim=[256] * 4, f_out_Dim=[256, 128, 64],
               gcn_in_dim=[512, 256, 128], gcn_out_dim=[256, 128, 64], graph_k=2, graph_layer_num=4,
               left_graph_dict=assets.load_graph_dict('left'), right_graph_dict=assets.load_graph_dict('right'),
               vertex_num=778, dense_coor=assets.synthetic_dense_coor(), num_attn_heads=4,
               upsample_weight=torch.from_numpy(assets.synthetic_upsample_weight()), dropout=dropout, mano_flag=True)
    return lm.HandNET_GCN(enc, mid, dec, False)


def lijun_fixture(mode):
    """net_lijun_eval.npz / net_lijun_train.npz: the second family, B=2, seeded image + weights; outputs, taps and (train)
    scalar loss, parameter-gradient signatures, BN running statistics."""
    torch.manual_seed(0)
    model = build_reference_model_b(dropout=0.0)
    sd = testing.deterministic_state(model.state_dict(), seed=4)
    model.load_state_dict(sd)
    model.train(mode == 'train')
    img = testing.seeded_image(2, seed=5)
    taps = {}
    hs = []
    for nm, mod in (('x1', model.encoder.resnet.layer4), ('x4', model.encoder.resnet.layer1),
                    ('fmap0', model.mid_model.convs[0]), ('fmap2', model.mid_model.convs[2]),
                    ('gcn0_left', model.decoder.dual_gcn.layers[0].graph_left),
                    ('imgex0_right', model.decoder.dual_gcn.layers[0].img_ex_right)):
        hs.append(mod.register_forward_hook(lambda m, i, o, nm=nm: taps.__setitem__(nm, o)))

    def dgl_hook(idx):
        def f(m, i, o):
            taps['dgl%d_L' % idx], taps['dgl%d_R' % idx] = o
        return f
    for i in range(3):
        hs.append(model.decoder.dual_gcn.layers[i].register_forward_hook(dgl_hook(i)))
    store = {}
    if mode == 'eval':
        with torch.no_grad():
            out = model(img)
    else:
        out = model(img)
    for h in hs:
        h.remove()
    assert out[3]['verts3d_MANO_list'] == {'left': [], 'right': []}
    for k, v in testing.flatten_outputs(out).items():
        pack(store, 'out/' + k, v)
    for k, v in taps.items():
        pack(store, 'tap/' + k, v)
    if mode == 'train':
        loss = net_oracle.scalar_loss(out)
        loss.backward()
        store['loss'] = np.float64(loss.item())
        names = []
        for k, p in model.named_parameters():
            if p.grad is None:
                continue
            names.append(k)
            st, sa = testing.signature(p.grad, nsamp=32)
            store['grad/' + k + '#stats'] = st
            store['grad/' + k + '#samp'] = sa
        store['grad_names'] = np.array(names)
        nsd = model.state_dict()
        for k in ('encoder.resnet.bn1.running_mean', 'encoder.resnet.layer4.2.bn3.running_var',
                  'mid_model.convs.0.2.running_mean', 'mid_model.convs.3.2.running_var',
                  'encoder.resnet.bn1.num_batches_tracked'):
            store['bnstat/' + k] = nsd[k].numpy()
    path = os.path.join(HERE, 'net_lijun_%s.npz' % mode)
    np.savez_compressed(path, **store)
    print('wrote', path, os.path.getsize(path), 'bytes,', len(store), 'arrays')


def build_reference_model_new(dropout=0.0):
    """common/myhand/lijun_model_newgraph.HandNET_GCN (`load_new_model`): encoder_lijun + decoder_lijun_mano with the
    REAL common/utils/mano.MANO wrapper and common/utils/manolayer.ManoLayer on synthetic MANO-shaped pickles; only the
    trainer's global config, `manopth` (imported, unused) and mmcv's focal loss are stand-ins, and `.cuda()` /
    `.to('cuda')` are made no-ops (this container has no GPU)."""
    import types
    ref_stubs.install()
    tmp = tempfile.mkdtemp()
    for side in ('left', 'right'):
        assets.write_synthetic_mano_pkl(os.path.join(tmp, 'MANO_%s.pkl' % side.upper()), side, seed=0)
    cfgm = types.ModuleType('main.config')
    cfgm.cfg = types.SimpleNamespace(mano_flag=True, render=False, normal=True, edge=True, vert2d=True, dice=False,
                                     sdf=False, lambda_sdf=1e6, lambda_render=100, lambda_normal=10, lambda_edge=100,
                                     sdf_thresh=0.01, data_type='x', mano_path=tmp, reverse=False)
    main = types.ModuleType('main')
    main.config = cfgm
    sys.modules['main'], sys.modules['main.config'] = main, cfgm
    sys.modules['manopth'] = types.ModuleType('manopth')
    fl = types.ModuleType('common.utils.focal_loss')
    fl.FocalLoss = type('FocalLoss', (torch.nn.Module,), {})
    sys.modules['common.utils.focal_loss'] = fl
    sys.modules.pop('common.utils.mano', None)              # a stand-in may be left over from the graph-model fixture
    torch.Tensor.cuda = lambda self, *a, **k: self
    if not getattr(torch.Tensor.to, '_rih_patched', False):
        _to = torch.Tensor.to

        def to(self, *a, **k):
            a = tuple(x for x in a if not (isinstance(x, str) and x.startswith('cuda')))
            if isinstance(k.get('device'), str) and k['device'].startswith('cuda'):
                k.pop('device')
            return _to(self, *a, **k) if (a or k) else self
        to._rih_patched = True
        torch.Tensor.to = to
    import common.myhand.lijun_model_newgraph as lm                     # reference
    from common.myhand.encoder_lijun import ResNetSimple as EncB, resnet_mid as MidB      # reference
    import common.myhand.decoder_lijun_mano as dm                       # reference
    enc = EncB(model_type='resnet50', pretrained=True, fmapDim=[128] * 4, handNum=2, heatmapDim=21)
    mid = MidB(model_type='resnet50', in_fmapDim=[2048, 1024, 512, 256], out_fmapDim=[256] * 4)
    cfg = types.SimpleNamespace(render=False, edge=True, normal=True, vert2d=True, dice=False)
    dec = dm.decoder(cfg, global_feature_dim=2048, f_in_Dim=[256] * 4, f_out_Dim=[256, 128, 64],
                     gcn_in_dim=[512, 256, 128], gcn_out_dim=[256, 128, 64], graph_k=2, graph_layer_num=4,
                     left_graph_dict=assets.load_graph_dict('left'), right_graph_dict=assets.load_graph_dict('right'),
                     vertex_num=778, dense_coor=assets.synthetic_dense_coor(), num_attn_heads=4,
                     upsample_weight=torch.from_numpy(assets.synthetic_upsample_weight()), dropout=dropout, mano_flag=True)
    return lm.HandNET_GCN(enc, mid, dec, False)


def newmodel_fixture():
    """net_newlijun_{eval,train}.npz + state_keys_newlijun.json: the MANO-in-the-forward model of the second family."""
    import json
    model = build_reference_model_new(0.0)
    sch = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(HERE, 'state_keys_newlijun.json'), 'w') as f:
        json.dump(sch, f)
    for mode in ('eval', 'train'):
        torch.manual_seed(0)
        model = build_reference_model_new(0.0)
        model.load_state_dict(testing.deterministic_state(model.state_dict(), seed=11))
        model.train(mode == 'train')
        img = testing.seeded_image(2, seed=12)
        store = {}
        with torch.set_grad_enabled(mode == 'train'):
            out = model(img)
        for k, v in testing.flatten_outputs(out).items():
            pack(store, 'out/' + k, v)
        if mode == 'train':
            loss = net_oracle.scalar_loss(out)
            loss.backward()
            store['loss'] = np.float64(loss.item())
            names = []
            for k, p in model.named_parameters():
                if p.grad is None:
                    continue
                names.append(k)
                st, sa = testing.signature(p.grad, nsamp=32)
                store['grad/' + k + '#stats'] = st
                store['grad/' + k + '#samp'] = sa
            store['grad_names'] = np.array(names)
        path = os.path.join(HERE, 'net_newlijun_%s.npz' % mode)
        np.savez_compressed(path, **store)
        print('wrote', path, os.path.getsize(path), 'bytes,', len(store), 'arrays')


def lijun_keys_fixture():
    import json
    sch = {k: list(v.shape) for k, v in build_reference_model_b(0.05).state_dict().items()}
    path = os.path.join(HERE, 'state_keys_lijun.json')
    with open(path, 'w') as f:
        json.dump(sch, f)
    print('wrote', path, len(sch), 'keys')


def metrics_fixture():
    """metrics.npz: the reference's evaluation helpers (common/utils/intag_eval.py: Jr, eval_hand2,
    batch_compute_similarity_transform_torch) on seeded prediction / ground-truth meshes of both hands."""
    import types
    tv = sys.modules['torchvision']
    if not hasattr(tv, 'transforms'):
        tv.transforms = types.ModuleType('torchvision.transforms')
        sys.modules['torchvision.transforms'] = tv.transforms
    import common.utils.intag_eval as ie                              # reference
    g = torch.Generator().manual_seed(77)
    B = 5
    store = {}
    lists = [{'left': [], 'right': []} for _ in range(4)]
    data = {}
    for side in ('left', 'right'):
        md = assets.synthetic_mano_dict(side)
        J16 = torch.from_numpy(np.asarray(md['J_regressor'].todense(), np.float32))
        jr = ie.Jr(J16, device='cpu')
        v_gt = torch.from_numpy(np.asarray(md['v_template'], np.float32)).unsqueeze(0).repeat(B, 1, 1) \
            + 0.01 * torch.randn(B, 778, 3, generator=g) + 0.3 * torch.randn(B, 1, 3, generator=g)
        # prediction = rotated, scaled, shifted, noisy ground truth (so that the alignment has work to do)
        A = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))[0]
        A = A * torch.sign(torch.det(A)).view(-1, 1, 1)
        v_pred = (v_gt @ A.transpose(1, 2)) * (0.8 + 0.4 * torch.rand(B, 1, 1, generator=g)) \
            + 0.05 * torch.randn(B, 1, 3, generator=g) + 0.004 * torch.randn(B, 778, 3, generator=g)
        j_gt, j_pred = jr(v_gt), jr(v_pred)
        data[side] = (v_gt, v_pred, j_gt, j_pred)
        store['in/Jreg_' + side] = jr.J_regressor.numpy()
        store['in/J16_' + side] = J16.numpy()
        store['in/v_gt_' + side], store['in/v_pred_' + side] = v_gt.numpy(), v_pred.numpy()
        store['out/j_gt_' + side], store['out/j_pred_' + side] = j_gt.numpy(), j_pred.numpy()
        # PA errors exactly as apps/eval_interhand.py:388-407 (root = joint 0)
        jp0, jg0 = j_pred - j_pred[:, 0:1], j_gt - j_gt[:, 0:1]
        vp0, vg0 = v_pred - j_pred[:, 0:1], v_gt - j_gt[:, 0:1]
        hat = ie.batch_compute_similarity_transform_torch(jp0, jg0)
        store['out/pa_mpjpe_' + side] = torch.sqrt(((hat - jg0) ** 2).sum(-1)).mean(-1).numpy()
        hat = ie.batch_compute_similarity_transform_torch(vp0, vg0)
        store['out/pa_mpvpe_' + side] = torch.sqrt(((hat - vg0) ** 2).sum(-1)).mean(-1).numpy()
        store['out/j_err_ori_' + side] = torch.linalg.norm(jp0 - jg0, dim=-1).numpy()
        store['out/v_err_ori_' + side] = torch.linalg.norm(vp0 - vg0, dim=-1).numpy()
    (vlg, vlp, jlg, jlp), (vrg, vrp, jrg, jrp) = data['left'], data['right']
    ie.eval_hand2(vlg, vrg, jlg, jrg, vlp, vrp, jlp, jrp, *lists)
    for side in ('left', 'right'):
        store['out/j_err_' + side] = lists[0][side][0]
        store['out/v_err_' + side] = lists[1][side][0]
    path = os.path.join(HERE, 'metrics.npz')
    np.savez_compressed(path, **store)
    print('wrote', path, os.path.getsize(path), 'bytes,', len(store), 'arrays')


def zlibseed(s):
    import zlib
    return zlib.crc32(s.encode()) & 0x7FFFFFFF


if __name__ == '__main__':
    which = sys.argv[1:] or ['eval', 'train', 'mano', 'keys', 'hrnet', 'loss', 'lijun', 'metrics', 'newmodel']
    if 'newmodel' in which:
        newmodel_fixture()
    if 'metrics' in which:
        metrics_fixture()
    if 'lijun' in which:
        lijun_keys_fixture()
        lijun_fixture('eval')
        lijun_fixture('train')
    if 'loss' in which:
        loss_fixture()
    if 'hrnet' in which:
        net_fixture('eval', 'hrnet32')
        net_fixture('train', 'hrnet32')
    if 'keys' in which:
        keys_fixture()
    if 'mano' in which:
        mano_fixture()
    if 'eval' in which:
        net_fixture('eval')
    if 'train' in which:
        net_fixture('train')
