"""ORACLE (test infrastructure, not product): CPU fp32 restatement of the reference network forward.

A functional, plain-PyTorch-CPU restatement of the RenderIH/IntagHand pose network
(`models.model.HandNET_GCN`, ResNet50 variant) driven by a reference-keyed `state_dict`.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file;
the product (`renderih_amd/`) never does.

Pinning: `tests/golden/make_golden.py` imports the *real* reference modules from /root/reference
(under import stubs) and stores their outputs for seeded inputs/weights in `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks this restatement against those fixtures (parity pinned by
reference-generated vectors; the reference's own tests pin nothing numeric for this path).

Each function cites the reference file:line it follows (paths relative to /root/reference).
Everything is differentiable through torch autograd, so the same code is the gradient oracle.
"""
import numpy as np
import torch
import torch.nn.functional as F

IMG_SIZE = 256  # dataset/dataset_utils.py:4


# ----------------------------------------------------------------------------- primitives
def _bn(sd, p, x, training, eps=1e-5):
    """nn.BatchNorm2d forward (batch statistics when training, running stats otherwise)."""
    return F.batch_norm(x, None if training else sd[p + 'running_mean'],
                        None if training else sd[p + 'running_var'],
                        sd[p + 'weight'], sd[p + 'bias'], training, 0.1, eps)


def _ln(sd, p, x, eps=1e-6):
    return F.layer_norm(x, (x.shape[-1],), sd[p + 'weight'], sd[p + 'bias'], eps)


def _lin(sd, p, x):
    return F.linear(x, sd[p + 'weight'], sd.get(p + 'bias'))


# ----------------------------------------------------------------------------- encoder
def _bottleneck(sd, p, x, stride, training):
    """torchvision Bottleneck v1.5 (stride on the 3x3), used via models/encoder.py:81-83."""
    out = F.relu(_bn(sd, p + 'bn1.', F.conv2d(x, sd[p + 'conv1.weight']), training))
    out = F.relu(_bn(sd, p + 'bn2.', F.conv2d(out, sd[p + 'conv2.weight'], stride=stride, padding=1), training))
    out = _bn(sd, p + 'bn3.', F.conv2d(out, sd[p + 'conv3.weight']), training)
    if (p + 'downsample.0.weight') in sd:
        x = _bn(sd, p + 'downsample.1.', F.conv2d(x, sd[p + 'downsample.0.weight'], stride=stride), training)
    return F.relu(out + x)


def resnet_trunk(sd, x, training, p='encoder.resnet.', layers=(3, 4, 6, 3)):
    """models/encoder.py:107-116 (ResNetSimple.forward, trunk part)."""
    x = F.conv2d(x, sd[p + 'conv1.weight'], stride=2, padding=3)
    x = F.relu(_bn(sd, p + 'bn1.', x, training))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li, n in enumerate(layers):
        for b in range(n):
            stride = 2 if (b == 0 and li > 0) else 1
            x = _bottleneck(sd, '%slayer%d.%d.' % (p, li + 1, b), x, stride, training)
        feats.append(x)
    x4, x3, x2, x1 = feats
    return x1, x2, x3, x4


def aux_decoder(sd, p, x, training):
    """models/encoder.py:21-64 ResNetSimple_decoder: [1x1 conv->ReLU->BN] then 3x[bilinear x2 -> 3x3 conv->ReLU->BN]."""
    fmaps = []
    x = _bn(sd, p + 'models.0.2.', F.relu(F.conv2d(x, sd[p + 'models.0.0.weight'])), training)
    fmaps.append(x)
    for i in (1, 2, 3):
        x = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)
        x = F.relu(F.conv2d(x, sd['%smodels.%d.1.weight' % (p, i)], padding=1))
        x = _bn(sd, '%smodels.%d.3.' % (p, i), x, training)
        fmaps.append(x)
    out = F.conv2d(x, sd[p + 'final_layer.weight'], sd[p + 'final_layer.bias'])
    return out, fmaps


def encoder_forward(sd, img, training):
    """models/encoder.py:107-126."""
    x1, x2, x3, x4 = resnet_trunk(sd, img, training)
    hms, hms_f = aux_decoder(sd, 'encoder.hms_decoder.', x1, training)
    out, dp_f = aux_decoder(sd, 'encoder.dp_decoder.', x1, training)
    return hms, out[:, :2], out[:, 2:], [x1, x2, x3, x4], hms_f, dp_f


def mid_forward(sd, img_f, hms_f, dp_f, training, p='mid_model.'):
    """models/encoder.py:165-173 resnet_mid.forward; conv1x1 = conv->ReLU->BN (model_zoo/__init__.py:56-62)."""
    gf = F.adaptive_avg_pool2d(img_f[0], 1).flatten(1)
    fmaps = []
    for i in range(4):
        x = torch.cat((hms_f[i], dp_f[i]), 1)
        if i > 0:
            x = torch.cat((x, img_f[i]), 1)
        x = F.relu(F.conv2d(x, sd['%sconvs.%d.0.weight' % (p, i)]))
        fmaps.append(_bn(sd, '%sconvs.%d.2.' % (p, i), x, training))
    return gf, fmaps



# ----------------------------------------------------------------------------- HRNet variant
def _cb(sd, p, x, training, stride=1, pad=0, relu=False):
    """Conv (bias optional) + BN (+ReLU): the Sequential(conv, bn[, relu]) pattern of model_zoo/hrnet.py."""
    x = _bn(sd, p + '1.', F.conv2d(x, sd[p + '0.weight'], sd.get(p + '0.bias'), stride=stride, padding=pad), training)
    return F.relu(x) if relu else x


def _basic_block(sd, p, x, training):
    """model_zoo/hrnet.py:28-58 BasicBlock (stride 1, no downsample inside HRNet branches)."""
    out = F.relu(_bn(sd, p + 'bn1.', F.conv2d(x, sd[p + 'conv1.weight'], padding=1), training))
    out = _bn(sd, p + 'bn2.', F.conv2d(out, sd[p + 'conv2.weight'], padding=1), training)
    return F.relu(out + x)


def _count(sd, prefix):
    """Number of consecutive integer children `prefix + '<i>.'` present in the state dict."""
    n = 0
    while any(k.startswith('%s%d.' % (prefix, n)) for k in sd):
        n += 1
    return n


def hr_module(sd, p, xs, training):
    """model_zoo/hrnet.py:102-238 HighResolutionModule.forward."""
    nb = len(xs)
    ys = []
    for i in range(nb):
        x = xs[i]
        for b in range(_count(sd, '%sbranches.%d.' % (p, i))):
            x = _basic_block(sd, '%sbranches.%d.%d.' % (p, i, b), x, training)
        ys.append(x)
    if nb == 1:
        return ys
    out = []
    for i in range(nb):
        acc = None
        for j in range(nb):
            if j == i:
                t = ys[j]
            elif j > i:       # 1x1 conv + BN + nearest upsample  (:170-183)
                t = _cb(sd, '%sfuse_layers.%d.%d.' % (p, i, j), ys[j], training)
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode='nearest')
            else:             # chain of stride-2 3x3 conv + BN, ReLU on all but the last  (:186-207)
                t = ys[j]
                for k in range(i - j):
                    t = _cb(sd, '%sfuse_layers.%d.%d.%d.' % (p, i, j, k), t, training, stride=2, pad=1,
                            relu=(k != i - j - 1))
            acc = t if acc is None else acc + t
        out.append(F.relu(acc))
    return out


def hrnet_trunk(sd, x, training, p='encoder.hrnet.'):
    """model_zoo/hrnet.py:490-527 HighResolutionNet.forward, head_type 'none'."""
    x = F.relu(_bn(sd, p + 'bn1.', F.conv2d(x, sd[p + 'conv1.weight'], stride=2, padding=1), training))
    x = F.relu(_bn(sd, p + 'bn2.', F.conv2d(x, sd[p + 'conv2.weight'], stride=2, padding=1), training))
    for b in range(_count(sd, p + 'layer1.')):
        x = _bottleneck(sd, '%slayer1.%d.' % (p, b), x, 1, training)
    ys = [x]
    for s in (2, 3, 4):
        tp = '%stransition%d.' % (p, s - 1)
        nb = len(ys) + 1
        xs = []
        for i in range(nb):
            if i < len(ys):
                if ('%s%d.0.weight' % (tp, i)) in sd:          # width change of an existing branch (stage 2 only)
                    xs.append(_cb(sd, '%s%d.' % (tp, i), ys[i] if s == 2 else ys[-1], training, pad=1, relu=True))
                else:
                    xs.append(ys[i])
            else:                                               # new branch from the LAST previous output (:507-523)
                t = ys[-1]
                for j in range(_count(sd, '%s%d.' % (tp, i))):
                    t = _cb(sd, '%s%d.%d.' % (tp, i, j), t, training, stride=2, pad=1, relu=True)
                xs.append(t)
        for m in range(_count(sd, '%sstage%d.' % (p, s))):
            xs = hr_module(sd, '%sstage%d.%d.' % (p, s, m), xs, training)
        ys = xs
    return ys


def hrnet_encoder_forward(sd, img, training):
    """models/encoder.py:223-240 HRnet_encoder.forward."""
    ys = hrnet_trunk(sd, img, training)
    size = ys[0].shape[2:]
    x = torch.cat([ys[0]] + [F.interpolate(y, size=size, mode='bilinear', align_corners=True) for y in ys[1:]], 1)

    def head(p):
        h = F.relu(_bn(sd, p + '1.', F.conv2d(x, sd[p + '0.weight'], sd[p + '0.bias']), training))
        return F.conv2d(h, sd[p + '3.weight'], sd[p + '3.bias'])
    hms = head('encoder.hms_decoder.')
    out = head('encoder.dp_decoder.')
    return hms, out[:, 0], out[:, 1:], ys[::-1], None, None


def hrnet_mid_forward(sd, img_f, training, p='mid_model.'):
    """models/encoder.py:333-352 hrnet_mid.forward (img_f coarsest first)."""
    fmaps = []
    for i in range(4):
        x = F.relu(F.conv2d(img_f[i], sd['%sconvs.%d.0.weight' % (p, i)]))
        fmaps.append(_bn(sd, '%sconvs.%d.2.' % (p, i), x, training))
    fine = img_f[::-1]
    y = _bottleneck(sd, p + 'incre_modules.0.0.', fine[0], 1, training)
    for i in range(3):
        d = _cb(sd, '%sdownsamp_modules.%d.' % (p, i), y, training, stride=2, pad=1, relu=True)
        y = _bottleneck(sd, '%sincre_modules.%d.0.' % (p, i + 1), fine[i + 1], 1, training) + d
    y = _cb(sd, p + 'final_layer.', y, training, relu=True)
    return F.adaptive_avg_pool2d(y, 1).flatten(1), fmaps

# ----------------------------------------------------------------------------- decoder blocks
def cheby(sd, p, x, L):
    """models/model_attn/gcn.py:34-69 with K=2: features interleaved as (fin, k)."""
    x1 = torch.matmul(L, x)                                  # B x V x F  (== mm(L, x0) per batch column)
    xc = torch.stack((x, x1), dim=-1).flatten(-2)            # B x V x (F*2), index = f*2+k
    return _lin(sd, p, xc)


def gcn_resblock(sd, p, x, L):
    """models/model_attn/gcn.py:99-110 (norm1 output is overwritten: dead, note N2)."""
    x1 = cheby(sd, p + 'fc1.', x, L)
    x1 = F.relu(_ln(sd, p + 'norm2.', x1))
    x1 = cheby(sd, p + 'fc2.', x1, L)
    x2 = _lin(sd, p + 'shortcut.', x)
    return _ln(sd, p + 'norm3.', x1 + x2)


def graph_layer(sd, p, x, L, n=4):
    """models/model_attn/gcn.py:131-138."""
    for i in range(n):
        x = gcn_resblock(sd, '%sGCN_blocks.%d.' % (p, i), x, L)
        if i != n - 1:
            x = F.relu(x)
    return x


def mlp_res(sd, p, x):
    """models/model_attn/self_attn.py:17-33."""
    return x + _lin(sd, p + 'fc2.', F.relu(_lin(sd, p + 'fc1.', _ln(sd, p + 'layer_norm.', x))))


def _mha(q, k, v, h):
    B, Sq, D = q.shape
    d = D // h
    q = q.view(B, Sq, h, d).transpose(1, 2)
    k = k.view(B, -1, h, d).transpose(1, 2)
    v = v.view(B, -1, h, d).transpose(1, 2)
    a = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / (d ** 0.5), -1)
    return torch.matmul(a, v).transpose(1, 2).contiguous().view(B, Sq, D)


def self_attn(sd, p, x, h=4):
    """models/model_attn/self_attn.py:63-85 (dropout = identity here)."""
    y = _ln(sd, p + 'layer_norm.', x)
    o = _mha(_lin(sd, p + 'w_qs.', y), _lin(sd, p + 'w_ks.', y), _lin(sd, p + 'w_vs.', y), h)
    x = x + _lin(sd, p + 'fc.', o)
    return mlp_res(sd, p + 'ff.', x)


def img_ex(sd, p, img, verts, patch):
    """models/model_attn/img_attn.py:51-67,79-92,109-113."""
    B = img.shape[0]
    g = F.relu(F.conv2d(img, sd[p + 'encoder.proj.weight'], sd[p + 'encoder.proj.bias'], stride=patch))
    g = g.view(B, g.shape[1], -1).transpose(-1, -2) + sd[p + 'encoder.position_embeddings.weight']
    g = self_attn(sd, p + 'encoder.self_attn.', g)
    g = _lin(sd, p + 'attn.fc.', g)
    V = verts.shape[1]
    x = self_attn(sd, p + 'attn.Attn.', torch.cat([verts, g], 1))
    return x[:, :V]


def inter_attn(sd, p, Lf, Rf, h=4):
    """models/model_attn/inter_attn.py:73-125 (shared w_qs/w_ks/w_vs/fc for both hands, note N5)."""
    Lf = self_attn(sd, p + 'L_self_attn_layer.', Lf)
    Rf = self_attn(sd, p + 'R_self_attn_layer.', Rf)
    L2 = _ln(sd, p + 'layer_norm1.', Lf)
    R2 = _ln(sd, p + 'layer_norm2.', Rf)
    Lq, Lk, Lv = (_lin(sd, p + w, L2) for w in ('w_qs.', 'w_ks.', 'w_vs.'))
    Rq, Rk, Rv = (_lin(sd, p + w, R2) for w in ('w_qs.', 'w_ks.', 'w_vs.'))
    feat_R2L = _mha(Lq, Rk, Rv, h)      # softmax(Lq Rk^T) Rv
    feat_L2R = _mha(Rq, Lk, Lv, h)
    Lf = mlp_res(sd, p + 'ffL.', Lf + _lin(sd, p + 'fc.', feat_R2L))
    Rf = mlp_res(sd, p + 'ffR.', Rf + _lin(sd, p + 'fc.', feat_L2R))
    return Lf, Rf


def dual_graph(sd, p, Lf, Rf, fmaps, L_left, L_right):
    """models/model_attn/DualGraph.py:62-91,130-139."""
    patches = (1, 2, 4)
    for i in range(3):
        q = '%slayers.%d.' % (p, i)
        pe = sd[q + 'position_embeddings.weight']
        Lf = graph_layer(sd, q + 'graph_left.', Lf + pe, L_left[i])
        Rf = graph_layer(sd, q + 'graph_right.', Rf + pe, L_right[i])
        Lf = img_ex(sd, q + 'img_ex_left.', fmaps[i], Lf, patches[i])
        Rf = img_ex(sd, q + 'img_ex_right.', fmaps[i], Rf, patches[i])
        Lf, Rf = inter_attn(sd, q + 'attn.', Lf, Rf)
        if i != 2:
            Lf = Lf.repeat_interleave(2, dim=1)      # nn.Upsample(nearest) along V, DualGraph.py:11-18
            Rf = Rf.repeat_interleave(2, dim=1)
    return Lf, Rf


# ----------------------------------------------------------------------------- second model family (common/myhand)
def lijun_mid_forward(sd, img_f, training, p='mid_model.'):
    """common/myhand/encoder_lijun.py:139-146: global feature = avg-pool of the coarsest trunk map; every trunk map goes
    through Conv1x1 -> ReLU -> BN (model_zoo/__init__.py:56-62)."""
    gf = F.adaptive_avg_pool2d(img_f[0], 1).flatten(1)
    fmaps = [_bn(sd, '%sconvs.%d.2.' % (p, i), F.relu(F.conv2d(x, sd['%sconvs.%d.0.weight' % (p, i)])), training)
             for i, x in enumerate(img_f)]
    return gf, fmaps


def lijun_resblock(sd, p, x):
    """common/myhand/model_attn/DualGraph_lijun.py:47-58 (MLP block; norm1 is live here)."""
    x1 = _lin(sd, p + 'fc1.', F.relu(_ln(sd, p + 'norm1.', x)))
    x1 = _lin(sd, p + 'fc2.', F.relu(_ln(sd, p + 'norm2.', x1)))
    return _ln(sd, p + 'norm3.', x1 + _lin(sd, p + 'shortcut.', x))


def lijun_graph_layer(sd, p, x, n=4):
    """DualGraph_lijun.py:82-88."""
    for i in range(n):
        x = lijun_resblock(sd, '%sGCN_blocks.%d.' % (p, i), x)
        if i != n - 1:
            x = F.relu(x)
    return x


def lijun_inter_attn(sd, p, Lf, Rf, h=4):
    """common/myhand/model_attn/inter_attn_lijun.py:79-125: both hands normalised from Lf+Rf; scores from each hand's own
    q.k^T, values from the other hand."""
    Lf = self_attn(sd, p + 'L_self_attn_layer.', Lf)
    Rf = self_attn(sd, p + 'R_self_attn_layer.', Rf)
    L2 = _ln(sd, p + 'layer_norm1.', Lf + Rf)
    R2 = _ln(sd, p + 'layer_norm2.', Rf + Lf)
    Lq, Lk, Lv = (_lin(sd, p + w, L2) for w in ('w_qs.', 'w_ks.', 'w_vs.'))
    Rq, Rk, Rv = (_lin(sd, p + w, R2) for w in ('w_qs.', 'w_ks.', 'w_vs.'))
    feat_R2L = _mha(Lq, Lk, Rv, h)      # softmax(Lq Lk^T) Rv   (:94,:110)
    feat_L2R = _mha(Rq, Rk, Lv, h)      # softmax(Rq Rk^T) Lv   (:95,:109)
    Lf = mlp_res(sd, p + 'ffL.', Lf + _lin(sd, p + 'fc.', feat_R2L))
    Rf = mlp_res(sd, p + 'ffR.', Rf + _lin(sd, p + 'fc.', feat_L2R))
    return Lf, Rf


def lijun_dual_graph(sd, p, Lf, Rf, fmaps):
    """DualGraph_lijun.py:136-163,197-207."""
    patches = (1, 2, 4)
    for i in range(3):
        q = '%slayers.%d.' % (p, i)
        pe = sd[q + 'position_embeddings.weight']
        Lf = lijun_graph_layer(sd, q + 'graph_left.', Lf + pe)
        Rf = lijun_graph_layer(sd, q + 'graph_right.', Rf + pe)
        Lf = img_ex(sd, q + 'img_ex_left.', fmaps[i], Lf, patches[i])
        Rf = img_ex(sd, q + 'img_ex_right.', fmaps[i], Rf, patches[i])
        Lf, Rf = lijun_inter_attn(sd, q + 'attn.', Lf, Rf)
        if i != 2:
            Lf = Lf.repeat_interleave(2, dim=1)
            Rf = Rf.repeat_interleave(2, dim=1)
    return Lf, Rf


def param_regressor(sd, p, v):
    """common/myhand/decoder_lijun_mano.py:112-160 (Hardswish MLP, rot6d -> rotation matrix -> axis-angle)."""
    from oracle import pose_oracle as po
    n = v.shape[0]
    feat = F.hardswish(_lin(sd, p + 'fc.2.', F.hardswish(_lin(sd, p + 'fc.0.', v.reshape(n, -1)))))
    rot6d = _lin(sd, p + 'fc_pose.2.', F.hardswish(_lin(sd, p + 'fc_pose.0.', feat)))
    R = po.rot6d_to_rotmat(rot6d)
    pose = po.rotation_matrix_to_angle_axis(R).reshape(n, -1)
    shape = _lin(sd, p + 'fc_shape.2.', F.hardswish(_lin(sd, p + 'fc_shape.0.', feat)))
    return pose, shape, R


def mano_tail(sd, mano, scale, trans2d, v3c, p='decoder.'):
    """decoder_lijun_mano.py:241-300: up-sampled meshes -> MANO parameters -> MANO layer -> root-centred, bone-length-
    normalised meshes.  `mano` = {'left': constants, 'right': constants} of oracle/mano_oracle.py (after the ctor's
    shapedirs sign fix, :167-169)."""
    from oracle import mano_oracle, pose_oracle as po
    v3d, root, pred, slen = {}, {}, {}, {}
    for side in ('left', 'right'):
        v3d[side] = F.linear(v3c[side].transpose(1, 2), sd[p + 'unsample_layer.weight']).transpose(1, 2)
        Jr = sd['%smano_%s.joint_regressor_torch' % (p, side)]
        root[side] = torch.einsum('bik,ji->bjk', v3d[side], Jr)[:, 0]
        pose, shape, _ = param_regressor(sd, p + 'param_regressor.', v3d[side])
        shape = torch.tanh(shape) * 3
        consts = dict(mano[side])
        # a persistent buffer of the model, registered under two names (mano_X.layer and mano_X_layer are the same
        # module, decoder_lijun_mano.py:160-163): load_state_dict applies the second key last
        consts['hands_components'] = sd['%smano_%s_layer.hands_components' % (p, side)]
        v, j = mano_oracle.mano_forward(consts, po.rodrigues_batch(pose[:, :3]), pose[:, 3:], shape, None, None,
                                        center_idx=None, use_pca=True, new_skel=False)
        v, j = v * 1000 / 1000, j * 1000 / 1000           # common/utils/manolayer.py:323-325 returns mm; :256-257 divide
        length = torch.linalg.norm(j[:, 9:10] - j[:, 0:1], dim=-1)
        slen[side] = (0.095 / length).reshape(-1, 1, 1)
        pred[side] = {'verts3d': (v - j[:, 0:1]) * slen[side], 'joints3d': j, 'mano_pose': pose, 'mano_shape': shape}
    root_rel = root['right'] - root['left']
    result = {'verts3d': {'left': pred['left']['verts3d'], 'right': pred['right']['verts3d'] + root_rel.reshape(-1, 1, 3)},
              'verts2d': {s_: projection_batch(scale[s_], trans2d[s_], pred[s_]['verts3d']) for s_ in ('left', 'right')},
              'v3d_left': v3d['left'], 'v3d_right': v3d['right']}
    params = {'scale': scale, 'trans2d': trans2d, 'scalelength_left': slen['left'], 'scalelength_right': slen['right'],
              'root_rel': root_rel}
    other = {'length': (slen['left'] + slen['right']) / 2, 'root_rel': root_rel,
             'verts3d_MANO_list': pred, 'verts2d_MANO_list': {'left': [], 'right': []}}
    return result, params, other


def is_family_b(sd):
    """lijun_model_graph.HandNET_GCN state: ResNet trunk, no auxiliary decoders, mid convs on the raw 2048-ch map."""
    return 'encoder.resnet.conv1.weight' in sd and 'encoder.hms_decoder.final_layer.weight' not in sd


def projection_batch(scale, trans2d, v, img_size=IMG_SIZE):
    """utils/manoutils.py:26-44."""
    s = (scale * img_size).view(-1, 1, 1)
    t = (trans2d * img_size / 2 + img_size / 2).unsqueeze(1)
    return s * v[..., :2] + t


def decoder_forward(sd, graph, gf, fmaps, p='decoder.', family_b=False):
    """models/decoder.py:128-174; family_b=True: common/myhand/decoder_lijun_graph.py:247-300 (same heads on
    DualGraph_lijun, `verts*_MANO_list` left empty).  `graph` = dict(left=..., right=...) with dense 'L' list (63,126,252),
    'perm' (1008) and 'perm_reverse' (778) as the reference ctor derives them (decoder.py:51-75)."""
    fmaps = fmaps[:-1]
    B = gf.shape[0]
    dc = sd[p + 'dense_coor'] * 2 - 1
    feats = {}
    for side in ('left', 'right'):
        pe = dc[graph[side]['perm']]                             # vert_to_GCN
        pe = pe.view(63, 16, 3).mean(1)                          # graph_avg_pool p=16 (graph_utils.py:35-42)
        g = _ln(sd, '%sgf_layer_%s.1.' % (p, side), _lin(sd, '%sgf_layer_%s.0.' % (p, side), gf))
        feats[side] = torch.cat([g.unsqueeze(1).repeat(1, 63, 1), pe.unsqueeze(0).repeat(B, 1, 1)], -1)
    if family_b:
        Lf, Rf = lijun_dual_graph(sd, p + 'dual_gcn.', feats['left'], feats['right'], fmaps)
    else:
        Lf, Rf = dual_graph(sd, p + 'dual_gcn.', feats['left'], feats['right'], fmaps,
                            graph['left']['L'], graph['right']['L'])
    out = {'left': Lf, 'right': Rf}
    scale, trans2d, v3c, v2c, v3, v2, v3m, v2m = ({} for _ in range(8))
    for side in ('left', 'right'):
        f = out[side]
        t = _lin(sd, p + 'avg_head.', f.transpose(-1, -2))[..., 0]
        t = _lin(sd, p + 'params_head.', t)
        scale[side], trans2d[side] = t[:, 0], t[:, 1:]
        v3c[side] = _lin(sd, p + 'coord_head.', f)
        v2c[side] = projection_batch(scale[side], trans2d[side], v3c[side])
        v3[side] = F.linear(v3c[side].transpose(1, 2), sd[p + 'unsample_layer.weight']).transpose(1, 2)
        v2[side] = projection_batch(scale[side], trans2d[side], v3[side])
        pr = graph[side]['perm_reverse']
        v3m[side] = [] if family_b else [v3c[side].repeat_interleave(4, dim=1)[:, pr]]
        v2m[side] = [] if family_b else [v2c[side].repeat_interleave(4, dim=1)[:, pr]]
    result = {'verts3d': v3, 'verts2d': v2}
    paramsDict = {'scale': scale, 'trans2d': trans2d}
    handDictList = [{'verts3d': v3c, 'verts2d': v2c}]
    otherInfo = {'verts3d_MANO_list': v3m, 'verts2d_MANO_list': v2m}
    return result, paramsDict, handDictList, otherInfo


def handnet_forward(sd, graph, img, training=False, taps=None, mano=None):
    """models/model.py:25-37 HandNET_GCN.forward (ResNet or HRNet encoder, told apart by the state-dict keys)."""
    if is_family_b(sd):             # common/myhand/lijun_model_graph.py:26-33
        x1, x2, x3, x4 = resnet_trunk(sd, img, training)
        img_f = [x1, x2, x3, x4]
        gf, fmaps = lijun_mid_forward(sd, img_f, training)
        if taps is not None:
            taps.update(x1=x1, x2=x2, x3=x3, x4=x4, gf=gf, fmap0=fmaps[0], fmap1=fmaps[1], fmap2=fmaps[2], fmap3=fmaps[3])
        out = decoder_forward(sd, graph, gf, fmaps, family_b=True)
        if 'decoder.mano_left.joint_regressor_torch' in sd:      # lijun_model_newgraph (MANO layer in the forward)
            _, params, hd, _ = out
            result, params, other = mano_tail(sd, mano, params['scale'], params['trans2d'], hd[0]['verts3d'])
            return result, params, hd, other
        return out
    if 'encoder.hrnet.conv1.weight' in sd:
        hms, mask, dp, img_f, hms_f, dp_f = hrnet_encoder_forward(sd, img, training)
        gf, fmaps = hrnet_mid_forward(sd, img_f, training)
        if taps is not None:
            taps.update(x1=img_f[0], x2=img_f[1], x3=img_f[2], x4=img_f[3], gf=gf,
                        fmap0=fmaps[0], fmap1=fmaps[1], fmap2=fmaps[2], fmap3=fmaps[3])
    else:
        hms, mask, dp, img_f, hms_f, dp_f = encoder_forward(sd, img, training)
        gf, fmaps = mid_forward(sd, img_f, hms_f, dp_f, training)
        if taps is not None:
            taps.update(x1=img_f[0], x2=img_f[1], x3=img_f[2], x4=img_f[3], gf=gf,
                        fmap0=fmaps[0], fmap1=fmaps[1], fmap2=fmaps[2], fmap3=fmaps[3],
                        hms_f3=hms_f[3], dp_f0=dp_f[0])
    result, paramsDict, handDictList, otherInfo = decoder_forward(sd, graph, gf, fmaps)
    otherInfo['hms'], otherInfo['mask'], otherInfo['dense'] = hms, mask, dp
    return result, paramsDict, handDictList, otherInfo


def graph_from_dicts(left_dict, right_dict):
    """What decoder.__init__ derives from the two graph dicts (decoder.py:51-75, gcn.py:79-86)."""
    g = {}
    for side, d in (('left', left_dict), ('right', right_dict)):
        Ls = list(d['coarsen_graphs_L'])[::-1][:3]             # reversed: 63, 126, 252
        g[side] = {'L': [torch.from_numpy(np.asarray(L.astype(np.float32).todense())).float() for L in Ls],
                   'perm': torch.as_tensor(np.asarray(d['graph_perm']), dtype=torch.long),
                   'perm_reverse': torch.as_tensor(np.asarray(d['graph_perm_reverse'])[:778], dtype=torch.long)}
    return g


def scalar_loss(outputs):
    """Fixed scalar used for gradient parity (covers every differentiable output of the 4-tuple)."""
    result, params, hd, other = outputs
    s = 0.
    for side in ('left', 'right'):
        s = s + result['verts3d'][side].abs().sum() + 1e-2 * result['verts2d'][side].abs().sum()
        s = s + hd[0]['verts3d'][side].pow(2).sum() + 1e-4 * hd[0]['verts2d'][side].pow(2).sum()
        s = s + params['scale'][side].sum() + params['trans2d'][side].pow(2).sum()
    if isinstance(other.get('verts3d_MANO_list', {}).get('left'), dict):   # MANO model: parameters, joints, raw meshes
        for side in ('left', 'right'):
            m = other['verts3d_MANO_list'][side]
            s = s + 10 * m['joints3d'].abs().sum() + m['mano_pose'].pow(2).sum() + m['mano_shape'].abs().sum()
            s = s + result['v3d_' + side].pow(2).sum()
        s = s + other['length'].sum() + other['root_rel'].abs().sum()
    if 'hms' in other:              # the second model family has no auxiliary heads
        s = s + 1e-3 * other['hms'].pow(2).sum() + 1e-3 * other['mask'].abs().sum() + 1e-3 * other['dense'].pow(2).sum()
    return s


def run(sd, graph, img, training, dtype=torch.float32, with_grad=False, mano=None):
    """Convenience for tests: forward (and scalar-loss backward) at a chosen dtype.
    Returns (flat outputs, {param name: grad}) with the flat names of renderih_amd.testing.flatten_outputs."""
    s = {}
    for k, v in sd.items():
        t = v.detach().clone()
        if t.is_floating_point():
            t = t.to(dtype)
            if with_grad and 'running' not in k and 'dense_coor' not in k and '.mano_' not in k:      # parameters only
                t.requires_grad_(True)
        s[k] = t
    g = {h: {'L': [L.to(dtype) for L in graph[h]['L']], 'perm': graph[h]['perm'],
             'perm_reverse': graph[h]['perm_reverse']} for h in graph}
    with torch.set_grad_enabled(with_grad):
        if mano is not None:
            mano = {h: {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in mano[h].items()}
                    for h in mano}
        out = handnet_forward(s, g, img.to(dtype), training=training, mano=mano)
    grads = {}
    if with_grad:
        scalar_loss(out).backward()
        grads = {k: v.grad for k, v in s.items() if v.is_floating_point() and v.grad is not None}
    result, params, hd, other = out
    flat = {}
    for side in ('left', 'right'):
        flat['result.verts3d.' + side] = result['verts3d'][side]
        flat['result.verts2d.' + side] = result['verts2d'][side]
        flat['params.scale.' + side] = params['scale'][side]
        flat['params.trans2d.' + side] = params['trans2d'][side]
        flat['hand0.verts3d.' + side] = hd[0]['verts3d'][side]
        flat['hand0.verts2d.' + side] = hd[0]['verts2d'][side]
        ml = other['verts3d_MANO_list'][side]
        if isinstance(ml, dict):
            for k2 in ('verts3d', 'joints3d', 'mano_pose', 'mano_shape'):
                flat['other.mano.%s.%s' % (side, k2)] = ml[k2]
            flat['result.v3d_' + side] = result['v3d_' + side]
            flat['params.scalelength_' + side] = params['scalelength_' + side]
        elif ml:
            flat['other.verts3d_MANO.' + side] = ml[0]
            flat['other.verts2d_MANO.' + side] = other['verts2d_MANO_list'][side][0]
    for k in ('hms', 'mask', 'dense', 'length', 'root_rel'):
        if k in other:
            flat['other.' + k] = other[k]
    return {k: v.detach() for k, v in flat.items()}, grads
