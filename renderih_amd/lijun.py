"""Second model family of the reference on the HIP ops: `common/myhand/lijun_model_graph.py` (`load_graph_model`), the
network that `apps/eval_interhand.py:238` and `core/graph_model.py:39` instantiate.

Differences from `models/model.py` (family a), all kept:
  * encoder_lijun.py:62-104   ResNet-50 trunk only -- no heat-map / dense-pose decoders;
  * encoder_lijun.py:107-146  `resnet_mid`: Conv1x1 -> ReLU -> BN straight on the four trunk maps (2048/1024/512/256 ch);
  * DualGraph_lijun.py:28-58  `GCN_ResBlock` is an MLP block (LN-ReLU-fc1-LN-ReLU-fc2 + linear shortcut, LN) -- no
                               Chebyshev features, and `norm1` is live here;
  * inter_attn_lijun.py:79-122 cross-hand attention: both hands are normalised from the SUM Lf+Rf, the scores are each
                               hand's OWN q.k^T and only the values come from the other hand;
  * decoder_lijun_graph.py:128-152,247-300  same heads; a `ParamRegressor` is constructed (main/config.py:80
                               mano_flag=True) and therefore present in checkpoints but never called; the
                               `verts*_MANO_list` outputs stay empty.

Parameter names / shapes equal the reference's `state_dict` (tests/golden/state_keys_lijun.json, dumped from the real
modules).  Every per-hand layer pair runs as one launch on hands-stacked activations X[2,B,V,D].
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .attn import DropCtx, MLP_res_block, SelfAttn, _drop_add, _lin_drop_res, _lin_pair, _xavier, img_ex
from . import pose_head
from .decoder import IMG_SIZE, decoder as _DecoderA
from .encoder import FoldableTrunk, ResNetTrunk, conv, conv1x1, conv_bn, flush_batches_tracked


# ------------------------------------------------------------------------------------------------ encoder / mid
class ResNetSimple(FoldableTrunk, nn.Module):
    """encoder_lijun.py:62-104: the torchvision ResNet trunk; forward returns [x1, x2, x3, x4] (coarsest first)."""

    def __init__(self, model_type='resnet50', pretrained=False, fmapDim=(256, 256, 256, 256), handNum=2, heatmapDim=21):
        super().__init__()
        layers = {'resnet50': (3, 4, 6, 3), 'resnet101': (3, 4, 23, 3), 'resnet152': (3, 8, 36, 3)}
        if model_type not in layers:
            raise NotImplementedError('bottleneck ResNets only (the reference path uses resnet50)')
        self.resnet = ResNetTrunk(layers[model_type])
        self.expansion = 4
        self.handNum = handNum

    def forward(self, img):
        """img: [B,3,256,256] NCHW fp32.  The maps come back NHWC (internal layout of this package)."""
        x4, x3, x2, x1 = self._trunk(ops.nchw_to_nhwc(img, cpad=4))
        flush_batches_tracked()
        return [x1, x2, x3, x4]


class resnet_mid(nn.Module):
    """encoder_lijun.py:107-146."""

    def __init__(self, model_type='resnet50', in_fmapDim=(256, 256, 256, 256), out_fmapDim=(256, 256, 256, 256)):
        super().__init__()
        self.expansion = 4
        self.img_fmaps_dim = [512 * 4, 256 * 4, 128 * 4, 64 * 4]
        self.convs = nn.ModuleList([conv1x1(in_fmapDim[i], out_fmapDim[i]) for i in range(len(out_fmapDim))])
        self.output_layer = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Flatten(start_dim=1))    # parameter-free
        self.global_feature_dim = 512 * self.expansion
        self.fmaps_dim = list(out_fmapDim)

    def get_info(self):
        return {'global_feature_dim': self.global_feature_dim, 'fmaps_dim': self.fmaps_dim}

    def forward(self, img_fmaps):
        gf = ops.global_avgpool(img_fmaps[0])
        fmaps = [conv_bn(seq[0], seq[2], x, conv_relu=True) for seq, x in zip(self.convs, img_fmaps)]
        flush_batches_tracked()
        return gf, fmaps


def load_encoder(cfg):
    """encoder_lijun.py:326-346 (resnet branch; `pretrained=True` there is a download and is not done here)."""
    et = cfg.MODEL.ENCODER_TYPE
    if et.find('resnet') == -1:
        raise NotImplementedError('family (b) is built for the resnet encoders (ENCODER_TYPE %s)' % et)
    encoder = ResNetSimple(model_type=et, pretrained=False, fmapDim=[128, 128, 128, 128], handNum=2, heatmapDim=21)
    mid_model = resnet_mid(model_type=et, in_fmapDim=[2048, 1024, 512, 256], out_fmapDim=cfg.MODEL.DECONV_DIMS)
    return encoder, mid_model


# ------------------------------------------------------------------------------------------------ graph blocks
class GCN_ResBlock(nn.Module):
    """DualGraph_lijun.py:28-58: x -> LN3( fc2(relu(LN2(fc1(relu(LN1(x)))))) [dropout] + shortcut(x) )."""

    def __init__(self, in_dim, out_dim, mid_dim, graph_L, graph_k, drop_out=0.01):
        super().__init__()
        self.graph_k = graph_k
        self.in_dim = in_dim
        self.norm1 = nn.LayerNorm(in_dim, eps=1e-6)
        self.fc1 = nn.Linear(in_dim, mid_dim)
        self.norm2 = nn.LayerNorm(out_dim, eps=1e-6)
        self.fc2 = nn.Linear(mid_dim, out_dim)
        self.dropout = nn.Dropout(drop_out)
        self.shortcut = nn.Linear(in_dim, out_dim)
        self.norm3 = nn.LayerNorm(out_dim, eps=1e-6)

    @staticmethod
    def forward_pair(L, R, X, dc, relu_out):
        # LN1 also hands X back as an alias: the shortcut reads that one, and its gradient is added to LN1's inside
        # the LayerNorm-backward kernel
        h, X = ops.LayerNormPairFn.apply(X, None, L.norm1.weight, R.norm1.weight, L.norm1.bias, R.norm1.bias,
                                         L.norm1.eps, True, True)
        h = ops.layernorm_pair(_lin_pair(L.fc1, R.fc1, h), L.norm2, R.norm2, relu=True)
        if dc.p > 0:
            x1 = _drop_add(dc, None, _lin_pair(L.fc2, R.fc2, h))
            return ops.layernorm_pair(x1, L.norm3, R.norm3, x2=_lin_pair(L.shortcut, R.shortcut, X), relu=relu_out)
        x2 = _lin_pair(L.shortcut, R.shortcut, X)
        return ops.layernorm_pair(_lin_pair(L.fc2, R.fc2, h, residual=x2), L.norm3, R.norm3, relu=relu_out)


class GraphLayer(nn.Module):
    """DualGraph_lijun.py:61-88."""

    def __init__(self, in_dim=256, out_dim=256, graph_L=None, graph_k=2, graph_layer_num=3, drop_out=0.01):
        super().__init__()
        assert graph_k > 1
        self.GCN_blocks = nn.ModuleList([GCN_ResBlock(in_dim, out_dim, out_dim, graph_L, graph_k, drop_out)])
        for _ in range(graph_layer_num - 1):
            self.GCN_blocks.append(GCN_ResBlock(out_dim, out_dim, out_dim, graph_L, graph_k, drop_out))
        for m in self.modules():
            _xavier(m)

    @staticmethod
    def forward_pair(L, R, X, dc):
        n = len(L.GCN_blocks)
        for i, (bl, br) in enumerate(zip(L.GCN_blocks, R.GCN_blocks)):
            X = GCN_ResBlock.forward_pair(bl, br, X, dc, relu_out=(i != n - 1))     # F.relu between blocks fused
        return X


class inter_attn(nn.Module):
    """inter_attn_lijun.py:38-125."""

    def __init__(self, f_dim, n_heads=4, d_q=None, d_v=None, dropout=0.1):
        super().__init__()
        self.L_self_attn_layer = SelfAttn(f_dim, n_heads=n_heads, hid_dim=f_dim, dropout=dropout)
        self.R_self_attn_layer = SelfAttn(f_dim, n_heads=n_heads, hid_dim=f_dim, dropout=dropout)
        d_q = f_dim // n_heads if d_q is None else d_q
        d_v = f_dim // n_heads if d_v is None else d_v
        self.n_heads, self.d_q, self.d_v, self.f_dim = n_heads, d_q, d_v, f_dim
        self.norm = d_q ** 0.5
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.w_qs = nn.Linear(f_dim, n_heads * d_q)
        self.w_ks = nn.Linear(f_dim, n_heads * d_q)
        self.w_vs = nn.Linear(f_dim, n_heads * d_v)
        self.fc = nn.Linear(n_heads * d_v, f_dim)
        self.layer_norm1 = nn.LayerNorm(f_dim, eps=1e-6)
        self.layer_norm2 = nn.LayerNorm(f_dim, eps=1e-6)
        self.ffL = MLP_res_block(f_dim, f_dim, dropout)
        self.ffR = MLP_res_block(f_dim, f_dim, dropout)
        for m in self.modules():
            _xavier(m)

    def forward_pair(self, X, dc):
        X = SelfAttn.forward_pair(self.L_self_attn_layer, self.R_self_attn_layer, X, dc)
        # Lf2 = LN1(Lf + Rf), Rf2 = LN2(Rf + Lf): the other hand is the second (fused-add) input of the norm
        X2 = ops.layernorm_pair(X, self.layer_norm1, self.layer_norm2, x2=torch.flip(X, (0,)))
        w = torch.cat([self.w_qs.weight, self.w_ks.weight, self.w_vs.weight], 0)
        b = torch.cat([self.w_qs.bias, self.w_ks.bias, self.w_vs.bias], 0)
        sd = (lambda: dc.seed()) if dc.p > 0 else (lambda: 0)
        # feat_R2L = softmax(Lq Lk^T) Rv, feat_L2R = softmax(Rq Rk^T) Lv  (inter_attn_lijun.py:94-112)
        feat = ops.cross_attention_stacked(ops.linear(X2, w, b), self.n_heads, dc.p, sd(), sd(), own_keys=True)
        return MLP_res_block.forward_pair(self.ffL, self.ffR, _lin_drop_res(dc, self.fc, feat, X), dc)


class DualGraphLayer(nn.Module):
    """DualGraph_lijun.py:91-163."""

    def __init__(self, verts_in_dim=256, verts_out_dim=256, graph_L_Left=None, graph_L_Right=None, graph_k=2,
                 graph_layer_num=4, img_size=64, img_f_dim=256, grid_size=8, grid_f_dim=128, n_heads=4, dropout=0.01):
        super().__init__()
        self.verts_num = graph_L_Left.shape[0]
        self.verts_in_dim, self.img_size, self.img_f_dim = verts_in_dim, img_size, img_f_dim
        self.position_embeddings = nn.Embedding(self.verts_num, self.verts_in_dim)
        self.graph_left = GraphLayer(verts_in_dim, verts_out_dim, graph_L_Left, graph_k, graph_layer_num, dropout)
        self.graph_right = GraphLayer(verts_in_dim, verts_out_dim, graph_L_Right, graph_k, graph_layer_num, dropout)
        self.img_ex_left = img_ex(img_size, img_f_dim, grid_size, grid_f_dim, verts_out_dim, n_heads, dropout)
        self.img_ex_right = img_ex(img_size, img_f_dim, grid_size, grid_f_dim, verts_out_dim, n_heads, dropout)
        self.attn = inter_attn(verts_out_dim, n_heads=n_heads, dropout=dropout)

    def forward_pair(self, X, img_f, dc):
        _, B, V, D = X.shape
        assert V == self.verts_num and D == self.verts_in_dim
        X = ops.add_rows_bcast(X.reshape(2 * B, V, D), self.position_embeddings.weight).view(2, B, V, D)
        X = GraphLayer.forward_pair(self.graph_left, self.graph_right, X, dc)
        X = img_ex.forward_pair(self.img_ex_left, self.img_ex_right, img_f, X, dc)
        return self.attn.forward_pair(X, dc)


class DualGraph(nn.Module):
    """DualGraph_lijun.py:166-207."""

    def __init__(self, verts_in_dim=(512, 256, 128), verts_out_dim=(256, 128, 64), graph_L_Left=None,
                 graph_L_Right=None, graph_k=(2, 2, 2), graph_layer_num=(4, 4, 4), img_size=(16, 32, 64),
                 img_f_dim=(256, 256, 256), grid_size=(8, 8, 16), grid_f_dim=(256, 128, 64), n_heads=4, dropout=0.01):
        super().__init__()
        for i in range(len(verts_in_dim) - 1):
            assert verts_out_dim[i] == verts_in_dim[i + 1]
            assert graph_L_Left[i + 1].shape[0] == 2 * graph_L_Left[i].shape[0]
            assert graph_L_Right[i + 1].shape[0] == 2 * graph_L_Right[i].shape[0]
        self.layers = nn.ModuleList()
        for i in range(len(verts_in_dim)):
            self.layers.append(DualGraphLayer(verts_in_dim[i], verts_out_dim[i], graph_L_Left[i], graph_L_Right[i],
                                              graph_k[i], graph_layer_num[i], img_size[i], img_f_dim[i], grid_size[i],
                                              grid_f_dim[i], n_heads, dropout))
        self._up = {}

    def _upsample2(self, x):
        V = x.shape[1]
        key = (V, x.device)
        if key not in self._up:
            self._up[key] = ops.RowIndex(np.arange(2 * V) // 2, V, x.device)       # graph_upsample(x, 2): row gather
        return self._up[key](x)

    def forward_stacked(self, X, img_f_list, dc):
        assert len(img_f_list) == len(self.layers)
        for i, layer in enumerate(self.layers):
            X = layer.forward_pair(X, img_f_list[i], dc)
            if i != len(self.layers) - 1:
                _, B, V, D = X.shape
                X = self._upsample2(X.reshape(2 * B, V, D)).view(2, B, 2 * V, D)
        return X

    def forward(self, Lf, Rf, img_f_list, dc=None):
        X = self.forward_stacked(torch.stack([Lf, Rf]), img_f_list, dc if dc is not None else DropCtx(0.0, False))
        return X[0], X[1]


# ------------------------------------------------------------------------------------------------ decoder
def make_linear_layers(feat_dims, relu_final=True, use_bn=False):
    """decoder_lijun_graph.py:112-124 (Hardswish MLP); only its parameters matter here (never called in this model)."""
    layers = []
    for i in range(len(feat_dims) - 1):
        layers.append(nn.Linear(feat_dims[i], feat_dims[i + 1]))
        if i < len(feat_dims) - 2 or (i == len(feat_dims) - 2 and relu_final):
            if use_bn:
                layers.append(nn.BatchNorm1d(feat_dims[i + 1]))
            layers.append(nn.Hardswish(inplace=True))
    return nn.Sequential(*layers)


class ParamRegressor(nn.Module):
    """decoder_lijun_graph.py / decoder_lijun_mano.py:126-160: mesh (778*3) -> 1024 -> 512 -> {16 x 6D rotations,
    10 shape} with Hardswish.  The graph model only carries its parameters; the MANO model (`decoder_mano`) calls it."""

    def __init__(self, joint_num=265):
        super().__init__()
        self.joint_num = joint_num
        self.fc = make_linear_layers([self.joint_num * 3, 1024, 512], use_bn=False)
        self.fc_pose = make_linear_layers([512, 128, 16 * 6], relu_final=False)
        self.fc_shape = make_linear_layers([512, 128, 10], relu_final=False)

    @staticmethod
    def _mlp(seq, x):
        for m in seq:
            x = ops.linear(x, m.weight, m.bias) if isinstance(m, nn.Linear) else pose_head.hardswish(x)
        return x

    def forward(self, pose_3d):
        """[N,778,3] -> axis-angle pose [N,48], shape [N,10], rotation matrices [N*16,3,3]   (:142-160)"""
        n = pose_3d.shape[0]
        feat = self._mlp(self.fc, pose_3d.reshape(n, self.joint_num * 3))
        rot6d = self._mlp(self.fc_pose, feat)
        pose_rotmat, aa = pose_head.rot6d_to_rotmat_aa(rot6d.view(n * 16, 6))
        return aa.view(n, 48), self._mlp(self.fc_shape, feat), pose_rotmat


class decoder(_DecoderA):
    """decoder_lijun_graph.py:162-300: the heads of family (a) (`models/decoder.py`) on `DualGraph_lijun`."""

    def __init__(self, cfg=None, global_feature_dim=2048, f_in_Dim=[256, 256, 256, 256], f_out_Dim=[128, 64, 32],
                 gcn_in_dim=[256, 128, 128], gcn_out_dim=[128, 128, 64], graph_k=2, graph_layer_num=4,
                 left_graph_dict={}, right_graph_dict={}, vertex_num=778, dense_coor=None, num_attn_heads=4,
                 upsample_weight=None, dropout=0.05, mano_flag=False):
        super().__init__(global_feature_dim=global_feature_dim, f_in_Dim=f_in_Dim, f_out_Dim=f_out_Dim,
                         gcn_in_dim=gcn_in_dim, gcn_out_dim=gcn_out_dim, graph_k=graph_k,
                         graph_layer_num=graph_layer_num, left_graph_dict=left_graph_dict,
                         right_graph_dict=right_graph_dict, vertex_num=vertex_num, dense_coor=dense_coor,
                         num_attn_heads=num_attn_heads, upsample_weight=upsample_weight, dropout=dropout,
                         dual_graph_cls=DualGraph, mano_lists=False)
        self.cfg = cfg
        self.mano = mano_flag
        if self.mano:
            self.param_regressor = ParamRegressor(joint_num=778)


class MANO(nn.Module):
    """common/utils/mano.py:40-94: the MANO layer (`layer`, center_idx=None, PCA pose) + the 21-joint regressor buffer
    (16 MANO joints + the finger-tip vertices 745/317/445/556/673, re-ordered)."""

    def __init__(self, mano_data, hand_type='right'):
        super().__init__()
        from .manolayer import ManoLayer
        self.hand_type = hand_type
        self.layer = ManoLayer(mano_data, center_idx=None)
        self.vertex_num = 778
        J = self.layer.J_regressor.detach().clone()
        tips = torch.zeros((5, J.shape[1]), dtype=J.dtype)
        for i, v in enumerate((745, 317, 445, 556, 673)):
            tips[i, v] = 1.0
        J = torch.cat([J.cpu(), tips], 0)[[0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]]
        self.register_buffer('joint_regressor_torch', J.float().contiguous())

    def get_3d_joints_T(self, verts_T):
        """einsum('bik,ji->bjk') of mano.py:82-92 on the transposed mesh [B,3,778] -> [B,3,21]."""
        return ops.linear(verts_T, self.joint_regressor_torch)


class decoder_mano(decoder):
    """common/myhand/decoder_lijun_mano.py:85-300 (`load_new_model`): the graph decoder, then per hand
    ParamRegressor -> rot6d -> axis-angle, 3*tanh shape, the MANO layer inside the forward, the mesh centred on the
    wrist and rescaled to a 9.5 cm wrist-to-middle-base bone, projected with the predicted camera."""

    def __init__(self, cfg=None, mano_left=None, mano_right=None, **kw):
        kw['mano_flag'] = True
        super().__init__(cfg, **kw)
        self.mano_left = MANO(mano_left, 'left')
        self.mano_left_layer = self.mano_left.layer            # registered under both names, as in the reference
        self.mano_right = MANO(mano_right, 'right')
        self.mano_right_layer = self.mano_right.layer
        with torch.no_grad():                                  # decoder_lijun_mano.py:167-169
            if torch.sum(torch.abs(self.mano_left_layer.shapedirs[:, 0, :] - self.mano_right_layer.shapedirs[:, 0, :])) < 1:
                self.mano_left_layer.shapedirs[:, 0, :] *= -1

    def forward(self, x, fmaps):
        assert x.shape[1] == self.gf_dim
        fmaps = fmaps[:-1]
        bs = x.shape[0]
        dc = DropCtx(self.dropout_p, self.training)
        Lf, Rf = self._initial_features(x)
        f = self.dual_gcn.forward_stacked(torch.stack([Lf, Rf]), fmaps, dc)
        temp, v3, v2, upT = self._stacked_heads(f.reshape(2 * bs, f.shape[2], f.shape[3]))
        up = upT.transpose(1, 2).contiguous()                              # result['v3d_*']: [2B,778,3]
        pose, shape, _ = self.param_regressor(up)                          # one regressor for both hands (:251-252)
        shape = pose_head.tanh_scale(shape, 3.0)
        root_R = pose_head.rodrigues(pose[:, :3])
        sides = ('left', 'right')
        scale, trans2d, verts3d, verts2d = {}, {}, {}, {}
        result = {'verts3d': {}, 'verts2d': {}}
        pred, slen, root = {}, {}, {}
        for h, side in enumerate(sides):
            sl = slice(h * bs, (h + 1) * bs)
            mano = self.mano_left if side == 'left' else self.mano_right
            scale[side], trans2d[side] = temp[sl, 0], temp[sl, 1:]
            verts3d[side], verts2d[side] = v3[sl], v2[sl]
            result['v3d_' + side] = up[sl]
            root[side] = mano.get_3d_joints_T(upT[sl])[:, :, 0]           # wrist of the regressed mesh, [B,3]
            mv, mj = mano.layer(root_R[sl], pose[sl, 3:], shape[sl])       # metres (the reference goes through mm)
            vn, s = pose_head.center_scale(mv, mj, root=0, bone=(9, 0), target=0.095)
            slen[side] = s.view(-1, 1, 1)
            pred[side] = {'verts3d': vn, 'joints3d': mj, 'mano_pose': pose[sl], 'mano_shape': shape[sl]}
            result['verts2d'][side] = ops.projection_batch(scale[side], trans2d[side], vn, IMG_SIZE)
        root_rel = root['right'] - root['left']
        result['verts3d']['left'] = pred['left']['verts3d']
        result['verts3d']['right'] = pred['right']['verts3d'] + root_rel.reshape(-1, 1, 3)
        handDictList = [{'verts3d': verts3d, 'verts2d': verts2d}]
        otherInfo = {'length': (slen['left'] + slen['right']) / 2, 'root_rel': root_rel,
                     'verts3d_MANO_list': {'left': pred['left'], 'right': pred['right']},
                     'verts2d_MANO_list': {'left': [], 'right': []}}
        paramsDict = {'scale': scale, 'trans2d': trans2d, 'scalelength_left': slen['left'],
                      'scalelength_right': slen['right'], 'root_rel': root_rel}
        return result, paramsDict, handDictList, otherInfo


class HandNET_GCN(nn.Module):
    """lijun_model_graph.py:18-33."""

    def __init__(self, encoder, mid_model, decoder, cliff=False):
        super().__init__()
        self.encoder = encoder
        self.mid_model = mid_model
        self.decoder = decoder
        self.cliff = cliff

    _half = None

    def use_fp16_backbone(self, enable=True):
        """Inference only: encoder + mid_model with fp16 storage and folded BatchNorm (renderih_amd/half.py); see
        renderih_amd.model.HandNET_GCN.use_fp16_backbone."""
        if enable:
            from .half import HalfBackboneB
            self._half = HalfBackboneB(self.encoder, self.mid_model)
        else:
            self._half = None
        return self

    def forward(self, img):
        with ops.owned_bounds():        # (parameter bounds measured at the top stay valid until the forward returns)
            return self._forward(img)

    def _forward(self, img):
        ops.begin_forward(self)         # operand bounds of the three-product GEMM engine are per forward pass
        if self._half is not None and not self.training and not torch.is_grad_enabled():
            global_feature, fmaps = self._half(img)
        else:
            img_fmaps = self.encoder(img)
            global_feature, fmaps = self.mid_model(img_fmaps)
        return self.decoder(global_feature, fmaps)


def load_decoder(cfg, encoder_info, asset_root=None, mano_flag=True):
    """decoder_lijun_graph.py:318-358; assets as renderih_amd.model.load_decoder (misc/*.pkl when present)."""
    from .model import load_decoder as _load_a
    a = _load_a(cfg, encoder_info, asset_root, decoder_cls=decoder, extra=dict(cfg=cfg, mano_flag=mano_flag))
    return a


def load_graph_model(cfg=None, cliff=False, mano_flag=True):
    """lijun_model_graph.py:36-70.  `mano_flag` is main/config.py:80 (True in the reference checkout): it only decides
    whether the unused ParamRegressor parameters exist.  MODEL_PRETRAIN_PATH is loaded when the file exists, with the
    reference's `module.` prefix handling."""
    import os
    from .config import load_cfg
    if cfg is None or isinstance(cfg, str):
        cfg = load_cfg(cfg)
    encoder, mid_model = load_encoder(cfg)
    dec = load_decoder(cfg, mid_model.get_info(), mano_flag=mano_flag)
    model = HandNET_GCN(encoder, mid_model, dec, cliff)
    path = str(cfg.MODEL_PARAM.MODEL_PRETRAIN_PATH) if hasattr(cfg, 'MODEL_PARAM') else 'none'
    if os.path.exists(path):
        state = torch.load(path, map_location='cpu')
        if any(k.startswith('module.') for k in state):
            state = {k[7:]: v for k, v in state.items()}
        model.load_state_dict(state, strict=False)
    return model


def _mano_data(cfg, side, root=None):
    """MISC.MANO_PATH/MANO_{LEFT,RIGHT}.pkl when present (licence-gated, not shipped); otherwise the seeded synthetic
    MANO-shaped model of renderih_amd.assets (tests, benchmarks)."""
    import os
    from . import assets
    base = root or os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    path = os.path.join(base, str(cfg.MISC.MANO_PATH), 'MANO_%s.pkl' % side.upper())
    return path if os.path.exists(path) else assets.synthetic_mano_dict(side)


def load_new_model(cfg=None, cliff=False):
    """common/myhand/lijun_model_newgraph.py:35-70: encoder_lijun + decoder_lijun_mano (MANO layer in the forward)."""
    from .config import load_cfg
    from .model import load_decoder as _load_a
    if cfg is None or isinstance(cfg, str):
        cfg = load_cfg(cfg)
    encoder, mid_model = load_encoder(cfg)
    dec = _load_a(cfg, mid_model.get_info(), None, decoder_cls=decoder_mano,
                  extra=dict(cfg=cfg, mano_left=_mano_data(cfg, 'left'), mano_right=_mano_data(cfg, 'right')))
    return HandNET_GCN(encoder, mid_model, dec, cliff)


def build_new_model(dropout=0.05):
    from .config import load_cfg
    cfg = load_cfg(None)
    cfg.TRAIN.dropout = dropout
    return load_new_model(cfg)


def build_graph_model(dropout=0.05):
    from .config import load_cfg
    cfg = load_cfg(None)
    cfg.TRAIN.dropout = dropout
    return load_graph_model(cfg)
