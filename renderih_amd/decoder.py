"""Mesh decoder head on the HIP ops (drop-in for the reference's models/decoder.py:30-174).

Same constructor arguments, attributes (`converter`, `unsample_layer`, `get_upsample_weight`, `dense_coor`
buffer) and output dictionaries as the reference.  Feature maps arrive NHWC from `renderih_amd.encoder`.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .attn import DualGraph, DropCtx, _xavier, _lin, _ln

IMG_SIZE = 256        # dataset/dataset_utils.py:4


class GCN_vert_convert():
    """models/model_zoo/__init__.py:85-96 (index maps between MANO vertex order and coarsened-graph order).
    Works on any device with plain indexing, like the reference (used by the loss, core/Loss.py:141-142)."""

    def __init__(self, vertex_num=1, graph_perm_reverse=[0], graph_perm=[0]):
        self.graph_perm_reverse = graph_perm_reverse[:vertex_num]
        self.graph_perm = graph_perm
        self._dev = {}

    def _index(self, name, device):
        """The index list as a device tensor, uploaded once per device: indexing with a host list would copy it to
        the GPU on every call (a synchronous transfer that also cannot be captured in a hipGraph)."""
        key = (name, device)
        if key not in self._dev:
            self._dev[key] = torch.as_tensor(np.asarray(getattr(self, name), dtype=np.int64), device=device)
        return self._dev[key]

    def vert_to_GCN(self, x):
        return x[:, self._index('graph_perm', x.device)]

    def GCN_to_vert(self, x):
        return x[:, self._index('graph_perm_reverse', x.device)]


class decoder(nn.Module):
    drops_last_fmap = True              # forward() never reads fmaps[-1] (models/decoder.py:130)

    def __init__(self, global_feature_dim=2048, f_in_Dim=[256, 256, 256, 256], f_out_Dim=[128, 64, 32],
                 gcn_in_dim=[256, 128, 128], gcn_out_dim=[128, 128, 64], graph_k=2, graph_layer_num=4,
                 left_graph_dict={}, right_graph_dict={}, vertex_num=778, dense_coor=None, num_attn_heads=4,
                 upsample_weight=None, dropout=0.05, dual_graph_cls=DualGraph, mano_lists=True):
        super().__init__()
        self.mano_lists = mano_lists        # family (b) (renderih_amd/lijun.py) leaves the *_MANO_list outputs empty
        assert len(f_in_Dim) == 4
        f_in_Dim = f_in_Dim[:-1]
        assert len(gcn_in_dim) == 3
        for i in range(len(gcn_out_dim) - 1):
            assert gcn_out_dim[i] == gcn_in_dim[i + 1]

        graph_dict = {'left': left_graph_dict, 'right': right_graph_dict}
        graph_L = {}
        for hand_type in ['left', 'right']:
            # the reference reverses the caller's list in place (decoder.py:53-54); we reverse a copy
            graph_L[hand_type] = list(graph_dict[hand_type]['coarsen_graphs_L'])[::-1]

        self.vNum_in = graph_L['left'][0].shape[0]
        self.vNum_out = graph_L['left'][2].shape[0]
        self.vNum_all = graph_L['left'][-1].shape[0]
        self.vNum_mano = vertex_num
        self.gf_dim = global_feature_dim
        self.gcn_in_dim = gcn_in_dim
        self.gcn_out_dim = gcn_out_dim
        self.dropout_p = dropout

        if dense_coor is not None:
            self.register_buffer('dense_coor', torch.from_numpy(np.asarray(dense_coor)).float())

        self.converter = {}
        self._perm = {}
        for hand_type in ['left', 'right']:
            self.converter[hand_type] = GCN_vert_convert(vertex_num=self.vNum_mano,
                                                         graph_perm_reverse=graph_dict[hand_type]['graph_perm_reverse'],
                                                         graph_perm=graph_dict[hand_type]['graph_perm'])
            self._perm[hand_type] = (np.asarray(graph_dict[hand_type]['graph_perm'], dtype=np.int64),
                                     np.asarray(graph_dict[hand_type]['graph_perm_reverse'], dtype=np.int64)[:vertex_num])

        self.dual_gcn = dual_graph_cls(verts_in_dim=self.gcn_in_dim, verts_out_dim=self.gcn_out_dim,
                                  graph_L_Left=graph_L['left'][:3], graph_L_Right=graph_L['right'][:3],
                                  graph_k=[graph_k] * 3, graph_layer_num=[graph_layer_num] * 3, img_size=[8, 16, 32],
                                  img_f_dim=f_in_Dim, grid_size=[8, 8, 8], grid_f_dim=f_out_Dim,
                                  n_heads=num_attn_heads, dropout=dropout)

        self.gf_layer_left = nn.Sequential(*(nn.Linear(self.gf_dim, self.gcn_in_dim[0] - 3),
                                             nn.LayerNorm(self.gcn_in_dim[0] - 3, eps=1e-6)))
        self.gf_layer_right = nn.Sequential(*(nn.Linear(self.gf_dim, self.gcn_in_dim[0] - 3),
                                              nn.LayerNorm(self.gcn_in_dim[0] - 3, eps=1e-6)))
        self.unsample_layer = nn.Linear(self.vNum_out, self.vNum_mano, bias=False)
        self.coord_head = nn.Linear(self.gcn_out_dim[-1], 3)
        self.avg_head = nn.Linear(self.vNum_out, 1)
        self.params_head = nn.Linear(self.gcn_out_dim[-1], 3)

        for m in (self.gf_layer_left, self.gf_layer_right):
            for mm in m.modules():
                _xavier(mm)
        _xavier(self.coord_head)
        _xavier(self.avg_head)
        _xavier(self.params_head)
        if upsample_weight is not None:
            self.unsample_layer.load_state_dict({'weight': upsample_weight.to(self.unsample_layer.weight.data.device)})
        else:
            _xavier(self.unsample_layer)
        self._cache = {}

    def get_upsample_weight(self):
        return self.unsample_layer.weight.data

    def get_converter(self):
        return self.converter

    def get_hand_pe(self, bs, num=None):
        """decoder.py:118-126: per-vertex positional code = avg-pooled (dense_coor*2-1) in graph order."""
        if num is None:
            num = self.vNum_in
        dc = self.dense_coor * 2 - 1
        out = []
        for hand_type in ('left', 'right'):
            pe = dc[torch.as_tensor(self._perm[hand_type][0], device=dc.device)]
            p = pe.shape[0] // num
            out.append(pe.view(num, p, -1).mean(1).unsqueeze(0).repeat(bs, 1, 1))
        return out[0], out[1]

    def _pe_const(self, bs):
        """Constant of the forward (a function of the `dense_coor` buffer only); rebuilt when the buffer changes."""
        key = (bs, self.dense_coor.device, self.dense_coor._version, self.dense_coor.data_ptr())
        if self._cache.get('pe_key') != key:
            with torch.no_grad():
                self._cache['pe'] = tuple(t.contiguous() for t in self.get_hand_pe(bs, self.vNum_in))
            self._cache['pe_key'] = key
        return self._cache['pe']

    def _rowidx(self, name, idx, vin, device):
        key = (name, device)
        if key not in self._cache:
            self._cache[key] = ops.RowIndex(idx, vin, device)
        return self._cache[key]

    def _initial_features(self, x):
        """decoder.py:133-139: per-hand global feature (Linear + LN) tiled over the 63 coarse vertices + positional code."""
        bs, dev = x.shape[0], x.device
        pel, per = self._pe_const(bs)
        rep = self._rowidx('rep', np.zeros(self.vNum_in, np.int64), 1, dev)       # .unsqueeze(1).repeat(1, V, 1)
        gl = _ln(self.gf_layer_left[1], _lin(self.gf_layer_left[0], x))
        gr = _ln(self.gf_layer_right[1], _lin(self.gf_layer_right[0], x))
        return torch.cat([rep(gl.unsqueeze(1)), pel], dim=-1), torch.cat([rep(gr.unsqueeze(1)), per], dim=-1)

    def _stacked_heads(self, f):
        """The output heads, shared between the hands (decoder.py:143-163), on the hands-stacked features [2B,V,D]:
        camera (scale | trans2d) [2B,3], coarse vertices [2B,252,3] and their projection, up-sampled vertices transposed
        [2B,3,778]."""
        temp = _lin(self.avg_head, f.transpose(-1, -2).contiguous())[..., 0]
        temp = _lin(self.params_head, temp)
        v3 = _lin(self.coord_head, f)
        v2 = ops.projection_batch(temp[:, 0], temp[:, 1:], v3, IMG_SIZE)
        upT = ops.linear(v3.transpose(1, 2).contiguous(), self.unsample_layer.weight)
        return temp, v3, v2, upT

    def forward(self, x, fmaps):
        assert x.shape[1] == self.gf_dim
        fmaps = fmaps[:-1]
        bs = x.shape[0]
        dev = x.device
        dc = DropCtx(self.dropout_p, self.training)
        Lf, Rf = self._initial_features(x)

        scale, trans2d = {}, {}
        verts3d, verts2d = {}, {}
        result = {'verts3d': {}, 'verts2d': {}}
        if ops.PAIR_HANDS:
            # the heads are shared between the hands (decoder.py:143-163): run them once over the stacked batch
            f = self.dual_gcn.forward_stacked(torch.stack([Lf, Rf]), fmaps, dc)
            f = f.reshape(2 * bs, f.shape[2], f.shape[3])
            temp, v3, v2, upT = self._stacked_heads(f)
            up = upT.transpose(1, 2).contiguous()
            up2 = ops.projection_batch(temp[:, 0], temp[:, 1:], up, IMG_SIZE)
            for h, hand_type in enumerate(['left', 'right']):
                sl = slice(h * bs, (h + 1) * bs)
                scale[hand_type], trans2d[hand_type] = temp[sl, 0], temp[sl, 1:]
                verts3d[hand_type], verts2d[hand_type] = v3[sl], v2[sl]
                result['verts3d'][hand_type], result['verts2d'][hand_type] = up[sl], up2[sl]
            feats = None
        else:
            Lf, Rf = self.dual_gcn(Lf, Rf, fmaps, dc)
            feats = {'left': Lf, 'right': Rf}
        for hand_type in (['left', 'right'] if feats is not None else []):
            f = feats[hand_type]
            temp = _lin(self.avg_head, f.transpose(-1, -2).contiguous())[..., 0]
            temp = _lin(self.params_head, temp)
            scale[hand_type] = temp[:, 0]
            trans2d[hand_type] = temp[:, 1:]
            verts3d[hand_type] = _lin(self.coord_head, f)
            verts2d[hand_type] = ops.projection_batch(scale[hand_type], trans2d[hand_type], verts3d[hand_type], IMG_SIZE)
            up = ops.linear(verts3d[hand_type].transpose(1, 2).contiguous(), self.unsample_layer.weight)
            result['verts3d'][hand_type] = up.transpose(1, 2).contiguous()
            result['verts2d'][hand_type] = ops.projection_batch(scale[hand_type], trans2d[hand_type],
                                                               result['verts3d'][hand_type], IMG_SIZE)
        paramsDict = {'scale': scale, 'trans2d': trans2d}
        handDictList = [{'verts3d': verts3d, 'verts2d': verts2d}]

        otherInfo = {'verts3d_MANO_list': {'left': [], 'right': []}, 'verts2d_MANO_list': {'left': [], 'right': []}}
        for hand_type in (['left', 'right'] if self.mano_lists else []):
            p = self.vNum_all // self.vNum_out
            # graph_upsample(x, p) then GCN_to_vert (decoder.py:165-172) == one gather with idx = perm_reverse // p
            g = self._rowidx('mano_' + hand_type, self._perm[hand_type][1] // p, self.vNum_out, dev)
            otherInfo['verts3d_MANO_list'][hand_type].append(g(verts3d[hand_type]))
            otherInfo['verts2d_MANO_list'][hand_type].append(g(verts2d[hand_type]))
        return result, paramsDict, handDictList, otherInfo
