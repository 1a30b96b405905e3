"""GCN + attention blocks of the mesh decoder on the HIP ops (drop-in for models/model_attn/*.py).

Parameter names follow the reference (`GCN_blocks.M.{norm1,fc1,norm2,fc2,shortcut,norm3}`, `w_qs/w_ks/w_vs/fc`,
`layer_norm`, `ff.{layer_norm,fc1,fc2}`, `L_self_attn_layer`, `ffL/ffR` ...).  nn.Linear / nn.LayerNorm /
nn.Embedding objects only hold parameters; compute goes through `renderih_amd.ops`.

Reference quirks kept (SURVEY.md notes N2, N5, N6): `norm1` of every GCN_ResBlock is dead; inter_attn shares
w_qs/w_ks/w_vs/fc between the hands; LayerNorm eps is 1e-6; Chebyshev features are interleaved (f, k).
"""
import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn

import os

from . import ops, streams

# the image-grid encoders of the decoder in a side stream (DualGraphLayer.forward_pair); needs streams.SIDE > 0 as well.
# Built, parity-tested (tests/test_gpu_model.py::test_decoder_side_stream_changes_nothing) and REFUTED by measurement: the
# ResNet50 step is 1.4 % slower with it (36.56 against 36.03 ms same-box, profiles/r03/ab/g4_decoder_fork_*.log) -- ten
# launch-floor kernels per level hidden, a fork / join pair per level and direction added to the graph -- so it is opt-in.
DECODER_FORK = os.environ.get('RIH_DECODER_FORK', '0') == '1'


def _xavier(layer):
    """models/model_attn/*.py `weights_init`: xavier_uniform_ on Conv2d/Linear weights, zero bias."""
    if isinstance(layer, (nn.Conv2d, nn.Linear)):
        nn.init.xavier_uniform_(layer.weight.data)
        if isinstance(layer, nn.Linear) and layer.bias is not None:
            nn.init.constant_(layer.bias.data, 0.0)


class DropCtx:
    """Dropout bookkeeping for one forward: probability, base seed and a call counter (each dropout site draws a
    distinct stream of the counter-based RNG; the mask is recomputed, not stored, in backward)."""

    def __init__(self, p=0.0, training=False):
        self.p = float(p) if training else 0.0
        if self.p > 0 and ops.DROPOUT_SEED_TENSOR is None:
            self.base = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())     # host RNG (torch.manual_seed controls it)
        else:
            self.base = 0x5EED      # the per-step entropy comes from the device-resident seed word (hipGraph replay)
        self.n = 0

    def seed(self):
        self.n += 1
        return (self.base << 20) + self.n * 0x9E3779B1


def _lin(m, x, residual=None, relu=False):
    return ops.linear(x, m.weight, m.bias, residual=residual, relu=relu)


def _lin_drop(dc, m, x, relu=False):
    """dropout(act(linear(x))): the dropout rides in the GEMM's epilogue when ops.GEMM_DROPOUT is set (same mask stream)."""
    if dc.p > 0 and ops.GEMM_DROPOUT:
        return ops.linear(x, m.weight, m.bias, relu=relu, drop=(dc.p, dc.seed()))
    h = _lin(m, x, relu=relu)
    return _drop_add(dc, None, h) if dc.p > 0 else h


def _lin_drop_pair(dc, mL, mR, X, relu=False):
    if dc.p > 0 and ops.GEMM_DROPOUT:
        return ops.linear_pair(X, mL, mR, relu=relu, drop=(dc.p, dc.seed()))
    h = ops.linear_pair(X, mL, mR, relu=relu)
    return _drop_add(dc, None, h) if dc.p > 0 else h


def _ln(m, x, x2=None, relu=False):
    return ops.layernorm(x, m.weight, m.bias, eps=m.eps, x2=x2, relu=relu)


def _drop_add(dc, a, b):
    """a + dropout(b)."""
    return ops.add_dropout(a, b, dc.p, dc.seed() if dc.p > 0 else 0)


def _lin_drop_res(dc, m, x, res):
    """res + dropout(linear(x)); the add is fused into the GEMM epilogue when dropout is off."""
    if dc.p > 0 and ops.GEMM_DROPOUT:
        return ops.linear(x, m.weight, m.bias, residual=res, drop=(dc.p, dc.seed()))
    if dc.p > 0:
        return _drop_add(dc, res, _lin(m, x))
    return _lin(m, x, residual=res)


def _seeds3(dc):
    """The three mask streams of (output projection, MLP hidden, MLP output) in the order the standalone sequence draws them."""
    return (dc.seed(), dc.seed(), dc.seed()) if dc.p > 0 else (0, 0, 0)


def _lin_pair(mL, mR, X, residual=None, relu=False):
    return ops.linear_pair(X, mL, mR, residual=residual, relu=relu)


def _lin_drop_res_pair(dc, mL, mR, X, res):
    if dc.p > 0 and ops.GEMM_DROPOUT:
        return ops.linear_pair(X, mL, mR, residual=res, drop=(dc.p, dc.seed()))
    if dc.p > 0:
        return _drop_add(dc, res, _lin_pair(mL, mR, X))
    return _lin_pair(mL, mR, X, residual=res)


# ---- hands-stacked execution -------------------------------------------------------------------------------------
# `forward_pair(left_module, right_module, X, ...)` below run the left- and the right-hand instance of a block on
# X [2, B, V, D] (slice 0 = left) with one launch per layer pair instead of one per hand; the arithmetic per hand is
# that of `forward`.  DualGraph.forward takes this route when ops.PAIR_HANDS is set.


class GraphCSR:
    """Device CSR of a graph Laplacian and of its transpose (for the backward), fp32 like the reference's dense L."""

    def __init__(self, L):
        L = sp.csr_matrix(L).astype(np.float32)
        L.sort_indices()
        Lt = sp.csr_matrix(L.T)
        Lt.sort_indices()
        self.host = (L, Lt)
        self.dev = None
        self.n = L.shape[0]

    def on(self, device):
        if self.dev is None or self.dev[0][0].device != device:
            def up(M):
                return (torch.as_tensor(M.indptr.astype(np.int32), device=device),
                        torch.as_tensor(M.indices.astype(np.int32), device=device),
                        torch.as_tensor(M.data.astype(np.float32), device=device))
            self.dev = (up(self.host[0]), up(self.host[1]))
        return self.dev


class GCN_ResBlock(nn.Module):
    """models/model_attn/gcn.py:72-110."""

    def __init__(self, in_dim, out_dim, mid_dim, graph_L, graph_k, drop_out=0.01):
        super().__init__()
        assert graph_k == 2, 'the reference configuration uses Chebyshev order K=2'
        if isinstance(graph_L, np.ndarray):
            dense = torch.from_numpy(graph_L).float()
        else:
            dense = torch.from_numpy(np.asarray(sp.csr_matrix(graph_L).astype(np.float32).todense())).float()
        self.register_buffer('graph_L', dense, persistent=False)     # same non-persistent buffer as the reference
        self._csr = GraphCSR(dense.numpy())
        self._pair_cache = {}
        self.graph_k = graph_k
        self.in_dim = in_dim
        self.norm1 = nn.LayerNorm(in_dim, eps=1e-6)                  # dead in the reference forward (N2); kept for keys
        self.fc1 = nn.Linear(in_dim * graph_k, mid_dim)
        self.norm2 = nn.LayerNorm(out_dim, eps=1e-6)
        self.fc2 = nn.Linear(mid_dim * graph_k, out_dim)
        self.dropout = nn.Dropout(drop_out)
        self.shortcut = nn.Linear(in_dim, out_dim)
        self.norm3 = nn.LayerNorm(out_dim, eps=1e-6)

    def forward(self, x, dc, relu_out):
        csr, csr_t = self._csr.on(x.device)
        x1 = _lin(self.fc1, ops.cheby_features(x, csr, csr_t))
        x1 = _ln(self.norm2, x1, relu=True)
        x1c = ops.cheby_features(x1, csr, csr_t)
        if dc.p > 0:
            x1 = _drop_add(dc, None, _lin(self.fc2, x1c))
            return _ln(self.norm3, x1, x2=_lin(self.shortcut, x), relu=relu_out)
        x2 = _lin(self.shortcut, x)
        return _ln(self.norm3, _lin(self.fc2, x1c, residual=x2), relu=relu_out)


    @staticmethod
    def forward_pair(L, R, X, dc, relu_out):
        _, B, V, _ = X.shape
        if L._same_graph(R):
            csr, csr_t = L._csr.on(X.device)

            def cheb(t):
                return ops.cheby_features(t.reshape(2 * B, V, t.shape[-1]), csr, csr_t).view(2, B, V, -1)
        else:           # user-supplied graphs that differ between the hands: per-hand gathers, stacked again
            cl, cr = L._csr.on(X.device), R._csr.on(X.device)

            def cheb(t):
                return torch.stack([ops.cheby_features(t[0], *cl), ops.cheby_features(t[1], *cr)])
        x1 = _lin_pair(L.fc1, R.fc1, cheb(X))
        x1 = ops.layernorm_pair(x1, L.norm2, R.norm2, relu=True)
        x1c = cheb(x1)
        if dc.p > 0:
            x1 = _lin_drop_pair(dc, L.fc2, R.fc2, x1c)
            return ops.layernorm_pair(x1, L.norm3, R.norm3, x2=_lin_pair(L.shortcut, R.shortcut, X), relu=relu_out)
        x2 = _lin_pair(L.shortcut, R.shortcut, X)
        return ops.layernorm_pair(_lin_pair(L.fc2, R.fc2, x1c, residual=x2), L.norm3, R.norm3, relu=relu_out)

    def _same_graph(self, other):
        key = ('same', id(other))
        if self._pair_cache.get('key') != key:
            self._pair_cache['key'] = key
            self._pair_cache['same'] = bool(self.graph_L.shape == other.graph_L.shape and
                                            torch.equal(self.graph_L.cpu(), other.graph_L.cpu()))
        return self._pair_cache['same']


class GraphLayer(nn.Module):
    """models/model_attn/gcn.py:113-138."""

    def __init__(self, in_dim=256, out_dim=256, graph_L=None, graph_k=2, graph_layer_num=3, drop_out=0.01):
        super().__init__()
        self.GCN_blocks = nn.ModuleList()
        self.GCN_blocks.append(GCN_ResBlock(in_dim, out_dim, out_dim, graph_L, graph_k, drop_out))
        for _ in range(graph_layer_num - 1):
            self.GCN_blocks.append(GCN_ResBlock(out_dim, out_dim, out_dim, graph_L, graph_k, drop_out))
        for m in self.modules():
            _xavier(m)

    def forward(self, x, dc):
        n = len(self.GCN_blocks)
        for i, blk in enumerate(self.GCN_blocks):
            x = blk(x, dc, relu_out=(i != n - 1))        # F.relu between blocks fused into norm3's kernel
        return x


    @staticmethod
    def forward_pair(L, R, X, dc):
        n = len(L.GCN_blocks)
        for i, (bl, br) in enumerate(zip(L.GCN_blocks, R.GCN_blocks)):
            X = GCN_ResBlock.forward_pair(bl, br, X, dc, relu_out=(i != n - 1))
        return X


class MLP_res_block(nn.Module):
    """models/model_attn/self_attn.py:17-33."""

    def __init__(self, in_dim, hid_dim, dropout=0.1):
        super().__init__()
        self.layer_norm = nn.LayerNorm(in_dim, eps=1e-6)
        self.fc1 = nn.Linear(in_dim, hid_dim)
        self.fc2 = nn.Linear(hid_dim, in_dim)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)

    def forward(self, x, dc):
        y, x = ops.layernorm_skip(x, self.layer_norm.weight, self.layer_norm.bias, eps=self.layer_norm.eps)
        h = _lin_drop(dc, self.fc1, y, relu=True)
        return _lin_drop_res(dc, self.fc2, h, x)


    @staticmethod
    def forward_pair(L, R, X, dc):
        y, X = ops.layernorm_pair_skip(X, L.layer_norm, R.layer_norm)
        h = _lin_drop_pair(dc, L.fc1, R.fc1, y, relu=True)
        return _lin_drop_res_pair(dc, L.fc2, R.fc2, h, X)


class SelfAttn(nn.Module):
    """models/model_attn/self_attn.py:36-85 (pre-LN multi-head self attention + MLP block)."""

    def __init__(self, f_dim, hid_dim=None, n_heads=4, d_q=None, d_v=None, dropout=0.1):
        super().__init__()
        d_q = f_dim // n_heads if d_q is None else d_q
        d_v = f_dim // n_heads if d_v is None else d_v
        hid_dim = f_dim if hid_dim is None else hid_dim
        assert d_q == d_v == f_dim // n_heads
        self.n_heads, self.d_q, self.d_v, self.f_dim = n_heads, d_q, d_v, f_dim
        self.norm = d_q ** 0.5
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.w_qs = nn.Linear(f_dim, n_heads * d_q)
        self.w_ks = nn.Linear(f_dim, n_heads * d_q)
        self.w_vs = nn.Linear(f_dim, n_heads * d_v)
        self.layer_norm = nn.LayerNorm(f_dim, eps=1e-6)
        self.fc = nn.Linear(n_heads * d_v, f_dim)
        self.ff = MLP_res_block(f_dim, hid_dim, dropout)

    def forward(self, x, dc):
        y, x = ops.layernorm_skip(x, self.layer_norm.weight, self.layer_norm.bias, eps=self.layer_norm.eps)
        # one fused QKV projection: the three nn.Linear parameters are stacked (a copy) into a [3D, D] operand
        w = torch.cat([self.w_qs.weight, self.w_ks.weight, self.w_vs.weight], 0)
        b = torch.cat([self.w_qs.bias, self.w_ks.bias, self.w_vs.bias], 0)
        o = ops.self_attention_packed(ops.linear(y, w, b), self.n_heads, dc.p, dc.seed() if dc.p > 0 else 0)
        x = _lin_drop_res(dc, self.fc, o, x)
        return self.ff(x, dc)


    @staticmethod
    def forward_pair(L, R, X, dc):
        _, B, S, D = X.shape
        # both hands' fused QKV operands stacked once into [2, 3D, D]
        w = torch.cat([L.w_qs.weight, L.w_ks.weight, L.w_vs.weight, R.w_qs.weight, R.w_ks.weight, R.w_vs.weight], 0)
        b = torch.cat([L.w_qs.bias, L.w_ks.bias, L.w_vs.bias, R.w_qs.bias, R.w_ks.bias, R.w_vs.bias], 0)
        y, X = ops.layernorm_pair_skip(X, L.layer_norm, R.layer_norm)
        qkv = ops.LinearPairFn.apply(y, w.view(2, 3 * D, D), None, b.view(2, 3 * D), None, None, False)
        o = ops.self_attention_packed(qkv.view(2 * B, S, 3 * D), L.n_heads, dc.p, dc.seed() if dc.p > 0 else 0)
        X = _lin_drop_res_pair(dc, L.fc, R.fc, o.view(2, B, S, D), X)
        return MLP_res_block.forward_pair(L.ff, R.ff, X, dc)


class img_feat_to_grid(nn.Module):
    """models/model_attn/img_attn.py:38-67: patch conv + ReLU -> 64 tokens + position embedding -> SelfAttn."""

    def __init__(self, img_size, img_f_dim, grid_size, grid_f_dim, n_heads=4, dropout=0.01):
        super().__init__()
        self.img_f_dim, self.img_size, self.grid_f_dim, self.grid_size = img_f_dim, img_size, grid_f_dim, grid_size
        self.position_embeddings = nn.Embedding(grid_size * grid_size, grid_f_dim)
        patch = img_size // grid_size
        self.proj = nn.Conv2d(img_f_dim, grid_f_dim, kernel_size=patch, stride=patch)
        self.self_attn = SelfAttn(grid_f_dim, n_heads=n_heads, hid_dim=grid_f_dim, dropout=dropout)

    def forward(self, img, dc):
        B, H, W, Cc = img.shape                                  # NHWC
        assert Cc == self.img_f_dim and H == self.img_size and W == self.img_size
        g = ops.conv2d(img, self.proj.weight, self.proj.bias, stride=self.proj.stride[0], pad=0, relu=True)
        g = g.view(B, self.grid_size * self.grid_size, self.grid_f_dim)      # token = (h, w) row-major, as the reference
        g = ops.add_rows_bcast(g, self.position_embeddings.weight)
        return self.self_attn(g, dc)


    @staticmethod
    def forward_pair(L, R, img, dc):
        B = img.shape[0]
        g = ops.patch_conv_pair(img, L.proj, R.proj)                 # [2,B,gs,gs,Dg], ReLU fused
        g = g.view(2, B, L.grid_size * L.grid_size, L.grid_f_dim)
        g = ops.add_rows_pair(g, L.position_embeddings.weight, R.position_embeddings.weight)
        return SelfAttn.forward_pair(L.self_attn, R.self_attn, g, dc)


class img_attn(nn.Module):
    """models/model_attn/img_attn.py:70-92."""

    def __init__(self, verts_f_dim, img_f_dim, n_heads=4, d_q=None, d_v=None, dropout=0.1):
        super().__init__()
        self.img_f_dim, self.verts_f_dim = img_f_dim, verts_f_dim
        self.fc = nn.Linear(img_f_dim, verts_f_dim)
        self.Attn = SelfAttn(verts_f_dim, n_heads=n_heads, hid_dim=verts_f_dim, dropout=dropout)

    def forward(self, verts_f, img_f, dc):
        V = verts_f.shape[1]
        x = torch.cat([verts_f, _lin(self.fc, img_f)], dim=1)    # token concat / slice: copies only
        x = self.Attn(x, dc)
        return x[:, :V]


    @staticmethod
    def forward_pair(L, R, X, img_f, dc):
        V = X.shape[2]
        x = torch.cat([X, _lin_pair(L.fc, R.fc, img_f)], dim=2)
        x = SelfAttn.forward_pair(L.Attn, R.Attn, x, dc)
        return x[:, :, :V]


class img_ex(nn.Module):
    """models/model_attn/img_attn.py:95-113."""

    def __init__(self, img_size, img_f_dim, grid_size, grid_f_dim, verts_f_dim, n_heads=4, dropout=0.01):
        super().__init__()
        self.verts_f_dim = verts_f_dim
        self.encoder = img_feat_to_grid(img_size, img_f_dim, grid_size, grid_f_dim, n_heads, dropout)
        self.attn = img_attn(verts_f_dim, grid_f_dim, n_heads=n_heads, dropout=dropout)
        for m in self.modules():
            _xavier(m)

    def forward(self, img, verts_f, dc):
        return self.attn(verts_f, self.encoder(img, dc), dc)


    @staticmethod
    def forward_pair(L, R, img, X, dc):
        return img_attn.forward_pair(L.attn, R.attn, X, img_feat_to_grid.forward_pair(L.encoder, R.encoder, img, dc), dc)


class inter_attn(nn.Module):
    """models/model_attn/inter_attn.py:38-125: per-hand SelfAttn, then cross-hand attention with shared projections."""

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

    def forward(self, Lf, Rf, dc):
        Lf = self.L_self_attn_layer(Lf, dc)
        Rf = self.R_self_attn_layer(Rf, dc)
        L2 = _ln(self.layer_norm1, Lf)
        R2 = _ln(self.layer_norm2, Rf)
        # shared projections (N5): stacked once into a [3D, D] operand, one fused QKV GEMM per hand
        w = torch.cat([self.w_qs.weight, self.w_ks.weight, self.w_vs.weight], 0)
        b = torch.cat([self.w_qs.bias, self.w_ks.bias, self.w_vs.bias], 0)
        sd = (lambda: dc.seed()) if dc.p > 0 else (lambda: 0)
        # feat_R2L = softmax(Lq Rk^T) Rv, feat_L2R = softmax(Rq Lk^T) Lv  (inter_attn.py:93-104)
        feat_R2L, feat_L2R = ops.cross_attention_packed(ops.linear(L2, w, b), ops.linear(R2, w, b), self.n_heads,
                                                        dc.p, sd(), sd())
        Lf = self.ffL(_lin_drop_res(dc, self.fc, feat_R2L, Lf), dc)
        Rf = self.ffR(_lin_drop_res(dc, self.fc, feat_L2R, Rf), dc)
        return Lf, Rf


    def forward_pair(self, X, dc):
        X = SelfAttn.forward_pair(self.L_self_attn_layer, self.R_self_attn_layer, X, dc)
        w = torch.cat([self.w_qs.weight, self.w_ks.weight, self.w_vs.weight], 0)
        b = torch.cat([self.w_qs.bias, self.w_ks.bias, self.w_vs.bias], 0)
        sd = (lambda: dc.seed()) if dc.p > 0 else (lambda: 0)
        # shared projections (N5): ONE fused QKV GEMM over both hands' rows, then the two cross-hand directions
        qkv = ops.linear(ops.layernorm_pair(X, self.layer_norm1, self.layer_norm2), w, b)
        feat = ops.cross_attention_stacked(qkv, self.n_heads, dc.p, sd(), sd())
        return MLP_res_block.forward_pair(self.ffL, self.ffR, _lin_drop_res(dc, self.fc, feat, X), dc)


class DualGraphLayer(nn.Module):
    """models/model_attn/DualGraph.py:21-91."""

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

    def forward(self, Lf, Rf, img_f, dc):
        assert Lf.shape[1] == self.verts_num and Lf.shape[2] == self.verts_in_dim
        pe = self.position_embeddings.weight
        Lf = self.graph_left(ops.add_rows_bcast(Lf, pe), dc)
        Rf = self.graph_right(ops.add_rows_bcast(Rf, pe), dc)
        Lf = self.img_ex_left(img_f, Lf, dc)
        Rf = self.img_ex_right(img_f, Rf, dc)
        return self.attn(Lf, Rf, dc)


    def forward_pair(self, X, img_f, dc):
        _, B, V, D = X.shape
        assert V == self.verts_num and D == self.verts_in_dim
        X = ops.add_rows_bcast(X.reshape(2 * B, V, D), self.position_embeddings.weight).view(2, B, V, D)
        # the image-grid encoder (patch convolution + a SelfAttn block on 64 tokens) reads only the feature map: it runs in a
        # side stream beside the graph convolutions (streams.fork_join; issue order -- and with it the dropout seeds -- unchanged)
        L, R = self.img_ex_left, self.img_ex_right
        if DECODER_FORK:
            X, grid = streams.fork_join([lambda: GraphLayer.forward_pair(self.graph_left, self.graph_right, X, dc),
                                         lambda: img_feat_to_grid.forward_pair(L.encoder, R.encoder, img_f, dc)],
                                        reads=[X, img_f])
        else:
            X = GraphLayer.forward_pair(self.graph_left, self.graph_right, X, dc)
            grid = img_feat_to_grid.forward_pair(L.encoder, R.encoder, img_f, dc)
        X = img_attn.forward_pair(L.attn, R.attn, X, grid, dc)
        return self.attn.forward_pair(X, dc)


class DualGraph(nn.Module):
    """models/model_attn/DualGraph.py:94-139."""

    def __init__(self, verts_in_dim=(512, 256, 128), verts_out_dim=(256, 128, 64), graph_L_Left=None,
                 graph_L_Right=None, graph_k=(2, 2, 2), graph_layer_num=(4, 4, 4), img_size=(16, 32, 64),
                 img_f_dim=(256, 256, 256), grid_size=(8, 8, 16), grid_f_dim=(256, 128, 64), n_heads=4, dropout=0.01):
        super().__init__()
        self.layers = nn.ModuleList()
        for i in range(len(verts_in_dim)):
            self.layers.append(DualGraphLayer(verts_in_dim[i], verts_out_dim[i], graph_L_Left[i], graph_L_Right[i],
                                              graph_k[i], graph_layer_num[i], img_size[i], img_f_dim[i], grid_size[i],
                                              grid_f_dim[i], n_heads, dropout))
        self._up = {}

    def _upsample2(self, x):
        """nn.Upsample(scale_factor=2, nearest) along V (DualGraph.py:11-18) as a row gather."""
        V = x.shape[1]
        key = (V, x.device)
        if key not in self._up:
            self._up[key] = ops.RowIndex(np.arange(2 * V) // 2, V, x.device)
        return self._up[key](x)

    def forward(self, Lf, Rf, img_f_list, dc):
        if ops.PAIR_HANDS:
            X = self.forward_stacked(torch.stack([Lf, Rf]), img_f_list, dc)
            return X[0], X[1]
        for i, layer in enumerate(self.layers):
            Lf, Rf = layer(Lf, Rf, img_f_list[i], dc)
            if i != len(self.layers) - 1:
                Lf, Rf = self._upsample2(Lf), self._upsample2(Rf)
        return Lf, Rf

    def forward_stacked(self, X, img_f_list, dc):
        """X [2,B,V,D] (left, right) -> [2,B,4V,D']: every per-hand layer pair is one launch."""
        for i, layer in enumerate(self.layers):
            X = layer.forward_pair(X, img_f_list[i], dc)
            if i != len(self.layers) - 1:
                _, B, V, D = X.shape
                X = self._upsample2(X.reshape(2 * B, V, D)).view(2, B, 2 * V, D)
        return X
