"""Hand-graph assets and synthetic stand-ins for the files the reference reads from misc/.

The reference decoder needs (models/decoder.py:177-210, utils/manoutils.py:77-103)
  misc/graph_{left,right}.pkl  -> Laplacians of the coarsened hand graph + vertex permutations
  misc/v_color.pkl             -> dense_coor [778,3]
  misc/upsample.pkl            -> upsample_weight [778,252]
  misc/mano/MANO_{LEFT,RIGHT}.pkl
misc.tar is not part of the reference checkout and MANO is licence gated.  The graph data is a pure
function of the MANO topology, which the reference ships as OBJ; `tests/golden/make_assets.py` ran the
reference's `build_graph` on it once and stored the result in `assets/hand_graph.npz`.  The other
files get seeded synthetic stand-ins with the right shapes/sparsity.  When the
real files exist (`load_graph_dict(path=...)`) they are used instead.
"""
import os
import pickle
import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_NPZ = os.path.join(_HERE, 'assets', 'hand_graph.npz')

MANO_PARENT = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]


def _npz():
    return np.load(_NPZ)


def load_graph_dict(side, path=None):
    """Return the dict `decoder(...)` consumes: 'coarsen_graphs_L' (list of 5 scipy CSR, fine->coarse,
    as `build_graph` returns them, coarsening.py:419-426), 'graph_perm', 'graph_perm_reverse',
    'mesh_faces'.  A fresh dict per call (the reference ctor reverses the L list in place)."""
    if path is not None and os.path.exists(path):
        with open(path, 'rb') as f:
            return pickle.load(f)
    z = _npz()
    Ls = []
    for lvl in range(5):
        n = int(z['%s_L%d_n' % (side, lvl)])
        Ls.append(sp.csr_matrix((z['%s_L%d_data' % (side, lvl)].astype(np.float32),
                                 z['%s_L%d_indices' % (side, lvl)],
                                 z['%s_L%d_indptr' % (side, lvl)]), shape=(n, n)))
    return {'mesh_faces': z['%s_faces' % side].astype(np.int64),
            'coarsen_graphs_L': Ls,
            'graph_perm': z['%s_perm' % side].astype(np.int64).tolist(),
            'graph_perm_reverse': z['%s_perm_reverse' % side].astype(np.int64)}


def hand_faces(side):
    return _npz()['%s_faces' % side].astype(np.int64)


def obj_template(side):
    return _npz()['%s_obj_verts' % side].astype(np.float32)


def synthetic_dense_coor(seed=0):
    """Stand-in for misc/v_color.pkl: [778,3] in [0,1]."""
    return np.random.RandomState(1000 + seed).rand(778, 3).astype(np.float32)


def synthetic_upsample_weight(seed=0):
    """Stand-in for misc/upsample.pkl: [778,252], rows sum to 1 (an interpolation matrix)."""
    rs = np.random.RandomState(2000 + seed)
    w = np.zeros((778, 252), np.float32)
    for v in range(778):
        idx = rs.choice(252, 3, replace=False)
        a = rs.rand(3).astype(np.float32) + 0.1
        w[v, idx] = a / a.sum()
    return w


def synthetic_mano_dict(side='right', seed=0):
    """A MANO-shaped dict (keys/shapes/sparsity of MANO_{LEFT,RIGHT}.pkl as models/manolayer.py:108-151
    reads them) built from the OBJ template and seeded random bases.  Not the licensed model."""
    rs = np.random.RandomState(3000 + seed + (0 if side == 'right' else 7))
    vt = obj_template(side).copy()
    vt = (vt - vt.mean(0)) / (np.abs(vt).max() * 8.0)          # ~ hand-sized, metres
    q, _ = np.linalg.qr(rs.randn(45, 45))
    weights = np.zeros((778, 16), np.float32)
    for v in range(778):
        j = rs.choice(16, 2, replace=False)
        a = rs.rand(2).astype(np.float32) + 0.05
        weights[v, j] = a / a.sum()
    rows, cols, vals = [], [], []
    for j in range(16):
        idx = rs.choice(778, 12, replace=False)
        a = rs.rand(12) + 0.05
        a = a / a.sum()
        rows += [j] * 12
        cols += idx.tolist()
        vals += a.tolist()
    J_reg = sp.csc_matrix((np.asarray(vals), (rows, cols)), shape=(16, 778))
    d = {
        'hands_components': q.astype(np.float64),
        'J_regressor': J_reg,
        'J': (J_reg @ vt).astype(np.float64),
        'weights': weights.astype(np.float64),
        'posedirs': (rs.randn(778, 3, 135) * 2e-3).astype(np.float64),
        'v_template': vt.astype(np.float64),
        'shapedirs': (rs.randn(778, 3, 10) * 5e-3).astype(np.float64),
        'hands_mean': (rs.randn(45) * 0.2).astype(np.float64),
        'f': hand_faces(side).astype(np.uint32),
        'kintree_table': np.array([[4294967295] + MANO_PARENT[1:], list(range(16))], dtype=np.int64),
    }
    return d


def write_synthetic_mano_pkl(path, side='right', seed=0):
    with open(path, 'wb') as f:
        pickle.dump(synthetic_mano_dict(side, seed), f)
    return path
