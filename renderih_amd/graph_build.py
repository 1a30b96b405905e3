"""Offline generator of the hand-graph assets (`misc/graph_{left,right}.pkl` of the reference) from the MANO mesh topology
alone, so that they can be rebuilt without `misc.tar`.  CPU / numpy only, not on the GPU path.

What it reproduces (reference: models/model_zoo/coarsening.py:48-428 `build_graph`, itself derived from Defferrard's
graph-CNN coarsening): four rounds of heavy-edge matching that pair the 778 mesh vertices into a balanced binary cluster
tree (padded with fake vertices to 1008 = 63 * 16), the vertex order that makes every pooling step "merge neighbours 2k,
2k+1", the permuted adjacency and the rescaled normalised Laplacian of each level (1008/504/252/126/63 nodes), and the
index maps between mesh order and graph order.  The matching is order dependent, so the visiting order, the tie rule
(first strict maximum), the float64 arithmetic of the pairing score and even the reference's use of "the first stored
weight of a node" in that score are kept; `tests/test_graph_build.py` checks the result against the packaged asset that
the reference's own `build_graph` produced (permutations exactly, Laplacians to 1e-5).

    python -m renderih_amd.graph_build --out misc       # writes misc/graph_left.pkl, misc/graph_right.pkl
"""
import numpy as np
import scipy.sparse as sp
from scipy.sparse.linalg import eigsh


def mesh_adjacency(faces, n=None):
    """Symmetric 0/1 adjacency (float64 CSR, like the reference's after its float arithmetic) of the triangle mesh
    (coarsening.py:350-376)."""
    faces = np.asarray(faces, np.int64)
    n = int(faces.max()) + 1 if n is None else n
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [0, 2]]], 0)
    e = np.concatenate([e, e[:, ::-1]], 0)
    e = np.unique(e[e[:, 0] != e[:, 1]], axis=0)
    return sp.csr_matrix((np.ones(len(e), np.float64), (e[:, 0], e[:, 1])), shape=(n, n))


def _match_level(W, order, weights):
    """One round of heavy-edge matching (coarsening.py:158-216): visit the nodes in `order`; an unmatched node takes the
    unmatched neighbour with the largest score (2 w_ij + f_i + f_j) / (d_i + d_j), f = first stored weight of a node,
    first strict maximum winning; float64 throughout.  Returns the cluster id of every node.
    The reference's row-length bookkeeping (coarsening.py:171-176) counts the first entry of every row for the row
    before it: node 0 therefore also scans the first stored entry of node 1, and the last node misses its last
    neighbour.  The published assets were generated that way, so the scan windows are reproduced."""
    C = sp.csc_matrix(W)
    C.sort_indices()
    ptr, nbr, val = C.indptr, C.indices, C.data.astype(np.float64)
    n = W.shape[0]
    first = val[ptr[:-1]]
    span = np.diff(ptr)
    if n > 1:
        span[0] += 1
        span[-1] -= 1
    wts = np.asarray(weights, np.float64)
    matched = np.zeros(n, bool)
    cluster = np.zeros(n, np.int32)
    nxt = 0
    for t in order:
        if matched[t]:
            continue
        matched[t] = True
        lo, hi = ptr[t], ptr[t] + span[t]
        cand = nbr[lo:hi]
        score = (2. * val[lo:hi] + first[t] + first[cand]) * 1. / (wts[t] + wts[cand] + 1e-9)
        score = np.where(matched[cand], 0., score)
        best = -1
        if len(score) and score.max() > 0:
            best = int(cand[int(np.argmax(score))])          # argmax = first occurrence of the maximum
        cluster[t] = nxt
        if best >= 0:
            cluster[best] = nxt
            matched[best] = True
        nxt += 1
    return cluster


def heavy_edge_matching(W, levels):
    """coarsening.py:72-154: graphs[0..levels] (coarser and coarser, edge weights summed) and parents[0..levels-1]."""
    W = sp.csr_matrix(W)
    graphs, parents = [W], []
    degree = np.asarray(W.sum(axis=0)).ravel() - W.diagonal()
    order = np.argsort(np.asarray(W.sum(axis=0)).squeeze())
    for _ in range(levels):
        cluster = _match_level(W, order, degree)
        parents.append(cluster)
        coo = W.tocoo()
        m = int(cluster.max()) + 1
        W = sp.csr_matrix((coo.data, (cluster[coo.row], cluster[coo.col])), shape=(m, m))
        W.eliminate_zeros()
        graphs.append(W)
        degree = np.asarray(W.sum(axis=0)).ravel()
        order = np.argsort(np.asarray(W.sum(axis=0)).squeeze())
    return graphs, parents


def tree_orders(parents):
    """coarsening.py:219-262: node order per level such that nodes 2k, 2k+1 of a level are the children of node k of the
    next one; missing children become fake nodes numbered after the real ones.  Finest level first."""
    if not parents:
        return []
    orders = [list(range(int(parents[-1].max()) + 1))]
    for par in parents[::-1]:
        fake = len(par)
        kids = [[] for _ in range(int(par.max()) + 1)]
        for child, p in enumerate(par):
            kids[p].append(child)
        layer = []
        for node in orders[-1]:
            ch = list(kids[node]) if node < len(kids) else []
            assert len(ch) <= 2
            while len(ch) < 2:
                ch.append(fake)
                fake += 1
            layer.extend(ch)
        orders.append(layer)
    return orders[::-1]


def permute_adjacency(A, order):
    """coarsening.py:270-295: pad with isolated fake nodes up to len(order) and renumber node k to its slot in `order`."""
    m = len(order)
    slot = np.argsort(order)
    coo = sp.coo_matrix(A)
    return sp.coo_matrix((coo.data, (slot[coo.row], slot[coo.col])), shape=(m, m))


def normalized_laplacian(W):
    """coarsening.py:10-29: I - D^-1/2 W D^-1/2 (CSR); isolated nodes get a plain 1 on the diagonal."""
    d = np.asarray(W.sum(axis=0)).ravel() + np.spacing(np.array(0, W.dtype))
    D = sp.diags(1 / np.sqrt(d), 0)
    L = sp.identity(d.size, dtype=W.dtype) - D * W * D
    return sp.csr_matrix(L)


def fold_fake_nodes(order, levels, n_real):
    """coarsening.py:379-394 `cut_perm`: every fake slot takes the vertex (block) of its sibling, level by level, so that
    gathering mesh vertices in graph order never reads a fake index."""
    p = np.asarray(order, np.int64).copy()
    p[p > n_real - 1] = -1
    for lvl in range(levels):
        blk = p.reshape(-1, 2 ** (lvl + 1))
        half = blk.shape[1] // 2
        left_fake = blk[:, 0] == -1
        blk[left_fake, :half] = blk[left_fake, half:]
        right_fake = blk[:, half] == -1
        blk[right_fake, half:] = blk[right_fake, :half]
        p = blk.reshape(-1)
    return p.tolist()


def build_graph(faces, coarsening_levels=4):
    """Same dictionary as the reference's `build_graph` (coarsening.py:397-428)."""
    import torch
    faces = np.asarray(faces)
    n = int(faces.max()) + 1
    adj = mesh_adjacency(faces, n)
    graphs, parents = heavy_edge_matching(adj, coarsening_levels)
    orders = tree_orders(parents)
    adjs, laps = [], []
    for i, A in enumerate(graphs):
        if i < coarsening_levels:
            A = permute_adjacency(A, orders[i])
        A = sp.csr_matrix(A)
        A.eliminate_zeros()
        adjs.append(A)
        laps.append(normalized_laplacian(A))
    perm = orders[0]
    for i in range(coarsening_levels):
        lmax = eigsh(laps[i], k=1, which='LM', return_eigenvectors=False)[0]
        L = laps[i]
        L /= lmax * 2                       # the reference divides by 2*lmax here (coarsening.py:36), kept
        L -= sp.identity(L.shape[0], format='csr', dtype=L.dtype)
        laps[i] = L
    inverse = np.empty(len(perm), np.int64)
    inverse[np.asarray(perm)] = np.arange(len(perm))
    return {'mesh_faces': faces, 'mesh_adj': adj,
            'graph_mask': torch.from_numpy((np.asarray(perm) < n).astype(float)).float(),
            'coarsen_graphs_adj': adjs, 'coarsen_graphs_L': laps,
            'graph_perm': fold_fake_nodes(perm, coarsening_levels, n), 'graph_perm_reverse': inverse}


def main():
    import argparse
    import os
    import pickle
    from . import assets
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--out', default='misc', help='directory for graph_left.pkl / graph_right.pkl')
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    for side in ('left', 'right'):
        g = build_graph(assets.hand_faces(side), 4)
        path = os.path.join(args.out, 'graph_%s.pkl' % side)
        with open(path, 'wb') as f:
            pickle.dump(g, f)
        print('wrote', path, [L.shape[0] for L in g['coarsen_graphs_L']])


if __name__ == '__main__':
    main()
