"""Regenerate renderih_amd/assets/hand_graph.npz from the reference (run in the build container).

The reference reads misc/graph_{left,right}.pkl (missing: misc.tar is not in the checkout) but can
rebuild them with models/model_zoo/coarsening.py:397-428 `build_graph(faces, 4)` on the MANO
topology shipped as OBJ in pose_data_optimize/helpful_py/{left,right}_hand_mano.obj.  We run that
reference code once and store its *outputs* (Laplacians as CSR float32, vertex permutations, faces,
OBJ vertex positions) as a small fixture; no reference source is copied.
"""
import os
import sys
import warnings
import numpy as np

sys.path.insert(0, os.path.dirname(__file__))
import ref_stubs  # noqa: E402

ref_stubs.install()
warnings.filterwarnings('ignore')
from models.model_zoo import build_graph  # noqa: E402  (reference)

OBJ = '/root/reference/pose_data_optimize/helpful_py/%s_hand_mano.obj'
OUT = os.path.join(os.path.dirname(__file__), '..', '..', 'renderih_amd', 'assets', 'hand_graph.npz')


def read_obj(path):
    v, f = [], []
    for line in open(path):
        if line.startswith('v '):
            v.append([float(t) for t in line.split()[1:4]])
        elif line.startswith('f '):
            f.append([int(t.split('/')[0]) - 1 for t in line.split()[1:4]])
    return np.asarray(v, np.float32), np.asarray(f, np.int32)


def main():
    out = {}
    for side in ('left', 'right'):
        verts, faces = read_obj(OBJ % side)
        assert verts.shape == (778, 3) and faces.shape == (1538, 3)
        g = build_graph(faces.astype(np.int64), 4)
        out['%s_faces' % side] = faces
        out['%s_obj_verts' % side] = verts
        out['%s_perm' % side] = np.asarray(g['graph_perm'], np.int32)
        out['%s_perm_reverse' % side] = np.asarray(g['graph_perm_reverse'], np.int32)
        # build_graph order: 1008, 504, 252, 126, 63 (decoder.py:53-54 reverses the list)
        for lvl, L in enumerate(g['coarsen_graphs_L']):
            L = L.tocsr()
            L.sort_indices()
            out['%s_L%d_indptr' % (side, lvl)] = L.indptr.astype(np.int32)
            out['%s_L%d_indices' % (side, lvl)] = L.indices.astype(np.int32)
            out['%s_L%d_data' % (side, lvl)] = L.data.astype(np.float32)
            out['%s_L%d_n' % (side, lvl)] = np.int32(L.shape[0])
    np.savez_compressed(OUT, **out)
    print('wrote', os.path.abspath(OUT), os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
