"""renderih_amd/graph_build.py (offline regeneration of misc/graph_*.pkl) against the packaged asset that
the reference's own `build_graph` produced (tests/golden/make_assets.py): vertex permutations exactly, Laplacians of all
five levels with the same sparsity and values to 1e-5 (the largest eigenvalue comes from ARPACK)."""
import numpy as np
import pytest

from renderih_amd import assets
from renderih_amd.graph_build import build_graph, tree_orders


def test_tree_orders_example():
    """The worked example of the reference (coarsening.py:265-266)."""
    got = tree_orders([np.array([4, 1, 1, 2, 2, 3, 0, 0, 3]), np.array([2, 1, 0, 1, 0])])
    assert got == [[3, 4, 0, 9, 1, 2, 5, 8, 6, 7, 10, 11], [2, 4, 1, 3, 0, 5], [0, 1, 2]]


@pytest.mark.parametrize('side', ['left', 'right'])
def test_build_graph_equals_reference_asset(side):
    ref = assets.load_graph_dict(side)
    got = build_graph(assets.hand_faces(side), 4)
    assert list(got['graph_perm']) == list(ref['graph_perm'])
    assert np.array_equal(np.asarray(got['graph_perm_reverse']), np.asarray(ref['graph_perm_reverse']))
    assert [L.shape[0] for L in got['coarsen_graphs_L']] == [1008, 504, 252, 126, 63]
    for lvl, (a, b) in enumerate(zip(got['coarsen_graphs_L'], ref['coarsen_graphs_L'])):
        a, b = a.tocsr(), b.tocsr()
        a.sort_indices()
        b.sort_indices()
        a.eliminate_zeros()
        assert a.nnz == b.nnz and np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices), lvl
        assert np.abs(a.data - b.data).max() < 1e-5, lvl
    assert float(got['graph_mask'].sum()) == 778
