"""CPU-side checks: C ABI surface, state_dict schema, config shim, asset loading, drop-in import paths."""
import json
import os
import re
import subprocess
import numpy as np
import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def test_library_exports_every_declared_symbol():
    from renderih_amd import _lib

    def declared_in(name):
        hdr = open(os.path.join(ROOT, 'include', name)).read()
        hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
        return set(re.findall(r'\b(rih_[a-z0-9_]+)\s*\(', hdr))
    declared = declared_in('renderih_amd.h')
    assert declared, 'no declarations parsed'
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_lib.SIGNATURES.keys())
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.lib_path()], capture_output=True, text=True).stdout
    exported = set(re.findall(r' T (rih_[a-z0-9_]+)', out))
    assert declared <= exported, declared - exported
    assert lib.rih_version() >= 1 and lib.rih_arch() == b'gfx950'


def test_gemm_desc_layout_matches_c():
    """ctypes mirror of rih_gemm_desc must have the C compiler's layout (checked with a tiny C program)."""
    from renderih_amd._lib import GemmDesc
    import ctypes
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "renderih_amd.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",' \
          'sizeof(rih_gemm_desc),offsetof(rih_gemm_desc,sA1),offsetof(rih_gemm_desc,sCsplit),' \
          'offsetof(rih_gemm_desc,alpha),offsetof(rih_gemm_desc,tile),offsetof(rih_gemm_desc,engine),offsetof(rih_gemm_desc,ones_row),' \
          'offsetof(rih_gemm_desc,sBias1),offsetof(rih_gemm_desc,sR1));return 0;}'
    import tempfile
    d = tempfile.mkdtemp()
    open(os.path.join(d, 'a.c'), 'w').write(src)
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), os.path.join(d, 'a.c'), '-o', os.path.join(d, 'a')])
    got = [int(x) for x in subprocess.check_output([os.path.join(d, 'a')]).split()]
    assert got == [ctypes.sizeof(GemmDesc), GemmDesc.sA1.offset, GemmDesc.sCsplit.offset, GemmDesc.alpha.offset,
                   GemmDesc.tile.offset, GemmDesc.engine.offset, GemmDesc.ones_row.offset, GemmDesc.sBias1.offset,
                   GemmDesc.sR1.offset]


def test_ops_refuse_cpu_tensors():
    from renderih_amd import ops
    with pytest.raises(RuntimeError):
        ops.relu(torch.zeros(4))
    with pytest.raises(RuntimeError):
        ops.linear(torch.zeros(2, 4), torch.zeros(3, 4))


def test_hconv_desc_layout_matches_c():
    """ctypes mirror of rih_hconv_desc against the C compiler's layout."""
    from renderih_amd._lib import HConvDesc
    import ctypes
    import tempfile
    fields = ['x', 'bias', 'res', 'y', 'N', 'Wo', 'ldx', 'Kpad', 'relu', 'out_f32']
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "renderih_amd.h"\nint main(){printf("%zu' + ' %zu' * len(fields) + \
          '\\n",sizeof(rih_hconv_desc),' + ','.join('offsetof(rih_hconv_desc,%s)' % f for f in fields) + ');return 0;}'
    d = tempfile.mkdtemp()
    open(os.path.join(d, 'a.c'), 'w').write(src)
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), os.path.join(d, 'a.c'), '-o', os.path.join(d, 'a')])
    got = [int(x) for x in subprocess.check_output([os.path.join(d, 'a')]).split()]
    assert got == [ctypes.sizeof(HConvDesc)] + [getattr(HConvDesc, f).offset for f in fields]


def test_new_host_modules_refuse_cpu_tensors():
    """fp16 backbone, batch preparation, SDF, graphed inference: no CPU path either."""
    import torch.nn as nn
    from renderih_amd import half, input_pipeline, sdf, graph
    with pytest.raises(RuntimeError):
        half.PackedConv(nn.Conv2d(8, 8, 1))
    with pytest.raises(RuntimeError):
        half.maxpool3x3s2(torch.zeros(1, 4, 4, 8, dtype=torch.float16))
    with pytest.raises(RuntimeError):
        input_pipeline.BatchPreparer(train=False)(torch.zeros(1, 8, 8, 3, dtype=torch.uint8), torch.zeros(1, 1598, 2),
                                                  torch.zeros(1, 1598, 3))
    with pytest.raises(RuntimeError):
        sdf.sdf(torch.zeros(4, 3, dtype=torch.int32), torch.zeros(1, 4, 3))
    with pytest.raises(RuntimeError):
        graph.GraphedInference(lambda t: t, torch.zeros(2))


def test_state_dict_schema_equals_reference():
    from renderih_amd.model import build_model
    ref = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'state_keys.json')))
    sd = build_model(0.05).state_dict()
    assert list(sd.keys()) == list(ref.keys())
    assert {k: list(v.shape) for k, v in sd.items()} == ref


def test_state_dict_schema_equals_reference_hrnet32():
    """HRNet-W32 variant (ENCODER_TYPE hrnet32): same key set and shapes as the reference's module tree."""
    from renderih_amd.model import build_model
    ref = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'state_keys_hrnet32.json')))
    sd = build_model(0.05, 'hrnet32').state_dict()
    assert set(sd.keys()) == set(ref.keys()), sorted(set(sd.keys()) ^ set(ref.keys()))[:10]
    assert {k: list(v.shape) for k, v in sd.items()} == ref


def test_state_dict_schema_equals_reference_family_b():
    """common/myhand/lijun_model_graph.load_graph_model (mano_flag=True, main/config.py:80): same keys, order, shapes."""
    from renderih_amd.lijun import build_graph_model
    ref = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'state_keys_lijun.json')))
    sd = build_graph_model(0.05).state_dict()
    assert list(sd.keys()) == list(ref.keys())
    assert {k: list(v.shape) for k, v in sd.items()} == ref


def test_dropin_import_paths_and_config(tmp_path):
    import models.model as mm
    import models.manolayer as ml
    import common.myhand.lijun_model_graph as lg
    import common.myhand.encoder_lijun as le
    import common.myhand.decoder_lijun_graph as ld
    assert callable(lg.load_graph_model) and lg.HandNET_GCN.__module__ == 'renderih_amd.lijun'
    assert callable(le.load_encoder) and callable(ld.load_decoder) and hasattr(ld, 'ParamRegressor')
    import common.myhand.lijun_model_newgraph as ln
    import common.myhand.decoder_lijun_mano as lmn
    assert callable(ln.load_new_model) and lmn.decoder.__name__ == 'decoder_mano'
    from renderih_amd.config import load_cfg
    assert mm.Model is mm.HandNET_GCN and callable(mm.load_model)
    assert hasattr(ml, 'ManoLayer') and hasattr(ml, 'rodrigues_batch')
    y = tmp_path / 'c.yaml'
    y.write_text('MODEL:\n  ENCODER_TYPE: resnet50\nTRAIN:\n  dropout: 0.0\nEXTRA:\n  k: 1\n')
    cfg = load_cfg(str(y))
    assert cfg.TRAIN.dropout == 0.0 and cfg.MODEL.GCN_IN_DIM == [512, 256, 128] and cfg.EXTRA.k == 1
    m = mm.load_model(str(y))
    assert m.decoder.dropout_p == 0.0
    assert m.decoder.unsample_layer.weight.shape == (778, 252)
    assert tuple(m.decoder.get_upsample_weight().shape) == (778, 252)
    v = torch.arange(778 * 3, dtype=torch.float32).view(1, 778, 3)
    conv = m.decoder.converter['left']
    g = conv.vert_to_GCN(v)
    assert g.shape == (1, 1008, 3)
    assert torch.equal(conv.GCN_to_vert(g), v)           # the permutation round-trips real vertices


def test_graph_assets_match_reference_statistics():
    """Laplacians 1008/504/252/126/63 with nnz 5638/2994/1572/842/423."""
    from renderih_amd import assets
    for side in ('left', 'right'):
        g = assets.load_graph_dict(side)
        assert [L.shape[0] for L in g['coarsen_graphs_L']] == [1008, 504, 252, 126, 63]
        if side == 'left':
            assert [L.nnz for L in g['coarsen_graphs_L']] == [5638, 2994, 1572, 842, 423]
        assert len(g['graph_perm']) == 1008 and len(g['graph_perm_reverse']) == 1008
        assert g['mesh_faces'].shape == (1538, 3)


def test_mano_helpers_cpu_roundtrip(tmp_path):
    """Host-side helpers (not the HIP path): axis<->pca and Rmat2axis round trips."""
    from renderih_amd import assets
    from renderih_amd.manolayer import ManoLayer, rodrigues_batch
    layer = ManoLayer(assets.synthetic_mano_dict('right'))
    g = torch.Generator().manual_seed(1)
    axis = torch.randn(4, 45, generator=g) * 0.5
    assert torch.allclose(layer.pca2axis(layer.axis2pca(axis)), axis, atol=1e-4)
    R = layer.axis2Rmat(axis)
    assert torch.allclose(layer.Rmat2axis(R), axis, atol=1e-3)
    fr = layer.get_local_frame(torch.zeros(2, 10))
    assert fr.shape == (2, 15, 3, 3)
    assert torch.allclose(rodrigues_batch(torch.zeros(1, 3)), torch.eye(3).unsqueeze(0), atol=1e-6)


def test_checkpoint_conventions_round_trip(tmp_path):
    """bare / {'epoch','network'} / 'module.'-prefixed files all load (core/gcn_trainer.py:90-100, 303-311)."""
    import torch.nn as nn
    from renderih_amd.checkpoint import load_checkpoint, save_checkpoint
    src = nn.Sequential(nn.Linear(4, 3), nn.BatchNorm1d(3))
    with torch.no_grad():
        src[0].weight.fill_(0.25)
    p1, p2, p3 = (str(tmp_path / n) for n in ('a.pth', 'b.pth', 'c.pth'))
    save_checkpoint(src, p1, 7)
    torch.save(src.state_dict(), p2)
    torch.save({'epoch': 3, 'network': {'module.' + k: v for k, v in src.state_dict().items()}}, p3)
    for path, epoch in ((p1, 7), (p2, None), (p3, 3)):
        dst = nn.Sequential(nn.Linear(4, 3), nn.BatchNorm1d(3))
        assert load_checkpoint(dst, path) == epoch
        assert torch.equal(dst[0].weight, src[0].weight)
    wrapped = nn.Module()
    wrapped.module = src
    save_checkpoint(wrapped, p1, 1)
    assert set(torch.load(p1)['network']) == set(src.state_dict())


def test_one_launch_adam_refuses_cpu_tensors():
    """renderih_amd.optim.Adam drives a HIP kernel: CPU parameters raise (no fallback), like every op of the package."""
    import pytest
    from renderih_amd import optim
    p = torch.zeros(8, requires_grad=True)
    p.grad = torch.ones(8)
    with pytest.raises(RuntimeError, match='GPU tensors'):
        optim.Adam([p], lr=1e-3).step()


def test_fork_join_hop_is_an_identity_with_a_node_of_its_own():
    """streams._hop (the autograd node fork_join puts on the calling stream behind every side-stream result): values and
    gradients pass unchanged, nested results keep their structure, tensors outside the autograd graph are returned as they are."""
    import torch
    from renderih_amd import streams
    x = torch.randn(5, requires_grad=True)
    c = torch.randn(5)
    a, (b, k), n = streams._hop([x * 2.0, (x + 1.0, c), None])
    assert k is c and n is None
    assert type(a.grad_fn).__name__ == '_HopBackward' and type(b.grad_fn).__name__ == '_HopBackward'
    assert torch.equal(a, x.detach() * 2.0) and torch.equal(b, x.detach() + 1.0)
    (a.sum() + 3.0 * b.sum()).backward()
    assert torch.equal(x.grad, torch.full((5,), 5.0))
    # CPU tensors: fork_join is a plain loop and adds no node
    outs = streams.fork_join([lambda: x * 2.0, lambda: x * 3.0], reads=[[x], [x]])
    assert [type(o.grad_fn).__name__ for o in outs] == ['MulBackward0', 'MulBackward0']
    assert streams.SIDE == int(os.environ.get('RIH_SIDE_STREAMS', '3')) and streams.HOP
