"""GPU tests of the opt-in / secondary code paths: fp16-storage inference backbone (csrc/rih_half.hip; LDS-DMA and
register-staged loaders), fused attention kernels (csrc/rih_attn.hip), batch
input preparation (csrc/rih_input.hip), SDF voxeliser (csrc/rih_sdf.hip), contact deviation, hipGraph-replayed inference and
the BatchNorm-folded fp32 trunk.  All of them passed their first hardware run in round 1's driver suite (they were wrapped as
xfail-on-failure then); they are ordinary tests now: a failure fails the suite."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return torch.device('cuda:0')


def _half_kernels(d):
    import test_half
    test_half.kernels_vs_torch(d)
    # larger than one tile in every direction, deep K
    test_half.conv_case(d, 4, 32, 32, 256, 512, 3, 1, 1, 'conv-bn', True, True, False, seed=11)
    test_half.conv_case(d, 8, 64, 64, 64, 64, 1, 1, 0, 'conv-bn', True, False, False, seed=12)
    test_half.conv_case(d, 2, 16, 16, 1024, 2048, 1, 2, 0, 'conv-bn', False, False, False, seed=13)


def test_fp16_conv_kernels_lds_dma():
    _half_kernels(dev())


def test_fp16_conv_kernels_register_staged():
    """RIH_HCONV_GLDS=0 is read once per process by the library, hence its own interpreter."""
    code = ('import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_paths as T; '
            'T._half_kernels(torch.device("cuda:0")); torch.cuda.synchronize(); print("REGSTAGE-OK")'
            % (ROOT, os.path.join(ROOT, 'tests')))
    p = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, RIH_HCONV_GLDS='0'))
    assert p.returncode == 0 and 'REGSTAGE-OK' in p.stdout, (p.stdout + p.stderr)[-3000:]


def test_fp16_backbone_close_to_fp32():
    """fp16-storage inference backbone (BASELINE configs[4]): kernel-level checks, then the WHOLE model with the fp16 backbone
    against the CPU ORACLE (the pinned restatement of the reference path, fp32 eval forward on the same seeded weights and
    images) -- not against this package's own fp32 path.  Inference mode with fp16 storage is not the 1e-4 parity path
    (DESIGN 3.9); measured 7e-4 of the mesh extent, bar 5e-3."""
    import test_half
    from oracle import net_oracle
    from renderih_amd import assets, testing
    from renderih_amd.model import build_model
    d = dev()
    test_half.backbone_vs_fp32(d, B=2)
    test_half.backbone_b_vs_fp32(d, B=2)
    m = build_model(0.0)
    sd = testing.deterministic_state(m.state_dict(), seed=1)
    m.load_state_dict(sd)
    m = m.to(d).eval()
    img = testing.seeded_image(2, 3)
    graph = net_oracle.graph_from_dicts(assets.load_graph_dict('left'), assets.load_graph_dict('right'))
    with torch.no_grad():
        want = testing.flatten_outputs(net_oracle.handnet_forward({k: v.clone() for k, v in sd.items()}, graph, img,
                                                                  training=False))
        m.use_fp16_backbone()
        got = testing.flatten_outputs(m(img.to(d)))
    worst = 0.0
    for k in ('result.verts3d.left', 'result.verts3d.right', 'hand0.verts3d.left', 'hand0.verts3d.right'):
        assert bool(torch.isfinite(got[k]).all())
        e = testing.rel_err(got[k], want[k])
        worst = max(worst, e)
        assert e < 5e-3, (k, e)
    print('fp16-storage backbone, whole model vs CPU oracle: worst rel. err %.2e' % worst)


def test_graphed_inference_replay_is_bit_identical():
    from renderih_amd import testing
    from renderih_amd.graph import GraphedInference
    from renderih_amd.model import build_model
    d = dev()
    m = build_model(0.0).to(d).eval()
    g = GraphedInference(m, testing.seeded_image(2, 3).to(d))
    img2 = testing.seeded_image(2, 4).to(d)
    with torch.no_grad():
        eager = testing.flatten_outputs(m(img2))
    replay = testing.flatten_outputs(g(img2))
    for k in ('result.verts3d.left', 'result.verts3d.right'):
        assert torch.equal(replay[k], eager[k]), k


def test_folded_batchnorm_fp32_trunk():
    import test_half
    test_half.conv2d_packed_vs_torch(dev())
    test_half.folded_fp32_trunk_vs_plain(dev(), B=2)


def test_contact_deviation_kernel():
    import test_metrics
    test_metrics.cdev_kernel_vs_fixture(dev())


def test_fused_attention_kernels(monkeypatch):
    import test_gpu_ops as G
    from renderih_amd import ops
    monkeypatch.setattr(ops, 'FUSED_ATTN', True)
    G.test_attention(2, 63, 63, 64, 4)
    G.test_attention(1, 150, 190, 128, 4)
    G.test_attention(1, 127, 127, 256, 4)
    G.test_attention(4, 316, 316, 128, 4)
    G.test_attention_dropout_matches_hash_mask()
    G.test_self_attention_packed(2, 40, 64, 4)
    G.test_cross_attention_packed(2, 63, 128, 4)
    G.test_cross_attention_stacked_and_rows_pair()


def test_batch_input_preparation_matches_reference_fixtures():
    import test_input_pipeline
    test_input_pipeline.prepare_vs_fixtures(dev())


def test_sdf_voxeliser():
    import test_sdf
    test_sdf.sdf_vs_oracle(dev(), G=16)
    test_sdf.sdf_vs_reference_golden(dev())         # outputs of the reference's own kernel (oracle/_ref, tests/golden/sdf_ref.npz)
