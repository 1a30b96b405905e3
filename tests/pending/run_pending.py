"""Runs ONE group of hardware checks for a code path that has not been on a GPU yet (see tests/test_zz_gpu_pending.py, which
starts this file in its own interpreter).  Exit code 0 = every check of the group passed on the GPU.

    python tests/pending/run_pending.py half_kernels | half_backbone | fused_attention | presplit | input_pipeline | sdf | graphed_inference | folded_fp32 | cdev | half_kernels_regstage
"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch        # noqa: E402


def half_kernels(dev):
    import test_half
    test_half.kernels_vs_torch(dev)
    # larger than one tile in every direction, deep K
    test_half.conv_case(dev, 4, 32, 32, 256, 512, 3, 1, 1, 'conv-bn', True, True, False, seed=11)
    test_half.conv_case(dev, 8, 64, 64, 64, 64, 1, 1, 0, 'conv-bn', True, False, False, seed=12)
    test_half.conv_case(dev, 2, 16, 16, 1024, 2048, 1, 2, 0, 'conv-bn', False, False, False, seed=13)


def half_kernels_regstage(dev):
    os.environ['RIH_HCONV_GLDS'] = '0'          # read by the library at its first rih_hconv call
    half_kernels(dev)


def half_backbone(dev):
    import test_half
    worst = test_half.backbone_vs_fp32(dev, B=2)
    print('fp16 backbone vs fp32 backbone, max relative error per tensor:', worst)
    print('second family:', test_half.backbone_b_vs_fp32(dev, B=2))
    # the full ResNet50 model: finite outputs, close to the fp32 path
    from renderih_amd.model import build_model
    from renderih_amd import testing
    m = build_model(0.0).to(dev).eval()
    img = testing.seeded_image(2, 3).to(dev)
    with torch.no_grad():
        ref = testing.flatten_outputs(m(img))
        m.use_fp16_backbone()
        got = testing.flatten_outputs(m(img))
    for k in ('result.verts3d.left', 'result.verts3d.right'):
        assert bool(torch.isfinite(got[k]).all())
        e = testing.rel_err(got[k], ref[k])
        print(k, 'fp16-vs-fp32 relative error', e)
        assert e < 3e-2, (k, e)


def graphed_inference(dev):
    """GraphedInference (fp32 path): a replay equals the eager launch sequence bit for bit."""
    from renderih_amd.model import build_model
    from renderih_amd import testing
    from renderih_amd.graph import GraphedInference
    m = build_model(0.0).to(dev).eval()
    g = GraphedInference(m, testing.seeded_image(2, 3).to(dev))
    img2 = testing.seeded_image(2, 4).to(dev)
    with torch.no_grad():
        eager = testing.flatten_outputs(m(img2))
    replay = testing.flatten_outputs(g(img2))
    for k in ('result.verts3d.left', 'result.verts3d.right'):
        assert torch.equal(replay[k], eager[k]), k


def folded_fp32(dev):
    import test_half
    test_half.conv2d_packed_vs_torch(dev)
    print('folded fp32 trunk vs plain eval path:', test_half.folded_fp32_trunk_vs_plain(dev, B=2))


def cdev(dev):
    import test_metrics
    test_metrics.cdev_kernel_vs_fixture(dev)


def fused_attention(dev):
    import test_gpu_ops as G
    from renderih_amd import ops
    ops.FUSED_ATTN = True
    G.test_attention(2, 63, 63, 64, 4)
    G.test_attention(1, 150, 190, 128, 4)
    G.test_attention(1, 127, 127, 256, 4)
    G.test_attention(4, 316, 316, 128, 4)
    G.test_attention_dropout_matches_hash_mask()
    G.test_self_attention_packed(2, 40, 64, 4)
    G.test_cross_attention_packed(2, 63, 128, 4)
    G.test_cross_attention_stacked_and_rows_pair()


def presplit(dev):
    import test_gpu_ops as G
    from renderih_amd import ops
    for act in (False, True):
        ops.PRESPLIT, ops.PRESPLIT_ACT = True, act
        for case in G.CONV_CASES:
            G.test_conv2d(case)


def input_pipeline(dev):
    import test_input_pipeline
    test_input_pipeline.prepare_vs_fixtures(dev)


def sdf(dev):
    import test_sdf
    test_sdf.sdf_vs_oracle(dev, G=16)


if __name__ == '__main__':
    assert torch.cuda.is_available(), 'needs a GPU'
    {'half_kernels': half_kernels, 'half_backbone': half_backbone, 'fused_attention': fused_attention,
     'presplit': presplit, 'input_pipeline': input_pipeline, 'sdf': sdf, 'graphed_inference': graphed_inference, 'folded_fp32': folded_fp32, 'cdev': cdev, 'half_kernels_regstage': half_kernels_regstage}[sys.argv[1]](torch.device('cuda:0'))
    torch.cuda.synchronize()
    print('PENDING-OK', sys.argv[1])
