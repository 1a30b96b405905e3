"""The pieces compose like one iteration of the reference's training loop (core/gcn_trainer.py:215-262):
folder sample -> collate -> GPU batch preparation -> network -> mesh loss -> backward.  Host logic only (ABI emulator)."""
import os
import pickle
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _write_sample(root, i, rs):
    from PIL import Image
    for d in ('img', 'ori_handdict', 'anno'):
        os.makedirs(os.path.join(root, 'train', d), exist_ok=True)
    Image.fromarray(rs.randint(0, 256, (256, 256, 3)).astype(np.uint8)).save(os.path.join(root, 'train', 'img', '%d.jpg' % i))
    hd = {}
    for side, off in (('left', -0.05), ('right', 0.05)):
        c = np.array([off, 0, 0.6])
        hd[side] = {'verts3d': (rs.randn(778, 3) * 0.04 + c).astype(np.float32), 'joints3d': (rs.randn(21, 3) * 0.04 + c).astype(np.float32),
                    'verts2d': (rs.rand(778, 2) * 256).astype(np.float32), 'joints2d': (rs.rand(21, 2) * 256).astype(np.float32),
                    'R': np.eye(3, dtype=np.float32), 'pose': np.zeros(45, np.float32), 'shape': np.zeros(10, np.float32)}
    np.save(os.path.join(root, 'train', 'ori_handdict', '%d.npy' % i), hd, allow_pickle=True)
    with open(os.path.join(root, 'train', 'anno', '%d.pkl' % i), 'wb') as f:
        pickle.dump({}, f)


def test_one_training_iteration_composes(tmp_path):
    import random
    from abi_emulator import emulated_abi
    from renderih_amd import assets
    from renderih_amd.dataset import InterHandFolder, collate_for_gpu
    from renderih_amd.input_pipeline import BatchPreparer
    from renderih_amd.loss import GraphLoss, FusedMeshLoss, calc_loss_GCN_fused
    from renderih_amd.manolayer import ManoLayer
    import test_half
    rs = np.random.RandomState(0)
    for i in range(2):
        _write_sample(str(tmp_path), i, rs)
    ds = InterHandFolder(str(tmp_path), 'train')
    imgs, p2, p3 = collate_for_gpu([ds[i] for i in range(len(ds))])
    random.seed(3)
    np.random.seed(3)
    with emulated_abi():
        dev = torch.device('cpu')
        batch = BatchPreparer(train=True)(imgs, p2, p3)
        assert batch.imgTensor.shape == (2, 3, 256, 256) and batch.v3d_l.shape == (2, 778, 3) and batch.root_rel.shape == (2, 3)
        # bone-length normalisation: mean |joint 9 - joint 0| of the two hands is BONE_LENGTH, roots are at the origin
        bl = (batch.j3d_l[:, 9] - batch.j3d_l[:, 0]).norm(dim=1) + (batch.j3d_r[:, 9] - batch.j3d_r[:, 0]).norm(dim=1)
        assert torch.allclose(bl / 2, torch.full((2,), 0.095), atol=1e-6)
        assert batch.j3d_l[:, 9].abs().max() < 1e-7 and batch.j3d_r[:, 9].abs().max() < 1e-7
        model = test_half.tiny_model().train()
        mano = {s: ManoLayer(assets.synthetic_mano_dict(s)) for s in ('left', 'right')}
        gl = {s: GraphLoss(mano[s].J_regressor, mano[s].get_faces(), level=4, device=dev) for s in ('left', 'right')}
        conv = model.decoder.converter
        fused = FusedMeshLoss(gl['left'], gl['right'], conv['left'], conv['right'])
        out = model(batch.imgTensor)
        loss, terms = calc_loss_GCN_fused(fused, 0, *out, batch.v2d_l.contiguous(), batch.v2d_r.contiguous(),
                                          batch.v3d_l.contiguous(), batch.v3d_r.contiguous(), batch.root_rel)
        loss.backward()
    assert torch.isfinite(loss) and set(terms) == {'vert2d_loss', 'vert3d_loss', 'joint_loss', 'norm_loss', 'edge_loss'}
    grads = [p.grad for p in model.parameters() if p.requires_grad and p.grad is not None]
    assert len(grads) > 500 and all(torch.isfinite(g).all() for g in grads)
