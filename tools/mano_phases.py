#!/usr/bin/env python
"""Phase timing of the fused MANO forward (shader-clock stamps of the first chunk of the workgroups (tile, group 0))."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderih_amd import assets, _lib
from renderih_amd.manolayer import ManoLayer, rodrigues_batch
dev = torch.device('cuda:0')
lib = _lib.load()
layer = ManoLayer(assets.synthetic_mano_dict('right')).to(dev)
B = 4096
g = torch.Generator().manual_seed(0)
root = rodrigues_batch(torch.randn(B, 3, generator=g)).to(dev)
pose, shape = (0.5 * torch.randn(B, 45, generator=g)).to(dev), torch.randn(B, 10, generator=g).to(dev)
buf = torch.zeros(13 * 16 + 32, dtype=torch.int64, device=dev)
with torch.no_grad():
    layer(root, pose, shape)
    torch.cuda.synchronize()
    lib.rih_mano_debug_stamps(buf.data_ptr())
    layer(root, pose, shape)
    torch.cuda.synchronize()
    lib.rih_mano_debug_stamps(0)
# backward, per-hand kernel (workgroup 0)
pg, sg, rg = pose.clone().requires_grad_(), shape.clone().requires_grad_(), root.clone().requires_grad_()
v, j = layer(rg, pg, sg)
torch.autograd.backward([v], [torch.randn_like(v)])
torch.cuda.synchronize()
lib.rih_mano_debug_stamps(buf.data_ptr())
v, j = layer(rg, pg, sg)
torch.autograd.backward([v], [torch.randn_like(v)])
torch.cuda.synchronize()
lib.rih_mano_debug_stamps(0)
raw = buf.cpu()
s = raw[:208].view(13, 16)
names = ['1a pca/beta', '1b rot/jt', '1c chain', '1d special', '1e post/joints', '2 mfma', '3 skin']
for tile in (0, 1, 12):
    d = [int(s[tile, i + 1] - s[tile, i]) for i in range(7)]
    print('tile %2d: ' % tile + ' | '.join('%s %d' % (n, x) for n, x in zip(names, d)) + ' | chunk total %d cycles' % (int(s[tile, 7] - s[tile, 0])))

d1 = [int(s[1, i]) for i in range(4)]
print('phase 1d of workgroup 0: blend of the special coordinates %d | barrier %d | their skinning %d | barrier %d | tips copy + barrier %d' % (d1[0] - int(s[0, 3]), d1[1] - d1[0], d1[2] - d1[1], d1[3] - d1[2], int(s[0, 4]) - d1[3]))
tn = ['issue + weights + blend 1st half', 'land 2nd half + barrier', 'issue next + blend 2nd half', 'park + barrier', 'skin', 'land next 1st half + barrier']
d = [int(s[0, 9 + i] - s[0, 8 + i]) for i in range(6)]
print('hand-major forward, third tile of workgroup 0: ' + ' | '.join('%s %d' % (n, x) for n, x in zip(tn, d)) + ' | tile total %d cycles' % (int(s[0, 14] - s[0, 8])))

raw = buf.cpu()
b = [int(raw[208 + i]) for i in range(8)]
bn = ['0 loads/axis', '1 dv_eff+sums', '3 joint side', '4 skinning', '5 dG=W^T M', '6-7 chain', 'hand-off']      # (until round 4 an extra label shifted the last four)
print('backward, per-hand kernel (workgroup 0): ' + ' | '.join('%s %d' % (n, b[i + 1] - b[i]) for i, n in enumerate(bn[:7])) + ' | total %d cycles' % (b[7] - b[0]))
