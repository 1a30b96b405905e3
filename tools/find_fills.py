"""Which Python lines issue the tiny fill/zero kernels of a training step (torch.profiler with stacks, one eager step)."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from renderih_amd import assets
from renderih_amd.model import build_model
from renderih_amd.loss import GraphLoss, FusedMeshLoss, calc_loss_GCN_fused
from renderih_amd.manolayer import ManoLayer
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = build_model(0.05).to(dev).train()
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=3e-4, weight_decay=1e-2, fused=True)
mano = {s: ManoLayer(assets.synthetic_mano_dict(s)) for s in ('left', 'right')}
gl = {s: GraphLoss(mano[s].J_regressor, mano[s].get_faces(), level=4, device=dev) for s in ('left', 'right')}
conv = model.decoder.converter
fused = FusedMeshLoss(gl['left'], gl['right'], conv['left'], conv['right'])
model.decoder.unsample_layer.weight.requires_grad_(False)
img, lab = synth_batch(8, dev, 0)
def step():
    opt.zero_grad(set_to_none=True)
    out = model(img)
    loss, _ = calc_loss_GCN_fused(fused, 0, *out, lab['v2d_l'], lab['v2d_r'], lab['v3d_l'], lab['v3d_r'], lab['root_rel'])
    loss.backward()
    opt.step()
step(); step(); torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
def chain(e, n=4):
    out = []
    p = e.cpu_parent
    while p is not None and len(out) < n:
        out.append(p.name[:40])
        p = p.cpu_parent
    return ' < '.join(out)


for names in (('aten::fill_',), ('aten::copy_',), ('aten::add', 'aten::add_', 'aten::mul', 'aten::cat', 'aten::index')):
    cnt = collections.Counter()
    for e in prof.events():
        if e.name in names:
            shp = str(e.input_shapes)[:34] if e.input_shapes else ''
            cnt[(e.name, chain(e), shp)] += 1
    print('==', names, 'total', sum(cnt.values()))
    for k, v in cnt.most_common(22):
        print(v, k)
