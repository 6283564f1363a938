"""Where the module path's step time goes (VideoModel.forward + the reference's loss assembly + backward + clip_grad_norm_ +
torch.optim.SGD, headline shape, one GPU): each section timed with a device synchronisation after it (host + device time of the
section), plus whether the parameter gradients autograd leaves in p.grad are views into ONE flat buffer (zero-copy)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from ta3n_amd.loss import attentive_entropy
from ta3n_amd.models import VideoModel
from ta3n_amd.synthetic import synth_batch

Bs, Bt, T, D, C = 128, 74, 5, 2048, 12
xs, xt, ys, yt = synth_batch(C, T, D, Bs, Bt, seed=1)
xs, xt, ys = xs.cuda(), xt.cuda(), ys.cuda()
m = VideoModel(C, "video", "trn-m", "RGB", train_segments=T, val_segments=T, base_model="resnet101", fc_dim=512, verbose=False).cuda()
m.train()
opt = torch.optim.SGD(m.parameters(), 3e-2, momentum=0.9, weight_decay=1e-4, nesterov=True)
acc = {}


def tick(name, t0):
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()


def step():
    t = time.perf_counter()
    o = m(xs, xt, [0.75, 0.75, 0.5], 0, True, False)
    t = tick("forward", t)
    loss = F.cross_entropy(o[1], ys)
    pd_all = []
    for l in range(3):
        ps, pt = o[3][l].reshape(-1, 2), o[8][l].reshape(-1, 2)
        lab = torch.cat((torch.zeros(ps.size(0)), torch.ones(pt.size(0)))).long().cuda()
        pd = torch.cat((ps, pt)); pd_all.append(pd)
        loss = loss + F.cross_entropy(pd, lab)
    loss = loss + 0.003 * attentive_entropy(torch.cat((o[1], o[6])), pd_all[1])
    t = tick("loss assembly", t)
    opt.zero_grad()
    loss.backward()
    t = tick("zero_grad + backward", t)
    torch.nn.utils.clip_grad_norm_(m.parameters(), 20)
    t = tick("clip_grad_norm_", t)
    opt.step()
    t = tick("optimizer.step", t)


if "--accel" in sys.argv:      # ta3n_amd.accel: clip_grad_norm_ / SGD.step as passes over the flat buffers (what compat/ and main.py install)
    from ta3n_amd import accel
    accel.install()
for _ in range(10):
    step()
acc.clear()
n = 100
for _ in range(n):
    step()
for k, v in acc.items():
    print(f"  {k:24s} {1e6 * v / n:8.1f} us")
print(f"  {'sum':24s} {1e6 * sum(acc.values()) / n:8.1f} us (each section followed by a device synchronisation)")
base = m._flat
views = sum(1 for p in m.parameters() if p.grad is not None and p.grad._base is not None)
bases = {p.grad._base.data_ptr() for p in m.parameters() if p.grad is not None and p.grad._base is not None}
print(f"  gradients that are views: {views} of {sum(1 for p in m.parameters() if p.grad is not None)}; distinct base buffers: {len(bases)}")
print("  optimizer foreach:", opt.param_groups[0].get("foreach"), " params:", len(list(m.parameters())))
