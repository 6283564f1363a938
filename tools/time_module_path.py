"""Step time of the loops a USER runs (not the resident-batch loop bench.py times), headline shape, one GPU:
 (a) module path: ta3n_amd.models.VideoModel + the reference's loss assembly + torch.optim.SGD (what main.py does per step,
     features already on the device);
 (b) engine path as train_ddp.py drives it: set_batch (device copy of the features) + train_step per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from ta3n_amd.engine import TrainEngine
from ta3n_amd.loss import attentive_entropy
from ta3n_amd.models import VideoModel
from ta3n_amd.synthetic import synth_batch

Bs, Bt, T, D, C = 128, 74, 5, 2048, 12
xs, xt, ys, yt = synth_batch(C, T, D, Bs, Bt, seed=1)
xs, xt, ys = xs.cuda(), xt.cuda(), ys.cuda()
m = VideoModel(C, "video", "trn-m", "RGB", train_segments=T, val_segments=T, base_model="resnet101", fc_dim=512, verbose=False).cuda()
m.train()
opt = torch.optim.SGD(m.parameters(), 3e-2, momentum=0.9, weight_decay=1e-4, nesterov=True)
dl = torch.cat((torch.zeros(Bs), torch.ones(Bt))).long().cuda()


def module_step():
    o = m(xs, xt, [0.75, 0.75, 0.5], 0, True, False)
    loss = F.cross_entropy(o[1], ys)
    pd_all = []
    for l in range(3):
        ps, pt = o[3][l].reshape(-1, 2), o[8][l].reshape(-1, 2)
        lab = torch.cat((torch.zeros(ps.size(0)), torch.ones(pt.size(0)))).long().cuda()
        pd = torch.cat((ps, pt)); pd_all.append(pd)
        loss = loss + F.cross_entropy(pd, lab)
    loss = loss + 0.003 * attentive_entropy(torch.cat((o[1], o[6])), pd_all[1])
    opt.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(m.parameters(), 20)
    opt.step()


from ta3n_amd import accel
for name, on in (("module path (VideoModel + torch loss + torch's per-tensor clip_grad_norm_ / SGD.step)", False),
                 ("module path with ta3n_amd.accel (flat clip + flat SGD step; what compat/ and main.py install)", True)):
    accel.install() if on else accel.uninstall()
    fn = module_step
    best = 1e9
    for rep in range(3):              # host-bound loop: best of three runs of 100 steps
        for _ in range(10):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 100
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, 1e6 * (time.perf_counter() - t0) / n)
    print(f"{name}: {best:.0f} us/step")
accel.uninstall()
for bf16 in (False, True):
    eng = TrainEngine(Bs, Bt, T, D, 512, C, bf16=bf16, bf16_store=bf16)
    for v in eng.param_views().values():
        v.normal_(0, 0.01)
    eng.refresh_bf16(params=True)
    for _ in range(10):
        eng.set_batch(xs, xt, ys); eng.train_step([0.75, 0.75, 0.5], 0.003, 0.03)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 200
    for _ in range(n):
        eng.set_batch(xs, xt, ys); eng.train_step([0.75, 0.75, 0.5], 0.003, 0.03)
    torch.cuda.synchronize()
    print(f"engine path with a fresh device batch per step ({'bf16' if bf16 else 'fp32'}): {1e6 * (time.perf_counter() - t0) / n:.0f} us/step")
