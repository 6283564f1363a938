"""Where does the distance of the bf16 arithmetic's GRADIENTS from the fp32 reference come from?  (the GPU tests
show relation-discriminator hidden-layer weight gradients 6.5-15 % (rel. L2) off the fp32 reference at the configs[3] / configs[4]
shapes - ReLU flips, or the per-segment rounding of the tuple activations in oracle/ta3n_oracle.py:_SegSumMatmulBf16?)

CPU only, the oracle's two modes: one step's gradients at the given shape under
  fp32                       the reference arithmetic (the yardstick)
  bf16                       the product's contract (BF16_POLICY, per-segment rounding)
  bf16, sum then round       the same with the relation discriminator's input rounded AFTER the tuple sum (SEGSUM_ROUND_AFTER_SUM)
  ablations                  only the forward products / only the input-gradient products / only the weight-gradient products rounded
  fp32, bf16's ReLU masks    fp32 products everywhere, but every ReLU uses the on/off pattern of the bf16 run: what mask flips alone cost
and the relative L2 distance of every gradient tensor from the fp32 one.  usage: python tools/bf16_relation_disc_deviation.py [config4|config5|headline]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from oracle import ta3n_oracle as orc
from ta3n_amd.synthetic import synth_batch, synth_state

SHAPES = {"config4": dict(Bs=512, Bt=512, T=9, D=2048, F=512, C=30), "config5": dict(Bs=128, Bt=128, T=12, D=1024, F=512, C=12),
          "headline": dict(Bs=128, Bt=74, T=5, D=2048, F=512, C=12)}
name = sys.argv[1] if len(sys.argv) > 1 else "config5"
sh = SHAPES[name]
torch.set_num_threads(min(8, os.cpu_count() or 1))
BETA, GAMMA = [0.75, 0.75, 0.5], 0.003


def grads(arith, policy=None, after_sum=False, relu_masks=None, record_masks=None):
    cfg = orc.Config(num_class=sh["C"], num_segments=sh["T"], feature_dim=sh["D"], fc_dim=sh["F"], dropout_i=0.0, dropout_v=0.0, arithmetic=arith)
    params = synth_state(orc.param_shapes(cfg), seed=11, scale="trained")
    xs, xt, ys, _ = synth_batch(sh["C"], sh["T"], sh["D"], sh["Bs"], sh["Bt"], seed=21)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    keep_policy, keep_flag, keep_relu = dict(orc.BF16_POLICY), orc.SEGSUM_ROUND_AFTER_SUM, F.relu
    counter = [0]

    def relu(x, inplace=False):      # record / impose the on-off pattern of every ReLU, in call order
        i = counter[0]; counter[0] += 1
        if record_masks is not None:
            record_masks.append((x.detach() > 0))
        if relu_masks is not None:
            return x * relu_masks[i].to(x.dtype)
        return keep_relu(x)
    try:
        if policy is not None:
            orc.BF16_POLICY.update(policy)
        orc.SEGSUM_ROUND_AFTER_SUM = after_sum
        if relu_masks is not None or record_masks is not None:
            orc.F.relu = relu
        src = orc.forward_domain(p, xs, BETA, cfg, domain="S")
        tgt = orc.forward_domain(p, xt, BETA, cfg, domain="T")
        loss, _ = orc.total_loss(src, tgt, ys, GAMMA, cfg)
        names = [k for k in p if orc.is_live(k)]
        g = torch.autograd.grad(loss, [p[k] for k in names], allow_unused=True)
    finally:
        orc.BF16_POLICY.clear(); orc.BF16_POLICY.update(keep_policy)
        orc.SEGSUM_ROUND_AFTER_SUM = keep_flag
        orc.F.relu = keep_relu
    return {k: v.double() for k, v in zip(names, g) if v is not None}


def dist(a, ref):
    return {k: float(((a[k] - ref[k]).pow(2).sum().sqrt() / (ref[k].pow(2).sum().sqrt() + 1e-300))) for k in ref}


def only(which):      # a policy with only one of (fwd, dgrad, wgrad) rounding, bias sums exact
    return {k: tuple(v[i] and i == which for i in range(3)) + (False,) for k, v in orc.BF16_POLICY.items()}


t0 = time.time()
ref = grads("fp32")
masks16 = []
runs = {"bf16 (product contract: per-segment rounding)": grads("bf16", record_masks=masks16)}
runs["bf16, tuple sum rounded once"] = grads("bf16", after_sum=True)
runs["bf16, only forward products rounded"] = grads("bf16", policy=only(0))
runs["bf16, only input-gradient products rounded"] = grads("bf16", policy=only(1))
runs["bf16, only weight-gradient products rounded"] = grads("bf16", policy=only(2))
runs["fp32 products, ReLU on/off pattern of the bf16 run"] = grads("fp32", relu_masks=masks16)
masks32 = []
grads("fp32", record_masks=masks32)
flips = [(int((a != b).sum()), a.numel()) for a, b in zip(masks16, masks32)]
print(f"# shape {name}: {sh}; trained-scale weights (seed 11), batch seed 21, beta {BETA}; {time.time() - t0:.0f} s")
print(f"# ReLU units on the other side in the bf16 run: {sum(f for f, _ in flips)} of {sum(n for _, n in flips)} "
      f"({100.0 * sum(f for f, _ in flips) / sum(n for _, n in flips):.3f} %)")
rel = sorted(k for k in ref if k.startswith("relation_domain_classifier_all") and k.endswith(".0.weight"))
print("# rel. L2 distance from the fp32 gradient: relation-discriminator hidden-layer weights (one column per relation), then the median over ALL tensors")
for label, g in runs.items():
    d = dist(g, ref)
    print(f"{label:58s} " + " ".join(f"{d[k]:.2e}" for k in rel) + f"   median {np.median(list(d.values())):.2e}  worst {max(d.values()):.2e} ({max(d, key=d.get)})")
