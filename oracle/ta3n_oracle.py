"""CPU oracle for the TA3N temporal-adversarial train step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``ta3n_amd/`` may import this module;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` do, and only as the checker / the timed CPU baseline.

It is a compact functional restatement of the reference's hot path (the
reference is a PyTorch program whose arithmetic is ATen CPU kernels, so the
restatement uses the same ATen CPU ops through ``torch`` on CPU tensors; the
backward pass is ``torch.autograd`` exactly as the reference's
``loss.backward()``).  Each function cites the reference lines it follows
(paths relative to cmhungsteve/TA3N).

Parity pin: ``tests/golden/*.npz`` were produced by running the reference's own
``models.VideoModel`` / ``main.train`` in the build container
(``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py`` checks this
module against every fixture.  The reference ships no tests or golden vectors
of its own.

Two arithmetics (``Config.arithmetic``):
  * ``"fp32"`` - the reference's own arithmetic (ATen CPU fp32); pinned by the goldens.
  * ``"bf16"`` - BASELINE configs[1]: the SAME graph, but every contraction the HIP path runs on the bf16 matrix cores
    rounds BOTH operands to bf16 (round to nearest even) and multiplies / accumulates in fp32 (``_MatmulBf16``).
    Which contractions those are is the arithmetic contract of the bf16 configuration, written down per reference layer
    in ``BF16_POLICY`` (and in DESIGN.md section 5); parameters, biases, softmax / entropy / losses, GradReverse scales and the
    optimiser stay fp32.  This mode is independent of the product's launch descriptors: it is derived from the reference's
    layer structure only, and it is what tests/test_gpu_bf16.py compares the HIP bf16 path with.

Scope: frame_aggregation='trn-m', baseline_type='video', share_params='Y',
use_bn='none', use_attn='TransAttn', add_fc=1, adv_DA='RevGrad' - the
UCF->HMDB_full TA3N configuration of script_train_val.sh.
"""
from __future__ import annotations

import itertools
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SUBSAMPLE_NUM = 3  # TRNmodule.py:32
NUM_BOTTLENECK = 256  # models.py:223

ARCH_FEATURE_DIM = {  # models.py:125-126 reads torchvision <arch>.fc.in_features
    "resnet18": 512, "resnet34": 512, "resnet50": 2048, "resnet101": 2048,
    "resnet152": 2048,
}


# ----------------------------------------------------------------------------
# integer layer (must be bit-exact)
# ----------------------------------------------------------------------------
def relation_scales(num_frames: int) -> List[int]:
    """TRNmodule.py:34  scales = [T, T-1, ..., 2]."""
    return [i for i in range(num_frames, 1, -1)]


def selected_relations(num_frames: int) -> List[List[Tuple[int, ...]]]:
    """Frame tuples actually used by RelationModuleMultiScale.forward.

    TRNmodule.py:36-41 (all C(T,s) sorted tuples per scale, via
    itertools.combinations, TRNmodule.py:84-86), :60 (scale 0 uses tuple 0),
    :68-71 (later scales use idx = int(ceil(i * n_total / n_select)) for
    i < min(3, n_total)).
    """
    out = []
    for sid, scale in enumerate(relation_scales(num_frames)):
        rel = list(itertools.combinations(range(num_frames), scale))
        if sid == 0:
            out.append([rel[0]])
            continue
        n_total = len(rel)
        n_sel = min(SUBSAMPLE_NUM, n_total)
        idx = [int(math.ceil(i * n_total / n_sel)) for i in range(n_sel)]
        out.append([rel[i] for i in idx])
    return out


def segment_indices_test_mode(num_frames: int, num_segments: int, new_length: int = 1) -> np.ndarray:
    """dataset.py:103-116 `_get_test_indices` (the only sampler used: every
    TSNDataSet in main.py:171-197 is built with test_mode=True).  1-based."""
    num_min = num_segments + new_length - 1
    num_select = num_frames - new_length + 1
    if num_frames >= num_min:
        tick = float(num_select) / float(num_segments)
        offsets = np.array([int(tick / 2.0 + tick * float(x)) for x in range(num_segments)])
    else:
        id_select = np.array([x for x in range(num_select)])
        id_expand = np.ones(num_segments - num_select, dtype=int) * id_select[id_select[0] - 1]
        offsets = np.append(id_select, id_expand)
    return offsets + 1


# ----------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------
@dataclass
class Config:
    num_class: int = 12
    num_segments: int = 5
    feature_dim: int = 2048          # models.py:125-126
    fc_dim: int = 512                # script_train_val.sh (fc_dim=512)
    dropout_i: float = 0.5
    dropout_v: float = 0.5
    place_adv: Tuple[str, str, str] = ("Y", "Y", "Y")   # opts.py:67
    add_loss_DA: str = "attentive_entropy"            # opts.py:54
    use_attn: str = "TransAttn"
    frame_aggregation: str = "trn-m"                  # 'avgpool' = TemPooling (models.py:246, 421-433; BASELINE configs[0])
    arithmetic: str = "fp32"                          # 'bf16': BASELINE configs[1] (bf16 MFMA operands, fp32 accumulation)
    compute_dead_branches: bool = False              # also evaluate what the reference computes and never uses (the frame classifier,
                                                      # models.py:617-618): for timing the CPU path fairly, no effect on any output
    dis_DA: str = "none"                              # 'DAN' | 'JAN' (opts.py:44; main.py:452-505): MMD losses on the feature list
    place_dis: Tuple[str, str, str] = ("N", "Y", "N")  # opts.py:65 (script_train_val.sh:148 passes N Y N); indexes feat = [Y, V, F1]
    use_bn: str = "none"                              # 'AdaBN' | 'AutoDIAL' (opts.py; models.py:195-198, 490-543, 569-570): domain-specific
                                                      # BatchNorm1d between the shared FC and its ReLU (alpha = 1: no batch mixing)
    ens_DA: str = "none"                              # 'MCD' (opts.py:49): second video classifier + a gradient-reversed second forward
    add_fc: int = 1
    bf16_twins: bool = True                           # bf16 only: operands are read from bf16 copies (TA3N_FLAG_BF16_STORE), so a bias
                                                      # gradient made by a weight-gradient launch sums ROUNDED values (BF16_POLICY bias16)

    @property
    def feat_dim(self) -> int:       # models.py:129
        return min(self.fc_dim, self.feature_dim)


def param_shapes(cfg: Config) -> Dict[str, Tuple[int, ...]]:
    """state_dict keys/shapes of VideoModel for the trn-m configuration
    (models.py:141-294, TRNmodule.py:44-54).  BatchNorm buffers omitted."""
    Fd, D, C, T, NB = cfg.feat_dim, cfg.feature_dim, cfg.num_class, cfg.num_segments, NUM_BOTTLENECK
    s: Dict[str, Tuple[int, ...]] = {}

    def lin(name, o, i):
        s[name + ".weight"] = (o, i)
        s[name + ".bias"] = (o,)

    if cfg.frame_aggregation == "avgpool":          # feat_aggregated_dim = feat_shared_dim (models.py:246-247), no TRN, no relation discriminators
        lin("fc_feature_shared_source", Fd, D)
        if cfg.use_bn != "none":
            for bn in ("bn_shared_S", "bn_shared_T"):
                s[bn + ".weight"] = (Fd,)
                s[bn + ".bias"] = (Fd,)
        lin("fc_feature_source", Fd, Fd)
        lin("fc_feature_domain", Fd, Fd)
        lin("fc_classifier_source", C, Fd)
        lin("fc_classifier_domain", 2, Fd)
        lin("fc_feature_video_source", Fd, Fd)
        lin("fc_feature_video_source_2", Fd, Fd)
        lin("fc_feature_domain_video", Fd, Fd)
        lin("fc_classifier_video_source", C, Fd)
        if cfg.ens_DA == "MCD":
            lin("fc_classifier_video_source_2", C, Fd)
        lin("fc_classifier_domain_video", 2, Fd)
        return s
    lin("fc_feature_shared_source", Fd, D)          # models.py:141
    if cfg.use_bn != "none":                        # :195-196 (the other BatchNorm layers the option creates are never used with trn-m)
        for bn in ("bn_shared_S", "bn_shared_T"):
            s[bn + ".weight"] = (Fd,)
            s[bn + ".bias"] = (Fd,)
    lin("fc_feature_source", Fd, Fd)                # :156  (never used in fwd)
    lin("fc_feature_domain", Fd, Fd)                # :161
    lin("fc_classifier_source", C, Fd)              # :166  (dead for baseline 'video')
    lin("fc_classifier_domain", 2, Fd)              # :170
    for i, sc in enumerate(relation_scales(T)):     # TRNmodule.py:44-54
        lin(f"TRN.fc_fusion_scales.{i}.1", NB, sc * Fd)
    for bn in ("bn_trn_S", "bn_trn_T"):             # models.py:225-226 (unused)
        s[bn + ".weight"] = (NB,)
        s[bn + ".bias"] = (NB,)
    lin("fc_feature_video_source", NB, NB)          # :258 (unused)
    lin("fc_feature_video_source_2", NB, NB)        # :262 (unused)
    lin("fc_feature_domain_video", NB, NB)          # :267
    lin("fc_classifier_video_source", C, NB)        # :272
    if cfg.ens_DA == "MCD":
        lin("fc_classifier_video_source_2", C, NB)  # :276-279
    lin("fc_classifier_domain_video", 2, NB)        # :281
    for i in range(T - 1):                          # :286-294
        lin(f"relation_domain_classifier_all.{i}.0", NB, NB)
        lin(f"relation_domain_classifier_all.{i}.2", 2, NB)
    return s


# parameters that never receive a gradient in this configuration
DEAD_PREFIXES = ("fc_feature_source.", "fc_classifier_source.", "bn_trn_S.", "bn_trn_T.",
                 "fc_feature_video_source.", "fc_feature_video_source_2.")


def is_live(name: str) -> bool:
    return not name.startswith(DEAD_PREFIXES)


# ----------------------------------------------------------------------------
# model forward
# ----------------------------------------------------------------------------
class _GradReverse(torch.autograd.Function):
    """models.py:20-29."""

    @staticmethod
    def forward(ctx, x, beta):
        ctx.beta = beta
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.neg() * ctx.beta, None


def rne_bf16(t: torch.Tensor) -> torch.Tensor:
    """fp32 -> nearest-even bf16 -> fp32 (what v_cvt_pk_bf16_f32 / a bf16 store does)."""
    return t.to(torch.bfloat16).to(torch.float32)


def _mm16(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """bf16(a) @ bf16(b) with "fp32 accumulation": the products of bf16 values are exact in fp32; their sum is formed in
    fp64 and rounded to fp32 once, i.e. the correctly rounded value every fp32 summation order approximates.  (Summing
    in fp32 here would add this machine's BLAS summation order to the comparison: ~1e-6 relative noise that flips the
    bf16 rounding of downstream operands and a few ReLU units - measured 5e-4 .. 7e-3 relative L2 on gradient tensors
    depending on the host CPU, against 4e-6 .. 4e-4 for the HIP kernels versus an fp64-accumulating model.)"""
    return (rne_bf16(a).double() @ rne_bf16(b).double()).float()


class _MatmulBf16(torch.autograd.Function):
    """x W^T of an nn.Linear (and its two backward products) with the operands of the flagged contractions rounded
    to bf16; products and sums are fp32 (a bf16 x bf16 product is exact in fp32; the summation order differs from the
    matrix core's, which is fp32 round-off).  fwd / dgrad / wgrad: which of  x W^T,  g W,  g^T x  round their operands.
    The bias is added (and its gradient summed) outside, in fp32; bias16: the bias gradient is the column sum of
    the ROUNDED g (a weight-gradient launch that reads bf16 copies only has those)."""

    @staticmethod
    def forward(ctx, x, w, fwd, dgrad, wgrad):
        ctx.save_for_backward(x, w)
        ctx.flags = (dgrad, wgrad)
        return _mm16(x, w.t()) if fwd else x @ w.t()

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        dgrad, wgrad = ctx.flags
        gx = _mm16(g, w) if dgrad else g @ w
        gw = _mm16(g.t(), x) if wgrad else g.t() @ x
        return gx, gw, None, None, None


class _BiasBf16Sum(torch.autograd.Function):
    """y = h + b whose bias gradient is the column sum of bf16-rounded upstream gradients."""

    @staticmethod
    def forward(ctx, h, b):
        return h + b

    @staticmethod
    def backward(ctx, g):
        return g, rne_bf16(g).sum(0)


# Experiment switch (tools/bf16_relation_disc_deviation.py): True = the relation discriminator's hidden layer
# rounds the SUM of its scale's tuple activations once, the way a literal bf16 autocast of models.py:475-479 would, instead of each
# tuple activation on its own.  Never set by the tests: the product reads the tuple activations as separate K segments.
SEGSUM_ROUND_AFTER_SUM = False


class _SegSumMatmulBf16(torch.autograd.Function):
    """(sum_t z_t) W^T computed as sum_t bf16(z_t) bf16(W)^T: the relation discriminator's hidden layer reads the
    tuple activations of its scale as separate K segments (each rounded on its own), never their sum.  Backward:
    every z_t receives bf16(g) bf16(W); dW = bf16(g)^T bf16(sum_t z_t) (the weight-gradient launch reads R_j).
    THIS FUNCTION RESTATES PRODUCT STRUCTURE (see the note above BF16_POLICY)."""

    @staticmethod
    def forward(ctx, w, *zs):
        ctx.save_for_backward(w, *zs)
        w16 = rne_bf16(w).double().t()
        if SEGSUM_ROUND_AFTER_SUM:
            r = zs[0]
            for z in zs[1:]:
                r = r + z
            return (rne_bf16(r).double() @ w16).float()
        out = None
        for z in zs:
            y = rne_bf16(z).double() @ w16
            out = y if out is None else out + y
        return out.float()

    @staticmethod
    def backward(ctx, g):
        w, *zs = ctx.saved_tensors
        gz = _mm16(g, w)
        r = zs[0]
        for z in zs[1:]:
            r = r + z
        return (_mm16(g.t(), r),) + tuple(gz for _ in zs)


# What in this bf16 mode is INDEPENDENT of the product and what RESTATES it (VERDICT r03 weak #2a).  The reference has no bf16 mode
# and ships no fixtures for one, so this mode cannot be pinned the way the fp32 mode is (tests/golden/*.npz from the reference itself):
# it is a model of "the reference's layers with bf16 matrix-core operands", and three of its choices are the product's, not the
# reference's:
#   (1) WHICH contractions round their operands (BF16_POLICY below): the nn.Linear layers that run on the MFMA tile kernel do, the
#       2-wide / C-wide output layers and the video discriminator's 256x256 layer inside the heads kernel do not - a different
#       kernel partition would give a different, equally legitimate table;
#   (2) _SegSumMatmulBf16: the relation discriminator's hidden layer rounds each tuple activation z_t on its own (K segments of one
#       tile) instead of their sum R_j - the plan's segment structure;
#   (3) bias16 / _BiasBf16Sum: bias gradients as column sums of ROUNDED upstream gradients - what a weight-gradient launch that reads
#       bf16 twins has.
# Independent of the product: the rounding itself (RNE on the fp32 bit pattern), exact products, fp64 accumulation (no summation
# order of any kernel), autograd through the reference's own formulas, everything outside the contractions in fp32.
# Consequently tests/test_gpu_bf16.py (HIP bf16 vs this mode) checks that the kernels IMPLEMENT this contract, not that the contract
# is the reference's; the distance of the contract from the reference is measured separately against the fp32 mode
# (tests/test_gpu_gradients.py::test_bf16_distance_from_the_fp32_reference_*, tests/test_gpu_training_equivalence.py), and
# tools/bf16_relation_disc_deviation.py attributes it to the three choices above.
#
# The arithmetic contract of the bf16 configuration, per reference layer: which of (forward x W^T, input gradient g W,
# weight gradient g^T x) run on the bf16 matrix cores, and whether the bias gradient sums rounded values.  Everything
# else - the 2-wide / C-wide output layers and the 256x256 video-discriminator layer inside the fused heads kernel,
# all softmax / entropy / loss math, the optimiser - is fp32.
BF16_POLICY = {
    # layer prefix:                        (fwd,   dgrad, wgrad, bias16)
    "fc_feature_shared_source":            (True,  False, True,  True),    # models.py:565-566 (no input gradient: features are data)
    "fc_feature_domain":                   (True,  True,  True,  True),    # :458-459
    "fc_classifier_domain":                (False, False, False, False),   # :460   heads kernel, fp32
    "TRN.fc_fusion_scales":                (True,  True,  True,  True),    # TRNmodule.py:60-79
    "relation_domain_classifier_all.*.0":  (True,  True,  True,  False),   # models.py:475-479 (forward: _SegSumMatmulBf16)
    "relation_domain_classifier_all.*.2":  (False, False, True,  False),   # :479   2-wide output layer: fp32 except dW
    "fc_feature_domain_video":             (False, False, True,  False),   # :466-467 heads kernel fp32; dW on the matrix cores
    "fc_classifier_video_source":          (False, False, True,  False),   # :686
    "fc_classifier_domain_video":          (False, False, True,  False),   # :468
}


def _policy(name: str):
    """BF16_POLICY entry of a layer name ('*' stands for the scale / relation index)."""
    import re
    for k, v in BF16_POLICY.items():
        if re.fullmatch(re.escape(k).replace("\\*", "[0-9]+") + "(\\.[0-9]+\\.1)?", name):
            return v
    raise KeyError(f"no bf16 policy for layer {name!r}")


def _linear(p, name, x, cfg: Optional["Config"] = None):
    if cfg is None or cfg.arithmetic == "fp32":
        return F.linear(x, p[name + ".weight"], p[name + ".bias"])
    fwd, dgrad, wgrad, bias16 = _policy(name)
    h = _MatmulBf16.apply(x, p[name + ".weight"], fwd, dgrad, wgrad)
    return _BiasBf16Sum.apply(h, p[name + ".bias"]) if (bias16 and cfg.bf16_twins) else h + p[name + ".bias"]


def _relu_m(x, mask=None):
    """ReLU, or - mask-synchronised comparisons (tests/test_gpu_masked_gradients.py) - the on/off pattern of ANOTHER computation of the
    same layer imposed on this one: x * mask, forward and backward.  With the pattern taken from the implementation under test a hidden
    unit within round-off of zero can no longer land on different sides in the two computations, so what is left of a gradient
    difference is arithmetic (summation order), not a flipped unit."""
    return F.relu(x) if mask is None else x * mask.to(x.dtype)


def trn_multiscale(p, x, cfg: Config, with_tuples: bool = False, masks=None):
    """TRNmodule.py:58-82.  x [B,T,F] -> [B,T-1,256] (with_tuples: also the per-scale lists of tuple activations).
    masks: optional [B, n_tuples, 256] on/off pattern of the tuple activations (tuples in visiting order), see _relu_m."""
    rel = selected_relations(cfg.num_segments)
    B = x.size(0)
    acts, parts = [], []
    t_idx = 0
    for sid, tuples in enumerate(rel):
        scale = len(tuples[0])
        acc = None
        zs = []
        for tup in tuples:
            a = x[:, list(tup), :].reshape(B, scale * cfg.feat_dim)
            a = _relu_m(_linear(p, f"TRN.fc_fusion_scales.{sid}.1", F.relu(a), cfg), None if masks is None else masks[:, t_idx])
            t_idx += 1
            zs.append(a)
            acc = a if acc is None else acc + a
        acts.append(acc.unsqueeze(1))
        parts.append(zs)
    out = torch.cat(acts, 1)
    return (out, parts) if with_tuples else out


def trans_attn(pred_domain):
    """models.py:351-357: w = 1 - H(softmax(pred_domain))."""
    ent = torch.sum(-F.softmax(pred_domain, 1) * F.log_softmax(pred_domain, 1), 1)
    return 1 - ent


def forward_domain(p, x, beta, cfg: Config, drop_i=None, drop_v=None, reverse_mu=None, domain="S", bn_running=None, bn_batch=None, masks=None):
    """One domain's pass through VideoModel.forward (models.py:545-722) for the
    trn-m / video / TransAttn configuration.  x [B,T,D].  drop_i / drop_v are
    optional multiplicative dropout masks already scaled by 1/(1-p)
    ([B*T,F] and [B,256]); None means dropout off (eval or p=0).
    masks: optional dict of on/off patterns {"F1" [B*T,F], "Hf" [B*T,F], "Z" [B,n_tuples,256], "Hr" [B,T-1,256], "Hv" [B,256]} imposed
    on the ReLUs (trn-m path; _relu_m).
    Returns dict with the reference's per-domain outputs."""
    B, T = x.size(0), cfg.num_segments
    mk = (lambda k: None) if masks is None else (lambda k: masks.get(k))
    z0 = _linear(p, "fc_feature_shared_source", x.reshape(-1, x.size(-1)), cfg)          # :565-566
    if cfg.use_bn != "none":
        # domainAlign 'shared' (:490-543, 569-570) with alpha = 1 (self.alpha is ones(1); AutoDIAL's Parameter never receives a
        # gradient, it is read with .item()): no source/target mixing, each domain through its own BatchNorm1d.  bn_running =
        # (mean, var): eval mode; else batch statistics (and bn_batch, a dict, receives them for the running-average update)
        w, b_ = p[f"bn_shared_{domain}.weight"], p[f"bn_shared_{domain}.bias"]
        if bn_running is not None:
            z0 = F.batch_norm(z0, bn_running[0], bn_running[1], w, b_, training=False, eps=1e-5)
        else:
            if bn_batch is not None:
                bn_batch[domain] = (z0.detach().mean(0), z0.detach().var(0, unbiased=True), z0.size(0))
            z0 = F.batch_norm(z0, None, None, w, b_, training=True, eps=1e-5)
    f = _relu_m(z0, mk("F1"))                                                        # :572
    if drop_i is not None:
        f = f * drop_i                                                               # :574-575
    feat_frame = f.view(B, T, -1)                                                    # :578
    # frame-level adversarial branch (:456-462, :606-610)
    h = _relu_m(_linear(p, "fc_feature_domain", _GradReverse.apply(f, beta[2]), cfg), mk("Hf"))
    pred_frame = _linear(p, "fc_classifier_domain", h, cfg).view(B, T, 2)
    if cfg.compute_dead_branches:
        _ = F.linear(f, p["fc_classifier_source.weight"], p["fc_classifier_source.bias"])       # :617-618, dead for baseline_type 'video'
    if cfg.frame_aggregation == "avgpool":
        # aggregate_frames, "1. averaging" (:421-433) without attention; attn is a placeholder column (:627-628)
        v = feat_frame.mean(1)
        vd = v * drop_v if drop_v is not None else v                                 # :679
        if reverse_mu is not None:                                                   # :682-684
            vd = _GradReverse.apply(vd, reverse_mu)
        y = _linear(p, "fc_classifier_video_source", vd, cfg)                        # :686
        hv = F.relu(_linear(p, "fc_feature_domain_video", _GradReverse.apply(vd, beta[1]), cfg))
        pred_video = _linear(p, "fc_classifier_domain_video", hv, cfg)
        y2 = y
        if cfg.ens_DA == "MCD":                                                      # :716-720
            y2 = F.linear(vd, p["fc_classifier_video_source_2.weight"], p["fc_classifier_video_source_2.bias"])
        return dict(attn=v[:, 0], out=y, out2=y2, pred_domain=[pred_video, pred_video, pred_frame], feat=[y, v, feat_frame])
    # TRN (:632-636)
    rel, rel_parts = trn_multiscale(p, feat_frame, cfg, with_tuples=True, masks=mk("Z"))
    # relation discriminators (:472-488)
    preds, hrs = [], []
    for i in range(T - 1):
        if cfg.arithmetic == "bf16":      # hidden layer on the scale's tuple activations as separate (separately rounded) K segments
            name = f"relation_domain_classifier_all.{i}.0"
            zs = [_GradReverse.apply(z, beta[0]) for z in rel_parts[i]]
            hr = _relu_m(_SegSumMatmulBf16.apply(p[name + ".weight"], *zs) + p[name + ".bias"], None if mk("Hr") is None else mk("Hr")[:, i])
        else:
            r = _GradReverse.apply(rel[:, i, :], beta[0])
            hr = _relu_m(_linear(p, f"relation_domain_classifier_all.{i}.0", r), None if mk("Hr") is None else mk("Hr")[:, i])
        hrs.append(hr)
        preds.append(_linear(p, f"relation_domain_classifier_all.{i}.2", hr, cfg).view(-1, 1, 2))
    pred_rel = torch.cat(preds, 1).view(-1, 2)
    # transferable attention (:379-388, :643-645)
    if cfg.use_attn == "TransAttn":
        w = trans_attn(pred_rel).view(-1, T - 1, 1)
        rel_attn = (w + 1) * rel
        attn = w[:, :, 0]
    else:
        rel_attn, attn = rel, rel[:, :, 0]
    v = torch.sum(rel_attn, 1)                                                       # :651
    vd = v * drop_v if drop_v is not None else v                                     # :679
    if reverse_mu is not None:                                                       # :682-684 (the MCD step's second forward)
        vd = _GradReverse.apply(vd, reverse_mu)
    y = _linear(p, "fc_classifier_video_source", vd, cfg)                            # :686
    hv = _relu_m(_linear(p, "fc_feature_domain_video", _GradReverse.apply(vd, beta[1]), cfg), mk("Hv"))  # :464-470
    pred_video = _linear(p, "fc_classifier_domain_video", hv, cfg)
    y2 = y
    if cfg.ens_DA == "MCD":                                                          # :716-720
        y2 = F.linear(vd, p["fc_classifier_video_source_2.weight"], p["fc_classifier_video_source_2.bias"])
    return dict(attn=attn, out=y, out2=y2,
                pred_domain=[pred_rel.view(B, T - 1, 2), pred_video, pred_frame],   # :697-707, :722 reversed
                feat=[y, v, feat_frame],                                             # :578, :675, :690, :722
                # the post-ReLU hidden activations (test infrastructure: mask-synchronised comparisons read their on/off patterns)
                hidden=dict(F1=f, Hf=h, Z=torch.stack([z for zs in rel_parts for z in zs], 1), Hr=torch.stack(hrs, 1), Hv=hv))


# ----------------------------------------------------------------------------
# losses (main.py:439-562, loss.py:15-25)
# ----------------------------------------------------------------------------
def attentive_entropy(pred, pred_domain):
    """loss.py:15-25."""
    ent_d = torch.sum(-F.softmax(pred_domain, 1) * F.log_softmax(pred_domain, 1), 1)
    w = 1 + ent_d
    return torch.mean(w * torch.sum(-F.softmax(pred, 1) * F.log_softmax(pred, 1), 1))


def gaussian_kernel(source, target, kernel_mul=2.0, kernel_num=5, fix_sigma=None):
    """loss.py:46-59: sum of kernel_num RBF kernels on the stacked [source; target] rows, bandwidths bw * mul^i around the mean
    pairwise squared distance (taken from .data: no gradient through the bandwidth)."""
    n = int(source.size(0)) + int(target.size(0))
    total = torch.cat([source, target], dim=0)
    l2 = ((total.unsqueeze(0) - total.unsqueeze(1)) ** 2).sum(2)
    bw = fix_sigma if fix_sigma else torch.sum(l2.detach()) / (n * n - n)
    bw = bw / kernel_mul ** (kernel_num // 2)
    return sum(torch.exp(-l2 / (bw * kernel_mul ** i)) for i in range(kernel_num))


def mmd_rbf(source, target, kernel_mul=2.0, kernel_num=5, fix_sigma=None):
    """loss.py:61-85, ver=2 (the only one main.py uses)."""
    b = int(source.size(0))
    k = gaussian_kernel(source, target, kernel_mul, kernel_num, fix_sigma)
    return torch.mean(k[:b, :b] + k[b:, b:] - k[:b, b:] - k[b:, :b])


def jan(source_list, target_list, kernel_muls=(2.0, 2.0), kernel_nums=(2, 5), fix_sigma_list=(None, None)):
    """loss.py:87-120, ver=2: the layers' kernels multiplied."""
    b = int(source_list[0].size(0))
    joint = None
    for i in range(len(source_list)):
        k = gaussian_kernel(source_list[i], target_list[i], kernel_muls[i], kernel_nums[i], fix_sigma_list[i])
        joint = k if joint is None else joint * k
    return torch.mean(joint[:b, :b] + joint[b:, b:] - joint[:b, b:] - joint[b:, :b])


def dis_mcd(out1, out2):
    """loss.py:27-29."""
    return torch.mean(torch.abs(F.softmax(out1, dim=1) - F.softmax(out2, dim=1)))


def discrepancy_loss(src, tgt, cfg: Config, n_src: int, n_tgt: int):
    """main.py:452-505: JAN on [Y, V]; DAN = mmd_rbf per enabled feature level, in batches of <= 256 rows."""
    fs = [f[:n_src] for f in src["feat"]]
    ft = [f[:n_tgt] for f in tgt["feat"]]
    kernel_muls, kernel_nums, fix_sigma = [2.0] * 2, [2, 5], [None] * 2
    if cfg.dis_DA == "JAN":
        fs, ft = fs[:-cfg.add_fc], ft[:-cfg.add_fc]
        n = min(fs[0].size(0), ft[0].size(0))
        return jan([f[:n] for f in fs], [f[:n] for f in ft], kernel_muls, kernel_nums, fix_sigma)
    kernel_muls += [kernel_muls[-1]] * cfg.add_fc
    kernel_nums += [kernel_nums[-1]] * cfg.add_fc
    fix_sigma += [fix_sigma[-1]] * cfg.add_fc
    loss = 0
    for l in range(cfg.add_fc + 2):
        if cfg.place_dis[l] != "Y":
            continue
        n = min(fs[l].size(0), ft[l].size(0))
        a, b = fs[l][:n], ft[l][:n]
        sb = min(256, n)
        a = a.reshape((-1, sb) + tuple(a.shape[1:]))
        b = b.reshape((-1, sb) + tuple(b.shape[1:]))
        parts = [mmd_rbf(a[t], b[t], kernel_muls[l], kernel_nums[l], fix_sigma[l]) for t in range(a.size(0))]
        loss = loss + sum(parts) / len(parts)
    return loss


def total_loss(src, tgt, label_source, gamma, cfg: Config,
               n_src: Optional[int] = None, n_tgt: Optional[int] = None, alpha: float = 0.0, tgt_rev=None):
    """main.py:421-422 (removeDummy), 439-451 (classification CE), 508-538
    (adversarial CE per enabled level), 559-562 (attentive entropy)."""
    n_src = src["out"].size(0) if n_src is None else n_src
    n_tgt = tgt["out"].size(0) if n_tgt is None else n_tgt
    out_s, out_t = src["out"][:n_src], tgt["out"][:n_tgt]
    loss_c = F.cross_entropy(out_s, label_source[:n_src])
    if cfg.ens_DA == "MCD":                                                # main.py:447
        loss_c = loss_c + F.cross_entropy(src["out2"][:n_src], label_source[:n_src])
    loss = loss_c
    parts = {"loss_c": loss_c}
    if cfg.dis_DA != "none":                                               # main.py:452-505
        loss_d = discrepancy_loss(src, tgt, cfg, n_src, n_tgt)
        loss = loss + alpha * loss_d
        parts["loss_d"] = loss_d
    pred_all = []
    loss_a = 0
    for l in range(3):
        if cfg.place_adv[l] == "Y":
            ps = src["pred_domain"][l][:n_src].reshape(-1, 2)
            pt = tgt["pred_domain"][l][:n_tgt].reshape(-1, 2)
            lab = torch.cat((torch.zeros(ps.size(0)), torch.ones(pt.size(0)))).long().to(ps.device)
            pd = torch.cat((ps, pt), 0)
            pred_all.append(pd)
            loss_a = loss_a + F.cross_entropy(pd, lab)
    if pred_all:
        loss = loss + loss_a
        parts["loss_a"] = loss_a
    if cfg.ens_DA == "MCD":                                                # main.py:548-556: from the second, reversed forward
        loss_s = -dis_mcd(tgt_rev["out"][:n_tgt], tgt_rev["out2"][:n_tgt])
        loss = loss + loss_s
        parts["loss_s"] = loss_s
    if cfg.add_loss_DA == "attentive_entropy" and cfg.use_attn != "none":
        # with MCD the reference has re-bound out_target to the second, reversed forward's output by now (main.py:550-552),
        # so the entropy term's gradient reaches the target features through GradReverse(mu)
        out_t_e = tgt_rev["out"][:n_tgt] if cfg.ens_DA == "MCD" else out_t
        loss_e = attentive_entropy(torch.cat((out_s, out_t_e), 0), pred_all[1])
        loss = loss + gamma * loss_e
        parts["loss_e"] = loss_e
    parts["loss"] = loss
    return loss, parts


# ----------------------------------------------------------------------------
# train step (main.py:348-352, 574-583, 620-621, 800-802)
# ----------------------------------------------------------------------------
def beta_dann(p: float) -> float:
    """main.py:351."""
    return float(2.0 / (1.0 + np.exp(-10 * p)) - 1)


def lr_dann(lr0: float, p: float) -> float:
    """main.py:800-802."""
    return lr0 / (1.0 + 10 * p) ** 0.75


@dataclass
class TrainState:
    params: Dict[str, torch.Tensor]
    momentum: Dict[str, torch.Tensor] = field(default_factory=dict)
    lr: float = 3e-2


def train_step(state: TrainState, xs, xt, label_source, beta, gamma, cfg: Config,
               momentum=0.9, weight_decay=1e-4, clip=20.0, drop_i=None, drop_v=None,
               n_src=None, n_tgt=None, grad_hook=None, alpha=0.0, mu=0.0, masks=None):
    """One optimisation step: forward both domains, total loss, backward,
    clip_grad_norm_ (main.py:578-581), Nesterov SGD with weight decay
    (main.py:83, 583; torch.optim.SGD semantics: g += wd*p; buf = mu*buf + g
    (buf = g on first use); g = g + mu*buf; p -= lr*g).  Parameters without a
    gradient are skipped (they are not in any BASELINE config's graph)."""
    p = {k: v.detach().clone().requires_grad_(True) for k, v in state.params.items()}
    di_s = di_t = dv_s = dv_t = None
    if drop_i is not None:
        di_s, di_t = drop_i
    if drop_v is not None:
        dv_s, dv_t = drop_v
    src = forward_domain(p, xs, beta, cfg, di_s, dv_s, domain="S", masks=None if masks is None else masks[0])
    tgt = forward_domain(p, xt, beta, cfg, di_t, dv_t, domain="T", masks=None if masks is None else masks[1])
    tgt_rev = None
    if cfg.ens_DA == "MCD":      # main.py:550: the whole model once more with reverse=True; only the target outputs are used
        tgt_rev = forward_domain(p, xt, beta, cfg, di_t, dv_t, reverse_mu=mu, domain="T")
    loss, parts = total_loss(src, tgt, label_source, gamma, cfg, n_src, n_tgt, alpha=alpha, tgt_rev=tgt_rev)
    names = [k for k in p if is_live(k)]
    grads = torch.autograd.grad(loss, [p[k] for k in names], allow_unused=True)
    g = {k: gi for k, gi in zip(names, grads) if gi is not None}
    if grad_hook is not None:            # e.g. a data-parallel all-reduce
        g = grad_hook(g)
    raw = {k: v.clone() for k, v in g.items()}
    total_norm = torch.linalg.vector_norm(
        torch.stack([torch.linalg.vector_norm(v) for v in g.values()]))
    if clip is not None:
        coef = torch.clamp(clip / (total_norm + 1e-6), max=1.0)      # clip_grad_norm_
        g = {k: v * coef for k, v in g.items()}
    new_params = {k: v.detach().clone() for k, v in state.params.items()}
    for k, gi in g.items():
        d = gi + weight_decay * new_params[k]
        buf = state.momentum.get(k)
        buf = d.clone() if buf is None else momentum * buf + d
        state.momentum[k] = buf
        d = d + momentum * buf
        new_params[k] = new_params[k] - state.lr * d
    state.params = new_params
    return dict(loss=loss.detach(), parts={k: v.detach() for k, v in parts.items()},
                src=src, tgt=tgt, grads=raw, clipped=g, total_norm=total_norm)
