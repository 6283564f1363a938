"""`VideoModel` - the reference's model surface (models.py:59-67 ctor, :545-722
forward, state_dict keys of models.py:141-294 / TRNmodule.py:44-54) on top of
libta3n_hip.so.  `main.py`/`test_models.py` of the reference construct it, call
`forward(input_source, input_target, beta, mu, is_train, reverse)`, read the
10-tuple, back-propagate through it with autograd and save/load its state_dict;
all of that works here, with every arithmetic op of forward and backward executed
by the HIP kernels (no eager-PyTorch or CPU fallback: calling forward without the
HIP library or without a GPU raises).

Supported configurations = the TA3N hot path and TemPooling:
frame_aggregation='trn-m' (use_attn in {'TransAttn','none'}) or 'avgpool' (use_attn 'none': BASELINE configs[0] and the
TemPooling + RevGrad rows), baseline_type='video', share_params='Y', use_bn='none', add_fc=1, ens_DA='none',
use_attn_frame='none'.
Anything else raises NotImplementedError at construction (several of those
branches are broken in the reference itself).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from . import _lib
from .TRNmodule import RelationModuleMultiScale

torch.manual_seed(1)          # models.py:14 seeds at import; kept so default inits are reproducible

# models.py:125-126 reads torchvision.models.<arch>(True).fc.in_features; torchvision is
# not a dependency here, the table below is that lookup.
ARCH_FEATURE_DIM = {"resnet18": 512, "resnet34": 512, "resnet50": 2048, "resnet101": 2048, "resnet152": 2048,
                    "c3d": 4096}


class GradReverse(torch.autograd.Function):
    """models.py:20-29.  Exported for API compatibility; inside VideoModel the
    reversal is folded into the discriminator input-gradient GEMMs as a -beta scale."""

    @staticmethod
    def forward(ctx, x, beta):
        ctx.beta = beta
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output.neg() * ctx.beta, None


def _bn_before_forward(model, plan, ws, train):
    """use_bn, eval mode: the running statistics go to the kernel (region "bn_run" [S, T][mean, var][F])."""
    if model.use_bn != 'none' and not train:
        off, n = plan.region("bn_run")
        ws[off:off + n].copy_(torch.stack((model.bn_shared_S.running_mean, model.bn_shared_S.running_var,
                                           model.bn_shared_T.running_mean, model.bn_shared_T.running_var)).reshape(-1))


def _bn_after_forward(model, plan, ws, train, Bs, Bt):
    """use_bn, train mode: nn.BatchNorm1d's buffer update (momentum 0.1, unbiased batch variance) from the kernel's batch
    statistics (region "bn_batch" [S, T][mean, biased var, 1/std][F])."""
    if model.use_bn == 'none' or not train:
        return
    off, n = plan.region("bn_batch")
    st = ws[off:off + n].view(2, 3, -1)
    T = model.train_segments
    for d, (mod, rows) in enumerate(((model.bn_shared_S, Bs * T), (model.bn_shared_T, Bt * T))):
        if rows > 0:
            with torch.no_grad():
                mod.running_mean.mul_(0.9).add_(st[d, 0], alpha=0.1)
                mod.running_var.mul_(0.9).add_(st[d, 1] * (rows / max(rows - 1, 1)), alpha=0.1)
                mod.num_batches_tracked += 1


def _backward_buffer(model, plan, dev):
    """(flat gradient buffer for one backward pass, fresh).  fresh: no parameter holds a gradient (the usual iteration after
    zero_grad(set_to_none=True)) - the pass writes into the model's PERSISTENT buffer, whose per-parameter views are cached, so
    handing the gradients out costs one attribute store per parameter.  Otherwise (a second backward before zero_grad: MCD's
    reversed pass, gradient accumulation) a temporary whose content is then added.  The alignment gaps between tensors and the
    parameters without a gradient read as zeros in either (ta3n_backward writes every live gradient in full, nothing else)."""
    items = model._flat_items(plan)
    if all(p.grad is None for _, _, _, p in items):
        buf = model._grad_buf
        if buf is None or buf.numel() != plan.param_floats or buf.device != dev:
            buf = model._grad_buf = torch.zeros(plan.param_floats, dtype=torch.float32, device=dev)
            model._grad_buf_views = [buf[off:off + p.numel()].view(shape) for _, off, shape, p in items]
        return buf, True
    return torch.zeros(plan.param_floats, dtype=torch.float32, device=dev), False


def _deliver_grads(model, plan, grads, fresh, unused, needs_grad) -> None:
    """Hands the parameter gradients of one backward pass to autograd's consumers WITHOUT going through 48 AccumulateGrad nodes
    (each clones what a Python Function returns: ~0.4 ms of copies per step): every parameter's .grad becomes a VIEW into one flat
    buffer laid out like the parameters (`model._grad_flat`; what clip_grad_norm_ / the optimiser then read - ta3n_amd.accel makes
    those two single passes over it).  Like torch with zero_grad(set_to_none=False), the buffer is reused by the next iteration.
    Parameters whose logits fed no loss keep grad None, like in the reference (see _HipForward.backward).
    grads, fresh: _backward_buffer's; needs_grad: per plan parameter."""
    items = model._flat_items(plan)
    live_n = plan.live_floats
    if fresh:
        model._grad_flat = target = grads
        views = model._grad_buf_views
    else:
        flat = model._grad_flat
        mine = flat is not None and flat.device == grads.device and flat.numel() == grads.numel()
        if mine:      # do the existing .grad tensors still live in the flat buffer (nobody replaced them)?
            base = flat.data_ptr()
            mine = all(p.grad is None or p.grad.data_ptr() == base + 4 * off for _, off, _, p in items)
        if mine:
            flat[:live_n].add_(grads[:live_n])
            target = flat
            views = model._grad_buf_views if flat is model._grad_buf else None
        else:         # somebody else's .grad tensors: accumulate into them one by one
            target = views = None
    for k, ((name, off, shape, p), (_, _, _, live), need) in enumerate(zip(items, plan.params, needs_grad)):
        if not live or not need or name.startswith(unused):
            continue
        if target is None:
            g = grads[off:off + p.numel()].view(shape)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.add_(g)
        elif p.grad is None:
            p.grad = views[k] if views is not None else target[off:off + p.numel()].view(shape)
    model._grad_live_floats = live_n
    # (ta3n_amd.accel: the tensor objects handed out, in parameters() order - a .grad somebody replaced is another object)
    model.__dict__["_grad_views"] = [p.grad for p in model.parameters() if p.grad is not None] if target is not None else None
    if model._grad_live_elems_plan is not plan:
        model._grad_live_elems = sum(p.numel() for (_, _, _, p), (_, _, _, live) in zip(items, plan.params) if live)
        model._grad_live_elems_plan = plan


class _HipForward(torch.autograd.Function):
    """One autograd node for the whole forward; backward = ta3n_backward."""

    @staticmethod
    def forward(ctx, model, xs, xt, beta, train, reverse_mu, *params):
        dev = model._flat.device
        Bs, Bt = xs.shape[0], xt.shape[0]
        plan = model._plan(Bs, Bt)
        x = torch.cat((xs.reshape(Bs * model.train_segments, -1), xt.reshape(Bt * model.train_segments, -1)), 0)
        x = x.to(device=dev, dtype=torch.float32).contiguous()
        ws = model._ws_checkout(plan, ctx, any(ctx.needs_input_grad))
        h = _lib.Hyper()
        h.beta[0], h.beta[1], h.beta[2] = float(beta[0]), float(beta[1]), float(beta[2])
        if reverse_mu is not None:           # forward(..., reverse=True): GradReverse(mu) behind dropout_v (models.py:682-684); the
            h.reverse, h.mu = 1, float(reverse_mu)      # workspace keeps the scalars, so this node's backward sees them again
        h.p_drop_i, h.p_drop_v = float(model.dropout_rate_i), float(model.dropout_rate_v)
        seeds = torch.randint(0, 2 ** 31 - 1, (2,))       # consumes the global torch RNG like nn.Dropout would
        h.seed_i, h.seed_v = int(seeds[0]), int(seeds[1])
        h.valid_source, h.valid_target, h.train = Bs, Bt, int(bool(train))
        h.inv_n_cls = h.inv_n_rel = h.inv_n_vid = h.inv_n_frm = h.inv_n_ent = 0.0   # the loss kernel is not used on this path
        L = _lib.lib()
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(L.ta3n_set_hyper(plan.handle, ws.data_ptr(), C.byref(h), stream), "ta3n_set_hyper")
        _bn_before_forward(model, plan, ws, train)
        _lib.check(L.ta3n_forward(plan.handle, x.data_ptr(), model._flat.data_ptr(), ws.data_ptr(), stream), "ta3n_forward")
        _bn_after_forward(model, plan, ws, train, Bs, Bt)
        ctx.model, ctx.plan, ctx.x, ctx.ws = model, plan, x, ws
        ctx.n_params = len(params)
        B, T, NR, Cn = Bs + Bt, model.train_segments, model.train_segments - 1, model.num_class

        def reg(name, shape):
            off, n = plan.region(name)
            return ws[off:off + n].view(shape).clone()

        attn, y = reg("attn", (B, NR)), reg("Y", (B, Cn))
        pr, pv, pf = reg("Pr", (B, NR, 2)), reg("Pv", (B, 2)), reg("Pf", (B, T, 2))
        v, f1 = reg("V", (B, -1)), reg("F1", (B, T, -1))
        y2 = reg("Y2", (B, Cn)) if model.ens_DA == 'MCD' else y.new_zeros(0)
        # the pooled video feature takes a gradient from the caller (dis_DA DAN / JAN on feat[1], main.py:452-505); the frame
        # features cannot be a loss operand in the reference either (loss.py:49 raises on 3-D features)
        ctx.mark_non_differentiable(f1)
        if model.ens_DA != 'MCD':
            ctx.mark_non_differentiable(y2)
        if not model._attn_on:
            ctx.mark_non_differentiable(attn)
        ctx.set_materialize_grads(False)      # an output that feeds no loss arrives as None in backward (see there)
        return attn, y, pr, pv, pf, v, f1, y2

    @staticmethod
    def backward(ctx, g_attn, g_y, g_pr, g_pv, g_pf, g_v, g_f1, g_y2):
        model, plan, ws = ctx.model, ctx.plan, ctx.ws
        dev = ws.device

        def put(name, g):
            off, n = plan.region(name)
            if g is None:
                ws[off:off + n].zero_()
            else:
                ws[off:off + n].copy_(g.reshape(-1))

        put("gY", g_y); put("gPr", g_pr); put("gPv", g_pv); put("gPf", g_pf)
        put("g_attn", g_attn if model._attn_on else None)
        put("gV_ext", g_v)
        if model.ens_DA == 'MCD':
            put("gY2", g_y2)
        grads, fresh = _backward_buffer(model, plan, dev)
        L = _lib.lib()
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(L.ta3n_backward(plan.handle, ctx.x.data_ptr(), model._flat.data_ptr(), grads.data_ptr(), ws.data_ptr(),
                                   stream), "ta3n_backward")
        model._ws_release(ctx)
        # A discriminator whose logits feed no loss (place_adv 'N', use_target none, ...) keeps grad None in the reference,
        # so torch.optim.SGD skips it - no weight decay either (main.py:508-538).  Same here: None, not zeros.
        unused = []
        if g_pf is None:
            unused += ["fc_feature_domain.", "fc_classifier_domain."]
        if g_pv is None:
            unused += ["fc_feature_domain_video.", "fc_classifier_domain_video."]
        if g_pr is None and not model._attn_on:      # with TransAttn the weights are not detached (models.py:351-357): the relation
            unused += ["relation_domain_classifier_all."]      # discriminators receive a gradient through V whatever the loss uses
        _deliver_grads(model, plan, grads, fresh, tuple(unused), ctx.needs_input_grad[6:])
        return (None,) * (6 + ctx.n_params)


class _HipForwardAvg(torch.autograd.Function):
    """frame_aggregation='avgpool' (TemPooling, models.py:421-433): forward = ta3n_forward of the general avgpool plan
    (F1 | Hf | mean | {Y, Hv} | {Pv, Pf}), backward = ta3n_backward from the caller's logit gradients."""

    @staticmethod
    def forward(ctx, model, xs, xt, beta, train, reverse_mu, *params):
        dev = model._flat.device
        Bs, Bt = xs.shape[0], xt.shape[0]
        T = int(xs.shape[1])                  # TemPooling averages whatever number of segments it is given (models.py:421-433):
        plan = model._plan(Bs, Bt, T)         # validation may use val_segments != train_segments (models.py:60, 555): a plan per count
        x = torch.cat((xs.reshape(Bs * T, -1), xt.reshape(Bt * T, -1)), 0).to(device=dev, dtype=torch.float32).contiguous()
        ws = model._ws_checkout(plan, ctx, any(ctx.needs_input_grad))
        h = _lib.Hyper()
        h.beta[0], h.beta[1], h.beta[2] = float(beta[0]), float(beta[1]), float(beta[2])
        if reverse_mu is not None:           # forward(..., reverse=True) (models.py:682-684)
            h.reverse, h.mu = 1, float(reverse_mu)
        h.p_drop_i, h.p_drop_v = float(model.dropout_rate_i), float(model.dropout_rate_v)
        seeds = torch.randint(0, 2 ** 31 - 1, (2,))
        h.seed_i, h.seed_v = int(seeds[0]), int(seeds[1])
        h.valid_source, h.valid_target, h.train = Bs, Bt, int(bool(train))
        L = _lib.lib()
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(L.ta3n_set_hyper(plan.handle, ws.data_ptr(), C.byref(h), stream), "ta3n_set_hyper")
        _bn_before_forward(model, plan, ws, train)
        _lib.check(L.ta3n_forward(plan.handle, x.data_ptr(), model._flat.data_ptr(), ws.data_ptr(), stream), "ta3n_forward")
        _bn_after_forward(model, plan, ws, train, Bs, Bt)
        ctx.model, ctx.plan, ctx.x, ctx.ws = model, plan, x, ws
        ctx.n_params = len(params)
        B, Cn = Bs + Bt, model.num_class

        def reg(name, shape):
            off, n = plan.region(name)
            return ws[off:off + n].view(shape).clone()

        y, pv, pf = reg("Y", (B, Cn)), reg("Pv", (B, 2)), reg("Pf", (B, T, 2))
        v, f1 = reg("V", (B, -1)), reg("F1", (B, T, -1))
        y2 = reg("Y2", (B, Cn)) if model.ens_DA == 'MCD' else y.new_zeros(0)
        ctx.mark_non_differentiable(f1)            # (V takes the caller's gradient: dis_DA on feat[1])
        if model.ens_DA != 'MCD':
            ctx.mark_non_differentiable(y2)
        ctx.set_materialize_grads(False)
        return y, pv, pf, v, f1, y2

    @staticmethod
    def backward(ctx, g_y, g_pv, g_pf, g_v, g_f1, g_y2):
        model, plan, ws = ctx.model, ctx.plan, ctx.ws
        dev = ws.device
        for name, g in (("gY", g_y), ("gPv", g_pv), ("gPf", g_pf), ("gV_ext", g_v)) + ((("gY2", g_y2),) if model.ens_DA == 'MCD' else ()):
            off, n = plan.region(name)
            if g is None:
                ws[off:off + n].zero_()
            else:
                ws[off:off + n].copy_(g.reshape(-1))
        grads, fresh = _backward_buffer(model, plan, dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(_lib.lib().ta3n_backward(plan.handle, ctx.x.data_ptr(), model._flat.data_ptr(), grads.data_ptr(), ws.data_ptr(),
                                            stream), "ta3n_backward")
        model._ws_release(ctx)
        unused = []                               # a discriminator that feeds no loss keeps grad None (see _HipForward.backward)
        if g_pf is None:
            unused += ["fc_feature_domain.", "fc_classifier_domain."]
        if g_pv is None:
            unused += ["fc_feature_domain_video.", "fc_classifier_domain_video."]
        _deliver_grads(model, plan, grads, fresh, tuple(unused), ctx.needs_input_grad[6:])
        return (None,) * (6 + ctx.n_params)


class VideoModel(nn.Module):
    def __init__(self, num_class, baseline_type, frame_aggregation, modality,
                 train_segments=5, val_segments=25,
                 base_model='resnet101', path_pretrained='', new_length=None,
                 before_softmax=True,
                 dropout_i=0.5, dropout_v=0.5, use_bn='none', ens_DA='none',
                 crop_num=1, partial_bn=True, verbose=True, add_fc=1, fc_dim=1024,
                 n_rnn=1, rnn_cell='LSTM', n_directions=1, n_ts=5,
                 use_attn='TransAttn', n_attn=1, use_attn_frame='none',
                 share_params='Y'):
        super().__init__()
        unsupported = []
        if frame_aggregation not in ('trn-m', 'avgpool'): unsupported.append(f"frame_aggregation={frame_aggregation!r}")
        if frame_aggregation == 'avgpool' and use_attn != 'none':
            unsupported.append("frame_aggregation='avgpool' with use_attn (the reference's script runs TemPooling with use_attn none)")
        if baseline_type != 'video': unsupported.append(f"baseline_type={baseline_type!r}")
        if share_params != 'Y': unsupported.append("share_params='N'")
        if use_bn not in ('none', 'AdaBN', 'AutoDIAL'): unsupported.append(f"use_bn={use_bn!r}")
        if use_bn != 'none' and frame_aggregation not in ('trn-m', 'avgpool'): unsupported.append("use_bn with this frame_aggregation")
        if ens_DA not in ('none', 'MCD'): unsupported.append(f"ens_DA={ens_DA!r}")
        if ens_DA == 'MCD' and frame_aggregation not in ('trn-m', 'avgpool'): unsupported.append("ens_DA='MCD' with this frame_aggregation")
        if use_attn not in ('TransAttn', 'none'): unsupported.append(f"use_attn={use_attn!r}")
        if use_attn_frame != 'none': unsupported.append(f"use_attn_frame={use_attn_frame!r}")
        if not before_softmax: unsupported.append("before_softmax=False")
        if add_fc < 1:
            raise ValueError('add at least one fc layer')          # models.py:137-138
        if add_fc != 1: unsupported.append(f"add_fc={add_fc}")
        if unsupported:
            raise NotImplementedError("ta3n_amd.VideoModel implements the TA3N hot path only; unsupported: " +
                                      ", ".join(unsupported))
        if base_model not in ARCH_FEATURE_DIM:
            raise ValueError(f"unknown base_model {base_model!r}")
        self.modality = modality
        self.train_segments, self.val_segments = train_segments, val_segments
        self.baseline_type, self.frame_aggregation = baseline_type, frame_aggregation
        self.reshape, self.before_softmax = True, before_softmax
        self.dropout_rate_i, self.dropout_rate_v = dropout_i, dropout_v
        self.use_bn, self.ens_DA, self.crop_num = use_bn, ens_DA, crop_num
        self.add_fc, self.fc_dim, self.share_params = add_fc, fc_dim, share_params
        self.n_layers, self.rnn_cell, self.n_directions, self.n_ts = n_rnn, rnn_cell, n_directions, n_ts
        self.use_attn, self.n_attn, self.use_attn_frame = use_attn, n_attn, use_attn_frame
        self.new_length = (1 if modality == "RGB" else 5) if new_length is None else new_length
        self.num_class = num_class
        self._attn_on = use_attn == 'TransAttn'
        self._avg = frame_aggregation == 'avgpool'
        if verbose:
            print(f"Initializing TSN with base model: {base_model}. input_modality: {modality}, "
                  f"num_segments: {train_segments}, new_length: {self.new_length}")

        # ---- parameters, named and initialised as models.py:119-325 ----
        self.feature_dim = ARCH_FEATURE_DIM[base_model]
        std = 0.001
        F_ = min(fc_dim, self.feature_dim) if fc_dim > 0 else self.feature_dim       # models.py:129
        NB = 256                                                                     # models.py:223

        def lin(i, o):
            m = nn.Linear(i, o)
            nn.init.normal_(m.weight, 0, std)
            nn.init.constant_(m.bias, 0)
            return m

        self.fc_feature_shared_source = lin(self.feature_dim, F_)        # :141
        self.fc_feature_source = lin(F_, F_)                             # :156 (unused in forward, kept for checkpoints)
        if use_bn != 'none':                                             # :194-198 AdaBN (ICLRW 2017): BN for source / target
            self.bn_shared_S = nn.BatchNorm1d(F_)                        # the two the trn-m forward uses (:515-516, 569-570)
            self.bn_shared_T = nn.BatchNorm1d(F_)
            self.bn_source_S = nn.BatchNorm1d(F_)                        # created by the reference, never used on this path
            self.bn_source_T = nn.BatchNorm1d(F_)
        self.fc_feature_domain = lin(F_, F_)                             # :161
        self.fc_classifier_source = lin(F_, num_class)                   # :166 (dead for baseline_type='video')
        self.fc_classifier_domain = lin(F_, 2)                           # :170
        self.num_bottleneck = NB
        if self._avg:                                                    # feat_aggregated_dim = feat_shared_dim (models.py:246-247)
            A = F_
        else:
            A = NB
            self.TRN = RelationModuleMultiScale(F_, NB, train_segments, verbose=verbose)   # :224 (default nn.Linear init)
            self.bn_trn_S = nn.BatchNorm1d(NB)                           # :225-226 (unused with use_bn='none')
            self.bn_trn_T = nn.BatchNorm1d(NB)
        self.fc_feature_video_source = lin(A, A)                         # :258 (unused)
        self.fc_feature_video_source_2 = lin(A, A)                       # :262 (unused)
        self.fc_feature_domain_video = lin(A, A)                         # :267
        self.fc_classifier_video_source = lin(A, num_class)              # :272
        if ens_DA == 'MCD':
            self.fc_classifier_video_source_2 = lin(A, num_class)        # :276-279 second classifier for self-ensembling
        self.fc_classifier_domain_video = lin(A, 2)                      # :281
        if not self._avg:
            self.relation_domain_classifier_all = nn.ModuleList(         # :286-294 (default init)
                nn.Sequential(nn.Linear(NB, NB), nn.ReLU(), nn.Linear(NB, 2)) for _ in range(train_segments - 1))
        if use_bn != 'none':                                             # :307-312 (unused with trn-m)
            self.bn_source_video_S = nn.BatchNorm1d(A)
            self.bn_source_video_T = nn.BatchNorm1d(A)
            self.bn_source_video_2_S = nn.BatchNorm1d(A)
            self.bn_source_video_2_T = nn.BatchNorm1d(A)
        self.alpha = torch.ones(1)                                       # :314 plain attribute
        if use_bn == 'AutoDIAL':                                         # :315-316; read with .item() in forward: it never gets a gradient
            self.alpha = nn.Parameter(self.alpha)
        self.relu = nn.ReLU(inplace=True)
        self.dropout_i = nn.Dropout(p=dropout_i)                         # kept as attributes; the HIP kernels apply them
        self.dropout_v = nn.Dropout(p=dropout_v)
        self._enable_pbn = partial_bn
        self._flat: Optional[torch.Tensor] = None
        self._grad_flat: Optional[torch.Tensor] = None      # the flat buffer the parameters' .grad tensors are views of (_deliver_grads)
        self._mom_flat: Optional[torch.Tensor] = None       # ... and the momentum buffers of ta3n_amd.accel's optimiser step
        self._items_cache = {}
        self._grad_live_floats = self._grad_live_elems = 0
        self._grad_live_elems_plan = None
        self._grad_buf: Optional[torch.Tensor] = None       # persistent flat gradient buffer and its per-parameter views (_backward_buffer)
        self._grad_buf_views = None
        self._plans: Dict[Tuple[int, int], _lib.Plan] = {}
        self._ws_init: Dict[int, torch.Tensor] = {}
        self._ws_pool: Dict[int, list] = {}
        self._feat_dim_F = F_

    # ---- reference API ----
    def partialBN(self, enable):                                          # models.py:348-349
        self._enable_pbn = enable

    def train(self, mode=True):
        # models.py:328-346 touches a non-existent self.base_model when partial BN is on (it only
        # works with --no_partialbn, the default, opts.py:89); there is no BatchNorm2d to freeze here.
        return super().train(mode)

    def get_trans_attn(self, pred_domain):                                # models.py:351-357 (utility, eager)
        p = torch.softmax(pred_domain, 1)
        return 1 - torch.sum(-p * torch.log_softmax(pred_domain, 1), 1)

    # ---- plumbing ----
    def _flags(self) -> int:
        # losses are assembled by the caller (main.py:439-562), so every discriminator has to be able to receive a gradient:
        # the adversarial flags only decide the plan's live parameter set here (the loss kernel is not used on this path)
        if self._avg:
            return (_lib.FLAG_ADV_VIDEO | _lib.FLAG_ADV_FRAME | _lib.FLAG_FEATURE_GRADS |
                    (_lib.FLAG_BN_SHARED if self.use_bn != 'none' else 0) | (_lib.FLAG_MCD if self.ens_DA == 'MCD' else 0))
        return (_lib.FLAG_ADV_RELATION | _lib.FLAG_ADV_VIDEO | _lib.FLAG_ADV_FRAME |
                (_lib.FLAG_TRANS_ATTN if self._attn_on else 0) |
                _lib.FLAG_FEATURE_GRADS |                                   # feat[1] may carry a discrepancy loss (dis_DA)
                (_lib.FLAG_BN_SHARED if self.use_bn != 'none' else 0) |
                (_lib.FLAG_MCD if self.ens_DA == 'MCD' else 0))

    def _plan(self, Bs: int, Bt: int, T: Optional[int] = None) -> _lib.Plan:
        T = self.train_segments if T is None else int(T)
        key = (Bs, Bt, T)
        if key not in self._plans:
            self._plans[key] = _lib.Plan(Bs, Bt, T, self.feature_dim, self._feat_dim_F, self.num_class,
                                         self._flags(), aggregation=_lib.AGG_AVGPOOL if self._avg else _lib.AGG_TRN_M)
        return self._plans[key]

    def _ws_template(self, plan: _lib.Plan) -> torch.Tensor:
        key = id(plan)
        t = self._ws_init.get(key)
        if t is None or t.device != self._flat.device:
            t = torch.zeros(plan.ws_floats, dtype=torch.float32, device=self._flat.device)
            stream = C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
            _lib.check(_lib.lib().ta3n_init_workspace(plan.handle, t.data_ptr(), stream), "ta3n_init_workspace")
            self._ws_init[key] = t
        return t

    def _ws_checkout(self, plan: _lib.Plan, ctx, keep: bool) -> torch.Tensor:
        """A workspace for one forward (and its backward).  Workspaces are pooled per plan: a train loop alternates forward /
        backward and therefore reuses ONE buffer (every region is rewritten in full by the launches, like the engine's), instead
        of cloning ~40 MB per call; a second forward before the first one's backward takes another buffer.  keep=False
        (nothing requires grad): the buffer goes straight back once the outputs have been copied out - the launches and the
        output copies are stream-ordered before any later reuse."""
        import weakref
        pool = self._ws_pool.setdefault(id(plan), [])
        tmpl = self._ws_template(plan)
        entry = next((e for e in pool if not e[1] and e[0].device == tmpl.device), None)
        if entry is None:
            entry = [tmpl.clone(), False]
            pool.append(entry)
        if keep:
            # Ownership token: the entry is free again only when THIS checkout gives it back.  The finalizer of an autograd node
            # runs when the node dies - which can be long after its backward released the entry and another forward checked the
            # same buffer out (the previous iteration's graph lives until `loss` is rebound) - so neither path may clear an
            # entry that carries somebody else's token.
            tok = object()
            entry[1] = tok
            ctx._ws_entry = (entry, tok)
            weakref.finalize(ctx, self._ws_give_back, entry, tok)      # a graph dropped without backward frees it too
        return entry[0]

    @staticmethod
    def _ws_give_back(entry, tok) -> None:
        if entry[1] is tok:
            entry[1] = False

    def _ws_release(self, ctx) -> None:
        e = getattr(ctx, "_ws_entry", None)
        if e is not None:
            self._ws_give_back(*e)

    def _named_flat_params(self, plan: _lib.Plan):
        named = dict(self.named_parameters())
        return [(name, off, shape, named[name]) for name, off, shape, _ in plan.params]

    def _flat_items(self, plan: _lib.Plan):
        """[(name, offset, shape, nn.Parameter)] in the plan's order, cached per plan (walking the module tree costs ~0.1 ms per
        forward); re-validated by identity against the owning modules' _parameters dicts (a replaced Parameter object drops it)."""
        key = id(plan)
        hit = self._items_cache.get(key)
        if hit is not None and all(owner._parameters.get(leaf) is p for owner, leaf, p in hit[0]):
            return hit[1]
        items = self._named_flat_params(plan)
        owners = []
        for name, _, _, p in items:
            mod = self
            *path, leaf = name.split(".")
            for part in path:
                mod = getattr(mod, part)
            owners.append((mod, leaf, p))
        self._items_cache[key] = (owners, items)
        return items

    def _ensure_flat(self, plan: _lib.Plan, device: torch.device) -> None:
        """Parameters are views into one flat fp32 buffer laid out as the plan wants (live
        parameters first: that prefix is the gradient all-reduce / optimiser operand).
        Re-established whenever .to()/.cuda()/load replaced the parameter storage."""
        items = self._flat_items(plan)
        ok = self._flat is not None and self._flat.device == device
        if ok:
            base = self._flat.data_ptr()
            ok = all(p.data.data_ptr() == base + 4 * off and p.device == device for _, off, _, p in items)
        if ok:
            return
        flat = torch.zeros(plan.param_floats, dtype=torch.float32, device=device)
        for name, off, shape, p in items:
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1).to(device=device, dtype=torch.float32))
            p.data = flat[off:off + n].view(shape)
        for b_name, b in list(self.named_buffers()):
            if b.device != device:
                mod = self
                *path, leaf = b_name.split(".")
                for part in path:
                    mod = getattr(mod, part)
                mod._buffers[leaf] = b.to(device)
        self._flat = flat
        self._grad_flat = self._mom_flat = self._grad_buf = self._grad_buf_views = None
        import weakref
        me = weakref.ref(self)
        for q in self.parameters():      # (ta3n_amd.accel recognises the model's parameters by this)
            q._ta3n_owner = me
        self._ws_init.clear()
        self._ws_pool.clear()

    def forward(self, input_source, input_target, beta, mu, is_train, reverse):
        """models.py:545-722.  Returns (attn_s, out_s, out_s2, pred_domain_s, feat_s, attn_t, out_t,
        out_t2, pred_domain_t, feat_t) with pred_domain = [relation [B,T-1,2], video [B,2],
        frame [B,T,2]] and feat = [class logits, video feature V, frame features F1]."""
        if not torch.cuda.is_available():
            raise _lib.Ta3nError("ta3n_amd.VideoModel.forward needs a HIP device; there is no CPU fallback")
        if self.use_bn != 'none' and float(self.alpha.detach()) != 1.0:
            raise NotImplementedError("use_bn with alpha != 1 (source/target batch mixing of domainAlign, models.py:497-508, 531-533): the "
                                      "reference's own program never changes alpha from its initial 1")
        num_segments = self.train_segments if is_train else self.val_segments
        if num_segments != self.train_segments and not self._avg:
            # (the reference's TRN is built for train_segments frames, models.py:222-224: another count fails there too; TemPooling
            # averages any number of segments and gets a plan per count)
            raise ValueError("val_segments must equal num_segments for frame_aggregation 'trn-m' (the relation module is built for train_segments)")
        if input_source.dim() != 3 or input_target.dim() != 3 or input_source.size(1) != num_segments or \
                input_source.size(2) != self.feature_dim or input_target.size(2) != self.feature_dim:
            raise ValueError("inputs must be [B, num_segments, feature_dim]")
        device = next(self.parameters()).device
        if device.type != "cuda":
            device = torch.device("cuda", torch.cuda.current_device())
        Bs, Bt = input_source.size(0), input_target.size(0)
        plan = self._plan(Bs, Bt, num_segments if self._avg else None)
        self._ensure_flat(plan, device)
        params = [p for _, _, _, p in self._flat_items(plan)]
        s, t = slice(0, Bs), slice(Bs, Bs + Bt)
        if self._avg:
            with torch.cuda.device(device):
                y, pv, pf, v, f1, y2 = _HipForwardAvg.apply(self, input_source, input_target, list(beta), self.training,
                                                            float(mu) if reverse else None, *params)
            if self.ens_DA != 'MCD':
                y2 = y
            # models.py:627-628 (attn placeholder = first feature column), :697-708 (the relation slot repeats the video logits)
            return (v[s][:, 0], y[s], y2[s], [pv[s], pv[s], pf[s]], [y[s], v[s], f1[s]],
                    v[t][:, 0], y[t], y2[t], [pv[t], pv[t], pf[t]], [y[t], v[t], f1[t]])
        with torch.cuda.device(device):
            attn, y, pr, pv, pf, v, f1, y2 = _HipForward.apply(self, input_source, input_target, list(beta), self.training,
                                                                float(mu) if reverse else None, *params)
        out_s, out_t = y[s], y[t]
        out_s2, out_t2 = (y2[s], y2[t]) if self.ens_DA == 'MCD' else (out_s, out_t)      # models.py:713-720
        return (attn[s], out_s, out_s2, [pr[s], pv[s], pf[s]], [y[s], v[s], f1[s]],
                attn[t], out_t, out_t2, [pr[t], pv[t], pf[t]], [y[t], v[t], f1[t]])
