"""Temporal relation modules - parameter containers with the reference's names
(TRNmodule.py:27-56) and the bit-exact frame-tuple tables.

Inside `VideoModel` the multi-scale relation network runs as one grouped GEMM
launch of libta3n_hip.so (the per-tuple gather+concat of TRNmodule.py:60-63,
75-77 is folded into the A-operand addressing; see csrc/ta3n_plan.cpp phase F2),
so this module's own `forward` is never called on main.py's path.
"""
from __future__ import annotations

from math import comb

import torch
from torch import nn

from . import _lib


class RelationModuleMultiScale(nn.Module):
    """Multi-scale TRN (TRNmodule.py:27-86): scales [T..2], at most 3 evenly spaced frame
    tuples per scale, one ReLU-Linear(scale*D -> bottleneck)-ReLU fusion per scale."""

    def __init__(self, img_feature_dim, num_bottleneck, num_frames, verbose=True):
        super().__init__()
        self.subsample_num = 3                                               # TRNmodule.py:32
        self.img_feature_dim = img_feature_dim
        self.num_frames = num_frames
        self.scales = [i for i in range(num_frames, 1, -1)]                  # :34
        # the reference enumerates all C(T,s) tuples (:36-41); only the selected ones are ever
        # used (:60, :71) and they are produced by unranking in the C library
        self.relations_selected = _lib.relation_table(num_frames)
        self.subsample_scales = [min(self.subsample_num, comb(num_frames, s)) for s in self.scales]
        self.fc_fusion_scales = nn.ModuleList(                               # :44-54 (default nn.Linear init)
            nn.Sequential(nn.ReLU(), nn.Linear(s * img_feature_dim, num_bottleneck), nn.ReLU()) for s in self.scales)
        if verbose:
            print('Multi-Scale Temporal Relation Network Module in use', ['%d-frame relation' % i for i in self.scales])

    def return_relationset(self, num_frames, num_frames_relation):           # :84-86 (utility)
        import itertools
        return list(itertools.combinations(range(num_frames), num_frames_relation))

    def _hip_state(self, B, T, D, NB, dev):
        import ctypes as C
        key = (B, str(dev))
        cache = self.__dict__.setdefault("_hip_cache", {})
        if key not in cache:
            plan = _lib.Plan(B, 0, T, D, D, 1, 0, num_bottleneck=NB)
            with torch.cuda.device(dev):
                ws = torch.zeros(plan.ws_floats, dtype=torch.float32, device=dev)
                _lib.check(_lib.lib().ta3n_init_workspace(plan.handle, ws.data_ptr(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                           "ta3n_init_workspace")
            cache[key] = (plan, ws, torch.zeros(plan.param_floats, dtype=torch.float32, device=dev),
                          torch.zeros(plan.param_floats, dtype=torch.float32, device=dev), torch.zeros(B * T, D, dtype=torch.float32, device=dev))
        return cache[key]

    def forward(self, input):
        """TRNmodule.py:58-82 on its own: input [B, T, D] -> [B, T-1, bottleneck].  Runs the SAME grouped tile-list launches the
        train step uses: forward = the tuple GEMMs (gather + concat folded into the operand addressing, bias + ReLU in the
        epilogue; csrc/ta3n_plan.cpp: spec_Z), backward = the launch of the TRN weight gradients and the scatter-free input
        gradient (push_trn_wgrads / push_f1_grad) - on a plan of B source videos; the per-scale sum of the (at most 3) tuple
        activations and its fan-out back through their ReLU masks are the only things done here."""
        if not torch.cuda.is_available():
            raise _lib.Ta3nError("RelationModuleMultiScale.forward needs a HIP device; there is no CPU fallback")
        if input.dim() != 3 or input.size(1) != self.num_frames or input.size(2) != self.img_feature_dim:
            raise ValueError("input must be [B, num_frames, img_feature_dim]")
        NB = self.fc_fusion_scales[0][1].out_features
        if NB != 256:
            raise NotImplementedError("the fused launch sequence exists for num_bottleneck = 256 (TA3N's value, models.py:223)")
        params = [t for seq in self.fc_fusion_scales for t in (seq[1].weight, seq[1].bias)]
        return _TrnStandalone.apply(self, input, *params)


class _TrnStandalone(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, input, *params):
        import ctypes as C
        B, T, D = input.shape
        NB = mod.fc_fusion_scales[0][1].out_features
        dev = input.device if input.is_cuda else torch.device("cuda", torch.cuda.current_device())
        plan, ws, flat, grads, x = mod._hip_state(B, T, D, NB, dev)
        offs = {n: (o, sh) for n, o, sh, _ in plan.params}
        for j in range(len(mod.fc_fusion_scales)):
            for nm, t in (("weight", params[2 * j]), ("bias", params[2 * j + 1])):
                o, sh = offs[f"TRN.fc_fusion_scales.{j}.1.{nm}"]
                flat[o:o + t.numel()].copy_(t.detach().reshape(-1))
        o_f1, n_f1 = plan.region("F1")
        # the reference applies a ReLU to the gathered frames first (TRNmodule.py:49): the launch reads F1 as stored, so it is applied here
        ws[o_f1:o_f1 + n_f1].copy_(torch.relu(input.detach().to(dev, torch.float32)).reshape(-1))
        L = _lib.lib()
        h = _lib.Hyper()
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            _lib.check(L.ta3n_set_hyper(plan.handle, ws.data_ptr(), C.byref(h), stream), "ta3n_set_hyper")
            _lib.check(L.ta3n_train_step_range(plan.handle, x.data_ptr(), flat.data_ptr(), grads.data_ptr(), ws.data_ptr(), 1, 1, stream),
                       "ta3n_train_step_range")      # launch 1 of the fused sequence: the tuple GEMMs (+ the frame discriminator's hidden layer)
        o_z, n_z = plan.region("Zr")
        z = ws[o_z:o_z + n_z].view(B, -1, NB)
        out, t0 = [], 0
        for tuples in mod.relations_selected:             # sum of the scale's tuple activations (TRNmodule.py:73-79)
            out.append(z[:, t0:t0 + len(tuples)].sum(1, keepdim=True))
            t0 += len(tuples)
        ctx.mod, ctx.dev, ctx.dims = mod, dev, (B, T, D, NB)
        ctx.in_device, ctx.in_dtype = input.device, input.dtype
        # What backward needs is kept PER CALL: the module's workspace is shared by every call with this batch size, and the reference
        # calls the module twice per forward - TRN(source), TRN(target), models.py:636-651 - before one backward (with
        # equal batch sizes the second call used to overwrite the first one's activations, silently).
        ctx.save_for_backward(ws[o_f1:o_f1 + n_f1].clone(), z.clone(), *[t.detach() for t in params])
        return torch.cat(out, 1)

    @staticmethod
    def backward(ctx, g_out):
        import ctypes as C
        mod, dev, (B, T, D, NB) = ctx.mod, ctx.dev, ctx.dims
        plan, ws, flat, grads, x = mod._hip_state(B, T, D, NB, dev)
        f1, z, *params = ctx.saved_tensors                                # this call's own activations and parameters back into the workspace
        o_f1, n_f1 = plan.region("F1")
        ws[o_f1:o_f1 + n_f1].copy_(f1)
        offs = {n: (o, sh) for n, o, sh, _ in plan.params}
        for j in range(len(mod.fc_fusion_scales)):
            for nm, t in (("weight", params[2 * j]), ("bias", params[2 * j + 1])):
                o, sh = offs[f"TRN.fc_fusion_scales.{j}.1.{nm}"]
                flat[o:o + t.numel()].copy_(t.reshape(-1))
        o_gz, n_gz = plan.region("gZ")
        gz = ws[o_gz:o_gz + n_gz].view(B, -1, NB)
        g = g_out.to(dev, torch.float32)
        t0 = 0
        for j, tuples in enumerate(mod.relations_selected):     # d(sum of the scale's tuples) through each tuple's ReLU
            for k in range(len(tuples)):
                gz[:, t0 + k] = g[:, j] * (z[:, t0 + k] > 0)
            t0 += len(tuples)
        o, n = plan.region("gHf")
        ws[o:o + n].zero_()                                        # (no frame discriminator behind the standalone module)
        L = _lib.lib()
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        n_launch = L.ta3n_num_phases(plan.handle, 4)
        with torch.cuda.device(dev):                              # second-to-last launch: TRN weight gradients + gradient at the frame features
            _lib.check(L.ta3n_train_step_range(plan.handle, x.data_ptr(), flat.data_ptr(), grads.data_ptr(), ws.data_ptr(), n_launch - 2, 1, stream),
                       "ta3n_train_step_range")
        o, n = plan.region("gZ1")
        g_in = ws[o:o + n].view(B, T, D).clone().to(ctx.in_device, ctx.in_dtype) if ctx.needs_input_grad[1] else None
        offs = {nm: (o_, sh) for nm, o_, sh, _ in plan.params}
        out = []
        for j in range(len(mod.fc_fusion_scales)):
            for nm in ("weight", "bias"):
                o_, sh = offs[f"TRN.fc_fusion_scales.{j}.1.{nm}"]
                cnt = 1
                for v in sh:
                    cnt *= v
                out.append(grads[o_:o_ + cnt].view(sh).clone())
        return (None, g_in, *out)


class RelationModule(nn.Module):
    """Single-scale TRN (TRNmodule.py:6-25).  The reference's 'trn' mode crashes in
    VideoModel.forward (models.py:639 uses relation_domain_classifier_all, built only for
    'trn-m'), so only the parameter container is provided."""

    def __init__(self, img_feature_dim, num_bottleneck, num_frames):
        super().__init__()
        self.num_frames, self.img_feature_dim, self.num_bottleneck = num_frames, img_feature_dim, num_bottleneck
        self.classifier = nn.Sequential(nn.ReLU(), nn.Linear(num_frames * img_feature_dim, num_bottleneck), nn.ReLU())

    def forward(self, input):
        raise NotImplementedError("frame_aggregation='trn' is not on the TA3N hot path")
