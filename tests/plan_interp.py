"""CPU interpreter of a ta3n plan (test infrastructure).

Executes the SAME descriptor lists the HIP kernels consume (Seg / Task / Phase
PODs exported by ta3n_debug_arrays) with numpy, so the whole wiring of the
train step - operand offsets, K-segments, GradReverse scales, epilogues, fan-out,
workspace layout - is validated against the oracle on a machine without a GPU.
The pointwise phases (pool fwd/bwd, loss, grad-norm, SGD) are executed from
their specification in the kernel headers.  Nothing here is used by the product.
"""
import ctypes as C

import numpy as np

from ta3n_amd import _lib

BASE_X, BASE_P, BASE_G, BASE_WS, BASE_P16 = 0, 1, 2, 3, 4
EPI_BIAS, EPI_ADD, EPI_RELU, EPI_MASK, EPI_DROP_I, EPI_DROP_V, EPI_SUMROWS8, EPI_SUMSQ, EPI_ROWSUM_A = 1, 2, 4, 8, 16, 32, 64, 128, 256
EPI_COLSUM = 1 << 12
EPI_SGD = 1 << 11
EPI_SPLITK = 1 << 15
(PH_GEMM, PH_POOL_FWD, PH_LOSS, PH_POOL_BWD, PH_GRAD_NORM, PH_SGD, PH_HEADS, PH_POOL_CLS, PH_POOL_AVG_FWD, PH_POOL_AVG_BWD,
 PH_BN_FWD, PH_BN_BWD) = range(12)
HEADS_RPW = 16


class Seg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("a_base", "b_base", "a_off", "b_off", "a_ld", "b_ld", "a_kmajor", "b_kmajor",
                                         "klen", "scale_kind")] + [("pad", C.c_int32 * 2)]


class Task(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("m0", "n0", "m_valid", "n_valid", "seg_begin", "seg_count")] +
                [("epi", C.c_uint32)] +
                [(n, C.c_int32) for n in ("alpha_kind", "gamma_kind", "c_base", "c_off", "c_ld", "bias_base", "bias_off",
                                          "aux_base", "aux_off", "aux_ld", "add_base", "add_off", "add_ld", "drop_ld",
                                          "fan_count", "fan_ld")] +
                [("fan_mask_off", C.c_int32 * 3), ("fan_out_off", C.c_int32 * 3), ("seg0", Seg), ("cost", C.c_int32),
                 ("pad", C.c_int32 * 4), ("sig", C.c_int32), ("wait_begin", C.c_int32), ("wait_count", C.c_int32), ("pad2", C.c_int32)])


class Phase(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("kind", "group", "task_begin", "task_count", "wm", "wn", "wk", "bf16", "rm", "rn", "chain_off", "chain_n")]


BN_COLS = 4      # ta3n_types.h: columns per workgroup of the BatchNorm launches


def round_bf16(a):
    """Round-to-nearest-even to bfloat16 precision (what v_cvt_pk_bf16_f32 does), returned in the input dtype."""
    f = np.ascontiguousarray(a, dtype=np.float32)
    u = f.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32).astype(a.dtype)


_GEOM_FIELDS = ["Bs", "Bt", "B", "T", "D", "F", "NB", "C", "n_tuples", "n_rel", "flags",
                "o_F1", "o_Hf", "o_Pf", "o_Zr", "o_Hr", "o_Pr", "o_R", "o_attn", "o_V", "o_Vd", "o_Y", "o_Hv", "o_Pv",
                "o_gY", "o_gPv", "o_gPr", "o_gPf", "o_gattn", "o_gHv", "o_gHf", "o_gVt", "o_gPrT", "o_gRa", "o_gHr",
                "o_gR", "o_gZ", "o_gZ1", "o_zeros", "o_ones", "o_losses", "o_norm_part", "o_grad_norm", "o_hyper", "o_labels",
                "o_tuple_first", "n_norm_blocks", "live_floats", "p_W2_0", "p_b2_0", "p_W2_stride", "p_b2_stride",
                "p_Wcd", "p_bcd", "p_Wdv", "p_bdv", "p_Wcv", "p_bcv", "p_Wcdv", "p_bcdv", "o_fh_part", "o_fh_bpart",
                "o_loss_part", "n_vid_wg", "n_frm_wg", "heads_rpw", "o_sumsq", "n_sumsq", "o_metrics", "o_confusion",
                "o_ws16", "o_p16", "o_x16", "ws16_span", "o_gV_ext", "o_Y2", "o_gY2", "o_Z0", "o_gZ0", "o_bn_batch", "o_bn_run",
                "p_bn_w0", "p_bn_w1", "p_bn_b0", "p_bn_b1", "o_p16b", "pair_delta", "heads_vpw"]


class Geom(C.Structure):
    _fields_ = [(n, C.c_uint32 if n == "flags" else C.c_int32) for n in _GEOM_FIELDS]


def struct_sizes_ok():
    s = [C.c_int32() for _ in range(5)]
    _lib.lib().ta3n_debug_struct_sizes(*[C.byref(x) for x in s])
    return (s[0].value == C.sizeof(Seg) and s[1].value == C.sizeof(Task) and s[2].value == C.sizeof(Phase)
            and s[3].value == C.sizeof(Geom) and s[4].value == C.sizeof(_lib.Hyper))


def plan_arrays(plan):
    L = _lib.lib()
    ptrs = [C.c_void_p() for _ in range(6)]
    ns = [C.c_int64() for _ in range(3)]
    L.ta3n_debug_arrays(plan.handle, C.byref(ptrs[0]), C.byref(ns[0]), C.byref(ptrs[1]), C.byref(ns[1]),
                        C.byref(ptrs[2]), C.byref(ns[2]), C.byref(ptrs[3]), C.byref(ptrs[4]), C.byref(ptrs[5]))
    segs = C.cast(ptrs[0], C.POINTER(Seg * ns[0].value)).contents
    tasks = C.cast(ptrs[1], C.POINTER(Task * ns[1].value)).contents
    phases = C.cast(ptrs[2], C.POINTER(Phase * ns[2].value)).contents
    geom = C.cast(ptrs[3], C.POINTER(Geom)).contents
    tf = np.ctypeslib.as_array(C.cast(ptrs[5], C.POINTER(C.c_int32)), shape=(geom.n_rel + 1,)).copy()
    tup = (np.ctypeslib.as_array(C.cast(ptrs[4], C.POINTER(C.c_int32)), shape=(geom.n_tuples, geom.T)).copy()
           if geom.n_tuples > 0 else np.zeros((0, geom.T), np.int32))      # TA3N_AGG_AVGPOOL has no relation tuples
    return segs, tasks, phases, geom, tup, tf


def plan_waits(plan):
    """[n, 2] int array of the (counter, target) pairs of the plan's chained launches."""
    w = C.c_void_p(); n = C.c_int64()
    _lib.lib().ta3n_debug_waits(plan.handle, C.byref(w), C.byref(n))
    if n.value == 0:
        return np.zeros((0, 2), np.int32)
    return np.ctypeslib.as_array(C.cast(w, C.POINTER(C.c_int32)), shape=(n.value, 2)).copy()


def mix32(x):
    x = np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF
    x ^= x >> 16; x = (x * 0x7feb352d) & 0xFFFFFFFF
    x ^= x >> 15; x = (x * 0x846ca68b) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def keep_mask(seed, idx, p):
    """Same stateless dropout stream as ta3n_kernels.h:keep_mask."""
    seed = np.uint64(seed)
    h = mix32(mix32((np.asarray(idx, np.uint64) + np.uint64(0x9E3779B9) * (seed | np.uint64(1))) & 0xFFFFFFFF) ^ seed)
    u = (h >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (u >= np.float32(p)).astype(np.float64)


class Interp:
    def __init__(self, plan, dtype=np.float64):
        assert struct_sizes_ok(), "ctypes mirrors out of date with ta3n_types.h"
        self.plan = plan
        self.segs, self.tasks, self.phases, self.g, self.tuples, self.tf = plan_arrays(plan)
        self.dtype = dtype
        g = self.g
        self.ws = np.zeros(plan.ws_floats, dtype)
        self.ws[g.o_ones:g.o_ones + g.B * g.T * 4] = 1.0
        self.P = np.zeros(plan.param_floats, dtype)
        self.G = np.zeros(plan.param_floats, dtype)
        self.M = np.zeros(plan.param_floats, dtype)
        self.X = None
        self.labels = np.zeros(g.B, np.int64)
        self.hy = None
        self.side = None          # scalars of the update that rides in a pipelined step's first launch: dict(lr, momentum, weight_decay, clip)

    # ---- helpers ----
    def set_params(self, state):
        for name, off, shape, live in self.plan.params:
            n = int(np.prod(shape))
            self.P[off:off + n] = state[name].detach().cpu().double().numpy().reshape(-1)

    def get_params(self, src=None):
        src = self.P if src is None else src
        return {name: src[off:off + int(np.prod(shape))].reshape(shape).copy() for name, off, shape, live in self.plan.params}

    def buf(self, base):
        return {BASE_X: self.X, BASE_P: self.P, BASE_G: self.G, BASE_WS: self.ws}[base]

    def scale(self, kind):
        h = self.hy
        if kind == 1: return -h["beta"][0]
        if kind == 2: return -h["beta"][1]
        if kind == 3: return -h["beta"][2]
        if kind == 4: return 1.0 / (1.0 - h["p_drop_i"]) if (h["train"] and h["p_drop_i"] > 0) else 1.0
        if kind == 5: return 1.0 / (1.0 - h["p_drop_v"]) if (h["train"] and h["p_drop_v"] > 0) else 1.0
        if kind == 6: return -h["mu"] if h.get("reverse", 0) else 1.0
        return 1.0

    def view2d(self, base, off, ld, rows, cols, r0=0, c0=0):
        b = self.buf(base)
        idx = off + (np.arange(r0, r0 + rows)[:, None] * ld) + np.arange(c0, c0 + cols)[None, :]
        return idx, b

    def operand(self, base, off, ld, kmajor, r0, nr, klen):
        """[nr, klen] matrix of element (r, k)."""
        b = self.buf(base)
        r = np.arange(r0, r0 + nr)[:, None]
        k = np.arange(klen)[None, :]
        idx = off + (k * ld + r if kmajor else r * ld + k)
        return b[idx]

    # ---- phases ----
    def chain_order(self, ph, adversarial=True, seed=0):
        """An execution order of a chained launch's tasks that respects ONLY the declared hand-offs (Task.sig / wait lists):
        among the tasks whose counters are reached, the highest index runs first (adversarial: consumers as early as their
        wait lists allow) or a random one.  If a wait list misses a producer, the consumer runs before it and reads stale data."""
        waits = plan_waits(self.plan)
        rng = np.random.default_rng(seed)
        ids = list(range(ph.task_begin, ph.task_begin + ph.task_count))
        cnt = np.zeros(max(ph.chain_n, 1), np.int64)
        pending = set(ids)
        order = []
        while pending:
            ready = [i for i in pending
                     if all(cnt[waits[w, 0]] >= waits[w, 1] for w in range(self.tasks[i].wait_begin, self.tasks[i].wait_begin + self.tasks[i].wait_count))]
            assert ready, "chained launch deadlocks: no task is ready"
            i = max(ready) if adversarial else int(rng.choice(ready))
            order.append(i)
            pending.discard(i)
            if self.tasks[i].sig >= 0:
                cnt[self.tasks[i].sig] += 1
        return order

    def run_gemm(self, ph, order=None):
        for ti in (order if order is not None else range(ph.task_begin, ph.task_begin + ph.task_count)):
            self.run_task(ph, ti)

    def run_task(self, ph, ti):
        BM, BN = 32 * ph.wm * max(ph.rm, 1), 32 * ph.wn * max(ph.rn, 1)
        if True:
            t = self.tasks[ti]
            if t.epi & EPI_SGD:          # optimiser side job of a pipelined step's first launch: update of params [4 pad0, 4 pad1)
                if self.side is None:
                    return
                g, sd = self.g, self.side
                total = np.sqrt(self.ws[g.o_sumsq:g.o_sumsq + g.n_sumsq].sum())
                coef = min(sd["clip"] / (total + 1e-6), 1.0) if sd["clip"] > 0 else 1.0
                lo, hi = 4 * t.pad[0], 4 * t.pad[1]
                d = self.G[lo:hi] * coef + sd["weight_decay"] * self.P[lo:hi]
                self.M[lo:hi] = sd["momentum"] * self.M[lo:hi] + d
                d = d + sd["momentum"] * self.M[lo:hi]
                self.P[lo:hi] -= sd["lr"] * d
                return
            if t.epi & EPI_COLSUM:       # exact column sums of a table of per-workgroup partials
                src, rows, ld = t.pad[0], t.pad[1], t.pad[2]
                n = np.arange(t.n0, t.n_valid)
                v = self.ws[src + np.arange(rows)[:, None] * ld + n[None, :]].sum(0)
                self.buf(t.c_base)[t.c_off + n] = v
                if t.epi & EPI_SUMSQ:
                    self.ws[t.pad[3]] = float((v * v).sum())
                return
            if t.seg_count == 0:
                return
            if t.epi & EPI_SUMROWS8:
                dst, src, rows = t.pad[0], t.pad[1], t.pad[2]
                self.ws[dst:dst + 8] = self.ws[src:src + 8 * rows].reshape(rows, 8).sum(0)
            nr = min(BM, t.m_valid - t.m0); nc = min(BN, t.n_valid - t.n0)
            assert nr > 0 and nc > 0
            acc = np.zeros((nr, nc), self.dtype)
            rowsum = np.zeros(nr, self.dtype)
            for si in range(t.seg_begin, t.seg_begin + t.seg_count):
                s = self.segs[si]
                if ph.bf16 & 16:
                    # the Seg addresses bf16 twins in units of two elements.  A twin holds round_bf16(original) - which is what
                    # the operand read below computes from the original - so the model reads the original; that the
                    # twin is up to date when the kernel reads it is what the GPU test checks against this model.
                    def untwin(off):
                        g = self.g
                        if off >= g.o_x16:
                            return BASE_X, (off - g.o_x16) * 2
                        assert not (g.o_p16 <= off < g.o_x16)      # parameter twins use BASE_P16
                        assert off >= g.o_ws16
                        return BASE_WS, (off - g.o_ws16) * 2
                    # (parameter twins are addressed relative to their own region: base BASE_P16, offset in pairs of elements)
                    ab, ao = (BASE_P, s.a_off * 2) if s.a_base == BASE_P16 else untwin(s.a_off)
                    bb, bo = (BASE_P, s.b_off * 2) if s.b_base == BASE_P16 else untwin(s.b_off)
                    assert s.a_base in (BASE_WS, BASE_P16) and s.b_base in (BASE_WS, BASE_P16)
                    A = self.operand(ab, ao, s.a_ld, s.a_kmajor, t.m0, nr, s.klen)
                    Bm = self.operand(bb, bo, s.b_ld, s.b_kmajor, t.n0, nc, s.klen)
                    if not (ph.bf16 & 32):      # (pair twins, bit 32: hi + lo planes = the original to 16 mantissa bits; modelled as exact)
                        A, Bm = round_bf16(A), round_bf16(Bm)
                    rowsum += A.sum(1)          # the bias gradient of a twin-reading tile sums the rounded values
                    acc += A @ Bm.T
                    acc *= self.scale(s.scale_kind)
                    continue
                A = self.operand(s.a_base, s.a_off, s.a_ld, s.a_kmajor, t.m0, nr, s.klen)
                rowsum += A.sum(1)
                Bm = self.operand(s.b_base, s.b_off, s.b_ld, s.b_kmajor, t.n0, nc, s.klen)
                if ph.bf16 and not (ph.bf16 & 32):      # TA3N_FLAG_BF16_MFMA: operands rounded to bf16, products and sums in the wide
                    # type (TA3N_FLAG_F32_SPLIT, bit 32, is fp32-grade: modelled as exact products like the fp32 MFMA)
                    acc += round_bf16(A) @ round_bf16(Bm).T
                else:
                    acc += A @ Bm.T
                acc *= self.scale(s.scale_kind)
            if t.epi & EPI_SPLITK:       # two tasks per tile, each over part of the K segments: the first to run leaves its partial, the second finishes
                part = self.__dict__.setdefault("_split_parts", {})
                key = int(t.pad[0])
                if key not in part:
                    part[key] = acc
                    return
                acc = acc + part.pop(key)
            v = acc
            m = np.arange(t.m0, t.m0 + nr)[:, None]
            n = np.arange(t.n0, t.n0 + nc)[None, :]
            if t.epi & EPI_BIAS:
                v = v + self.buf(t.bias_base)[t.bias_off + n]
            v = v * self.scale(t.alpha_kind)
            if t.epi & EPI_ADD:
                v = v + self.buf(t.add_base)[t.add_off + m * t.add_ld + n]
            if t.epi & EPI_RELU:
                v = np.maximum(v, 0)
            if t.epi & EPI_MASK:
                v = np.where(self.buf(t.aux_base)[t.aux_off + m * t.aux_ld + n] > 0, v, 0)
            if (t.epi & (EPI_DROP_I | EPI_DROP_V)) and self.hy["train"]:
                seed = self.hy["seed_i"] if t.epi & EPI_DROP_I else self.hy["seed_v"]
                p = self.hy["p_drop_i"] if t.epi & EPI_DROP_I else self.hy["p_drop_v"]
                v = v * keep_mask(seed, m * t.drop_ld + n, p)
            v = v * self.scale(t.gamma_kind)
            self.buf(t.c_base)[t.c_off + m * t.c_ld + n] = v
            if t.epi & EPI_ROWSUM_A:
                self.buf(t.bias_base)[t.bias_off + np.arange(t.m0, t.m0 + nr)] = rowsum
            if t.epi & EPI_SUMSQ:
                self.ws[t.pad[3]] = float((v * v).sum()) + (float((rowsum * rowsum).sum()) if t.epi & EPI_ROWSUM_A else 0.0)
            for f in range(t.fan_count):
                mk = self.ws[t.fan_mask_off[f] + m * t.fan_ld + n]
                self.ws[t.fan_out_off[f] + m * t.fan_ld + n] = np.where(mk > 0, v, 0)

    @staticmethod
    def soft2(z):
        m = z.max(-1, keepdims=True)
        e = np.exp(z - m); s = e.sum(-1, keepdims=True)
        p = e / s; lp = z - m - np.log(s)
        H = -(p * lp).sum(-1)
        return p, lp, H

    def r(self, off, shape):
        n = int(np.prod(shape))
        return self.ws[off:off + n].reshape(shape)

    def run_pool_fwd(self):
        g, h = self.g, self.hy
        B, NR, NB, NT = g.B, g.n_rel, g.NB, g.n_tuples
        Hr = self.r(g.o_Hr, (B, NR, NB)); Zr = self.r(g.o_Zr, (B, NT, NB))
        R = self.r(g.o_R, (B, NR, NB)); Pr = self.r(g.o_Pr, (B, NR, 2)); attn = self.r(g.o_attn, (B, NR))
        attn_on = bool(g.flags & _lib.FLAG_TRANS_ATTN)
        V = np.zeros((B, NB), self.dtype)
        for j in range(NR):
            W2 = self.P[g.p_W2_0 + j * g.p_W2_stride:][:2 * NB].reshape(2, NB)
            b2 = self.P[g.p_b2_0 + j * g.p_b2_stride:][:2]
            Pr[:, j, :] = Hr[:, j, :] @ W2.T + b2
            R[:, j, :] = Zr[:, self.tf[j]:self.tf[j + 1], :].sum(1)
            if attn_on:
                _, _, H = self.soft2(Pr[:, j, :])
                w = 1 - H
                attn[:, j] = w
                V += (w[:, None] + 1) * R[:, j, :]
            else:
                attn[:, j] = R[:, j, 0]
                V += R[:, j, :]
        self.r(g.o_V, (B, NB))[:] = V
        Vd = V
        if h["train"] and h["p_drop_v"] > 0:
            idx = np.arange(B)[:, None] * NB + np.arange(NB)[None, :]
            Vd = V * keep_mask(h["seed_v"], idx, h["p_drop_v"]) * self.scale(5)
        self.r(g.o_Vd, (B, NB))[:] = Vd

    def run_pool_cls(self):
        """TA3N_AGG_AVGPOOL: mean over segments, dropout_v, classifier, CE on the valid source rows and the way back to gZ1
        (ta3n_pointwise.hip: pool_cls_kernel)."""
        g, h = self.g, self.hy
        B, T, F, Cn = g.B, g.T, g.F, g.C
        F1 = self.r(g.o_F1, (B, T, F))
        V = F1.mean(1)
        mk = np.ones((B, F), self.dtype)
        if h["train"] and h["p_drop_v"] > 0:
            idx = np.arange(B)[:, None] * F + np.arange(F)[None, :]
            mk = keep_mask(h["seed_v"], idx, h["p_drop_v"]) * self.scale(5)
        Vd = V * mk
        Wcv = self.P[g.p_Wcv:g.p_Wcv + Cn * F].reshape(Cn, F); bcv = self.P[g.p_bcv:g.p_bcv + Cn]
        Y = Vd @ Wcv.T + bcv
        self.r(g.o_V, (B, F))[:] = V; self.r(g.o_Vd, (B, F))[:] = Vd; self.r(g.o_Y, (B, Cn))[:] = Y
        b = np.arange(B)
        on = (b < g.Bs) & (b < h["valid_source"])
        m = Y.max(1, keepdims=True); lp = Y - m - np.log(np.exp(Y - m).sum(1, keepdims=True)); p = np.exp(lp)
        onehot = np.zeros_like(Y); onehot[b[on], self.labels[on]] = 1
        gY = np.where(on[:, None], (p - onehot) * h["inv_n_cls"], 0.0)
        self.r(g.o_gY, (B, Cn))[:] = gY
        lpart = self.r(g.o_loss_part, (B, 8)); lpart[:] = 0
        lpart[b[on], 0] = lpart[b[on], 1] = -lp[b[on], self.labels[on]] * h["inv_n_cls"]
        gV = (gY @ Wcv) * mk / T * self.scale(4)
        self.r(g.o_gZ1, (B, T, F))[:] = np.where(F1 > 0, gV[:, None, :], 0.0)

    def run_pool_avg_fwd(self):
        """TA3N_AGG_AVGPOOL, general: V = mean over the segments, Vd = dropout_v(V) (ta3n_pointwise.hip: pool_avg_fwd_kernel)."""
        g, h = self.g, self.hy
        B, T, F = g.B, g.T, g.F
        V = self.r(g.o_F1, (B, T, F)).mean(1)
        mk = np.ones((B, F), self.dtype)
        if h["train"] and h["p_drop_v"] > 0:
            idx = np.arange(B)[:, None] * F + np.arange(F)[None, :]
            mk = keep_mask(h["seed_v"], idx, h["p_drop_v"]) * self.scale(5)
        self.r(g.o_V, (B, F))[:] = V
        self.r(g.o_Vd, (B, F))[:] = V * mk
        self.ws[g.o_losses:g.o_losses + 8] = 0

    def run_pool_avg_bwd(self):
        """gVt / T spread over the segments: gZ1 directly (no frame discriminator) or the additive base of its launch."""
        g = self.g
        B, T, F = g.B, g.T, g.F
        gv = self.r(g.o_gVt, (B, F))
        if g.o_gV_ext > 0:       # TA3N_FLAG_FEATURE_GRADS: the caller's gradient at V
            gv = gv + self.r(g.o_gV_ext, (B, F))
        base = np.repeat(gv / T, T, axis=0)
        if g.o_gHf < 0:
            self.r(g.o_gZ1, (B * T, F))[:] = np.where(self.r(g.o_F1, (B * T, F)) > 0, base * self.scale(4), 0.0)
        else:
            self.r(g.o_gRa, (B * T, F))[:] = base

    def run_loss(self):
        g, h = self.g, self.hy
        B, NR, T, Cn = g.B, g.n_rel, g.T, g.C
        Y = self.r(g.o_Y, (B, Cn)); Pv = self.r(g.o_Pv, (B, 2)); Pr = self.r(g.o_Pr, (B * NR, 2)); Pf = self.r(g.o_Pf, (B * T, 2))
        b = np.arange(B); is_src = b < g.Bs
        valid = np.where(is_src, b < h["valid_source"], (b - g.Bs) < h["valid_target"])
        d = (~is_src).astype(int)
        m = Y.max(1, keepdims=True); lp = Y - m - np.log(np.exp(Y - m).sum(1, keepdims=True)); p = np.exp(lp)
        Hc = -(p * lp).sum(1)
        cls_on = is_src & valid
        onehot = np.zeros_like(Y); onehot[b[cls_on], self.labels[cls_on]] = 1
        gY = np.where(cls_on[:, None], (p - onehot) * h["inv_n_cls"], 0.0)
        l_cls = float((-lp[b[cls_on], self.labels[cls_on]]).sum() * h["inv_n_cls"])
        q, lq, Hd = self.soft2(Pv)
        gPv = np.zeros_like(Pv); l_vid = 0.0; l_ent = 0.0
        oh2 = np.zeros_like(Pv); oh2[b, d] = 1
        # avgpool (no relation rows): the relation slot of pred_domain is the video logits once more (models.py:707-708)
        vmult = (1 if g.flags & _lib.FLAG_ADV_VIDEO else 0) + (1 if (NR == 0 and g.flags & _lib.FLAG_ADV_RELATION) else 0)
        if vmult:
            gPv += np.where(valid[:, None], (q - oh2) * h["inv_n_vid"] * vmult, 0)
            l_vid = float((-lq[b, d] * valid).sum() * h["inv_n_vid"] * vmult)
        if g.flags & _lib.FLAG_ATTN_ENTROPY:
            ce = h["gamma"] * h["inv_n_ent"]
            l_ent = float(((1 + Hd) * Hc * valid).sum() * h["inv_n_ent"])
            gY += np.where(valid[:, None], ce * (1 + Hd)[:, None] * (-p * (lp + Hc[:, None])), 0)
            gPv += np.where(valid[:, None], ce * Hc[:, None] * (-q * (lq + Hd[:, None])), 0)
        self.r(g.o_gY, (B, Cn))[:] = gY; self.r(g.o_gPv, (B, 2))[:] = gPv

        def rows(P, per, flag, inv_n):
            bb = np.repeat(b, per); vv = np.repeat(valid, per); dd = np.repeat(d, per)
            qq, lqq, _ = self.soft2(P)
            oh = np.zeros_like(P); oh[np.arange(P.shape[0]), dd] = 1
            if not (g.flags & flag):
                return np.zeros_like(P), 0.0
            return np.where(vv[:, None], (qq - oh) * inv_n, 0), float((-lqq[np.arange(P.shape[0]), dd] * vv).sum() * inv_n)
        gPr, l_rel = rows(Pr, NR, _lib.FLAG_ADV_RELATION, h["inv_n_rel"])
        gPf, l_frm = rows(Pf, T, _lib.FLAG_ADV_FRAME, h["inv_n_frm"])
        self.r(g.o_gPr, (B * NR, 2))[:] = gPr; self.r(g.o_gPf, (B * T, 2))[:] = gPf
        self.ws[g.o_losses:g.o_losses + 6] = [l_cls + l_rel + l_vid + l_frm + h["gamma"] * l_ent, l_cls, l_rel, l_vid, l_frm, l_ent]

    def _bn_rows(self, dom):
        g = self.g
        return (0, g.Bs * g.T) if dom == 0 else (g.Bs * g.T, g.Bt * g.T)

    def run_bn_fwd(self):
        """bn_shared_fwd_kernel: F1 = dropout_i(relu(BatchNorm_domain(Z0))), batch statistics to region bn_batch."""
        g, h = self.g, self.hy
        F, BT = g.F, g.B * g.T
        Z0 = self.r(g.o_Z0, (BT, F)); F1 = self.r(g.o_F1, (BT, F))
        for dom, (pw, pb) in enumerate(((g.p_bn_w0, g.p_bn_b0), (g.p_bn_w1, g.p_bn_b1))):
            r0, n = self._bn_rows(dom)
            if n == 0:
                continue
            z = Z0[r0:r0 + n]
            if h["train"]:
                mean = z.mean(0); var = ((z - mean) ** 2).mean(0); inv = 1.0 / np.sqrt(var + 1e-5)
                st = self.r(g.o_bn_batch + dom * 3 * F, (3, F)); st[0] = mean; st[1] = var; st[2] = inv
                run = self.r(g.o_bn_run + dom * 2 * F, (2, F))      # nn.BatchNorm1d's buffer update on the device (momentum 0.1, unbiased variance)
                run[0] = 0.9 * run[0] + 0.1 * mean; run[1] = 0.9 * run[1] + 0.1 * var * (n / max(n - 1, 1))
            else:
                run = self.r(g.o_bn_run + dom * 2 * F, (2, F)); mean = run[0]; inv = 1.0 / np.sqrt(run[1] + 1e-5)
            y = np.maximum((z - mean) * inv * self.P[pw:pw + F] + self.P[pb:pb + F], 0)
            if h["train"] and h["p_drop_i"] > 0:
                idx = (np.arange(r0, r0 + n)[:, None] * F + np.arange(F)[None, :])
                y = y * keep_mask(h["seed_i"], idx, h["p_drop_i"])
            F1[r0:r0 + n] = y * self.scale(4)

    def run_bn_bwd(self):
        """bn_shared_bwd_kernel: gZ0 and the BatchNorm weight / bias gradients from gZ1 (train mode)."""
        g = self.g
        F, BT = g.F, g.B * g.T
        Z0 = self.r(g.o_Z0, (BT, F)); gy = self.r(g.o_gZ1, (BT, F)); gZ0 = self.r(g.o_gZ0, (BT, F))
        for dom, (pw, pb) in enumerate(((g.p_bn_w0, g.p_bn_b0), (g.p_bn_w1, g.p_bn_b1))):
            r0, n = self._bn_rows(dom)
            if n == 0:
                if g.n_sumsq > 0:
                    nb = (F + BN_COLS - 1) // BN_COLS
                    first = g.o_sumsq + g.n_sumsq - 2 * nb + dom * nb
                    self.ws[first:first + nb] = 0
                continue
            st = self.r(g.o_bn_batch + dom * 3 * F, (3, F))
            xh = (Z0[r0:r0 + n] - st[0]) * st[2]
            gg = gy[r0:r0 + n]
            sg, sgx = gg.sum(0), (gg * xh).sum(0)
            self.G[pw:pw + F] = sgx; self.G[pb:pb + F] = sg
            if g.n_sumsq > 0:      # fused step: the launch's workgroups (16 columns x one domain each) leave their share of the gradient norm in the last slots of ws["sumsq"]
                nb = (F + BN_COLS - 1) // BN_COLS
                q = np.zeros(nb * BN_COLS); q[:F] = sgx.astype(np.float64) ** 2 + sg.astype(np.float64) ** 2
                first = g.o_sumsq + g.n_sumsq - 2 * nb + dom * nb
                self.ws[first:first + nb] = q.reshape(nb, BN_COLS).sum(1)
            gZ0[r0:r0 + n] = self.P[pw:pw + F] * st[2] * (gg - sg / n - xh * sgx / n)

    def run_pool_bwd(self):
        g = self.g
        B, NR, NB = g.B, g.n_rel, g.NB
        gVt = self.r(g.o_gVt, (B, NB)); R = self.r(g.o_R, (B, NR, NB)); Pr = self.r(g.o_Pr, (B, NR, 2))
        if g.o_gV_ext > 0:       # TA3N_FLAG_FEATURE_GRADS: the caller's gradient at the pooled feature joins here
            gVt = gVt + self.r(g.o_gV_ext, (B, NB))
        gPr = self.r(g.o_gPr, (B, NR, 2)); gattn = self.r(g.o_gattn, (B, NR)); Hr = self.r(g.o_Hr, (B, NR, NB))
        gPrT = self.r(g.o_gPrT, (B, NR, 2)); gRa = self.r(g.o_gRa, (B, NR, NB)); gHr = self.r(g.o_gHr, (B, NR, NB))
        attn_on = bool(g.flags & _lib.FLAG_TRANS_ATTN)
        for j in range(NR):
            W2 = self.P[g.p_W2_0 + j * g.p_W2_stride:][:2 * NB].reshape(2, NB)
            gp = gPr[:, j, :].copy(); w1 = np.ones(B, self.dtype)
            if attn_on:
                dot = (R[:, j, :] * gVt).sum(1) + gattn[:, j]
                p, lp, H = self.soft2(Pr[:, j, :])
                gp += dot[:, None] * p * (lp + H[:, None])
                w1 = 1 + (1 - H)
            gPrT[:, j, :] = gp
            gRa[:, j, :] = w1[:, None] * gVt
            gHr[:, j, :] = np.where(Hr[:, j, :] > 0, gp @ W2, 0)

    def run_heads(self):
        """Specification of the fused heads kernel (ta3n_heads.hip): the unfused phases it replaces,
        in dependency order, plus the per-workgroup partial sums it hands to the next GEMM level."""
        g, h = self.g, self.hy
        B, T, NB, F, Cn = g.B, g.T, g.NB, g.F, g.C
        P = self.P

        def lin(xv, w_off, b_off, out, inp):
            return xv @ P[w_off:w_off + out * inp].reshape(out, inp).T + P[b_off:b_off + out]
        Hf = self.r(g.o_Hf, (B * T, F))
        self.r(g.o_Pf, (B * T, 2))[:] = lin(Hf, g.p_Wcd, g.p_bcd, 2, F)
        self.run_pool_fwd()
        Vd = self.r(g.o_Vd, (B, NB))
        self.r(g.o_Y, (B, Cn))[:] = lin(Vd, g.p_Wcv, g.p_bcv, Cn, NB)
        Hv = np.maximum(lin(Vd, g.p_Wdv, g.p_bdv, NB, NB), 0)
        self.r(g.o_Hv, (B, NB))[:] = Hv
        self.r(g.o_Pv, (B, 2))[:] = lin(Hv, g.p_Wcdv, g.p_bcdv, 2, NB)
        self.run_loss()
        gPv = self.r(g.o_gPv, (B, 2)); gY = self.r(g.o_gY, (B, Cn)); gPf = self.r(g.o_gPf, (B * T, 2))
        Wcdv = P[g.p_Wcdv:g.p_Wcdv + 2 * NB].reshape(2, NB); Wdv = P[g.p_Wdv:g.p_Wdv + NB * NB].reshape(NB, NB)
        Wcv = P[g.p_Wcv:g.p_Wcv + Cn * NB].reshape(Cn, NB); Wcd = P[g.p_Wcd:g.p_Wcd + 2 * F].reshape(2, F)
        gHv = np.where(Hv > 0, gPv @ Wcdv, 0)
        self.r(g.o_gHv, (B, NB))[:] = gHv
        gVt = -h["beta"][1] * (gHv @ Wdv) + gY @ Wcv
        if h["train"] and h["p_drop_v"] > 0:
            idx = np.arange(B)[:, None] * NB + np.arange(NB)[None, :]
            gVt = gVt * keep_mask(h["seed_v"], idx, h["p_drop_v"])
        self.r(g.o_gVt, (B, NB))[:] = gVt * self.scale(5)
        self.r(g.o_gattn, (B, g.n_rel))[:] = 0          # the fused step does not consume an upstream attention gradient
        self.run_pool_bwd()
        self.r(g.o_gHf, (B * T, F))[:] = np.where(Hf > 0, gPf @ Wcd, 0)
        lp = self.r(g.o_loss_part, (g.n_vid_wg + g.n_frm_wg, 8))
        lp[:] = 0
        lp[0, :6] = self.ws[g.o_losses:g.o_losses + 6]   # any split over workgroups sums to the same scalars
        self.ws[g.o_losses:g.o_losses + 8] = 0
        part = self.r(g.o_fh_part, (g.n_frm_wg, 2 * F)); bpart = self.r(g.o_fh_bpart, (g.n_frm_wg, 2))
        for w in range(g.n_frm_wg):
            rows = slice(w * g.heads_rpw, min((w + 1) * g.heads_rpw, B * T))
            part[w] = (gPf[rows].T @ Hf[rows]).reshape(-1)
            bpart[w] = gPf[rows].sum(0)

    def run_sgd(self, fused_norm=False):
        """fused_norm: global norm from the per-tile partials the fused step's gradient tiles left in ws["sumsq"]
        (ta3n_sgd_step_fused) instead of a pass over the gradient buffer."""
        g, h = self.g, self.hy
        n = g.live_floats
        total = np.sqrt((self.G[:n] ** 2).sum())
        if fused_norm:
            total = np.sqrt(self.ws[g.o_sumsq:g.o_sumsq + g.n_sumsq].sum())
        coef = min(h["clip"] / (total + 1e-6), 1.0) if h["clip"] > 0 else 1.0
        self.ws[g.o_grad_norm] = total; self.ws[g.o_grad_norm + 1] = coef
        d = self.G[:n] * coef + h["weight_decay"] * self.P[:n]
        self.M[:n] = h["momentum"] * self.M[:n] + d
        d = d + h["momentum"] * self.M[:n]
        self.P[:n] -= h["lr"] * d

    def run_group(self, group, fused_norm=False):
        for ph in self.phases:
            if ph.group != group:
                continue
            if ph.kind == PH_GEMM: self.run_gemm(ph)
            elif ph.kind == PH_POOL_FWD: self.run_pool_fwd()
            elif ph.kind == PH_LOSS: self.run_loss()
            elif ph.kind == PH_POOL_BWD: self.run_pool_bwd()
            elif ph.kind == PH_HEADS: self.run_heads()
            elif ph.kind == PH_POOL_CLS: self.run_pool_cls()
            elif ph.kind == PH_POOL_AVG_FWD: self.run_pool_avg_fwd()
            elif ph.kind == PH_POOL_AVG_BWD: self.run_pool_avg_bwd()
            elif ph.kind == PH_BN_FWD: self.run_bn_fwd()
            elif ph.kind == PH_BN_BWD: self.run_bn_bwd()
            elif ph.kind == PH_GRAD_NORM: pass
            elif ph.kind == PH_SGD: self.run_sgd(fused_norm)
