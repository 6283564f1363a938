// Discrepancy losses of the paper's DA baselines (dis_DA DAN / JAN, reference loss.py:46-120, called from main.py:452-505): the
// O(n^2 d) parts as HIP kernels - the pairwise squared distances of the stacked [source; target] features with the
// data-dependent bandwidth and the multi-bandwidth RBF sum (guassian_kernel, loss.py:46-59), and the contraction that carries a
// gradient at the kernel matrix back to the features.  fp32 in the reference's explicit-difference form ((x - y)^2 summed over the
// feature dimension - not |x|^2 + |y|^2 - 2 x.y, which cancels catastrophically for near-duplicate rows).  Sums that cross
// workgroups are per-tile partials added in a fixed order: bitwise reproducible.
//   K[p][q]  = sum_i exp(-L2[p][q] / bw_i),   bw_i = bw0 * mul^i,   bw0 = (fix_sigma or sum(L2) / (n^2 - n)) / mul^(num / 2)
//   Kp[p][q] = dK / dL2 = sum_i -exp(-L2 / bw_i) / bw_i          (the bandwidth carries no gradient: loss.py:55 uses .data)
//   rowdiff:  out[p][:] = sum_q C[p][q] (t[p][:] - t[q][:])      (gradient at t for C = 2 Kp (gK + gK^T))
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/ta3n_hip.h"
#include "ta3n_kernels.h"
#include "ta3n_plan.h"

using namespace ta3n;

namespace {

constexpr int TL = 16;      // L2 tile: 16 x 16 pairs per workgroup (256 threads), feature dimension in chunks of 64

int fail(int code, const std::string &msg) {
    ta3n::set_error(msg);
    return code;
}

__global__ __launch_bounds__(256) void mmd_l2_kernel(const float *__restrict__ t, int n, int d, float *__restrict__ l2, float *__restrict__ part) {
    __shared__ float a[TL][65], b[TL][65];
    __shared__ float red[4];
    const int ti = threadIdx.x / TL, tj = threadIdx.x % TL;
    const int p0 = blockIdx.y * TL, q0 = blockIdx.x * TL;
    float acc = 0.f;
    for (int k0 = 0; k0 < d; k0 += 64) {
        for (int e = threadIdx.x; e < TL * 64; e += 256) {
            const int r = e / 64, k = e % 64;
            a[r][k] = (p0 + r < n && k0 + k < d) ? t[(size_t)(p0 + r) * d + k0 + k] : 0.f;
            b[r][k] = (q0 + r < n && k0 + k < d) ? t[(size_t)(q0 + r) * d + k0 + k] : 0.f;
        }
        __syncthreads();
#pragma unroll 16
        for (int k = 0; k < 64; ++k) {
            const float df = a[ti][k] - b[tj][k];
            acc = fmaf(df, df, acc);
        }
        __syncthreads();
    }
    const bool ok = p0 + ti < n && q0 + tj < n;
    if (ok) l2[(size_t)(p0 + ti) * n + q0 + tj] = acc;
    float s = wave_allreduce_sum(ok ? acc : 0.f);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void mmd_kernel_kernel(float *__restrict__ k_io, float *__restrict__ kp, const float *__restrict__ part, int n_part,
                                                         int n, float mul, int num, float fix_sigma) {
    __shared__ float red[4];
    float bw;
    if (fix_sigma > 0.f) {
        bw = fix_sigma;
    } else {                                   // every workgroup adds the partials up in the same order
        float acc = 0.f;
        for (int i = threadIdx.x; i < n_part; i += 256) acc += part[i];
        acc = wave_allreduce_sum(acc);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        bw = ((red[0] + red[1]) + (red[2] + red[3])) / ((float)n * (float)n - (float)n);
    }
    for (int i = 0; i < num / 2; ++i) bw /= mul;     // loss.py:57: bandwidth /= kernel_mul ** (kernel_num // 2)
    const size_t total = (size_t)n * n;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const float l2 = k_io[e];
        float k = 0.f, dk = 0.f, b = bw;
        for (int i = 0; i < num; ++i) {
            const float ex = expf(-l2 / b);
            k += ex;
            dk -= ex / b;
            b *= mul;
        }
        k_io[e] = k;
        if (kp) kp[e] = dk;
    }
}

__global__ __launch_bounds__(256) void mmd_rowdiff_kernel(const float *__restrict__ c, const float *__restrict__ t, int n, int d, float scale,
                                                          float *__restrict__ out) {
    extern __shared__ float crow[];            // [n] coefficients of this row
    const int p = blockIdx.x;
    for (int q = threadIdx.x; q < n; q += 256) { crow[q] = c[(size_t)p * n + q]; }
    __syncthreads();
    for (int k = threadIdx.x; k < d; k += 256) {
        const float tp = t[(size_t)p * d + k];
        float acc = 0.f;                                   // the differences themselves are summed (not cs t_p - sum c_q t_q): identical
        for (int q = 0; q < n; ++q) acc = fmaf(crow[q], tp - t[(size_t)q * d + k], acc);   // rows contribute exactly nothing
        out[(size_t)p * d + k] = scale * acc;
    }
}

void launch_gaussian(const float *total, int n, int d, float mul, int num, float fix_sigma, float *k_out, float *kp_out, float *scratch, hipStream_t s) {
    const int nt = (n + TL - 1) / TL;
    hipLaunchKernelGGL(mmd_l2_kernel, dim3(nt, nt), dim3(256), 0, s, total, n, d, k_out, scratch);
    const int blocks = (int)std::min<int64_t>(((int64_t)n * n + 255) / 256, 1024);
    hipLaunchKernelGGL(mmd_kernel_kernel, dim3(blocks), dim3(256), 0, s, k_out, kp_out, scratch, nt * nt, n, mul, num, fix_sigma);
}

// ---- the whole discrepancy term of one rank's step without a framework in between (ta3n_discrepancy) ----
// total[p][:] = feature row of source video src0 + p (p < half) or target video tgt0 + p - half: the stacked [source; target] rows of loss.py:47
__global__ __launch_bounds__(256) void mmd_stack_kernel(const float *__restrict__ feat, int ld, int src0, int tgt0, int half, int d, float *__restrict__ total) {
    const int p = blockIdx.x;
    const float *__restrict__ src = feat + (size_t)(p < half ? src0 + p : tgt0 + p - half) * ld;
    for (int k = threadIdx.x; k < d; k += 256) total[(size_t)p * d + k] = src[k];
}
// The four-quadrant mean of loss.py:70-75 / :113-118 is  sum_pq s_p s_q J[p][q] / half^2  with s = +1 on source rows, -1 on target rows and
// J = K (mmd_rbf) or the product of the layers' kernels (JAN); its gradient at K_self is  s_p s_q K_other / half^2  (symmetric), which
// _GaussianKernelHip.backward turns into the row-difference coefficients  c = 2 Kp (gK + gK^T) = coef s_p s_q Kp_self K_other.  In place:
// kp_self becomes c.  part[block] (may be null) = this block's share of  sum s_p s_q J, added up in a fixed order by mmd_loss_add_kernel.
__global__ __launch_bounds__(256) void mmd_coef_kernel(const float *__restrict__ k_self, float *__restrict__ kp_self, const float *__restrict__ k_other,
                                                       int n, int half, float coef, float *__restrict__ part) {
    __shared__ float red[4];
    const size_t total = (size_t)n * n;
    float acc = 0.f;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int p = (int)(e / (size_t)n), q = (int)(e - (size_t)p * n);
        const float sgn = ((p < half) == (q < half)) ? 1.f : -1.f;
        const float other = k_other ? k_other[e] : 1.f;
        acc += sgn * (k_self[e] * other);
        kp_self[e] = coef * sgn * (kp_self[e] * other);
    }
    if (part) {
        acc = wave_allreduce_sum(acc);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}
__global__ __launch_bounds__(64) void mmd_loss_add_kernel(const float *__restrict__ part, int n_part, float scale, float *__restrict__ loss) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_part; i += 64) acc += part[i];
    acc = wave_allreduce_sum(acc);
    if (threadIdx.x == 0) *loss += scale * acc;
}
// dst[video row][:d] += grad[p][:]  (the gradient rows of the stacked features back to the videos they came from)
__global__ __launch_bounds__(256) void mmd_scatter_add_kernel(const float *__restrict__ grad, int d, int half, int src0, int tgt0, float *__restrict__ dst, int ld) {
    const int p = blockIdx.x;
    float *__restrict__ out = dst + (size_t)(p < half ? src0 + p : tgt0 + p - half) * ld;
    for (int k = threadIdx.x; k < d; k += 256) out[k] += grad[(size_t)p * d + k];
}

}  // namespace

extern "C" {

int64_t ta3n_gaussian_kernel_scratch_floats(int n) { return n > 0 ? (int64_t)((n + TL - 1) / TL) * ((n + TL - 1) / TL) : 0; }

int ta3n_gaussian_kernel(const float *total, int n, int d, float kernel_mul, int kernel_num, float fix_sigma, float *k_out, float *kp_out,
                         float *scratch, void *stream) {
    if (!total || !k_out || !scratch) return fail(TA3N_ERR_INVALID, "null argument");
    if (n < 2 || d < 1 || kernel_num < 1 || kernel_num > 16 || !(kernel_mul > 0.f)) return fail(TA3N_ERR_INVALID, "bad kernel arguments");
    launch_gaussian(total, n, d, kernel_mul, kernel_num, fix_sigma, k_out, kp_out, scratch, static_cast<hipStream_t>(stream));
    return hipGetLastError() == hipSuccess ? TA3N_OK : fail(TA3N_ERR_HIP, std::string("kernel-matrix launch failed: ") + hipGetErrorString(hipGetLastError()));
}

int ta3n_mmd_rowdiff(const float *c, const float *total, int n, int d, float scale, float *out, void *stream) {
    if (!c || !total || !out) return fail(TA3N_ERR_INVALID, "null argument");
    if (n < 1 || d < 1 || n > 12288) return fail(TA3N_ERR_INVALID, "bad sizes (n <= 12288: one row of coefficients is staged in LDS)");
    hipLaunchKernelGGL(mmd_rowdiff_kernel, dim3(n), dim3(256), (size_t)n * sizeof(float), static_cast<hipStream_t>(stream), c, total, n, d, scale, out);
    return hipGetLastError() == hipSuccess ? TA3N_OK : fail(TA3N_ERR_HIP, std::string("rowdiff launch failed: ") + hipGetErrorString(hipGetLastError()));
}

// Scratch of ta3n_discrepancy for batches of up to n_max stacked rows: two feature stacks + their gradients, two kernel matrices with
// derivatives, the distance partials of ta3n_gaussian_kernel and the loss partials.
static int64_t discrepancy_floats(int n, int c, int f) {
    const int64_t nn = (int64_t)n * n;
    return 2 * (int64_t)n * (c + f) + 4 * nn + ta3n_gaussian_kernel_scratch_floats(n) + 1024 + 64;
}
int64_t ta3n_discrepancy_scratch_floats(int batch_source, int batch_target, int num_class, int feat_dim) {
    const int half = std::min(batch_source, batch_target);
    return half > 0 ? discrepancy_floats(2 * half, num_class, feat_dim) : 64;
}

int ta3n_discrepancy(float *ws, int64_t o_logits, int num_class, int64_t o_feature, int feat_dim, int64_t o_grad_logits, int64_t o_grad_feature,
                     int batch_source, int batch_target, int valid_source, int valid_target, int kind, int place_logits, int place_feature,
                     float alpha, float *scratch, int64_t scratch_floats, float *loss_out, void *stream) {
    if (!ws || !scratch || !loss_out) return fail(TA3N_ERR_INVALID, "null argument");
    if (kind != 1 && kind != 2) return fail(TA3N_ERR_INVALID, "kind: 1 = DAN, 2 = JAN");
    if (num_class < 1 || feat_dim < 1 || batch_source < 0 || batch_target < 0 || valid_source < 0 || valid_source > batch_source ||
        valid_target < 0 || valid_target > batch_target)
        return fail(TA3N_ERR_INVALID, "bad sizes");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int B = batch_source + batch_target;
    if (hipMemsetAsync(loss_out, 0, sizeof(float), s) != hipSuccess ||
        hipMemsetAsync(ws + o_grad_feature, 0, (size_t)B * feat_dim * sizeof(float), s) != hipSuccess)
        return fail(TA3N_ERR_HIP, "memset failed");
    const int size = std::min(valid_source, valid_target);        // main.py:467, 482: the first min(#source, #target) videos of each domain
    if (size == 0) return TA3N_OK;
    // DAN (main.py:459-476): mmd_rbf per selected feature, in chunks of <= 256 videos whose losses are averaged; JAN (main.py:478-505): ONE joint
    // kernel of both features over all `size` videos
    const int half = kind == 1 ? std::min(256, size) : size;
    if (size % half != 0) return fail(TA3N_ERR_INVALID, "DAN: min(valid source, valid target) above 256 must be a multiple of 256 (main.py:463-466 reshapes)");
    const int chunks = size / half, n = 2 * half;
    if (n > 12288) return fail(TA3N_ERR_INVALID, "more than 6 144 videos per domain in one kernel matrix");
    if (scratch_floats < discrepancy_floats(n, num_class, feat_dim)) return fail(TA3N_ERR_INVALID, "scratch too small (ta3n_discrepancy_scratch_floats)");
    const int64_t nn = (int64_t)n * n;
    const int dims[2] = {num_class, feat_dim};
    const int64_t feat_off[2] = {o_logits, o_feature}, grad_off[2] = {o_grad_logits, o_grad_feature};
    const float muls[2] = {2.f, 2.f};
    const int nums[2] = {2, 5};                                   // main.py:456-457: kernel_muls = [2.0] * 2, kernel_nums = [2, 5]
    float *total[2], *grad[2], *K[2], *Kp[2];
    float *q = scratch;
    for (int l = 0; l < 2; ++l) { total[l] = q; q += (int64_t)n * dims[l]; grad[l] = q; q += (int64_t)n * dims[l]; }
    for (int l = 0; l < 2; ++l) { K[l] = q; q += nn; Kp[l] = q; q += nn; }
    float *gscr = q; q += ta3n_gaussian_kernel_scratch_floats(n);
    float *part = q;
    const int cblocks = (int)std::min<int64_t>((nn + 255) / 256, 1024);
    const bool use[2] = {kind == 2 || place_logits != 0, kind == 2 || place_feature != 0};
    for (int t = 0; t < chunks; ++t) {
        const int src0 = t * half, tgt0 = batch_source + t * half;
        for (int l = 0; l < 2; ++l) {
            if (!use[l]) continue;
            hipLaunchKernelGGL(mmd_stack_kernel, dim3(n), dim3(256), 0, s, ws + feat_off[l], dims[l], src0, tgt0, half, dims[l], total[l]);
            launch_gaussian(total[l], n, dims[l], muls[l], nums[l], 0.f, K[l], Kp[l], gscr, s);
        }
        const float inv = 1.f / ((float)half * (float)half);
        if (kind == 1) {
            for (int l = 0; l < 2; ++l) {
                if (!use[l]) continue;
                hipLaunchKernelGGL(mmd_coef_kernel, dim3(cblocks), dim3(256), 0, s, K[l], Kp[l], (const float *)nullptr, n, half, 4.f * alpha * inv / (float)chunks, part);
                hipLaunchKernelGGL(mmd_loss_add_kernel, dim3(1), dim3(64), 0, s, part, cblocks, inv / (float)chunks, loss_out);
            }
        } else {
            hipLaunchKernelGGL(mmd_coef_kernel, dim3(cblocks), dim3(256), 0, s, K[0], Kp[0], K[1], n, half, 4.f * alpha * inv, part);
            hipLaunchKernelGGL(mmd_loss_add_kernel, dim3(1), dim3(64), 0, s, part, cblocks, inv, loss_out);
            hipLaunchKernelGGL(mmd_coef_kernel, dim3(cblocks), dim3(256), 0, s, K[1], Kp[1], K[0], n, half, 4.f * alpha * inv, (float *)nullptr);
        }
        for (int l = 0; l < 2; ++l) {
            if (!use[l]) continue;
            hipLaunchKernelGGL(mmd_rowdiff_kernel, dim3(n), dim3(256), (size_t)n * sizeof(float), s, Kp[l], total[l], n, dims[l], 1.f, grad[l]);
            hipLaunchKernelGGL(mmd_scatter_add_kernel, dim3(n), dim3(256), 0, s, grad[l], dims[l], half, src0, tgt0, ws + grad_off[l], dims[l]);
        }
    }
    return hipGetLastError() == hipSuccess ? TA3N_OK : fail(TA3N_ERR_HIP, std::string("discrepancy launch failed: ") + hipGetErrorString(hipGetLastError()));
}

}  // extern "C"
