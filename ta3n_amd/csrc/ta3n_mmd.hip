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

}  // namespace

extern "C" {

int64_t ta3n_gaussian_kernel_scratch_floats(int n) { return n > 0 ? (int64_t)((n + TL - 1) / TL) * ((n + TL - 1) / TL) : 0; }

int ta3n_gaussian_kernel(const float *total, int n, int d, float kernel_mul, int kernel_num, float fix_sigma, float *k_out, float *kp_out,
                         float *scratch, void *stream) {
    if (!total || !k_out || !scratch) return fail(TA3N_ERR_INVALID, "null argument");
    if (n < 2 || d < 1 || kernel_num < 1 || kernel_num > 16 || !(kernel_mul > 0.f)) return fail(TA3N_ERR_INVALID, "bad kernel arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nt = (n + TL - 1) / TL;
    hipLaunchKernelGGL(mmd_l2_kernel, dim3(nt, nt), dim3(256), 0, s, total, n, d, k_out, scratch);
    const int blocks = (int)std::min<int64_t>(((int64_t)n * n + 255) / 256, 1024);
    hipLaunchKernelGGL(mmd_kernel_kernel, dim3(blocks), dim3(256), 0, s, k_out, kp_out, scratch, nt * nt, n, kernel_mul, kernel_num, fix_sigma);
    return hipGetLastError() == hipSuccess ? TA3N_OK : fail(TA3N_ERR_HIP, std::string("kernel-matrix launch failed: ") + hipGetErrorString(hipGetLastError()));
}

int ta3n_mmd_rowdiff(const float *c, const float *total, int n, int d, float scale, float *out, void *stream) {
    if (!c || !total || !out) return fail(TA3N_ERR_INVALID, "null argument");
    if (n < 1 || d < 1 || n > 12288) return fail(TA3N_ERR_INVALID, "bad sizes (n <= 12288: one row of coefficients is staged in LDS)");
    hipLaunchKernelGGL(mmd_rowdiff_kernel, dim3(n), dim3(256), (size_t)n * sizeof(float), static_cast<hipStream_t>(stream), c, total, n, d, scale, out);
    return hipGetLastError() == hipSuccess ? TA3N_OK : fail(TA3N_ERR_HIP, std::string("rowdiff launch failed: ") + hipGetErrorString(hipGetLastError()));
}

}  // extern "C"
