// The Sinkhorn matcher's inner operation on the GPU (outer loop, row f2; EPSM/utils/matcher.py:51-63 calls
// geomloss.SamplesLoss("sinkhorn", p = 2, blur = 0.01, scaling = 0.9), whose "online" backend never forms the cost matrix):
//
//     out_i = -eps * log sum_j exp( h_j - |x_i - y_j|^2 / (2 eps) )              ("softmin" of the dual update)
//     w_i   = sum_j p_ij y_j,   p_ij = softmax_j( h_j - |x_i - y_j|^2 / (2 eps) )  (optional: d out_i / d x_i = x_i - w_i)
//
// for two point clouds of n and m points in D <= 8 dimensions (the matcher's are 5-D: r, g, b, x, y).  The plain-torch
// restatement (epsm_mitsuba3_amd/matcher.py) forms four n x m matrices -- 17 GB each at the 256 x 256 matching resolution
// of exp/human.py and exp/glassslab.py, ~200 passes over them per call, 6.7 s -- this streams y through LDS and keeps a
// running (max, sum) per row: nothing but x, y, h is read, 4.3e9 pair evaluations per call at that size.
//
// Decomposition: a workgroup owns 256 rows (one per lane) and ONE of S column ranges, so that small clouds fill the chip
// too; the S partial (max, sum, weighted sum) triples of a row are merged by a second small kernel.  The column tile of
// 256 points sits in LDS as 8 floats per point (coordinates, then h in the last word): every lane reads the SAME point at
// the same time -- a broadcast, two ds_read_b128 per pair -- and does ~20 VALU operations on it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/epsm.h"
#include "epsm_common.h"

using epsm_host::fail;

namespace {

constexpr int kMaxD = 7;                 // coordinates per point (word 7 of a staged point is h)
constexpr float kLow = -1e30f;           // "minus infinity" that survives subtraction

// Arithmetic: with s = log2(e) / eps the exponent h_j - |x_i - y_j|^2 / (2 eps), in base 2, is
//     (h_j log2 e - |y_j|^2 s / 2)  +  (s x_i) . y_j  -  |x_i|^2 s / 2
// -- a per-column constant (computed when the tile is staged), D multiply-adds per pair, and a per-row constant that
// stays out of the running sums.  Eight columns at a time: one rescaling of the sums by exp2(old max - new max) per
// group instead of one per pair (v_exp_f32 costs a wave two issue slots).  Per pair D + ~4 simple operations and 1 1/8
// exponentials (first version: 2 D + 8 and 2; 0.46 -> 0.385 s per matcher call at 256 x 256).
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;

// A staged point is read by every lane at the same time, and a broadcast ds_read_b128 still occupies the LDS for the
// full 64 x 16 bytes: with one row per lane the kernel was bound by exactly that (two reads per pair = 16 LDS clocks per
// wave and pair: 0.36 s of the 0.385 s per matcher call at 256 x 256).  kRows rows per lane share each read: 0.23 s; with
// the chunk loop unrolled twice 0.214 s.  What bounds it now is the vector ALU: per 64 pairs the loop issues 160
// v_pk_fma_f32 (4 clocks each), 73 v_exp_f32 (8), ~290 subtractions / additions / maxima / moves (2): ~1 800 SIMD clocks,
// i.e. 0.77 ms of the 1.0 ms one softmin of 65 536^2 pairs takes (eight rows per lane, scalar instead of packed
// multiply-adds: no change).
constexpr int kRows = 4;
typedef float F2 __attribute__((ext_vector_type(2)));

template <int D, bool WSUM>
__global__ __launch_bounds__(256) void softmin_partial_kernel(int64_t n, int64_t m, const float *x, const float *y, const float *h,
                                                              const float *dual, float logw, float inv_eps,
                                                              float s, int64_t cols_per_split, float *part) {
    __shared__ float s_y[256 * 8];
    const int64_t i0 = (int64_t) blockIdx.x * (256 * kRows) + threadIdx.x;       // rows i0 + 256 r
    const int64_t j0 = (int64_t) blockIdx.y * cols_per_split, j1 = j0 + cols_per_split < m ? j0 + cols_per_split : m;
    float xs[kRows][D], mx[kRows], sum[kRows], w[kRows][D];
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int64_t i = i0 + 256 * r;
#pragma unroll
        for (int k = 0; k < D; ++k) { xs[r][k] = (i < n ? x[i * D + k] : 0.f) * s; w[r][k] = 0.f; }
        mx[r] = kLow; sum[r] = 0.f;
    }
    for (int64_t t0 = j0; t0 < j1; t0 += 256) {
        __syncthreads();
        {
            const int64_t j = t0 + threadIdx.x;
            float *dst = s_y + threadIdx.x * 8;
            float yy = 0.f;
#pragma unroll
            for (int k = 0; k < D; ++k) { const float v = j < j1 ? y[j * D + k] : 0.f; dst[k] = v; yy = fmaf(v, v, yy); }
            // h given, or h = log-weight + dual / eps of a Sinkhorn update (epsm_sinkhorn_update)
            const float hj = j < j1 ? (h ? h[j] : (dual ? fmaf(dual[j], inv_eps, logw) : logw)) : 0.f;
            dst[7] = j < j1 ? fmaf(hj, kLog2e, -0.5f * s * yy) : kLow;
        }
        __syncthreads();
#pragma unroll 2
        for (int jj = 0; jj < 256; jj += 8) {
            float v[kRows][8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float *p = s_y + (jj + c) * 8;
                float pk[D];
#pragma unroll
                for (int k = 0; k < D; ++k) pk[k] = p[k];
                const float ph = p[7];
#ifdef EPSM_AB_MATCHER_SCALAR_FMA
#pragma unroll
                for (int r = 0; r < kRows; ++r) {
                    float a = ph;
#pragma unroll
                    for (int k = 0; k < D; ++k) a = fmaf(xs[r][k], pk[k], a);
                    v[r][c] = a;
                }
#else
                // two rows per v_pk_fma_f32 (the point's coordinate goes to both halves through op_sel)
#pragma unroll
                for (int r = 0; r < kRows; r += 2) {
                    F2 a = {ph, ph};
#pragma unroll
                    for (int k = 0; k < D; ++k) a = __builtin_elementwise_fma(F2{xs[r][k], xs[r + 1][k]}, F2{pk[k], pk[k]}, a);
                    v[r][c] = a.x; v[r + 1][c] = a.y;
                }
#endif
            }
            float sc[kRows];
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
                const float cm = fmaxf(fmaxf(fmaxf(v[r][0], v[r][1]), fmaxf(v[r][2], v[r][3])), fmaxf(fmaxf(v[r][4], v[r][5]), fmaxf(v[r][6], v[r][7])));
                const float mn = fmaxf(mx[r], cm);
                sc[r] = __builtin_amdgcn_exp2f(mx[r] - mn);
                mx[r] = mn;
                sum[r] *= sc[r];
#pragma unroll
                for (int c = 0; c < 8; ++c) { v[r][c] = __builtin_amdgcn_exp2f(v[r][c] - mn); sum[r] += v[r][c]; }
            }
            if (WSUM) {
#pragma unroll
                for (int r = 0; r < kRows; ++r) {
#pragma unroll
                    for (int k = 0; k < D; ++k) w[r][k] *= sc[r];
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float *p = s_y + (jj + c) * 8;
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        const float pk = p[k];
#pragma unroll
                        for (int r = 0; r < kRows; ++r) w[r][k] = fmaf(v[r][c], pk, w[r][k]);
                    }
                }
            }
        }
    }
    constexpr int kStride = WSUM ? 2 + D : 2;
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int64_t i = i0 + 256 * r;
        if (i < n) {
            float *o = part + ((int64_t) blockIdx.y * n + i) * kStride;
            o[0] = mx[r]; o[1] = sum[r];
            if (WSUM) {
#pragma unroll
                for (int k = 0; k < D; ++k) o[2 + k] = w[r][k];
            }
        }
    }
}

template <int D, bool WSUM>
__global__ __launch_bounds__(256) void softmin_merge_kernel(int64_t n, int splits, const float *part, const float *x, float s, float eps,
                                                            const float *prev, float *out, float *wsum) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    constexpr int kStride = WSUM ? 2 + D : 2;
    float M = kLow;
    for (int q = 0; q < splits; ++q) M = fmaxf(M, part[((int64_t) q * n + i) * kStride]);
    float S = 0.f, W[D];
#pragma unroll
    for (int k = 0; k < D; ++k) W[k] = 0.f;
    for (int q = 0; q < splits; ++q) {
        const float *p = part + ((int64_t) q * n + i) * kStride;
        const float a = __builtin_amdgcn_exp2f(p[0] - M);
        S = fmaf(p[1], a, S);
        if (WSUM) {
#pragma unroll
            for (int k = 0; k < D; ++k) W[k] = fmaf(p[2 + k], a, W[k]);
        }
    }
    float xx = 0.f;
#pragma unroll
    for (int k = 0; k < D; ++k) { const float v = x[i * D + k]; xx = fmaf(v, v, xx); }
    const float val = -eps * kLn2 * (M - 0.5f * s * xx + __builtin_amdgcn_logf(S));       // v_log_f32 is log2
    out[i] = prev ? 0.5f * (prev[i] + val) : val;                                           // the symmetric update averages
    if (WSUM) {
        const float r = 1.f / S;
#pragma unroll
        for (int k = 0; k < D; ++k) wsum[i * D + k] = W[k] * r;
    }
}

template <int D, bool WSUM>
hipError_t run(int64_t n, int64_t m, const float *x, const float *y, const float *h, const float *dual, float logw, const float *prev,
               float eps, float *out, float *wsum, float *scratch, int splits, hipStream_t s) {
    const int64_t row_blocks = (n + 255) / 256, row_groups = (n + 256 * kRows - 1) / (256 * kRows);
    const int64_t cols = ((m + splits - 1) / splits + 255) / 256 * 256;
    hipLaunchKernelGGL((softmin_partial_kernel<D, WSUM>), dim3((unsigned) row_groups, (unsigned) splits), dim3(256), 0, s,
                       n, m, x, y, h, dual, logw, 1.f / eps, kLog2e / eps, cols, scratch);
    hipLaunchKernelGGL((softmin_merge_kernel<D, WSUM>), dim3((unsigned) row_blocks), dim3(256), 0, s, n, splits, scratch, x, kLog2e / eps, eps, prev, out, wsum);
    return hipGetLastError();
}

}  // namespace

extern "C" int epsm_sinkhorn_splits(int64_t n, int64_t m) {
    // enough workgroups for four per CU, no column range shorter than one tile
    const int64_t row_blocks = (n + 256 * kRows - 1) / (256 * kRows), tiles = (m + 255) / 256;
    int64_t s = (1024 + row_blocks - 1) / (row_blocks > 0 ? row_blocks : 1);
    if (s > tiles) s = tiles;
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    return (int) s;
}

extern "C" size_t epsm_sinkhorn_scratch_bytes(int64_t n, int64_t m, int D) {
    return (size_t) epsm_sinkhorn_splits(n, m) * (size_t) n * (size_t) (2 + D) * sizeof(float);
}

static int softmin_entry(const char *who, int64_t n, int64_t m, int D, const float *x, const float *y, const float *h, const float *dual,
                         float logw, const float *prev, float eps, float *out, float *wsum, void *scratch, size_t scratch_bytes, void *stream) {
    epsm_host::err_buf()[0] = 0;
    char msg[160];
    auto bad = [&](const char *what) { snprintf(msg, sizeof(msg), "%s: %s", who, what); return fail(EPSM_EINVAL, msg); };
    if (n < 0 || m < 0 || D < 1 || D > kMaxD) return bad("need n, m >= 0 and 1 <= D <= 7");
    if (n == 0) return EPSM_OK;
    if (m == 0) return bad("empty second cloud");
    if (!x || !y || !out || !scratch) return bad("NULL argument");
    if (!(eps > 0.f)) return bad("eps must be positive");
    if (scratch_bytes < epsm_sinkhorn_scratch_bytes(n, m, D)) return bad("scratch too small (epsm_sinkhorn_scratch_bytes)");
    const int splits = epsm_sinkhorn_splits(n, m);
    hipStream_t s = (hipStream_t) stream;
    float *sc = (float *) scratch;
    hipError_t e = hipSuccess;
#define EPSM_SOFTMIN_CASE(DD) case DD: e = wsum ? run<DD, true>(n, m, x, y, h, dual, logw, prev, eps, out, wsum, sc, splits, s) \
                                                : run<DD, false>(n, m, x, y, h, dual, logw, prev, eps, out, nullptr, sc, splits, s); break;
    switch (D) {
        EPSM_SOFTMIN_CASE(1) EPSM_SOFTMIN_CASE(2) EPSM_SOFTMIN_CASE(3) EPSM_SOFTMIN_CASE(4)
        EPSM_SOFTMIN_CASE(5) EPSM_SOFTMIN_CASE(6) EPSM_SOFTMIN_CASE(7)
    }
#undef EPSM_SOFTMIN_CASE
    if (e != hipSuccess) return epsm_host::hip_fail(who, e);
    return EPSM_OK;
}

extern "C" int epsm_sinkhorn_softmin(int64_t n, int64_t m, int D, const float *x, const float *y, const float *h, float eps,
                                     float *out, float *wsum, void *scratch, size_t scratch_bytes, void *stream) {
    if (!h && n > 0 && m > 0) { epsm_host::err_buf()[0] = 0; return fail(EPSM_EINVAL, "epsm_sinkhorn_softmin: NULL argument"); }
    return softmin_entry("epsm_sinkhorn_softmin", n, m, D, x, y, h, nullptr, 0.f, nullptr, eps, out, wsum, scratch, scratch_bytes, stream);
}

extern "C" int epsm_sinkhorn_update(int64_t n, int64_t m, int D, const float *x, const float *y, const float *dual, float log_weight,
                                    float eps, const float *prev, float *out, float *wsum, void *scratch, size_t scratch_bytes, void *stream) {
    return softmin_entry("epsm_sinkhorn_update", n, m, D, x, y, nullptr, dual, log_weight, prev, eps, out, wsum, scratch, scratch_bytes, stream);
}
