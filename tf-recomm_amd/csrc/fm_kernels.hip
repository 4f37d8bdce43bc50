// fm_kernels.hip - second-order factorisation-machine kernels on CSR rows (BASELINE config 5).
// Restates forward.py:21-22:   y(x) = mu + x.W + 0.5 * ( ||x V||^2 - sum_j x_j^2 ||V_j||^2 )
// on the design matrix fm.py:61-93 builds (one-hot user/item blocks + count-valued blocks).
// forward.py:22 writes the last term as x.dot(V**2) - without squaring x - which is the same
// thing on 0/1 features only; this kernel computes the general (libFM) form.
// One lane group (G lanes x VEC floats = one V row) per CSR row; the row's non-zeros are walked
// four at a time so four 4*D-byte row gathers are in flight per group.  HBM-bound gather.
//
// TRAIN mode (the reference trains in the external libFM binary, fm.py:154-155) also
// emits what the backward needs: s_r = x V per row, g_r = d loss / d y_r, and per non-zero
// (r, j, x) the coefficients a = g_r x and b = lam - g_r x^2, so that the feature-row gradient
//      dV_j = sum_r  a * s_r + b * V_j          dW_j = sum_r  a + lam * W_j
// is exactly the form the SVD segmented reduce (K3) computes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "svd_kernels.h"

namespace tfr {

// NTV: V rows by non-temporal loads - a V that outgrows the Infinity Cache streams past it and W, the indices and the values
// keep it (C5, one gpurun call: 387 -> 376 us; loading a row's indices / values / weights one per lane and handing
// them round by ds_bpermute instead: no change, and slower together with NTV - not kept).
template <int G, int VEC, bool TRAIN, bool NTV>
__global__ __launch_bounds__(256) void k_fm_forward(FmArgs a) {
    warm_args(a);
    constexpr int GPB = 256 / G;
    const int gl = threadIdx.x % G;
    const int d0 = gl * VEC;
    const int D = a.D;
    const float mu = *a.mu;
    bool oob = false;
    // full-width rows (block-uniform): unguarded 16-byte loads.  Behind a `d0 < D` guard each row load is merged with a
    // zero-initialised copy afterwards - a register move behind a FULL wait, so the four rows "in flight" arrived one after the
    // other (ISA: global_load_dwordx4 / s_waitcnt vmcnt(0) four times in a row)
    const bool full = D == G * VEC;
    float acc3[3] = {0.f, 0.f, 0.f};           // TRAIN: data loss, -, sum g
    for (int64_t row = (int64_t)blockIdx.x * GPB + threadIdx.x / G; row < a.n_rows;
         row += (int64_t)gridDim.x * GPB) {
        const int64_t lo = a.indptr[row], hi = a.indptr[row + 1];
        float s[VEC], q[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) { s[e] = 0.f; q[e] = 0.f; }
        float lin = 0.f;
        for (int64_t p = lo; p < hi; p += 4) {
            int32_t f[4];
            float x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {                // branch-free: an address that is there, the value dropped
                const bool ok = p + j < hi;
                const int64_t pj = ok ? p + j : hi - 1;
                const int32_t fj = a.indices[pj];
                const float xj = a.data[pj];
                f[j] = ok ? fj : 0;
                x[j] = ok ? xj : 0.f;
                if ((uint64_t)(int64_t)f[j] >= (uint64_t)a.F) { oob = true; f[j] = 0; x[j] = 0.f; }
            }
            float v[4][VEC];
            float w[4];
            if (full) {
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = a.W[f[j]];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float* vr = a.V + (size_t)f[j] * D + d0;
                    if constexpr (VEC == 4) {
                        typedef float f4v __attribute__((ext_vector_type(4)));
                        f4v t;
                        if constexpr (NTV) t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(vr));
                        else t = *reinterpret_cast<const f4v*>(vr);
                        v[j][0] = t.x; v[j][1] = t.y; v[j][2] = t.z; v[j][3] = t.w;
                    } else {
                        v[j][0] = NTV ? __builtin_nontemporal_load(vr) : *vr;
                    }
                }
            } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* vr = a.V + (size_t)f[j] * D;
                if constexpr (VEC == 4) {
                    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (d0 < D) {
                        if constexpr (NTV) {
                            const float* q4 = vr + d0;
                            t.x = __builtin_nontemporal_load(q4); t.y = __builtin_nontemporal_load(q4 + 1);
                            t.z = __builtin_nontemporal_load(q4 + 2); t.w = __builtin_nontemporal_load(q4 + 3);
                        } else {
                            t = *reinterpret_cast<const float4*>(vr + d0);
                        }
                    }
                    v[j][0] = t.x; v[j][1] = t.y; v[j][2] = t.z; v[j][3] = t.w;
                } else {
                    v[j][0] = (d0 < D) ? (NTV ? __builtin_nontemporal_load(vr + d0) : vr[d0]) : 0.f;
                }
                w[j] = a.W[f[j]];
            }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float xv = x[j] * v[j][e];
                    s[e] += xv;
                    q[e] = fmaf(xv, xv, q[e]);
                }
                lin = fmaf(x[j], w[j], lin);
            }
        }
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc += s[e] * s[e] - q[e];
#pragma unroll
        for (int o = G / 2; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
        const float yhat = (mu + lin) + 0.5f * acc;
        if (gl == 0 && a.out) a.out[row] = yhat;
        if constexpr (TRAIN) {
            const float yt = a.y[row];
            float g, l;
            if (a.loss == 0) { g = yhat - yt; l = 0.5f * g * g; }
            else {
                g = 1.f / (1.f + __expf(-yhat)) - yt;
                l = fmaxf(yhat, 0.f) - yhat * yt + log1pf(__expf(-fabsf(yhat)));
            }
            if (d0 < D) {
                if constexpr (VEC == 4) *reinterpret_cast<float4*>(a.s_rows + (size_t)row * D + d0) = make_float4(s[0], s[1], s[2], s[3]);
                else a.s_rows[(size_t)row * D + d0] = s[0];
            }
            if (gl == 0) {
                acc3[0] += l;
                acc3[2] += g;
            }
            for (int64_t p = lo + gl; p < hi; p += G) {       // per non-zero coefficients for the backward
                const float x = a.data[p];
                // one 16-byte record per non-zero {row, g x, lam - g x^2, -}: the backward reads one request per entry, not three
                a.ent[p] = make_int4((int32_t)row, __float_as_int(g * x), __float_as_int(a.lam - g * x * x), 0);
            }
        }
    }
    if (oob) atomicOr(a.err, 1);
    if constexpr (TRAIN) {
        __shared__ float red[4][3];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float t = acc3[c];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) t += __shfl_down(t, o, 64);
            if (lane == 0) red[wave][c] = t;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                a.partials[(size_t)blockIdx.x * 4 + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        }
    }
}

int fm_grid(int64_t n_rows, int G, bool train) {
    const int gpb = 256 / G;
    int64_t nb = (n_rows + gpb - 1) / gpb;
    // (training forward, one gpurun call, whole step: 2.10-2.12 ms at 2048 blocks, 2.07 at 4096 and at 8192; with its V rows
    // loaded non-temporally like the inference form's: 2.13-2.15 - the backward reads the same rows again and wants them cached)
    const int64_t cap = train ? 4096 : 8192;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return (int)nb;
}

void launch_fm(const FmArgs& a, bool train, int G, int VEC, int grid, hipStream_t s) {
#define TFR_FM_CASE(g, v)                                                                              \
    if (G == g && VEC == v) {                                                                          \
        if (train) hipLaunchKernelGGL((k_fm_forward<g, v, true, false>), dim3(grid), dim3(256), 0, s, a);     \
        else if (a.variant & 2) hipLaunchKernelGGL((k_fm_forward<g, v, false, true>), dim3(grid), dim3(256), 0, s, a);   \
        else hipLaunchKernelGGL((k_fm_forward<g, v, false, false>), dim3(grid), dim3(256), 0, s, a);   \
        return;                                                                                        \
    }
    TFR_FM_CASE(4, 4) TFR_FM_CASE(8, 4) TFR_FM_CASE(16, 4) TFR_FM_CASE(32, 4) TFR_FM_CASE(64, 4)
    TFR_FM_CASE(4, 1) TFR_FM_CASE(8, 1) TFR_FM_CASE(16, 1) TFR_FM_CASE(32, 1) TFR_FM_CASE(64, 1)
#undef TFR_FM_CASE
}

}  // namespace tfr
