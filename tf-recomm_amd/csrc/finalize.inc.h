// finalize.inc.h - K4 body, shared by k_finalize (svd_kernels.hip) and the fused csort scan
// launch (sort.hip): fixed-order reduction of the forward's per-block partials -> {data loss,
// regulariser, sum g}; dense update of bias_global (ApplyAdam dense kernel [TF1-lib]:
// m += (g-m)(1-b1); v += (g*g-v)(1-b2); var -= alpha*m/(sqrt(v)+eps); or var -= lr*g).
// Every thread of the block must call it; only threads 0..255 carry data.
#pragma once
#include <hip/hip_runtime.h>
#include "svd_kernels.h"

namespace tfr {

__device__ __forceinline__ void finalize_body(const FinArgs& a) {
    __shared__ float fin_red[4][3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float acc[3] = {0.f, 0.f, 0.f};
    if (tid < 256) {
        for (int b = tid; b < a.nblk; b += 256 * 8) {     // eight independent loads per round trip, added in order
            float x[8][3];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int bb = (b + q * 256 < a.nblk) ? b + q * 256 : b;         // in-bounds address; masked below
#pragma unroll
                for (int c = 0; c < 3; ++c) x[q][c] = a.partials[(size_t)bb * 4 + c];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (b + q * 256 < a.nblk) { acc[0] += x[q][0]; acc[1] += x[q][1]; acc[2] += x[q][2]; }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s = acc[c];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0 && wave < 4) fin_red[wave][c] = s;
    }
    __syncthreads();
    if (tid == 0) {
        float tot[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) tot[c] = (fin_red[0][c] + fin_red[1][c]) + (fin_red[2][c] + fin_red[3][c]);
        a.scalars[0] = tot[0];
        a.scalars[1] = tot[1];
        a.scalars[2] = tot[2];
        if (a.out) {
            a.out[0] = tot[0]; a.out[1] = tot[1]; a.out[2] = tot[2];
            if (a.out_err) a.out[3] = (float)*a.err;
        }
        if (a.update_mu && *a.err == 0) {
            const float g = tot[2];
            float w = *a.mu;
            if (a.opt == 0) {
                float m = *a.mu_m, v = *a.mu_v;
                m += (g - m) * (1.f - a.b1);
                v += (g * g - v) * (1.f - a.b2);
                w -= (a.alpha * m) / (sqrtf(v) + a.eps);
                *a.mu_m = m;
                *a.mu_v = v;
            } else {
                w -= a.lr * g;
            }
            *a.mu = w;
        }
        if (a.clear_partials) {
            float* pp = const_cast<float*>(a.partials);
            pp[0] = 0.f; pp[1] = 0.f; pp[2] = 0.f; pp[3] = 0.f;
        }
    }
}

}  // namespace tfr
