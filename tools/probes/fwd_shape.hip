// Probe 2: launch-shape variants of the D=128 gather-dot forward at the north-star sizes, P rows loaded
// non-temporally (fwd_policy.hip: keeps Q / biases in the Infinity Cache).  Compiler-scheduled loads.
//   hipcc --offload-arch=gfx950 -O3 -o fwd_shape tools/probes/fwd_shape.hip && ./fwd_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

// IDS: 0 vector loads, 1 scalar loads (s_load of the wave's 2*UNR consecutive ids)
// BIAS: 0 none, 1 per-group broadcast vector loads, 2 wave-wide vector load, 3 scalar loads
// DYN: 0 static grid-stride, 1 chunks handed out by an atomic counter (one returning atomic per wave-iteration,
//      issued one iteration ahead)
template <int UNR, int IDS, int BIAS, int DYN, bool PNT>
__global__ __launch_bounds__(256) void k_fwd(const float* __restrict__ P, const float* __restrict__ Q,
                                             const float* __restrict__ bu, const float* __restrict__ bi,
                                             const int32_t* __restrict__ iu, const int32_t* __restrict__ ii, size_t n, float* __restrict__ out,
                                             unsigned long long* __restrict__ counter) {
    const int lane = threadIdx.x & 63, gl = lane & 31, sub = lane >> 5;
    const size_t wave = __builtin_amdgcn_readfirstlane((int)((((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6)));
    const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    constexpr int SPI = 2 * UNR;
    size_t k0 = wave * SPI, knext = 0;
    if (DYN) {
        unsigned long long t = 0;
        if (lane == 0) t = atomicAdd(counter, 2ull);           // this chunk and the next one
        t = __shfl(t, 0, 64);
        k0 = (size_t)__builtin_amdgcn_readfirstlane((int)t) * SPI;
        knext = k0 + SPI;
    }
    while (k0 < n) {
        unsigned long long tn = 0;
        if (DYN) { if (lane == 0) tn = atomicAdd(counter, 1ull); }     // chunk after next, behind this iteration's loads
        int32_t u[UNR], it[UNR];
        if (IDS == 1) {
            const int32_t* su = iu + k0; const int32_t* si = ii + k0;   // uniform addresses -> s_load
#pragma unroll
            for (int k = 0; k < UNR; ++k) {
                const int32_t a0 = su[2 * k], a1 = su[2 * k + 1], b0 = si[2 * k], b1 = si[2 * k + 1];
                u[k] = sub ? a1 : a0; it[k] = sub ? b1 : b0;
            }
        } else {
#pragma unroll
            for (int k = 0; k < UNR; ++k) { const size_t kk = k0 + k * 2 + sub; u[k] = kk < n ? iu[kk] : 0; it[k] = kk < n ? ii[kk] : 0; }
        }
        v4f a[UNR], b[UNR];
        float x[UNR], y[UNR], wx = 0.f, wy = 0.f;
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const v4f* pa = reinterpret_cast<const v4f*>(P + (size_t)u[k] * 128 + gl * 4);
            a[k] = PNT ? __builtin_nontemporal_load(pa) : *pa;
            b[k] = *reinterpret_cast<const v4f*>(Q + (size_t)it[k] * 128 + gl * 4);
            x[k] = 0.f; y[k] = 0.f;
            if (BIAS == 1) { x[k] = bu[u[k]]; y[k] = bi[it[k]]; }
            if (BIAS == 3) {
                const int32_t u0 = __builtin_amdgcn_readlane(u[k], 0), u1 = __builtin_amdgcn_readlane(u[k], 32);
                const int32_t i0 = __builtin_amdgcn_readlane(it[k], 0), i1 = __builtin_amdgcn_readlane(it[k], 32);
                const float x0 = bu[u0], x1 = bu[u1], y0 = bi[i0], y1 = bi[i1];
                x[k] = sub ? x1 : x0; y[k] = sub ? y1 : y0;
            }
        }
        if (BIAS == 2) {
            int32_t mu_ = 0, mi_ = 0;
#pragma unroll
            for (int k = 0; k < UNR; ++k) {
                const int32_t s0 = __shfl(u[k], 0, 64), s1 = __shfl(u[k], 32, 64);
                const int32_t t0 = __shfl(it[k], 0, 64), t1 = __shfl(it[k], 32, 64);
                if (lane == 2 * k) { mu_ = s0; mi_ = t0; }
                if (lane == 2 * k + 1) { mu_ = s1; mi_ = t1; }
            }
            if (lane < SPI) { wx = bu[mu_]; wy = bi[mi_]; }
#pragma unroll
            for (int k = 0; k < UNR; ++k) { x[k] = __shfl(wx, 2 * k + sub, 64); y[k] = __shfl(wy, 2 * k + sub, 64); }
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            float s = a[k].x * b[k].x + a[k].y * b[k].y + a[k].z * b[k].z + a[k].w * b[k].w;
            for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
            s = (s + x[k]) + y[k];
            const size_t kk = k0 + k * 2 + sub;
            if (gl == 0 && kk < n) out[kk] = s;
        }
        if (DYN) {
            k0 = knext;
            tn = __shfl(tn, 0, 64);
            knext = (size_t)__builtin_amdgcn_readfirstlane((int)tn) * SPI;
        } else k0 += nw * SPI;
    }
}

__global__ void k_fill(float* p, size_t n, uint32_t salt) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)(i * 2654435761u) ^ salt; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((float)(h & 0xffff) - 32768.f) * (1.f / 65536.f);
    }
}
__global__ void k_ref(const float* P, const float* Q, const float* bu, const float* bi, const int32_t* iu, const int32_t* ii, size_t n, float* out, int with_bias) {
    const int gl = threadIdx.x & 31;
    for (size_t k = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5; k < n; k += ((size_t)gridDim.x * blockDim.x) >> 5) {
        const float4 a = reinterpret_cast<const float4*>(P + (size_t)iu[k] * 128)[gl], b = reinterpret_cast<const float4*>(Q + (size_t)ii[k] * 128)[gl];
        float s = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
        for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        if (gl == 0) out[k] = with_bias ? (s + bu[iu[k]]) + bi[ii[k]] : (s + 0.f) + 0.f;
    }
}
__global__ void k_cmp(const float* a, const float* b, size_t n, int* bad) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (a[i] != b[i]) atomicAdd(bad, 1);
}

struct Ctx { float *P, *Q, *bu, *bi, *res, *ref, *ref0; int32_t *iu, *ii; size_t n; unsigned long long* counter; hipStream_t s2; };

template <int UNR, int IDS, int BIAS, int DYN, bool PNT>
static void run(const Ctx& c, int blocks, const char* tag) {
    auto launch = [&](size_t off, size_t n, hipStream_t st) {
        if (DYN) CHK(hipMemsetAsync(c.counter, 0, 8, st));
        hipLaunchKernelGGL((k_fwd<UNR, IDS, BIAS, DYN, PNT>), dim3(blocks), dim3(256), 0, st, c.P, c.Q, c.bu, c.bi, c.iu + off, c.ii + off, n, c.res + off, c.counter);
    };
    // correctness
    CHK(hipMemset(c.res, 0, c.n * 4));
    launch(0, c.n, 0);
    int* bad; CHK(hipMalloc(&bad, 4)); CHK(hipMemset(bad, 0, 4));
    hipLaunchKernelGGL(k_cmp, dim3(1024), dim3(256), 0, 0, c.res, BIAS ? c.ref : c.ref0, c.n, bad);
    int hb = 0; CHK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CHK(hipFree(bad));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float t8 = 0.f, t1 = 0.f, tb = 0.f;
    for (int r = 0; r < 12; ++r) {                   // 8 batches per launch
        CHK(hipEventRecord(e0)); launch(0, c.n, 0); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) t8 += ms / 10;
    }
    for (int r = 0; r < 26; ++r) {                   // one batch per launch, event pair around each launch
        CHK(hipEventRecord(e0)); launch((size_t)(r % 8) * 262144, 262144, 0); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) t1 += ms / 24;
    }
    {                                                // 64 single-batch launches back to back on one stream: time per launch
        CHK(hipEventRecord(e0));
        for (int r = 0; r < 64; ++r) launch((size_t)(r % 8) * 262144, 262144, 0);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); tb = ms / 64;
    }
    float t2 = 0.f;
    {                                                // the same, alternating between two streams (independent batches)
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0, 0));
        CHK(hipStreamWaitEvent(c.s2, e0, 0));
        for (int r = 0; r < 64; ++r) launch((size_t)(r % 8) * 262144, 262144, (r & 1) ? c.s2 : 0);
        hipEvent_t e2; CHK(hipEventCreate(&e2)); CHK(hipEventRecord(e2, c.s2)); CHK(hipStreamWaitEvent(0, e2, 0));
        CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); t2 = ms / 64; CHK(hipEventDestroy(e2));
    }
    const double fr = 262144.0 * 1044 / 1e3 / 8.0;   // us at 8 TB/s... frac = fr / t_us / 1e3
    printf("%-8s unr=%d ids=%d bias=%d dyn=%d pnt=%d blocks=%5d : 8x %.2f (%.3f) | 1x ev %.2f (%.3f) | 1x b2b %.2f (%.3f) | 2-stream %.2f (%.3f)%s\n", tag, UNR, IDS, BIAS, DYN,
           (int)PNT, blocks, t8 * 1e3 / 8, fr / (t8 * 1e3 / 8) / 1e3, t1 * 1e3, fr / (t1 * 1e3) / 1e3, tb * 1e3, fr / (tb * 1e3) / 1e3, t2 * 1e3, fr / (t2 * 1e3) / 1e3,
           hb ? "  !! MISMATCH" : "");
    fflush(stdout);
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
}

int main() {
    const size_t prow = 10000000, qrow = 1000000, n = 262144 * 8;
    Ctx c; c.n = n;
    CHK(hipStreamCreateWithFlags(&c.s2, hipStreamNonBlocking));
    CHK(hipMalloc(&c.P, prow * 512)); CHK(hipMalloc(&c.Q, qrow * 512));
    CHK(hipMalloc(&c.bu, prow * 4)); CHK(hipMalloc(&c.bi, qrow * 4)); CHK(hipMalloc(&c.res, n * 4));
    CHK(hipMalloc(&c.ref, n * 4)); CHK(hipMalloc(&c.ref0, n * 4)); CHK(hipMalloc(&c.counter, 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, c.P, prow * 128, 1u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, c.Q, qrow * 128, 2u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, c.bu, prow, 3u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, c.bi, qrow, 4u);
    std::vector<int32_t> hu(n), hi(n);
    uint64_t s = 1234567891234567ull;
    for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hu[i] = (int32_t)(s % prow); s ^= s << 13; s ^= s >> 7; s ^= s << 17; hi[i] = (int32_t)(s % qrow); }
    CHK(hipMalloc(&c.iu, n * 4)); CHK(hipMalloc(&c.ii, n * 4));
    CHK(hipMemcpy(c.iu, hu.data(), n * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(c.ii, hi.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_ref, dim3(4096), dim3(256), 0, 0, c.P, c.Q, c.bu, c.bi, c.iu, c.ii, n, c.ref, 1);
    hipLaunchKernelGGL(k_ref, dim3(4096), dim3(256), 0, 0, c.P, c.Q, c.bu, c.bi, c.iu, c.ii, n, c.ref0, 0);
    CHK(hipDeviceSynchronize());

    run<4, 0, 1, 0, false>(c, 8192, "r01");
    run<4, 0, 1, 0, true>(c, 8192, "pnt");
    run<4, 0, 2, 0, true>(c, 8192, "wbias");
    run<4, 0, 3, 0, true>(c, 8192, "sbias");
    run<4, 1, 1, 0, true>(c, 8192, "sids");
    run<4, 1, 2, 0, true>(c, 8192, "sids+wb");
    run<4, 1, 3, 0, true>(c, 8192, "sids+sb");
    run<4, 0, 0, 0, true>(c, 8192, "nobias");
    run<2, 0, 2, 0, true>(c, 16384, "wbias");
    run<2, 0, 2, 0, true>(c, 8192, "wbias");
    run<8, 0, 2, 0, true>(c, 4096, "wbias");
    run<8, 0, 2, 0, true>(c, 2048, "wbias");
    run<4, 0, 2, 1, true>(c, 2048, "dyn");
    run<4, 0, 2, 1, true>(c, 4096, "dyn");
    run<4, 1, 2, 1, true>(c, 2048, "dyn+sids");
    run<2, 0, 2, 1, true>(c, 2048, "dyn");
    return 0;
}
