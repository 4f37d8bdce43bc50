// Probe: cache-policy and bias-load variants of the D=128 gather-dot forward's access pattern at the
// north-star sizes (P 10M x 512 B, Q 1M x 512 B, bu 40 MB, bi 4 MB, uniform ids).
//   hipcc --offload-arch=gfx950 -O3 -o fwd_policy tools/probes/fwd_policy.hip && ./fwd_policy
// Policies (gfx950 global_load modifiers): 0 default, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0
// Loads are inline asm (the compiler cannot see them), so every batch of loads ends in s_waitcnt vmcnt(0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

template <int POL> __device__ __forceinline__ void ld16(v4f& v, const void* p) {
    if constexpr (POL == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (POL == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (POL == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
}
template <int POL> __device__ __forceinline__ void ld4(float& v, const void* p) {
    if constexpr (POL == 0) asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (POL == 1) asm volatile("global_load_dword %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (POL == 2) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (POL == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (POL == 4) asm volatile("global_load_dword %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
}

// BIAS: 0 none, 1 per-group broadcast loads (what k_forward does), 2 one wave-wide load per table for the
// wave-iteration's ratings (lane j < 2*UNR loads the bias of rating j; distributed by shuffles)
template <int UNR, int PP, int PQ, int PB, int BIAS>
__global__ __launch_bounds__(256) void k_fwd(const float* __restrict__ P, const float* __restrict__ Q,
                                             const float* __restrict__ bu, const float* __restrict__ bi,
                                             const int32_t* __restrict__ iu, const int32_t* __restrict__ ii, size_t n, float* out) {
    const int lane = threadIdx.x & 63, gl = lane & 31, sub = lane >> 5;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    constexpr int SPI = 2 * UNR;
    for (size_t k0 = wave * SPI; k0 < n; k0 += nw * SPI) {
        int32_t u[UNR], it[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const size_t kk = k0 + k * 2 + sub;
            u[k] = kk < n ? iu[kk] : 0; it[k] = kk < n ? ii[kk] : 0;
        }
        v4f a[UNR], b[UNR];
        float x[UNR], y[UNR], wx = 0.f, wy = 0.f;
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            ld16<PP>(a[k], P + (size_t)u[k] * 128 + gl * 4);
            ld16<PQ>(b[k], Q + (size_t)it[k] * 128 + gl * 4);
            x[k] = 0.f; y[k] = 0.f;
            if constexpr (BIAS == 1) { ld4<PB>(x[k], bu + u[k]); ld4<PB>(y[k], bi + it[k]); }
        }
        if constexpr (BIAS == 2) {
            // lane j (j < SPI) takes rating k0 + j: that is (k = j / 2, sub = j % 2); fetch its ids by shuffle
            int32_t mu_ = 0, mi_ = 0;
#pragma unroll
            for (int k = 0; k < UNR; ++k) {
                const int32_t s0 = __shfl(u[k], 0, 64), s1 = __shfl(u[k], 32, 64);
                const int32_t t0 = __shfl(it[k], 0, 64), t1 = __shfl(it[k], 32, 64);
                if (lane == 2 * k) { mu_ = s0; mi_ = t0; }
                if (lane == 2 * k + 1) { mu_ = s1; mi_ = t1; }
            }
            if (lane < SPI) { ld4<PB>(wx, bu + mu_); ld4<PB>(wy, bi + mi_); }
        }
        // the loaded registers are tied to the wait so no use can be scheduled above it
        static_assert(UNR == 4, "operand list below is written for UNR = 4");
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]),
                       "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(wx), "+v"(wy)
                     :: "memory");
        if constexpr (BIAS == 2) {
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
    }
}

__global__ void k_fill(float* p, size_t n, uint32_t salt) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)(i * 2654435761u) ^ salt; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((float)(h & 0xffff) - 32768.f) * (1.f / 65536.f);
    }
}
__global__ void k_ref(const float* P, const float* Q, const float* bu, const float* bi, const int32_t* iu, const int32_t* ii, size_t n, float* out) {
    const int gl = threadIdx.x & 31;
    for (size_t k = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5; k < n; k += ((size_t)gridDim.x * blockDim.x) >> 5) {
        const float4 a = reinterpret_cast<const float4*>(P + (size_t)iu[k] * 128)[gl], b = reinterpret_cast<const float4*>(Q + (size_t)ii[k] * 128)[gl];
        float s = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
        for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        if (gl == 0) out[k] = (s + bu[iu[k]]) + bi[ii[k]];
    }
}

template <typename F>
static float time_us(F launch, int reps = 12) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch();
    CHK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0.f;
    for (int i = 0; i < reps; ++i) {
        CHK(hipEventRecord(e0)); launch(); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        sum += ms;
    }
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    return sum / reps * 1e3f;       // average: what bench.py reports
}

static const char* PN[] = {"def", "nt", "sc1", "sc0sc1", "sc1nt", "sc0"};

struct Ctx { float *P, *Q, *bu, *bi, *res, *ref, *ref0; int32_t *iu, *ii; size_t n; };
__global__ void k_cmp(const float* a, const float* b, size_t n, int* bad) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (a[i] != b[i]) atomicAdd(bad, 1);
}

template <int PP, int PQ, int PB, int BIAS>
static void run(const Ctx& c, int blocks, const char* tag) {
    const size_t n8 = c.n;
    float t8 = time_us([&] { hipLaunchKernelGGL((k_fwd<4, PP, PQ, PB, BIAS>), dim3(blocks), dim3(256), 0, 0, c.P, c.Q, c.bu, c.bi, c.iu, c.ii, n8, c.res); });
    // one batch per launch, rotating over the 8 batches
    int rot = 0;
    float t1 = time_us([&] {
        const size_t off = (size_t)(rot++ % 8) * 262144;
        hipLaunchKernelGGL((k_fwd<4, PP, PQ, PB, BIAS>), dim3(blocks), dim3(256), 0, 0, c.P, c.Q, c.bu, c.bi, c.iu + off, c.ii + off, (size_t)262144, c.res);
    }, 24);
    {   // the policy kernel must give exactly what the plain kernel gives (rows-only runs: the bias-free reference)
        CHK(hipMemset(c.res, 0, n8 * 4));
        hipLaunchKernelGGL((k_fwd<4, PP, PQ, PB, BIAS>), dim3(blocks), dim3(256), 0, 0, c.P, c.Q, c.bu, c.bi, c.iu, c.ii, n8, c.res);
        int* bad; CHK(hipMalloc(&bad, 4)); CHK(hipMemset(bad, 0, 4));
        hipLaunchKernelGGL(k_cmp, dim3(1024), dim3(256), 0, 0, c.res, BIAS ? c.ref : c.ref0, n8, bad);
        int hb = 0; CHK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CHK(hipFree(bad));
        if (hb) printf("  !! %d mismatches vs the plain kernel\n", hb);
    }
    printf("%-10s P=%-6s Q=%-6s b=%-6s bias=%d blocks=%5d : 8x %.2f us/batch (%.3f of 8 TB/s) | 1x %.2f us (%.3f)\n", tag, PN[PP], PN[PQ], PN[PB], BIAS,
           blocks, t8 / 8, 262144.0 * 1044 / (t8 / 8) / 1e6 / 8.0, t1, 262144.0 * 1044 / t1 / 1e6 / 8.0);
    fflush(stdout);
}

int main() {
    const size_t prow = 10000000, qrow = 1000000, n = 262144 * 8;
    Ctx c; c.n = n;
    CHK(hipMalloc(&c.P, prow * 512)); CHK(hipMalloc(&c.Q, qrow * 512));
    CHK(hipMalloc(&c.bu, prow * 4)); CHK(hipMalloc(&c.bi, qrow * 4)); CHK(hipMalloc(&c.res, n * 4));
    CHK(hipMalloc(&c.ref, n * 4)); CHK(hipMalloc(&c.ref0, n * 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, c.P, prow * 128, 1u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, c.Q, qrow * 128, 2u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, c.bu, prow, 3u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, c.bi, qrow, 4u);
    CHK(hipDeviceSynchronize());
    std::vector<int32_t> hu(n), hi(n);
    uint64_t s = 1234567891234567ull;
    for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hu[i] = (int32_t)(s % prow); s ^= s << 13; s ^= s >> 7; s ^= s << 17; hi[i] = (int32_t)(s % qrow); }
    CHK(hipMalloc(&c.iu, n * 4)); CHK(hipMalloc(&c.ii, n * 4));
    CHK(hipMemcpy(c.iu, hu.data(), n * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(c.ii, hi.data(), n * 4, hipMemcpyHostToDevice));

    hipLaunchKernelGGL(k_ref, dim3(4096), dim3(256), 0, 0, c.P, c.Q, c.bu, c.bi, c.iu, c.ii, n, c.ref);
    {   // bias-free reference: zero biases
        float *z1, *z2; CHK(hipMalloc(&z1, prow * 4)); CHK(hipMalloc(&z2, qrow * 4)); CHK(hipMemset(z1, 0, prow * 4)); CHK(hipMemset(z2, 0, qrow * 4));
        hipLaunchKernelGGL(k_ref, dim3(4096), dim3(256), 0, 0, c.P, c.Q, z1, z2, c.iu, c.ii, n, c.ref0);
        CHK(hipDeviceSynchronize()); CHK(hipFree(z1)); CHK(hipFree(z2));
    }
    // reference points
    run<0, 0, 0, 0>(c, 8192, "rows-only");
    run<0, 0, 0, 1>(c, 8192, "baseline");
    run<0, 0, 0, 1>(c, 2048, "baseline");
    run<0, 0, 0, 1>(c, 4096, "baseline");
    run<0, 0, 0, 2>(c, 8192, "wavebias");
    // P policy
    run<1, 0, 0, 1>(c, 8192, "P");
    run<2, 0, 0, 1>(c, 8192, "P");
    run<3, 0, 0, 1>(c, 8192, "P");
    run<4, 0, 0, 1>(c, 8192, "P");
    run<5, 0, 0, 1>(c, 8192, "P");
    // P and Q policy
    run<1, 1, 0, 1>(c, 8192, "PQ");
    run<2, 2, 0, 1>(c, 8192, "PQ");
    run<4, 4, 0, 1>(c, 8192, "PQ");
    run<1, 0, 0, 0>(c, 8192, "rows-only");
    run<1, 1, 0, 0>(c, 8192, "rows-only");
    // bias policy
    run<0, 0, 1, 1>(c, 8192, "b");
    run<0, 0, 2, 1>(c, 8192, "b");
    run<0, 0, 5, 1>(c, 8192, "b");
    run<1, 0, 2, 1>(c, 8192, "P+b");
    run<1, 0, 5, 1>(c, 8192, "P+b");
    run<1, 1, 5, 1>(c, 8192, "PQ+b");
    // wave-wide bias loads with the best-looking policies
    run<1, 0, 0, 2>(c, 8192, "wavebias");
    run<1, 1, 0, 2>(c, 8192, "wavebias");
    run<4, 0, 0, 2>(c, 8192, "wavebias");
    run<1, 0, 0, 2>(c, 4096, "wavebias");
    run<1, 0, 0, 2>(c, 2048, "wavebias");
    return 0;
}
