// Probe: what read-only bandwidth does this GPU deliver to (a) a streaming read, (b) a random gather
// of 512-byte rows (the gather-dot forward's access pattern, without the arithmetic)?
//   hipcc --offload-arch=gfx950 -O3 -o read_bw tools/probes/read_bw.hip && ./read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int UNR>
__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ x, size_t n4, float* out) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x * UNR;
    for (size_t i = (size_t)blockIdx.x * blockDim.x * UNR + threadIdx.x; i < n4; i += stride) {
        float4 v[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) { const size_t j = i + (size_t)k * blockDim.x; v[k] = j < n4 ? x[j] : make_float4(0, 0, 0, 0); }
#pragma unroll
        for (int k = 0; k < UNR; ++k) acc += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}

// one 32-lane group per row (512 B = 32 x float4); UNR rows in flight per group
template <int UNR>
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ tab, const int32_t* __restrict__ ids, size_t n, float* out) {
    const int gl = threadIdx.x & 31;
    const size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const size_t ng = ((size_t)gridDim.x * blockDim.x) >> 5;
    float acc = 0.f;
    for (size_t k0 = g * UNR; k0 < n; k0 += ng * UNR) {
        int32_t id[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) id[k] = (k0 + k < n) ? ids[k0 + k] : 0;
        float4 v[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) v[k] = tab[(size_t)id[k] * 32 + gl];
#pragma unroll
        for (int k = 0; k < UNR; ++k) acc += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}

// the forward's full access pattern: two row tables, optionally two 4-byte bias gathers and a 4-byte
// result per rating (MODE bit 0: biases, bit 1: store)
template <int UNR, int MODE, int NT = 0>
__global__ __launch_bounds__(256) void k_pair(const float4* __restrict__ P, const float4* __restrict__ Q,
                                              const float* __restrict__ bu, const float* __restrict__ bi,
                                              const int32_t* __restrict__ iu, const int32_t* __restrict__ ii, size_t n, float* out) {
    const int gl = threadIdx.x & 31;
    const size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const size_t ng = ((size_t)gridDim.x * blockDim.x) >> 5;
    float tot = 0.f;
    for (size_t k0 = g * UNR; k0 < n; k0 += ng * UNR) {
        int32_t u[UNR], it[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) { u[k] = (k0 + k < n) ? iu[k0 + k] : 0; it[k] = (k0 + k < n) ? ii[k0 + k] : 0; }
        float4 a[UNR], b[UNR];
        float x[UNR], y[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            if (NT) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f va = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(&P[(size_t)u[k] * 32 + gl]));
                a[k] = make_float4(va.x, va.y, va.z, va.w);
                if (NT == 1) {
                    const v4f vb = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(&Q[(size_t)it[k] * 32 + gl]));
                    b[k] = make_float4(vb.x, vb.y, vb.z, vb.w);
                } else b[k] = Q[(size_t)it[k] * 32 + gl];
            } else {
                a[k] = P[(size_t)u[k] * 32 + gl];
                b[k] = Q[(size_t)it[k] * 32 + gl];
            }
            x[k] = (MODE & 1) ? bu[u[k]] : 0.f;
            y[k] = (MODE & 1) ? bi[it[k]] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            float s = a[k].x * b[k].x + a[k].y * b[k].y + a[k].z * b[k].z + a[k].w * b[k].w + x[k] + y[k];
            if (MODE & 2) {
                for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
                if (gl == 0 && k0 + k < n) out[k0 + k] = s;
            } else tot += s;
        }
    }
    if (tot == 12345.678f) out[blockIdx.x] = tot;
}

template <typename F>
static float time_us(F launch, int reps = 20) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CHK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) {
        CHK(hipEventRecord(e0)); launch(); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best * 1e3f;
}

int main() {
    const size_t n4 = (size_t)1 << 28;                  // 4 GiB of float4
    float4* x; float* out;
    CHK(hipMalloc(&x, n4 * 16)); CHK(hipMalloc(&out, 1 << 20));
    CHK(hipMemset(x, 1, n4 * 16));
    printf("streaming read of 4 GiB:\n");
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
        float t1 = time_us([&] { hipLaunchKernelGGL(k_stream<1>, dim3(blocks), dim3(256), 0, 0, x, n4, out); });
        float t4 = time_us([&] { hipLaunchKernelGGL(k_stream<4>, dim3(blocks), dim3(256), 0, 0, x, n4, out); });
        float t8 = time_us([&] { hipLaunchKernelGGL(k_stream<8>, dim3(blocks), dim3(256), 0, 0, x, n4, out); });
        printf("  blocks=%5d  unr1 %.2f TB/s  unr4 %.2f TB/s  unr8 %.2f TB/s\n", blocks, n4 * 16 / t1 / 1e6, n4 * 16 / t4 / 1e6, n4 * 16 / t8 / 1e6);
    }
    // random rows: table = the same 4 GiB viewed as 8M rows of 512 B; n row reads per launch
    const size_t rows = n4 / 32;
    for (size_t n : {(size_t)524288, (size_t)4194304}) {
        std::vector<int32_t> h(n);
        uint64_t s = 88172645463325252ull;
        for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int32_t)(s % rows); }
        int32_t* ids; CHK(hipMalloc(&ids, n * 4)); CHK(hipMemcpy(ids, h.data(), n * 4, hipMemcpyHostToDevice));
        printf("random gather of %zu rows x 512 B (%.0f MB):\n", n, n * 512 / 1e6);
        for (int blocks : {2048, 4096, 8192, 16384}) {
            float t1 = time_us([&] { hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(256), 0, 0, x, ids, n, out); });
            float t2 = time_us([&] { hipLaunchKernelGGL(k_gather<2>, dim3(blocks), dim3(256), 0, 0, x, ids, n, out); });
            float t4 = time_us([&] { hipLaunchKernelGGL(k_gather<4>, dim3(blocks), dim3(256), 0, 0, x, ids, n, out); });
            float t8 = time_us([&] { hipLaunchKernelGGL(k_gather<8>, dim3(blocks), dim3(256), 0, 0, x, ids, n, out); });
            printf("  blocks=%5d  unr1 %.2f  unr2 %.2f  unr4 %.2f  unr8 %.2f TB/s\n", blocks, n * 512 / t1 / 1e6, n * 512 / t2 / 1e6, n * 512 / t4 / 1e6, n * 512 / t8 / 1e6);
        }
        CHK(hipFree(ids));
    }
    {
        // P: 8M rows (4 GiB); Q: the first 1M rows of a second 0.5 GiB table; biases 32 MB / 4 MB
        const size_t n = 262144 * 8;                     // 8 batches' worth per launch: no reuse inside the Infinity Cache
        const size_t prow = n4 / 32, qrow = 1 << 20;
        float4* Q; float *bu, *bi, *res;
        CHK(hipMalloc(&Q, qrow * 512)); CHK(hipMemset(Q, 1, qrow * 512));
        CHK(hipMalloc(&bu, prow * 4)); CHK(hipMalloc(&bi, qrow * 4)); CHK(hipMalloc(&res, n * 4));
        CHK(hipMemset(bu, 0, prow * 4)); CHK(hipMemset(bi, 0, qrow * 4));
        std::vector<int32_t> hu(n), hi(n);
        uint64_t s = 1234567891234567ull;
        for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hu[i] = (int32_t)(s % prow); s ^= s << 13; s ^= s >> 7; s ^= s << 17; hi[i] = (int32_t)(s % qrow); }
        int32_t *iu, *ii; CHK(hipMalloc(&iu, n * 4)); CHK(hipMalloc(&ii, n * 4));
        CHK(hipMemcpy(iu, hu.data(), n * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(ii, hi.data(), n * 4, hipMemcpyHostToDevice));
        printf("pair gather (P 8M rows, Q 1M rows, %zu ratings/launch), row bytes only (1024 B/rating):\n", n);
        for (int blocks : {2048, 8192}) {
            float r0 = time_us([&] { hipLaunchKernelGGL((k_pair<4, 0>), dim3(blocks), dim3(256), 0, 0, x, Q, bu, bi, iu, ii, n, res); });
            float r1 = time_us([&] { hipLaunchKernelGGL((k_pair<4, 1>), dim3(blocks), dim3(256), 0, 0, x, Q, bu, bi, iu, ii, n, res); });
            float r2 = time_us([&] { hipLaunchKernelGGL((k_pair<4, 2>), dim3(blocks), dim3(256), 0, 0, x, Q, bu, bi, iu, ii, n, res); });
            float r3 = time_us([&] { hipLaunchKernelGGL((k_pair<4, 3>), dim3(blocks), dim3(256), 0, 0, x, Q, bu, bi, iu, ii, n, res); });
            float n1 = time_us([&] { hipLaunchKernelGGL((k_pair<4, 1, 1>), dim3(blocks), dim3(256), 0, 0, x, Q, bu, bi, iu, ii, n, res); });
            float n3 = time_us([&] { hipLaunchKernelGGL((k_pair<4, 3, 1>), dim3(blocks), dim3(256), 0, 0, x, Q, bu, bi, iu, ii, n, res); });
            float n0 = time_us([&] { hipLaunchKernelGGL((k_pair<4, 0, 1>), dim3(blocks), dim3(256), 0, 0, x, Q, bu, bi, iu, ii, n, res); });
            float p3 = time_us([&] { hipLaunchKernelGGL((k_pair<4, 3, 2>), dim3(blocks), dim3(256), 0, 0, x, Q, bu, bi, iu, ii, n, res); });
            float p0 = time_us([&] { hipLaunchKernelGGL((k_pair<4, 0, 2>), dim3(blocks), dim3(256), 0, 0, x, Q, bu, bi, iu, ii, n, res); });
            printf("  blocks=%5d  NT for P only: rows only %.1f us | +both %.1f us per 262144\n", blocks, p0 / 8, p3 / 8);
            printf("  blocks=%5d  NT rows: rows only %.1f us | +biases %.1f us | +both %.1f us per 262144\n", blocks, n0 / 8, n1 / 8, n3 / 8);
            {
                const size_t n1b = 262144;                   // one batch per launch, a different batch each time would be ideal; ids differ per offset
                float best = 1e30f;
                hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
                for (int rep = 0; rep < 24; ++rep) {
                    const size_t off = (size_t)(rep % 8) * n1b;
                    CHK(hipEventRecord(e0));
                    hipLaunchKernelGGL((k_pair<4, 3>), dim3(blocks), dim3(256), 0, 0, x, Q, bu, bi, iu + off, ii + off, n1b, res);
                    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep >= 8 && ms < best) best = ms;
                }
                printf("  blocks=%5d  one batch (262144) per launch, +both: %.1f us\n", blocks, best * 1e3f);
            }
            printf("  blocks=%5d  rows only %.2f | +biases %.2f | +reduce+store %.2f | +both %.2f TB/s (of row bytes); us/262144 ratings: %.1f %.1f %.1f %.1f\n", blocks,
                   n * 1024 / r0 / 1e6, n * 1024 / r1 / 1e6, n * 1024 / r2 / 1e6, n * 1024 / r3 / 1e6, r0 / 8, r1 / 8, r2 / 8, r3 / 8);
        }
    }
    return 0;
}

