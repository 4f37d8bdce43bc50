// copy_bw.hip - what read+write rate a copy kernel reaches on this box, by kernel shape (the yardstick the fused
// training kernels are priced against; MI355X_MICROARCH.md quotes 6.29 TB/s for "float4 copy").
//   hipcc --offload-arch=gfx950 -O3 -o copy_bw copy_bw.hip && ./copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int UN, int NT>
__global__ void __launch_bounds__(256) k_copy(const f4* __restrict__ src, f4* __restrict__ dst, long n4) {
    const long stride = (long)gridDim.x * blockDim.x;
    long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; k + (UN - 1) * stride < n4; k += UN * stride) {
        f4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) v[u] = (NT & 1) ? __builtin_nontemporal_load(&src[k + u * stride]) : src[k + u * stride];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (NT & 2) __builtin_nontemporal_store(v[u], &dst[k + u * stride]); else dst[k + u * stride] = v[u];
        }
    }
    for (; k < n4; k += stride) dst[k] = src[k];
}

// block-contiguous: each workgroup owns a contiguous chunk (DRAM page locality per CU)
template <int UN, int NT>
__global__ void __launch_bounds__(256) k_copy_chunk(const f4* __restrict__ src, f4* __restrict__ dst, long n4) {
    const long per = (n4 + gridDim.x - 1) / gridDim.x;
    const long b0 = (long)blockIdx.x * per, b1 = (b0 + per < n4) ? b0 + per : n4;
    long k = b0 + threadIdx.x;
    for (; k + (UN - 1) * 256 < b1; k += UN * 256) {
        f4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) v[u] = (NT & 1) ? __builtin_nontemporal_load(&src[k + u * 256]) : src[k + u * 256];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (NT & 2) __builtin_nontemporal_store(v[u], &dst[k + u * 256]); else dst[k + u * 256] = v[u];
        }
    }
    for (; k < b1; k += 256) dst[k] = src[k];
}

template <int UN>
__global__ void __launch_bounds__(256) k_read(const f4* __restrict__ src, f4* __restrict__ dst, long n4) {
    const long stride = (long)gridDim.x * blockDim.x;
    long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (; k + (UN - 1) * stride < n4; k += UN * stride) {
        f4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) v[u] = src[k + u * stride];
#pragma unroll
        for (int u = 0; u < UN; ++u) acc += v[u];
    }
    if (acc.x == 123.456f) dst[0] = acc;
}

template <int UN>
__global__ void __launch_bounds__(256) k_write(f4* __restrict__ dst, long n4) {
    const long stride = (long)gridDim.x * blockDim.x;
    long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (; k < n4; k += stride) dst[k] = v;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename F>
static void timeit(const char* name, double bytes, F launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0.f;
    for (int r = -2; r < 10; ++r) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r < 0) continue;
        if (ms < best) best = ms;
        sum += ms;
    }
    printf("%-44s best %7.1f GB/s  mean %7.1f GB/s\n", name, bytes / best / 1e6, bytes / (sum / 10) / 1e6);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const long bytes = (argc > 1 ? atol(argv[1]) : 1024L) << 20;
    const long n4 = bytes / 16;
    f4 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    char nm[128];
    for (int g : {2048, 4096, 8192, 16384, 65536}) {
        snprintf(nm, sizeof nm, "copy stride un1 grid %d", g);      timeit(nm, 2.0 * bytes, [&] { k_copy<1, 0><<<g, 256>>>(a, b, n4); });
        snprintf(nm, sizeof nm, "copy stride un4 grid %d", g);      timeit(nm, 2.0 * bytes, [&] { k_copy<4, 0><<<g, 256>>>(a, b, n4); });
        snprintf(nm, sizeof nm, "copy stride un4 nt-store grid %d", g); timeit(nm, 2.0 * bytes, [&] { k_copy<4, 2><<<g, 256>>>(a, b, n4); });
        snprintf(nm, sizeof nm, "copy stride un4 nt-both grid %d", g);  timeit(nm, 2.0 * bytes, [&] { k_copy<4, 3><<<g, 256>>>(a, b, n4); });
        snprintf(nm, sizeof nm, "copy chunk un4 grid %d", g);       timeit(nm, 2.0 * bytes, [&] { k_copy_chunk<4, 0><<<g, 256>>>(a, b, n4); });
        snprintf(nm, sizeof nm, "copy chunk un4 nt-both grid %d", g);   timeit(nm, 2.0 * bytes, [&] { k_copy_chunk<4, 3><<<g, 256>>>(a, b, n4); });
    }
    {   // one float4 per thread, no loop
        const long g = (n4 + 255) / 256;
        timeit("copy one-f4-per-thread", 2.0 * bytes, [&] { k_copy<1, 0><<<(unsigned)g, 256>>>(a, b, n4); });
        timeit("copy one-f4-per-thread nt-both", 2.0 * bytes, [&] { k_copy_chunk<1, 3><<<(unsigned)g, 256>>>(a, b, n4); });
    }
    timeit("hipMemcpyDtoD", 2.0 * bytes, [&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); });
    for (int g : {2048, 8192}) {
        snprintf(nm, sizeof nm, "read only un4 grid %d", g);  timeit(nm, 1.0 * bytes, [&] { k_read<4><<<g, 256>>>(a, b, n4); });
        snprintf(nm, sizeof nm, "write only grid %d", g);     timeit(nm, 1.0 * bytes, [&] { k_write<1><<<g, 256>>>(b, n4); });
    }
    return 0;
}
