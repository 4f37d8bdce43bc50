// piece_walk.hip - why does a lane group that walks ~800 row-sized pieces 16 KB apart take ~0.3 us per piece?
//   hipcc --offload-arch=gfx950 -O3 -o piece_walk piece_walk.hip && ./piece_walk
// One 32-lane group adds NP pieces of 128 floats, WIDE loads in flight, for several piece strides (in 512-B rows).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int WIDE>
__global__ void walk(const float4* __restrict__ src, float4* out, int np, long stride_rows) {
    const int lane = threadIdx.x;            // 32 lanes x float4 = one 512-B row
    if (lane >= 32) return;
    float4 t = {0, 0, 0, 0};
    for (int p = 0; p < np; p += WIDE) {
        float4 x[WIDE];
#pragma unroll
        for (int q = 0; q < WIDE; ++q) x[q] = src[(size_t)(p + q) * stride_rows * 32 + lane];
#pragma unroll
        for (int q = 0; q < WIDE; ++q) { t.x += x[q].x; t.y += x[q].y; t.z += x[q].z; t.w += x[q].w; }
    }
    out[lane] = t;
}
int main() {
    const int NP = 800;
    const long strides[] = {1, 2, 8, 31, 32, 33, 34, 40, 64, 65};
    size_t bytes = (size_t)NP * 66 * 512 + 4096;
    float4 *src, *out;
    hipMalloc(&src, bytes); hipMalloc(&out, 4096);
    hipMemset(src, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // evict caches between runs: stream a 1 GB buffer
    char* junk; hipMalloc(&junk, 1ull << 30);
    for (long s : strides) {
        for (int w : {8, 32}) {
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(junk, rep, 1ull << 30);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                if (w == 8) hipLaunchKernelGGL(walk<8>, dim3(1), dim3(64), 0, 0, src, out, NP, s);
                else hipLaunchKernelGGL(walk<32>, dim3(1), dim3(64), 0, 0, src, out, NP, s);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("stride %3ld rows (%6ld B)  wide %2d: %8.1f us  (%.3f us/piece)\n", s, s * 512, w, best * 1000, best * 1000 / NP);
        }
    }
    return 0;
}
