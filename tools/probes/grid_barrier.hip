// Probe: what does an in-kernel grid barrier cost on this GPU, against a kernel boundary?
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier tools/probes/grid_barrier.hip && ./grid_barrier
// Barrier = one device-scope atomic counter per round (monotonic, no reset), bounded spin so a
// scheduling surprise ends in an error flag, never a hang.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, int* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();                                  // release this block's writes (agent scope)
        atomicAdd(counter, 1u);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { atomicExch(err, 1); ok = false; break; }
        }
        __threadfence();                                  // acquire the other blocks' writes
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(1024) void k_rounds(unsigned* counter, int* err, float* data, int rounds, int n) {
    // each round: every block writes a slice, barrier, reads a slice written by another block
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        const int i = (blockIdx.x * blockDim.x + threadIdx.x) % n;
        data[(size_t)(r & 1) * n + i] = acc + (float)r;
        if (!grid_barrier(counter, (unsigned)(r + 1) * gridDim.x, err)) return;
        const int j = (int)(((size_t)(blockIdx.x + 97) % gridDim.x) * blockDim.x + threadIdx.x) % n;
        acc += data[(size_t)(r & 1) * n + j];
    }
    if (acc == -1.f) data[0] = acc;
}

__global__ __launch_bounds__(1024) void k_one(float* data, int r, int n) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) % n;
    const int j = (int)(((size_t)(blockIdx.x + 97) % gridDim.x) * blockDim.x + threadIdx.x) % n;
    data[(size_t)(r & 1) * n + i] = data[(size_t)((r + 1) & 1) * n + j] + (float)r;
}

int main() {
    unsigned* counter; int* err; float* data;
    const int n = 1 << 20;
    CHK(hipMalloc(&counter, 4)); CHK(hipMalloc(&err, 4)); CHK(hipMalloc(&data, (size_t)2 * n * 4));
    CHK(hipMemset(data, 0, (size_t)2 * n * 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int blocks : {64, 256, 320, 512}) {
        for (int threads : {256, 1024}) {
            int maxb = 0;
            CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&maxb, k_rounds, threads, 0));
            if (blocks > maxb * 256) { printf("blocks=%d threads=%d: not co-resident (max %d/CU), skipped\n", blocks, threads, maxb); continue; }
            const int rounds = 200;
            float best = 1e9f;
            int herr = 0;
            for (int rep = 0; rep < 5; ++rep) {
                CHK(hipMemset(counter, 0, 4)); CHK(hipMemset(err, 0, 4));
                CHK(hipDeviceSynchronize());
                CHK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_rounds, dim3(blocks), dim3(threads), 0, 0, counter, err, data, rounds, n);
                CHK(hipEventRecord(e1));
                CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                CHK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
                if (herr) break;
            }
            float bestk = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CHK(hipDeviceSynchronize());
                CHK(hipEventRecord(e0));
                for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k_one, dim3(blocks), dim3(threads), 0, 0, data, r, n);
                CHK(hipEventRecord(e1));
                CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < bestk) bestk = ms;
            }
            printf("blocks=%4d threads=%4d: barrier round %.2f us%s | kernel-per-round %.2f us\n", blocks, threads,
                   best * 1e3f / rounds, herr ? " (SPIN LIMIT HIT)" : "", bestk * 1e3f / rounds);
            fflush(stdout);
        }
    }
    return 0;
}
