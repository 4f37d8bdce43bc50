// flag_fuse.hip - can the two dependent launches of the small-table step become one, with the second part's blocks waiting on
// a counter the first part's blocks bump (one-directional, no grid barrier)?
//   hipcc --offload-arch=gfx950 -O3 -o flag_fuse flag_fuse.hip && ./flag_fuse
// Part A: NA blocks x 1024 threads do ~`work` us of dependent loads, write SCR bytes of "pieces" to UNCACHED device memory,
// then one agent-scope atomic add each.  Part B: NB blocks x 256 threads preload their table rows, spin (one lane, s_sleep)
// until the counter reaches NA, read the pieces (uncached), write the rows.  Compared with the same work as two launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Args {
    const int* chain; float* pieces; float* table; unsigned* flag; unsigned long long* stamps; int* err;
    int na, chain_len, piece_floats, rows_per_block, D, step, nflags;
};

__device__ __forceinline__ void part_a(const Args& a, int blk) {
    // a dependent chain of loads (the gather -> dot -> sort -> reduce latency of k_tile_step)
    int p = (blk * 1024 + threadIdx.x) & 0xffff;
    for (int k = 0; k < a.chain_len; ++k) p = a.chain[p];
    float* dst = a.pieces + (size_t)blk * a.piece_floats;
    for (int i = threadIdx.x; i < a.piece_floats; i += 1024) dst[i] = (float)(p & 7) + (float)a.step;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // uncached stores: just drain them
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int f = 0; f < a.nflags; ++f) __hip_atomic_fetch_add(a.flag + 32 * f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.stamps[blk] = wall_clock64();
    }
}

__device__ __forceinline__ void part_b(const Args& a, int blk, bool wait) {
    const int D = a.D;
    const int row = blk * a.rows_per_block + threadIdx.x / 16;
    const int d0 = (threadIdx.x % 16) * 4;
    float4 w = *reinterpret_cast<const float4*>(a.table + (size_t)row * D + d0);   // preloaded before the wait
    if (wait) {
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            const unsigned target = (unsigned)a.na * (unsigned)(a.step + 1);
            const unsigned* fl = a.flag + 32 * (blk % a.nflags);
            while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1u << 20)) { atomicExch(a.err, 1); break; }
            }
            if (blk == 0) a.stamps[64] = wall_clock64();
        }
        __syncthreads();
    }
    // read "pieces" from every part-A block for this row (uncached memory)
    float4 acc = w;
    for (int t = 0; t < a.na; ++t) {
        const float* src = a.pieces + (size_t)t * a.piece_floats + ((size_t)(row * D + d0) % (size_t)(a.piece_floats - 4));
        float4 x = *reinterpret_cast<const float4*>((const float*)((size_t)src & ~(size_t)15));
        acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
    }
    *reinterpret_cast<float4*>(a.table + (size_t)row * D + d0) = acc;
}

__global__ __launch_bounds__(1024) void k_fused(Args a) {
    if ((int)blockIdx.x < a.na) { part_a(a, blockIdx.x); return; }
    if (threadIdx.x >= 256) return;
    part_b(a, blockIdx.x - a.na, true);
}
__global__ __launch_bounds__(1024) void k_a(Args a) { part_a(a, blockIdx.x); }
__global__ __launch_bounds__(256) void k_b(Args a) { part_b(a, blockIdx.x, false); }

int main() {
    const int NA = 10, D = 64, ROWS = 9984, RPB = 16, NB = ROWS / RPB;
    const int PF = 65536;                                 // 256 KB of pieces per part-A block
    int* chain; float *pieces, *table; unsigned* flag; unsigned long long* stamps; int* err;
    CHK(hipMalloc(&chain, 65536 * 4));
    { int* h = (int*)malloc(65536 * 4); for (int i = 0; i < 65536; ++i) h[i] = (i * 40503 + 977) & 0xffff; CHK(hipMemcpy(chain, h, 65536 * 4, hipMemcpyHostToDevice)); free(h); }
    CHK(hipExtMallocWithFlags((void**)&pieces, (size_t)NA * PF * 4, hipDeviceMallocUncached));
    CHK(hipExtMallocWithFlags((void**)&flag, 4096, hipDeviceMallocUncached));
    CHK(hipMalloc(&table, (size_t)ROWS * D * 4)); CHK(hipMemset(table, 0, (size_t)ROWS * D * 4));
    CHK(hipMalloc(&stamps, 1024)); CHK(hipMalloc(&err, 4)); CHK(hipMemset(err, 0, 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int chain_len : {8, 16, 24}) {
        for (int nflags : {1, 8}) {
            Args a = {chain, pieces, table, flag, stamps, err, NA, chain_len, PF, RPB, D, 0, nflags};
            const int steps = 200;
            float best_f = 1e9f, best_s = 1e9f, best_a = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CHK(hipMemset(flag, 0, 4096)); CHK(hipDeviceSynchronize());
                CHK(hipEventRecord(e0));
                for (int s = 0; s < steps; ++s) { a.step = s; hipLaunchKernelGGL(k_fused, dim3(NA + NB), dim3(1024), 0, 0, a); }
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_f) best_f = ms;
                CHK(hipEventRecord(e0));
                for (int s = 0; s < steps; ++s) { a.step = s; hipLaunchKernelGGL(k_a, dim3(NA), dim3(1024), 0, 0, a); hipLaunchKernelGGL(k_b, dim3(NB), dim3(256), 0, 0, a); }
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_s) best_s = ms;
                CHK(hipEventRecord(e0));
                for (int s = 0; s < steps; ++s) { a.step = s; hipLaunchKernelGGL(k_a, dim3(NA), dim3(1024), 0, 0, a); }
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_a) best_a = ms;
            }
            int herr = 0; CHK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            unsigned long long st[65]; CHK(hipMemcpy(st, stamps, sizeof(st), hipMemcpyDeviceToHost));
            unsigned long long last = 0; for (int i = 0; i < NA; ++i) if (st[i] > last) last = st[i];
            printf("chain %2d flags %d: part A alone %.2f us/step; two launches %.2f us/step; one launch with the counter %.2f us/step%s\n",
                   chain_len, nflags, best_a * 1000 / steps, best_s * 1000 / steps, best_f * 1000 / steps, herr ? "  (SPIN LIMIT HIT)" : "");
        }
    }
    return 0;
}
