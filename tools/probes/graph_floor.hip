// graph_floor.hip - what a dependent kernel launch costs on the GPU when the chain is submitted launch by launch on a stream and
// when the same chain is captured into a hipGraph and submitted at once (the headline step is two dependent launches of ~5 and
// ~2 us of work; the guide suggests graphs for launch-bound loops).  Build: hipcc --offload-arch=gfx950 -O3 graph_floor.hip -o graph_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_work(float* p, int n, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = p[i];
    for (int k = 0; k < iters; ++k) x = x * 1.0001f + 0.5f;
    p[i] = x;
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

int main() {
    const int n = 1 << 20, pairs = 200;
    float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemset(d, 0, n * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int itA : {0, 400, 1500}) {
        const int itB = itA / 3;
        auto chain = [&]() {
            for (int k = 0; k < pairs; ++k) {
                hipLaunchKernelGGL(k_work, dim3(256), dim3(1024), 0, s, d, n, itA);
                hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, d, n / 4, itB);
            }
        };
        // a single launch of each, for the work itself
        float a_ms = 0, b_ms = 0, t = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s)); hipLaunchKernelGGL(k_work, dim3(256), dim3(1024), 0, s, d, n, itA); CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&a_ms, e0, e1));
            CK(hipEventRecord(e0, s)); hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, d, n / 4, itB); CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&b_ms, e0, e1));
        }
        float best_stream = 1e9f, best_graph = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, s)); chain(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&t, e0, e1)); if (t < best_stream) best_stream = t;
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); chain(); CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&t, e0, e1)); if (t < best_graph) best_graph = t;
        }
        printf("work per kernel (event pair around one launch): A %.2f us, B %.2f us | per pair of dependent launches: stream %.2f us, graph %.2f us\n",
               a_ms * 1e3, b_ms * 1e3, best_stream * 1e3 / pairs, best_graph * 1e3 / pairs);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
