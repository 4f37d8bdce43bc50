// mt_blocks.hip - how fast can ONE workgroup run the MT19937 recurrence (k_mt_blocks of rng.hip)?
//   hipcc --offload-arch=gfx950 -O3 -I ../../tf-recomm_amd/csrc -I ../../include -o mt_blocks mt_blocks.hip && ./mt_blocks
// Variant A: the product kernel (640 threads, one word each, the form of the regeneration chosen per wave).
// Variant B: 320 threads, two words each (i and i + 320), one masked three-term form for every lane (no branches).
#include "../../tf-recomm_amd/csrc/rng.hip"
#include <cstdio>
#include <vector>
using namespace tfr;

__device__ __forceinline__ uint32_t word_masked(const uint32_t* o, int i) {
    // new[i] = o[base] ^ tw(o[i], o[i+1]) ^ [i >= 227] tw(o[i-227], o[i-226]) ^ [i >= 454] tw(o[i-454], o[i-453]),  base = i+397 / i+170 / i-57
    const int base = i < 227 ? i + 397 : (i < 454 ? i + 170 : i - 57);
    const int c = i < 227 ? 0 : i - 227, e = i < 454 ? 0 : i - 454;
    const uint32_t m2 = i < 227 ? 0u : 0xffffffffu, m3 = i < 454 ? 0u : 0xffffffffu;
    const uint32_t a0 = o[i], a1 = o[i + 1], bs = o[base], c0 = o[c], c1 = o[c + 1], e0 = o[e], e1 = o[e + 1];
    return bs ^ mt_tw(a0, a1) ^ (mt_tw(c0, c1) & m2) ^ (mt_tw(e0, e1) & m3);
}

__global__ __launch_bounds__(320) void k_blocks2(const uint32_t* __restrict__ state, uint32_t* __restrict__ raw, int32_t nb) {
    __shared__ uint32_t st[2][648];
    const int t = threadIdx.x, ia = t, ib = t + 320;
    st[0][ia] = state[ia]; st[0][ib] = ib < 624 ? state[ib] : 0u;
    st[1][ia] = 0u; st[1][ib] = 0u;
    if (t < 8) { st[0][640 + t] = 0u; st[1][640 + t] = 0u; }
    raw[ia] = state[ia]; if (ib < 624) raw[ib] = state[ib];
    __syncthreads();
    int cur = 0;
    for (int32_t b = 1; b <= nb; ++b) {
        const uint32_t* o = st[cur];
        const uint32_t wa = word_masked(o, ia);
        uint32_t wb = word_masked(o, ib < 623 ? ib : 622);
        if (ib == 623) {
            const uint32_t n396 = o[566] ^ mt_tw(o[169], o[170]) ^ mt_tw(o[396], o[397]);
            const uint32_t n0 = o[397] ^ mt_tw(o[0], o[1]);
            wb = n396 ^ mt_tw(o[623], n0);
        }
        uint32_t* dst = raw + (size_t)b * 624;
        st[cur ^ 1][ia] = wa; dst[ia] = wa;
        if (ib < 624) { st[cur ^ 1][ib] = wb; dst[ib] = wb; }
        __syncthreads();
        cur ^= 1;
    }
}

int main() {
    const int NB = 4000;
    uint32_t *state, *rawA, *rawB; int32_t* hdr;
    hipMalloc(&state, 625 * 4); hipMalloc(&rawA, (size_t)(NB + 1) * 624 * 4); hipMalloc(&rawB, (size_t)(NB + 1) * 624 * 4); hipMalloc(&hdr, 16);
    std::vector<uint32_t> h(625);
    h[0] = 5489u; for (int i = 1; i < 624; ++i) h[i] = 1812433253u * (h[i - 1] ^ (h[i - 1] >> 30)) + (uint32_t)i;
    h[624] = 624;
    hipMemcpy(state, h.data(), 625 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best[2] = {1e9f, 1e9f};
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mt_blocks, dim3(1), dim3(MT_THREADS), 0, 0, state, rawA, hdr, NB);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best[0]) best[0] = ms;
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_blocks2, dim3(1), dim3(320), 0, 0, state, rawB, NB);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); if (ms < best[1]) best[1] = ms;
    }
    std::vector<uint32_t> a((size_t)(NB + 1) * 624), b((size_t)(NB + 1) * 624);
    hipMemcpy(a.data(), rawA, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), rawB, b.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0; for (size_t k = 0; k < a.size(); ++k) bad += a[k] != b[k];
    printf("%d blocks: 640 threads x 1 word %.1f us (%.3f us/block); 320 threads x 2 words %.1f us (%.3f us/block); %zu words differ\n",
           NB, best[0] * 1000, best[0] * 1000 / NB, best[1] * 1000, best[1] * 1000 / NB, bad);
    return 0;
}
