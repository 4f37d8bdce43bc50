// rng.hip - the id draw of dataio.ShuffleIterator on the device.
//
// dataio.py:115  ids = np.random.randint(0, len(self.inputs), (self.batch_size,))  draws from NumPy's
// legacy global RandomState (MT19937, seeded np.random.seed(13575) at svd_train_val.py:15).  For a
// range that fits 32 bits NumPy's legacy randint is [NumPy-lib: _bounded_integers / distributions.c,
// masked rejection]:  mask = smallest 2^k - 1 >= high - 1;  per output, draw 32-bit Mersenne-Twister
// words until (word & mask) <= high - 1.  k_mt_draw replays exactly that stream, so the minibatch ids
// never leave the GPU and index work stays bit-exact (tests: against NumPy itself and against
// tests/golden/iter_streams.npz, which the reference's own dataio.py produced).
//
// MT19937 is one sequential recurrence, x[n+624] = x[n+397] ^ tw(x[n], x[n+1]), but a whole 624-word
// block is a function of the previous block alone: substituting the recurrence into itself gives every
// word of the new block from <= 7 words of the old one (up to three tw terms), so ONE workgroup
// regenerates a block in one LDS round trip and one barrier.  The same 624 threads temper their word,
// apply the rejection test and compact the accepted values in stream order (ballot ranks inside a
// wave, a 10-entry wave-count table across waves).  The barrier that publishes the new block also
// publishes the wave counts: one barrier per 624 draws.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "svd_kernels.h"

namespace tfr {

__device__ __forceinline__ uint32_t mt_tw(uint32_t a, uint32_t b) {
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// word i of the block that follows block o (the in-place regeneration loop of genrand, unrolled so
// that only old words appear on the right-hand side)
__device__ __forceinline__ uint32_t mt_next_word(const uint32_t* o, int i) {
    if (i < 227) return o[i + 397] ^ mt_tw(o[i], o[i + 1]);
    if (i < 454) return o[i + 170] ^ mt_tw(o[i - 227], o[i - 226]) ^ mt_tw(o[i], o[i + 1]);
    if (i < 623) return o[i - 57] ^ mt_tw(o[i - 454], o[i - 453]) ^ mt_tw(o[i - 227], o[i - 226]) ^ mt_tw(o[i], o[i + 1]);
    const uint32_t n396 = o[566] ^ mt_tw(o[169], o[170]) ^ mt_tw(o[396], o[397]);
    const uint32_t n0 = o[397] ^ mt_tw(o[0], o[1]);
    return n396 ^ mt_tw(o[623], n0);
}

// state: key[624] then pos (0..624, 624 = block exhausted), the layout of np.random.get_state()[1:3].
// Draws `need` accepted values into out[0..need) (as int64, like NumPy) and leaves the state exactly
// where NumPy's would be.  One block of MT_THREADS threads; every thread reaches every barrier and
// the loop ends for all threads at once (produced is computed identically by all of them).
constexpr int MT_THREADS = 640;
constexpr int MT_WAVES = MT_THREADS / 64;

__global__ __launch_bounds__(MT_THREADS) void k_mt_draw(uint32_t* __restrict__ state, int64_t* __restrict__ out,
                                                        int64_t need, uint32_t rng, uint32_t mask) {
    __shared__ uint32_t st[2][624];
    __shared__ int32_t cnt[2][MT_WAVES];
    const int i = threadIdx.x, lane = i & 63, wave = i >> 6;
    const bool own = i < 624;
    if (own) st[0][i] = state[i];
    int start = (int)state[624];
    __syncthreads();
    int cur = 0, par = 0;
    int64_t produced = 0;
    while (produced < need) {
        uint32_t w = 0, nxt = 0;
        if (own) { w = st[cur][i]; nxt = mt_next_word(st[cur], i); }
        const uint32_t val = mt_temper(w) & mask;
        const bool acc = own && i >= start && val <= rng;
        const unsigned long long bal = __ballot(acc);
        const int rank = __popcll(bal & ((1ull << lane) - 1ull));
        if (own) st[cur ^ 1][i] = nxt;
        if (lane == 0) cnt[par][wave] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int q = 0; q < MT_WAVES; ++q) {
            const int c = cnt[par][q];
            if (q < wave) before += c;
            total += c;
        }
        const int64_t off = produced + before + rank;
        if (acc && off < need) {
            out[off] = (int64_t)val;
            if (off == need - 1) state[624] = (uint32_t)(i + 1);     // the draw after the last one consumed
        }
        produced += total;
        if (produced >= need) break;                     // block `cur` is the state to keep
        cur ^= 1; par ^= 1; start = 0;
    }
    if (need > 0 && own) state[i] = st[cur][i];
}

void launch_mt_draw(uint32_t* d_state, int64_t* d_out, int64_t need, uint32_t rng, uint32_t mask, hipStream_t s) {
    hipLaunchKernelGGL(k_mt_draw, dim3(1), dim3(MT_THREADS), 0, s, d_state, d_out, need, rng, mask);
}

}  // namespace tfr
