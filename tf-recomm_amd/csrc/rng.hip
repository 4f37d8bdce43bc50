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
//
// The generator is a single workgroup, so its speed is ONE CU's vector issue rate: measured (TFR_RNG_DEBUG=1,
// in-kernel s_memtime at 2.4 GHz) ~1500 cycles per 624-word block = ~1 ns per draw, i.e. ~14 us for a
// 10000-id batch of a 900188-rating store and ~400 us for 262144 ids of a 90M store - in both cases less than the
// training step that consumes them, and the draws run on their own stream ahead of the steps.  In-kernel stamps
// showed each wave spending ~10 cycles per instruction (ten waves share four SIMDs), so what counts is the
// instruction total: the three forms of the regeneration (one, two or three tw terms) are dealt out per WAVE (scalar
// branches on the wave index, all LDS reads of a form issued together, one LDS round trip per block), the two waves
// that straddle a form boundary use masked variants, the block's last word (five tw terms) is computed on the
// lightest SIMD, and block k's ranking / stores overlap block k+2's regeneration (software pipeline).
// Tried and dropped: the same generator on four waves (one per SIMD) with three words per thread, to pay the per-block fixed
// costs four times instead of ten - 1800-1900 cycles per block against 1290-1630 here (one gpurun call, same box).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include "svd_kernels.h"

namespace tfr {

__device__ __forceinline__ uint32_t mt_tw(uint32_t a, uint32_t b) {
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1u)) & 0x9908b0dfu);
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// word i of the block that follows block o (the in-place regeneration loop of genrand, unrolled so
// that only old words appear on the right-hand side).  Reference form; the kernel below evaluates the
// same expressions with the case split made per wave.
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
constexpr int MT_PAD = 648;                              // words per buffer: lanes 624..639 read (and ignore) the padding

__global__ __launch_bounds__(MT_THREADS) void k_mt_draw(uint32_t* __restrict__ state, int64_t* __restrict__ out,
                                                        int32_t need, uint32_t rng, uint32_t mask, unsigned long long* __restrict__ dbg,
                                                        const int32_t* __restrict__ rest) {
    if (rest) {                                          // the tail of a wide draw that came up short: {ids still to draw, ids already out}
        need = rest[0];
        out += rest[1];
        if (need <= 0) return;                           // (the usual case: nothing left)
    }
    const unsigned long long t_c0 = dbg ? __builtin_amdgcn_s_memtime() : 0ull, t_r0 = dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    __shared__ uint32_t st[2][MT_PAD];
    __shared__ int32_t cnt[2][16];
    const int i = threadIdx.x, lane = i & 63;
    const int wave = __builtin_amdgcn_readfirstlane(i >> 6);
    const bool own = i < 624;
    st[0][i] = own ? state[i] : 0u;
    st[1][i] = 0u;
    if (i < 8) { st[0][640 + i] = 0u; st[1][640 + i] = 0u; }
    int start = (int)state[624];
    __syncthreads();
    // per-lane constants of the masked forms: wave 3 holds words 192..255 (form 1 below 227, form 2 from 227),
    // wave 7 holds 448..511 (form 2 below 454, form 3 from 454)
    const int b3 = i < 227 ? i + 397 : i + 170;          // wave 3: base word
    const int p3 = i < 227 ? 0 : i - 227;                //         second pair (masked off below 227)
    const uint32_t m3 = i < 227 ? 0u : 0xffffffffu;
    const int b7 = i < 454 ? i + 170 : i - 57;           // wave 7: base word
    const int p7 = i < 454 ? 0 : i - 454;                //         third pair (masked off below 454)
    const uint32_t m7 = i < 454 ? 0u : 0xffffffffu;
    // word i of the block after block o (the forms dealt out per wave; see mt_next_word for the plain statement).
    // Every form reads all its operands before it combines them: one LDS round trip per block.  The block's last
    // word (two NEW words among its inputs: five tw terms) is computed by wave 2 - the lightest SIMD (waves 2 and 6) -
    // and returned in `last`; its lane 0 stores it.
    auto regen = [&](const uint32_t* o, uint32_t& w, uint32_t& last) -> uint32_t {
        if (wave < 3) {                                                                        // words 0..191
            const uint32_t a0 = o[i], a1 = o[i + 1], bs = o[i + 397];
            w = a0;
            if (wave == 2) {
                const uint32_t x566 = o[566], x169 = o[169], x170 = o[170], x396 = o[396], x397 = o[397], x0 = o[0], x1 = o[1], x623 = o[623];
                const uint32_t n396 = x566 ^ mt_tw(x169, x170) ^ mt_tw(x396, x397);
                const uint32_t n0 = x397 ^ mt_tw(x0, x1);
                last = n396 ^ mt_tw(x623, n0);
            }
            return bs ^ mt_tw(a0, a1);
        }
        if (wave == 3) {                                                                       // 192..255
            const uint32_t a0 = o[i], a1 = o[i + 1], bs = o[b3], c0 = o[p3], c1 = o[p3 + 1];
            w = a0;
            return bs ^ mt_tw(a0, a1) ^ (mt_tw(c0, c1) & m3);
        }
        if (wave < 7) {                                                                        // 256..447
            const uint32_t a0 = o[i], a1 = o[i + 1], bs = o[i + 170], c0 = o[i - 227], c1 = o[i - 226];
            w = a0;
            return bs ^ mt_tw(a0, a1) ^ mt_tw(c0, c1);
        }
        if (wave == 7) {                                                                       // 448..511
            const uint32_t a0 = o[i], a1 = o[i + 1], bs = o[b7], c0 = o[i - 227], c1 = o[i - 226], e0 = o[p7], e1 = o[p7 + 1];
            w = a0;
            return bs ^ mt_tw(a0, a1) ^ mt_tw(c0, c1) ^ (mt_tw(e0, e1) & m7);
        }
        const int c = own ? i : 623;                     // 512..623 (lanes 624..639 compute padding; word 623 comes from wave 2)
        const uint32_t a0 = o[i], a1 = o[i + 1], bs = o[c - 57], c0 = o[c - 227], c1 = o[c - 226], e0 = o[c - 454], e1 = o[c - 453];
        w = a0;
        return bs ^ mt_tw(a0, a1) ^ mt_tw(c0, c1) ^ mt_tw(e0, e1);
    };
    // Software pipeline: while block `cur`'s accepted draws are ranked and stored, the block after the next is already
    // being regenerated - the generator is a chain of dependent LDS round trips, and all ten waves walk it in step, so
    // the only latency hiding there is comes from putting independent work side by side.
    //   registers: val / acc / rank of block cur;   st[cur ^ 1]: block cur + 1;   cnt[cur]: block cur's wave counts
    int cur = 0;
    int32_t produced = 0;
    uint32_t val;
    bool acc;
    int rank;
    {
        uint32_t w, last = 0;
        const uint32_t nxt = regen(st[0], w, last);
        val = mt_temper(w) & mask;
        acc = own && i >= start && val <= rng;
        const unsigned long long bal = __ballot(acc);
        rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        __syncthreads();                                 // every wave has read its words of st[0] and st[1] is unused so far
        if (i < 623) st[1][i] = nxt;
        if (i == 128) st[1][623] = last;
        if (lane == 0) cnt[0][wave] = __popcll(bal);
        __syncthreads();
    }
    while (need > 0) {
        // (1) block cur + 2 from block cur + 1
        int c = lane < MT_WAVES ? cnt[cur][lane] : 0;    // block cur's wave counts: read in the same LDS round trip
        uint32_t w2, last2 = 0;
        const uint32_t nxt2 = regen(st[cur ^ 1], w2, last2);
        // (2) block cur: prefix over the ten wave counts (lane q holds count q; inclusive scan by row shifts), then the ids
        c += __builtin_amdgcn_update_dpp(0, c, 0x111, 0xf, 0xf, true);     // row_shr:1
        c += __builtin_amdgcn_update_dpp(0, c, 0x112, 0xf, 0xf, true);     // row_shr:2
        c += __builtin_amdgcn_update_dpp(0, c, 0x114, 0xf, 0xf, true);     // row_shr:4
        c += __builtin_amdgcn_update_dpp(0, c, 0x118, 0xf, 0xf, true);     // row_shr:8
        const int total = __builtin_amdgcn_readlane(c, MT_WAVES - 1);
        const int before = wave ? __builtin_amdgcn_readlane(c, wave - 1) : 0;
        const int32_t off = produced + before + rank;
        if (acc && off < need) {
            out[off] = (int64_t)val;
            if (off == need - 1) state[624] = (uint32_t)(i + 1);     // the draw after the last one consumed
        }
        produced += total;
        if (produced >= need) break;                     // block `cur` (still intact in st[cur]) is the state to keep
        // (3) block cur + 1 becomes the current one
        val = mt_temper(w2) & mask;
        acc = own && val <= rng;
        const unsigned long long bal = __ballot(acc);
        rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (i < 623) st[cur][i] = nxt2;
        if (i == 128) st[cur][623] = last2;
        if (lane == 0) cnt[cur ^ 1][wave] = __popcll(bal);
        __syncthreads();
        cur ^= 1;
    }
    if (need > 0 && own) state[i] = st[cur][i];
    if (dbg && i == 0) {                                 // diagnostic only (TFR_RNG_DEBUG): shader cycles, 100 MHz ticks
        dbg[0] = __builtin_amdgcn_s_memtime() - t_c0;
        dbg[1] = __builtin_amdgcn_s_memrealtime() - t_r0;
    }
}

// ---- the wide form, for draws of tens of thousands of ids and more ----------------------------------------------------
// Measured (one gpurun call, TFR_RNG_WIDE=0/1): 262144 ids of a 99M store 326 -> 214 us; the C3-shaped step at dim 32, which the
// draw bounds, 330 -> 235 us; the headline step (chunks of up to 60000 ids) 20.2 -> 19.5 us.  What is left is the recurrence
// itself: ~700-900 cycles per 624-word block for ten waves in lock step on one CU (LDS round trip + ~35 instructions + barrier).
// Five waves with two words each and one masked three-term form for every lane (no per-wave branches) are slower: 0.383 against
// 0.295 us per block (tools/probes/mt_blocks.hip, same words out).
// k_mt_draw does everything on one CU and is bound by its instruction count (~1500 cycles per 624 words).  Only the
// recurrence itself is sequential: k_mt_blocks runs it alone (one LDS round trip, ~40 instructions and one barrier per
// block) and writes the raw blocks to memory; tempering, the rejection test and the compaction are then ordinary
// data-parallel work over those blocks - k_mt_count (accepted draws per block) and k_mt_emit (each block adds up the counts
// before it, ranks its own accepted draws and stores them; the block that holds the last id asked for leaves the state
// where NumPy's would be).  The number of blocks is fixed on the host from the acceptance rate plus six standard
// deviations; if the stream still comes up short (never observed) the last block records how much is missing and a
// k_mt_draw launch, which always follows and normally returns at once, draws the rest from there.
struct MtForms {                                         // the per-wave case split of the regeneration (see k_mt_draw)
    int i, wave, b3, p3, b7, p7;
    uint32_t m3, m7;
    bool own;
    __device__ __forceinline__ explicit MtForms(int tid) {
        i = tid; wave = __builtin_amdgcn_readfirstlane(tid >> 6); own = tid < 624;
        b3 = i < 227 ? i + 397 : i + 170; p3 = i < 227 ? 0 : i - 227; m3 = i < 227 ? 0u : 0xffffffffu;
        b7 = i < 454 ? i + 170 : i - 57;  p7 = i < 454 ? 0 : i - 454; m7 = i < 454 ? 0u : 0xffffffffu;
    }
    __device__ __forceinline__ uint32_t next(const uint32_t* o, uint32_t& last) const {
        if (wave < 3) {
            const uint32_t a0 = o[i], a1 = o[i + 1], bs = o[i + 397];
            if (wave == 2) {
                const uint32_t x566 = o[566], x169 = o[169], x170 = o[170], x396 = o[396], x397 = o[397], x0 = o[0], x1 = o[1], x623 = o[623];
                const uint32_t n396 = x566 ^ mt_tw(x169, x170) ^ mt_tw(x396, x397);
                const uint32_t n0 = x397 ^ mt_tw(x0, x1);
                last = n396 ^ mt_tw(x623, n0);
            }
            return bs ^ mt_tw(a0, a1);
        }
        if (wave == 3) {
            const uint32_t a0 = o[i], a1 = o[i + 1], bs = o[b3], c0 = o[p3], c1 = o[p3 + 1];
            return bs ^ mt_tw(a0, a1) ^ (mt_tw(c0, c1) & m3);
        }
        if (wave < 7) {
            const uint32_t a0 = o[i], a1 = o[i + 1], bs = o[i + 170], c0 = o[i - 227], c1 = o[i - 226];
            return bs ^ mt_tw(a0, a1) ^ mt_tw(c0, c1);
        }
        if (wave == 7) {
            const uint32_t a0 = o[i], a1 = o[i + 1], bs = o[b7], c0 = o[i - 227], c1 = o[i - 226], e0 = o[p7], e1 = o[p7 + 1];
            return bs ^ mt_tw(a0, a1) ^ mt_tw(c0, c1) ^ (mt_tw(e0, e1) & m7);
        }
        const int c = own ? i : 623;
        const uint32_t a0 = o[i], a1 = o[i + 1], bs = o[c - 57], c0 = o[c - 227], c1 = o[c - 226], e0 = o[c - 454], e1 = o[c - 453];
        return bs ^ mt_tw(a0, a1) ^ mt_tw(c0, c1) ^ mt_tw(e0, e1);
    }
};

// raw[0..624) = the current block (words from hdr[0] = pos on are still unconsumed), raw[b * 624 ..) = the b-th block after
// it, b = 1..nb; hdr = {pos, 0, 0, 0} ({.., ids still to draw, ids already out} of a short stream start at "nothing").
__global__ __launch_bounds__(MT_THREADS) void k_mt_blocks(const uint32_t* __restrict__ state, uint32_t* __restrict__ raw,
                                                          int32_t* __restrict__ hdr, int32_t nb) {
    __shared__ uint32_t st[2][MT_PAD];
    const MtForms f(threadIdx.x);
    const int i = threadIdx.x;
    const uint32_t w0 = f.own ? state[i] : 0u;
    st[0][i] = w0;
    st[1][i] = 0u;
    if (i < 8) { st[0][640 + i] = 0u; st[1][640 + i] = 0u; }
    if (f.own) raw[i] = w0;
    if (i < 4) hdr[i] = i == 0 ? (int32_t)state[624] : 0;
    __syncthreads();
    int cur = 0;
    for (int32_t b = 1; b <= nb; ++b) {
        uint32_t last = 0;
        const uint32_t nxt = f.next(st[cur], last);
        uint32_t* dst = raw + (size_t)b * 624;
        if (i < 623) { st[cur ^ 1][i] = nxt; dst[i] = nxt; }
        if (i == 128) { st[cur ^ 1][623] = last; dst[623] = last; }
        __syncthreads();                                 // block b complete in st[cur ^ 1]; every wave is done reading st[cur]
        cur ^= 1;
    }
}

// accepted draws of raw block blockIdx.x (block 0: from word pos on)
__global__ __launch_bounds__(MT_THREADS) void k_mt_count(const uint32_t* __restrict__ raw, const int32_t* __restrict__ hdr,
                                                         int32_t* __restrict__ counts, uint32_t rng, uint32_t mask) {
    __shared__ int32_t wc[MT_WAVES];
    const int i = threadIdx.x, lane = i & 63, wave = i >> 6, b = blockIdx.x;
    const int start = b == 0 ? hdr[0] : 0;
    const uint32_t val = i < 624 ? mt_temper(raw[(size_t)b * 624 + i]) & mask : 0xffffffffu;
    const bool acc = i < 624 && i >= start && val <= rng;
    const unsigned long long bal = __ballot(acc);
    if (lane == 0) wc[wave] = __popcll(bal);
    __syncthreads();
    if (i == 0) {
        int32_t t = 0;
#pragma unroll
        for (int w = 0; w < MT_WAVES; ++w) t += wc[w];
        counts[b] = t;
    }
}

// out[..] = the accepted draws in stream order; the state NumPy would be left with
__global__ __launch_bounds__(MT_THREADS) void k_mt_emit(const uint32_t* __restrict__ raw, int32_t* __restrict__ hdr,
                                                        const int32_t* __restrict__ counts, uint32_t* __restrict__ state,
                                                        int64_t* __restrict__ out, int32_t need, int32_t nb, uint32_t rng, uint32_t mask) {
    __shared__ int32_t wc[MT_WAVES];
    __shared__ int32_t red[MT_WAVES];
    __shared__ int32_t fin;
    const int i = threadIdx.x, lane = i & 63, wave = i >> 6, b = blockIdx.x;
    if (i == 0) fin = 0;
    int32_t before = 0;                                  // accepted draws of the blocks before this one
    for (int q = i; q < b; q += MT_THREADS) before += counts[q];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) before += __shfl_down(before, o, 64);
    if (lane == 0) red[wave] = before;
    const int start = b == 0 ? hdr[0] : 0;
    const uint32_t word = i < 624 ? raw[(size_t)b * 624 + i] : 0u;
    const uint32_t val = i < 624 ? mt_temper(word) & mask : 0xffffffffu;
    const bool acc = i < 624 && i >= start && val <= rng;
    const unsigned long long bal = __ballot(acc);
    const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
    if (lane == 0) wc[wave] = __popcll(bal);
    __syncthreads();
    int32_t base = 0, mine = 0;
#pragma unroll
    for (int w = 0; w < MT_WAVES; ++w) { base += red[w]; if (w < wave) mine += wc[w]; }
    const int32_t off = base + mine + rank;
    if (acc && off < need) {
        out[off] = (int64_t)val;
        if (off == need - 1) fin = i + 1;                // the draw after the last one consumed
    }
    __syncthreads();
    int32_t total = base;
#pragma unroll
    for (int w = 0; w < MT_WAVES; ++w) total += wc[w];
    const bool short_end = b == nb && total < need;      // the whole stream of nb blocks holds fewer than `need` ids
    if (fin || short_end) {                              // block-uniform
        if (i < 624) state[i] = word;
        if (i == 0) {
            state[624] = fin ? (uint32_t)fin : 624u;
            if (short_end) { hdr[2] = need - total; hdr[3] = total; }
        }
    }
}

void launch_mt_draw(uint32_t* d_state, int64_t* d_out, int64_t need, uint32_t rng, uint32_t mask, hipStream_t s,
                    unsigned long long* d_dbg, const MtScratch* ws) {
    // one launch draws < 2^31 ids (32-bit offsets inside the kernel); larger requests are cut here - the
    // state carries over from launch to launch on the stream
    const double p = ((double)rng + 1.0) / ((double)mask + 1.0);             // acceptance rate of one word, in (0.5, 1]
    while (need > 0) {
        if (ws && ws->raw && need >= MT_WIDE_MIN && !d_dbg) {
            int64_t n = need < MT_WIDE_MAX ? need : MT_WIDE_MAX;
            // words for n ids: n / p on average, standard deviation sqrt(n (1 - p)) / p; the current block may be spent
            int64_t nb = (int64_t)(((double)n + 6.0 * sqrt((double)n * (1.0 - p)) + 8.0) / p / 624.0) + 2;
            {   // TFR_RNG_WIDE_TRIM=<percent>: generate only that share of the blocks, so that the stream comes up short and the
                // k_mt_draw tail has work to do (tests)
                static int trim = -1;
                if (trim < 0) { const char* e = getenv("TFR_RNG_WIDE_TRIM"); trim = e ? atoi(e) : 100; }
                if (trim > 0 && trim < 100) { nb = nb * trim / 100; if (nb < 1) nb = 1; }
            }
            if (nb + 1 > ws->cap_blocks) {                                   // scratch sized for MT_WIDE_MAX ids at p = 1/2
                nb = ws->cap_blocks - 1;
                const int64_t fit = (int64_t)((double)(nb - 2) * 624.0 * p * 0.98);
                if (fit < n) n = fit;
            }
            hipLaunchKernelGGL(k_mt_blocks, dim3(1), dim3(MT_THREADS), 0, s, d_state, ws->raw, ws->hdr, (int32_t)nb);
            hipLaunchKernelGGL(k_mt_count, dim3((unsigned)(nb + 1)), dim3(MT_THREADS), 0, s, ws->raw, ws->hdr, ws->counts, rng, mask);
            hipLaunchKernelGGL(k_mt_emit, dim3((unsigned)(nb + 1)), dim3(MT_THREADS), 0, s, ws->raw, ws->hdr, ws->counts, d_state, d_out,
                               (int32_t)n, (int32_t)nb, rng, mask);
            hipLaunchKernelGGL(k_mt_draw, dim3(1), dim3(MT_THREADS), 0, s, d_state, d_out, 0, rng, mask, (unsigned long long*)nullptr,
                               (const int32_t*)(ws->hdr + 2));
            d_out += n;
            need -= n;
            continue;
        }
        const int64_t n = need < ((int64_t)1 << 30) ? need : ((int64_t)1 << 30);
        hipLaunchKernelGGL(k_mt_draw, dim3(1), dim3(MT_THREADS), 0, s, d_state, d_out, (int32_t)n, rng, mask, d_dbg, (const int32_t*)nullptr);
        d_out += n;
        need -= n;
    }
}

}  // namespace tfr
