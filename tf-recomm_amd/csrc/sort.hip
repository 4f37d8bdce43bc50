// sort.hip - stable key sort of batch positions by row id (the integer half of the
// deterministic segmented scatter).  Restates tf.unique + the batch-order walk of
// unsorted_segment_sum [TF1-lib] as "stable sort by id, then walk each run in order".
//
// Two paths, both stable, both pure integer work (bit-exact against np.argsort(kind="stable")):
//   csort  - hand-written one-pass counting sort for small tables (every row id is its own
//            bin, all bins in LDS): 3 launches sort BOTH id columns.  This is the
//            MovieLens-scale path, where launch count, not bytes, bounds the step.
//   rsort  - hand-written LSD radix sort (8-bit digits) for big tables: 3 launches per pass,
//            ceil(bits/8) passes, both columns per launch.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include "svd_kernels.h"
#include "finalize.inc.h"

namespace tfr {

// ------------------------------------------------------------------------------------
// csort pass 1: per tile of CSORT_TILE batch entries, the rank of every entry among the
// equal keys that precede it in the tile, and the tile's histogram.
//   in-wave: 64 lanes find their equal-key peers with one ballot per key bit;
//   across the 16 waves of the tile: waves take turns (in order) bumping the LDS counter of
//   their keys, so ranks follow batch order -> the sort is stable.
__global__ __launch_bounds__(CSORT_TILE) void k_csort_rank(CSortArgs a) {
    warm_args(a);
    extern __shared__ int32_t cnt[];
    const int col = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const int nb = a.nbins[col];
    for (int b = tid; b < nb; b += CSORT_TILE) cnt[b] = 0;
    __syncthreads();
    const int64_t k = (int64_t)tile * CSORT_TILE + tid;
    const bool valid = k < a.B;
    const int32_t key = valid ? (a.keys[col][k] & (nb - 1)) : 0;
    unsigned long long mask = __ballot(valid);
    for (int bit = 1; bit < nb; bit <<= 1) {
        const unsigned long long m = __ballot((key & bit) != 0);
        mask &= (key & bit) ? m : ~m;
    }
    const int lane = tid & 63, wave = tid >> 6;
    const unsigned long long below = mask & ((1ull << lane) - 1ull);
    const int rank_in_wave = __popcll(below);
    const int group_size = __popcll(mask);
    int base = 0;
    for (int w = 0; w < CSORT_TILE / 64; ++w) {
        if (wave == w && valid) {
            base = cnt[key];
            if (below == 0) cnt[key] = base + group_size;     // one writer per key per wave
        }
        __syncthreads();
    }
    if (valid) a.lrank[col][k] = base + rank_in_wave;
    for (int b = tid; b < nb; b += CSORT_TILE) a.hist[col][(size_t)tile * nb + b] = cnt[b];   // tile-major
}

// csort pass 2 (one block per column): offs[tile][bin] = number of entries with this key in
// earlier tiles; binbase[bin] = number of entries with a smaller key.  Thread t owns bins
// t, t+1024, ...: loads are coalesced across threads and all issued before the first use.
// blockIdx.y == 2 (optional) runs the step's finalize (K4) instead, saving a launch.
struct ScanFinArgs { CSortArgs c; FinArgs f; };

template <int NT>
__device__ __forceinline__ int32_t tile_prefix(const int32_t* __restrict__ h, int32_t* __restrict__ offs,
                                               int nb, int nt, int b) {
    int32_t v[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) v[t] = (t < nt) ? h[(size_t)t * nb + b] : 0;
    int32_t run = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t < nt) offs[(size_t)t * nb + b] = run;
        run += v[t];
    }
    return run;
}

__global__ __launch_bounds__(1024) void k_csort_scan(ScanFinArgs sa) {
    if (blockIdx.y == 2) {                               // the step's finalize rides in this launch
        if (blockIdx.x == 0) finalize_body(sa.f);
        return;
    }
    const CSortArgs& a = sa.c;
    __shared__ int32_t wsum[16];
    const int col = blockIdx.y, tid = threadIdx.x;
    const int nb = a.nbins[col], nt = a.ntiles;
    if ((int)blockIdx.x * 1024 >= nb) return;            // block-uniform
    const int32_t* __restrict__ h = a.hist[col];
    int32_t* __restrict__ offs = a.offs[col];
    const int lane = tid & 63, wave = tid >> 6;
    // phase A: this block's 1024 bins - per-bin prefix over the tiles
    const int b = blockIdx.x * 1024 + tid;
    int32_t tot = 0;
    if (b < nb) {
        if (nt <= 16) tot = tile_prefix<16>(h, offs, nb, nt, b);
        else {
            for (int t = 0; t < nt; ++t) {
                const int32_t v = h[(size_t)t * nb + b];
                offs[(size_t)t * nb + b] = tot;
                tot += v;
            }
        }
    }
    // phase B: exclusive scan of the bin totals inside the block; the block's grand total goes to
    // blocktot[] and the scatter kernel adds the totals of the preceding blocks.
    int32_t incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int32_t t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int32_t wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wsum[w];
    if (b < nb) a.binbase[col][b] = wbase + incl - tot;
    if (tid == 1023) a.blocktot[col][blockIdx.x] = wbase + incl;
}

// csort pass 3: scatter (key, batch position) to its sorted slot.
__global__ __launch_bounds__(CSORT_TILE) void k_csort_scatter(CSortArgs a) {
    warm_args(a);
    const int col = blockIdx.y, tile = blockIdx.x;
    const int64_t k = (int64_t)tile * CSORT_TILE + threadIdx.x;
    if (k >= a.B) return;
    const int32_t key = a.keys[col][k];
    const int nb = a.nbins[col];
    const int32_t bin = key & (nb - 1);
    int32_t base = 0;                                    // entries in the 1024-bin blocks before this one
    for (int q = 0; q < (bin >> 10); ++q) base += a.blocktot[col][q];
    const int32_t dst = base + a.binbase[col][bin] + a.offs[col][(size_t)tile * nb + bin] + a.lrank[col][k];
    a.ks[col][dst] = key;
    a.ps[col][dst] = (int32_t)k;
}

bool csort_eligible(int64_t B, int bits_u, int bits_i) {
    const int bits = bits_u > bits_i ? bits_u : bits_i;
    if (B < 1 || (1 << bits) > CSORT_MAX_BINS) return false;
    const int64_t ntiles = (B + CSORT_TILE - 1) / CSORT_TILE;
    return ((int64_t)1 << bits) * ntiles <= (1 << 20);       // keeps the one-block scan short
}

void launch_csort(const CSortArgs& a, const FinArgs* fin, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_csort_rank),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, CSORT_MAX_BINS * 4);
        attr_done = true;
    }
    const int nbmax = a.nbins[0] > a.nbins[1] ? a.nbins[0] : a.nbins[1];
    const dim3 grid(a.ntiles, 2);
    hipLaunchKernelGGL(k_csort_rank, grid, dim3(CSORT_TILE), (size_t)nbmax * 4, s, a);
    ScanFinArgs sa;
    sa.c = a;
    if (fin) sa.f = *fin; else memset(&sa.f, 0, sizeof(sa.f));
    hipLaunchKernelGGL(k_csort_scan, dim3((nbmax + 1023) / 1024, fin ? 3 : 2), dim3(1024), 0, s, sa);
    hipLaunchKernelGGL(k_csort_scatter, grid, dim3(CSORT_TILE), 0, s, a);
}

// scan (+ optional K4) and scatter only: the rank pass already ran inside k_front
void launch_csort_tail(const CSortArgs& a, const FinArgs* fin, hipStream_t s) {
    const int nbmax = a.nbins[0] > a.nbins[1] ? a.nbins[0] : a.nbins[1];
    ScanFinArgs sa;
    sa.c = a;
    if (fin) sa.f = *fin; else memset(&sa.f, 0, sizeof(sa.f));
    hipLaunchKernelGGL(k_csort_scan, dim3((nbmax + 1023) / 1024, fin ? 3 : 2), dim3(1024), 0, s, sa);
    hipLaunchKernelGGL(k_csort_scatter, dim3(a.ntiles, 2), dim3(CSORT_TILE), 0, s, a);
}

// ------------------------------------------------------------------------------------
// rsort: LSD radix sort, 8-bit digits, for tables too big for csort.  One pass = three launches
// (rank / scan / scatter) that handle up to two key columns at once (blockIdx.y).  Each pass is
// the csort idea with 256 bins: stable tile-local ranks by ballots + ordered wave turns, a
// bin-major histogram [256][ntiles], a flat exclusive scan of it, and a scatter of (key, value).
// Stable passes from the least significant digit up give a stable sort.
__global__ __launch_bounds__(CSORT_TILE) void k_rsort_rank(RSortArgs a) {
    warm_args(a);
    __shared__ int32_t cnt[256];
    const int col = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    if (tid < 256) cnt[tid] = 0;
    __syncthreads();
    const int64_t k = (int64_t)tile * CSORT_TILE + tid;
    const bool valid = k < a.B;
    int32_t key = 0;
    if (valid) {
        if (a.ids) {                                     // fused gather: this pass reads the store records itself
            int64_t id = a.ids[k];
            if ((uint64_t)id >= (uint64_t)a.N) { if (a.err) atomicOr(a.err, 2); id = 0; }
            const int4 rec = a.store[id];
            key = col == 0 ? rec.x : rec.y;
            if (col == 0) { a.u_out[k] = rec.x; a.i_out[k] = rec.y; a.r_out[k] = __int_as_float(rec.z); }
        } else {
            key = a.keys_in[col][k];
        }
    }
    const int32_t digit = (key >> a.shift) & 255;
    if (a.err && a.shift == 0) {                         // range check rides in the first pass
        const bool bad = valid && (uint32_t)key >= (uint32_t)a.limit[col];
        if (__any(bad) && (tid & 63) == 0) atomicOr(a.err, 1);
    }
    unsigned long long mask = __ballot(valid);
#pragma unroll
    for (int bit = 1; bit < 256; bit <<= 1) {
        const unsigned long long m = __ballot((digit & bit) != 0);
        mask &= (digit & bit) ? m : ~m;
    }
    const int lane = tid & 63, wave = tid >> 6;
    const unsigned long long below = mask & ((1ull << lane) - 1ull);
    const int rank_in_wave = __popcll(below);
    const int group_size = __popcll(mask);
    int base = 0;
    for (int w = 0; w < CSORT_TILE / 64; ++w) {
        if (wave == w && valid) {
            base = cnt[digit];
            if (below == 0) cnt[digit] = base + group_size;
        }
        __syncthreads();
    }
    if (valid) a.lrank[col][k] = base + rank_in_wave;
    if (tid < 256) a.hist[col][(size_t)tid * a.ntiles + tile] = cnt[tid];        // bin-major
}

// exclusive scan of hist[256 * ntiles] -> offs, split over blocks of 4096 contiguous entries:
// each block scans its chunk locally (one coalesced int4 per thread) and publishes its total; the
// scatter adds the totals of the preceding blocks.
__global__ __launch_bounds__(1024) void k_rsort_scan(RSortArgs a) {
    warm_args(a);
    __shared__ int32_t wsum[16];
    const int col = blockIdx.y, tid = threadIdx.x;
    const int32_t* __restrict__ h = a.hist[col];
    int32_t* __restrict__ offs = a.offs[col];
    const int64_t total = (int64_t)256 * a.ntiles;        // a multiple of 256
    const int64_t lo = (int64_t)blockIdx.x * 4096 + (int64_t)tid * 4;      // chunk = 4096: one int4 per thread
    int4 v = make_int4(0, 0, 0, 0);
    if (lo < total) v = *reinterpret_cast<const int4*>(h + lo);
    const int32_t sum = (v.x + v.y) + (v.z + v.w);
    const int lane = tid & 63, wave = tid >> 6;
    int32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int32_t t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int32_t run = incl - sum;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    if (lo < total) *reinterpret_cast<int4*>(offs + lo) = make_int4(run, run + v.x, run + v.x + v.y, run + v.x + v.y + v.z);
    if (tid == 1023) a.blocktot[col][blockIdx.x] = run + sum;
}

__global__ __launch_bounds__(CSORT_TILE) void k_rsort_scatter(RSortArgs a) {
    warm_args(a);
    const int col = blockIdx.y, tile = blockIdx.x;
    const int64_t k = (int64_t)tile * CSORT_TILE + threadIdx.x;
    // millions of keys (FM non-zeros: 8.4 M keys = 512 scan blocks): the totals of the preceding scan blocks are
    // prefix-summed once per block in LDS instead of being added up key by key (FM training step 3.1 -> 2.56 ms)
    constexpr int PREF = 2 * CSORT_TILE;
    __shared__ int32_t pref[PREF];
    __shared__ int32_t pw[CSORT_TILE / 64];
    const int64_t nb = ((int64_t)256 * a.ntiles + a.chunk - 1) / a.chunk;
    const bool use_pref = nb > 32;                           // block-uniform
    if (use_pref) {
        const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
        const int32_t x0 = (2 * t < nb && 2 * t < PREF) ? a.blocktot[col][2 * t] : 0;
        const int32_t x1 = (2 * t + 1 < nb && 2 * t + 1 < PREF) ? a.blocktot[col][2 * t + 1] : 0;
        int32_t incl = x0 + x1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t y = __shfl_up(incl, o, 64);
            if (lane >= o) incl += y;
        }
        if (lane == 63) pw[wave] = incl;
        __syncthreads();
        int32_t run = incl - (x0 + x1);
        for (int w = 0; w < wave; ++w) run += pw[w];
        pref[2 * t] = run;                                   // exclusive: totals of the blocks before 2t
        pref[2 * t + 1] = run + x0;
        __syncthreads();
    }
    if (k >= a.B) return;
    const int32_t key = a.keys_in[col][k];
    const int32_t digit = (key >> a.shift) & 255;
    const int64_t idx = (int64_t)digit * a.ntiles + tile;
    int32_t base = 0;
    const int qn = (int)(idx / a.chunk);
    if (use_pref) {
        const int covered = qn < PREF ? qn : PREF - 1;       // pref[PREF-1] = totals of blocks 0..PREF-2
        base = pref[covered];
        for (int q = covered; q < qn; ++q) base += a.blocktot[col][q];
    } else {
        for (int q = 0; q < qn; ++q) base += a.blocktot[col][q];
    }
    const int32_t dst = base + a.offs[col][idx] + a.lrank[col][k];
    a.keys_out[col][dst] = key;
    a.vals_out[col][dst] = a.vals_in[col] ? a.vals_in[col][k] : (int32_t)k;
}

void launch_rsort_pass(const RSortArgs& a, int ncols, hipStream_t s) {
    const dim3 grid(a.ntiles, ncols);
    hipLaunchKernelGGL(k_rsort_rank, grid, dim3(CSORT_TILE), 0, s, a);
    const int64_t total = (int64_t)256 * a.ntiles;
    hipLaunchKernelGGL(k_rsort_scan, dim3((unsigned)((total + a.chunk - 1) / a.chunk), ncols), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(k_rsort_scatter, grid, dim3(CSORT_TILE), 0, s, a);
}

// ------------------------------------------------------------------------------------
// rsort, wide tiles: the same LSD pass from 65536 keys up (the big-table step's batches, FM non-zeros: 8.4 M, AUC scores).  The
// 1024-key scatter above writes 4 bytes at a time to 256 x ntiles different places - at 8.4 M keys that is 285 us per pass, four
// times what the bytes cost.  Here a 1024-thread workgroup owns a tile of 4096 keys, ranks them itself (wave-private LDS counters + ballots: no
// lrank array to write and read back), stages key + value in LDS in digit order and writes every digit's run (16 keys on
// average) as one contiguous piece.  Launches per pass: histogram, scan (the kernel above), scatter.
constexpr int RW_TILE = 4096;
constexpr int RW_KPT = 4;                                // keys per thread: a wave owns 256 contiguous keys, four rounds

__global__ __launch_bounds__(1024) void k_rsortw_hist(RSortArgs a) {
    warm_args(a);
    __shared__ int32_t cnt[256];
    const int col = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    if (tid < 256) cnt[tid] = 0;
    __syncthreads();
    const int64_t k0 = (int64_t)tile * RW_TILE + (tid >> 6) * 256 + (tid & 63);
    bool bad = false;
    // the thread's four keys in ONE round of loads (two with the fused gather: ids, then records), not one round per key: the
    // wave's loads were issued and waited for key by key, and a 64-tile launch is nothing but latency.  Keys past the end read
    // the last key's address and are dropped.
    int32_t key[RW_KPT];
    bool ok[RW_KPT];
    const int64_t last = a.B - 1;
    if (a.ids) {                                         // fused gather (first pass only): this pass reads the store records itself
        int64_t id[RW_KPT];
#pragma unroll
        for (int r = 0; r < RW_KPT; ++r) {
            const int64_t k = k0 + r * 64;
            ok[r] = k < a.B;
            id[r] = a.ids[ok[r] ? k : last];
        }
        int4 rec[RW_KPT];
#pragma unroll
        for (int r = 0; r < RW_KPT; ++r) {
            if ((uint64_t)id[r] >= (uint64_t)a.N) { if (a.err && ok[r]) atomicOr(a.err, 2); id[r] = 0; }
            rec[r] = a.store[id[r]];
        }
#pragma unroll
        for (int r = 0; r < RW_KPT; ++r) {
            const int64_t k = k0 + r * 64;
            key[r] = col == 0 ? rec[r].x : rec[r].y;
            if (col == 0 && ok[r]) { a.u_out[k] = rec[r].x; a.i_out[k] = rec[r].y; a.r_out[k] = __int_as_float(rec[r].z); }
        }
    } else {
#pragma unroll
        for (int r = 0; r < RW_KPT; ++r) {
            const int64_t k = k0 + r * 64;
            ok[r] = k < a.B;
            key[r] = a.keys_in[col][ok[r] ? k : last];
        }
    }
#pragma unroll
    for (int r = 0; r < RW_KPT; ++r) {
        if (!ok[r]) continue;
        if (a.err && a.shift == 0) bad |= (uint32_t)key[r] >= (uint32_t)a.limit[col];   // range check rides in the first pass
        atomicAdd(&cnt[(key[r] >> a.shift) & 255], 1);
    }
    if (a.err && a.shift == 0 && __any(bad) && (tid & 63) == 0) atomicOr(a.err, 1);
    __syncthreads();
    if (tid < 256) a.hist[col][(size_t)tid * a.ntiles + tile] = cnt[tid];        // bin-major, as k_rsort_scan expects
}

__global__ __launch_bounds__(1024) void k_rsortw_scatter(RSortArgs a) {
    warm_args(a);
    __shared__ int32_t wcnt[16][256];
    __shared__ int32_t dstart[256], gbase[256];
    __shared__ int32_t stage_k[RW_TILE], stage_v[RW_TILE];
    constexpr int PREF = 2048;
    __shared__ int32_t pref[PREF];
    __shared__ int32_t pw[16];
    __shared__ int32_t wsum[4];
    const int col = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t t0 = (int64_t)tile * RW_TILE;
    const int nt = (int)((a.B - t0 < RW_TILE) ? a.B - t0 : RW_TILE);       // keys in this tile
    // the tile's keys + values into registers (all loads in flight)
    int32_t key[RW_KPT], val[RW_KPT];
#pragma unroll
    for (int r = 0; r < RW_KPT; ++r) {
        const int p = wave * 256 + r * 64 + lane;
        const bool ok = p < nt;
        key[r] = ok ? a.keys_in[col][t0 + p] : 0;
        val[r] = ok ? (a.vals_in[col] ? a.vals_in[col][t0 + p] : (int32_t)(t0 + p)) : 0;
    }
    for (int q = tid; q < 16 * 256; q += 1024) (&wcnt[0][0])[q] = 0;
    // totals of the scan blocks in front of an entry of the histogram matrix: prefix them once per workgroup
    const int64_t nb = ((int64_t)256 * a.ntiles + a.chunk - 1) / a.chunk;
    {
        const int32_t x0 = (2 * tid < nb && 2 * tid < PREF) ? a.blocktot[col][2 * tid] : 0;
        const int32_t x1 = (2 * tid + 1 < nb && 2 * tid + 1 < PREF) ? a.blocktot[col][2 * tid + 1] : 0;
        int32_t incl = x0 + x1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t y = __shfl_up(incl, o, 64);
            if (lane >= o) incl += y;
        }
        if (lane == 63) pw[wave] = incl;
        __syncthreads();
        int32_t run = incl - (x0 + x1);
        for (int w = 0; w < wave; ++w) run += pw[w];
        pref[2 * tid] = run;                             // exclusive: totals of the blocks before 2 * tid
        pref[2 * tid + 1] = run + x0;
    }
    __syncthreads();
    // (A) digits of the own chunk, per wave
#pragma unroll
    for (int r = 0; r < RW_KPT; ++r)
        if (wave * 256 + r * 64 + lane < nt) atomicAdd(&wcnt[wave][(key[r] >> a.shift) & 255], 1);
    __syncthreads();
    // (B) per digit: where it starts inside the tile (dstart), where the tile's run of it starts in the output (gbase), and
    //     the waves' offsets inside the tile
    if (tid < 256) {
        int32_t tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += wcnt[w][tid];
        int32_t incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t y = __shfl_up(incl, o, 64);
            if (lane >= o) incl += y;
        }
        if (lane == 63) wsum[wave] = incl;
        dstart[tid] = incl - tot;
        const int64_t idx = (int64_t)tid * a.ntiles + tile;
        const int qn = (int)(idx / a.chunk);
        const int covered = qn < PREF ? qn : PREF - 1;
        int32_t base = pref[covered];
        for (int q = covered; q < qn; ++q) base += a.blocktot[col][q];
        gbase[tid] = base + a.offs[col][idx];
    }
    __syncthreads();
    if (tid < 256) {
        int32_t run = dstart[tid];
        for (int w = 0; w < wave; ++w) run += wsum[w];
        dstart[tid] = run;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int32_t c = wcnt[w][tid];
            wcnt[w][tid] = run;
            run += c;
        }
    }
    __syncthreads();
    // (C) rank inside the tile, round by round (stable: waves, rounds, lanes in order) -> LDS in digit order
#pragma unroll
    for (int r = 0; r < RW_KPT; ++r) {
        const bool valid = wave * 256 + r * 64 + lane < nt;
        const int32_t digit = (key[r] >> a.shift) & 255;
        unsigned long long mask = __ballot(valid);
#pragma unroll
        for (int bit = 1; bit < 256; bit <<= 1) {
            const unsigned long long mb = __ballot((digit & bit) != 0);
            mask &= (digit & bit) ? mb : ~mb;
        }
        const unsigned long long below = mask & ((1ull << lane) - 1ull);
        int32_t base = 0;
        if (valid && below == 0) {
            base = wcnt[wave][digit];
            wcnt[wave][digit] = base + __popcll(mask);
        }
        const int leader = valid ? __ffsll((long long)mask) - 1 : lane;
        base = __shfl(base, leader, 64);
        if (valid) {
            const int loc = base + __popcll(below);
            stage_k[loc] = key[r];
            stage_v[loc] = val[r];
        }
    }
    __syncthreads();
    // (D) out: consecutive threads write consecutive places of a digit's run
#pragma unroll
    for (int j = 0; j < RW_KPT; ++j) {
        const int p = tid + j * 1024;
        if (p < nt) {
            const int32_t k = stage_k[p];
            const int32_t d = (k >> a.shift) & 255;
            const int32_t dst = gbase[d] + (p - dstart[d]);
            a.keys_out[col][dst] = k;
            a.vals_out[col][dst] = stage_v[p];
        }
    }
}

bool rsortw_eligible(int64_t B) {
    static int64_t lo = -1;                              // TFR_RSORT_WIDE_MIN: smallest key count that takes the wide tiles
    if (lo < 0) { const char* e = getenv("TFR_RSORT_WIDE_MIN"); lo = e ? atoll(e) : 65536; }   // C3's 262144-key sort: 71 -> 66 us
    return B >= lo;
}

// a.ntiles / a.chunk are set here (tiles of 4096 keys); hist / offs / blocktot are the buffers of the 1024-key form (larger)
void launch_rsortw_pass(RSortArgs a, int ncols, hipStream_t s) {
    a.ntiles = (int32_t)((a.B + RW_TILE - 1) / RW_TILE);
    const dim3 grid(a.ntiles, ncols);
    hipLaunchKernelGGL(k_rsortw_hist, grid, dim3(1024), 0, s, a);
    if (a.ids) {                                         // the gathered columns are what the scatter of this pass reads
        a.keys_in[0] = a.u_out; a.keys_in[1] = a.i_out;
        a.ids = nullptr;
    }
    const int64_t total = (int64_t)256 * a.ntiles;
    hipLaunchKernelGGL(k_rsort_scan, dim3((unsigned)((total + a.chunk - 1) / a.chunk), ncols), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(k_rsortw_scatter, grid, dim3(1024), 0, s, a);
}

}  // namespace tfr
