// svd_kernels.hip - gfx950 kernels of the SVD minibatch step (wave64, no MFMA: this is a
// batched gather-dot and a segmented scatter, HBM-bound).
//
// Row geometry: one table row of D floats is spread over a lane group of G lanes, VEC
// floats per lane (G*VEC >= D), so a wave64 holds 64/G rows at once and a full group
// reads its row as one contiguous, 16-byte-per-lane burst (D=128: 32 lanes x float4 =
// 512 B).  D % 4 == 0 -> VEC=4 (global_load_dwordx4); otherwise VEC=1 (rows are then not
// 16-byte aligned; correctness path for the reference's dim=5/15).
//
// Reference semantics restated per kernel; file:line into the reference tree.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "svd_kernels.h"
#include "finalize.inc.h"

namespace tfr {

// ------------------------------------------------------------------------------------
template <int VEC> struct Frag { float v[VEC]; };

typedef float floatx4 __attribute__((ext_vector_type(4)));

// NT: non-temporal (streaming) load - rows that are read once should not displace the
// small, re-used tables (biases, ids) from L2 / Infinity Cache.
template <int VEC, bool NT = false>
__device__ __forceinline__ Frag<VEC> load_frag(const float* __restrict__ row, int d0, int D) {
    Frag<VEC> f;
    if constexpr (VEC == 4) {
        if (d0 < D) {
            floatx4 t;
            if constexpr (NT) t = __builtin_nontemporal_load(reinterpret_cast<const floatx4*>(row + d0));
            else t = *reinterpret_cast<const floatx4*>(row + d0);
            f.v[0] = t.x; f.v[1] = t.y; f.v[2] = t.z; f.v[3] = t.w;
        } else {
            f.v[0] = f.v[1] = f.v[2] = f.v[3] = 0.f;
        }
    } else {
        if constexpr (NT) f.v[0] = (d0 < D) ? __builtin_nontemporal_load(row + d0) : 0.f;
        else f.v[0] = (d0 < D) ? row[d0] : 0.f;
    }
    return f;
}

template <int VEC>
__device__ __forceinline__ void store_frag(float* __restrict__ row, int d0, int D, const Frag<VEC>& f) {
    if (d0 < D) {
        if constexpr (VEC == 4) {
            *reinterpret_cast<float4*>(row + d0) = make_float4(f.v[0], f.v[1], f.v[2], f.v[3]);
        } else {
            row[d0] = f.v[0];
        }
    }
}

template <int G>
__device__ __forceinline__ float group_sum(float x) {
    // butterfly over the G lanes of a group (G is a power of two <= 64); every lane of the
    // group ends with the same sum, in a fixed order -> deterministic.
#pragma unroll
    for (int o = G / 2; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// The same reduction staged through LDS memory instead of the cross-lane network: every lane
// parks its partial in LDS, the group's first lane reads the G values back as 16-byte vectors and
// adds them in lane order.  Kept for the A/B recorded in DESIGN.md (the butterfly is the default).
template <int G>
__device__ __forceinline__ float group_sum_lds(float x, float* wave_slot /* 64 floats of this wave */, int lane) {
    wave_slot[lane] = x;
    __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): the wave's own LDS writes have landed
    const int g0 = lane & ~(G - 1);
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < G; q += 4) {
        const float4 v = *reinterpret_cast<const float4*>(wave_slot + g0 + q);
        s += (v.x + v.y) + (v.z + v.w);
    }
    return s;
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_down(x, o, 64);
    return x;   // valid in lane 0
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// block-level sum of NV values per thread -> out[NV] written by thread 0.  NW waves per block.
template <int NV, int NW>
__device__ __forceinline__ void block_sum_store(float (&val)[NV], float* out) {
    __shared__ float red[NW][NV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const float s = wave_sum(val[c]);
        if (lane == 0) red[wave][c] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w += 4) t += (red[w][c] + red[w + 1][c]) + (red[w + 2][c] + red[w + 3][c]);
            out[c] = t;
        }
    }
}

// ------------------------------------------------------------------------------------
// K1  gather-dot forward (ops.py:13-14,37-38 embedding_lookup x4; ops.py:44-47 dot + biases)
//     MODE_INFER: logits only                          (svd_train_val.py:120-122)
//     MODE_TRAIN: + g = dcost/dlogit, per-block partial {data loss, regulariser, sum g}
//                 (ops.py:81-89 regulariser over gathered rows; ops.py:124 / 125-126 loss)
//     MODE_EVAL : per-block partial {sum (infer-rate)^2, count infer==rate}
//                 (svd_train_val.py:144-149)
// A lane group owns one rating at a time; UNR ratings are in flight per group so each
// lane has 2*UNR independent 16-byte loads outstanding.
template <int G, int VEC, int MODE, int UNR, int NW>
__device__ __forceinline__ void forward_body(const FwdArgs& a, int block, int nblocks) {
    constexpr int SPW = 64 / G;        // ratings per wave per pass; UNR passes in flight
    constexpr int SPI = SPW * UNR;     // ratings per wave-iteration
    const int lane = threadIdx.x & 63;
    const int sub = lane / G;
    const int gl = lane % G;
    const int d0 = gl * VEC;
    const int D = a.D;
    const int64_t wave_id = (int64_t)block * NW + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)nblocks * NW * SPI;
    const float mu = *a.mu;

    float acc[3] = {0.f, 0.f, 0.f};    // TRAIN: loss, reg, sum g | EVAL: sse, n_equal, -
    bool oob = false, oob_store = false;

    // The ids of iteration n+1 are fetched while the rows of iteration n are in flight, so the
    // only exposed memory round trip per iteration is the row gather itself.
    int32_t u_n[UNR], it_n[UNR];
    float rr_n[UNR];
    auto fetch_ids = [&](int64_t base) {
        if (a.ids) {
            // fused ShuffleIterator gather (dataio.py:115-117): id -> (user, item, rate) from the
            // HBM-resident store
            int64_t id[UNR];
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
                const int64_t k = base + j * SPW + sub;
                id[j] = (k < a.B) ? a.ids[k] : 0;
                if ((uint64_t)id[j] >= (uint64_t)a.N) { oob_store = true; id[j] = 0; }
            }
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
                const int4 rec = a.store[id[j]];          // one 16-byte record: {user, item, rate bits, -}
                u_n[j] = rec.x;
                it_n[j] = rec.y;
                rr_n[j] = __int_as_float(rec.z);
            }
        } else {
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
                const int64_t k = base + j * SPW + sub;
                const bool ok = k < a.B;
                u_n[j] = ok ? a.u[k] : 0;
                it_n[j] = ok ? a.it[k] : 0;
                rr_n[j] = (MODE != MODE_INFER && ok) ? a.r[k] : 0.f;
            }
        }
    };

    int64_t base = wave_id * SPI;
    if (base < a.B) fetch_ids(base);
    for (; base < a.B; base += stride) {
        int64_t k[UNR];
        int32_t u[UNR], it[UNR];
        bool ok[UNR];
        float rr[UNR];
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            k[j] = base + j * SPW + sub;
            ok[j] = k[j] < a.B;
            u[j] = u_n[j]; it[j] = it_n[j]; rr[j] = rr_n[j];
            if (a.ids && a.u_out && gl == 0 && ok[j]) { a.u_out[k[j]] = u[j]; a.it_out[k[j]] = it[j]; }   // kept for the backward
            if ((uint64_t)(int64_t)u[j] >= (uint64_t)a.U) { oob = true; u[j] = 0; }
            if ((uint64_t)(int64_t)it[j] >= (uint64_t)a.I) { oob = true; it[j] = 0; }
        }
        Frag<VEC> p[UNR], q[UNR];
        float bu_[UNR], bi_[UNR];
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            p[j] = load_frag<VEC>(a.P + (size_t)u[j] * D, d0, D);
            q[j] = load_frag<VEC>(a.Q + (size_t)it[j] * D, d0, D);
            bu_[j] = a.bu[u[j]];
            bi_[j] = a.bi[it[j]];
        }
        if (base + stride < a.B) fetch_ids(base + stride);      // next iteration's ids, behind the rows
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            float s = 0.f, sq = 0.f;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float qv = q[j].v[e];
                s = fmaf(p[j].v[e], a.item_abs ? fabsf(qv) : qv, s);
                if constexpr (MODE == MODE_TRAIN) sq = fmaf(p[j].v[e], p[j].v[e], fmaf(qv, qv, sq));
            }
            if (a.lds_reduce) {
                __shared__ float stage[NW][64];
                s = group_sum_lds<G>(s, stage[threadIdx.x >> 6], lane);
            } else {
                s = group_sum<G>(s);
            }
            const float logit = ((s + mu) + bu_[j]) + bi_[j];      // ops.py:45-47 order
            if (gl == 0 && ok[j]) {
                if constexpr (MODE == MODE_INFER) {
                    a.logits[k[j]] = logit;
                } else if constexpr (MODE == MODE_TRAIN) {
                    if (a.logits) a.logits[k[j]] = logit;
                    const float r = rr[j];
                    float g, l;
                    if (a.loss == 0) {                 // ops.py:124  l2_loss(infer - rate)
                        g = logit - r;
                        l = 0.5f * g * g;
                    } else {                           // ops.py:125-126 sigmoid cross-entropy
                        g = sigmoidf_(logit) - r;
                        l = fmaxf(logit, 0.f) - logit * r + log1pf(__expf(-fabsf(logit)));
                    }
                    a.g[k[j]] = g;
                    acc[0] += l;
                    acc[2] += g;
                    if (a.reg_bias) sq = fmaf(bu_[j], bu_[j], fmaf(bi_[j], bi_[j], sq));
                } else {
                    const float r = rr[j];
                    float inf = logit;                 // canonical infer = logits (README.md:33)
                    if (a.loss != 0) inf = rintf(sigmoidf_(logit));   // ops.py:77-78, half-to-even
                    const float d = inf - r;
                    acc[0] += d * d;
                    acc[1] += (inf == r) ? 1.f : 0.f;
                }
            }
            if constexpr (MODE == MODE_TRAIN) {
                if (ok[j]) acc[1] += 0.5f * sq;        // tf.nn.l2_loss = sum(x^2)/2
            }
        }
    }
    if (oob) atomicOr(a.err, 1);
    if (oob_store) atomicOr(a.err, 2);
    if constexpr (MODE != MODE_INFER) block_sum_store<3, NW>(acc, a.partials + (size_t)block * 4);
}

template <int G, int VEC, int MODE, int UNR>
__global__ __launch_bounds__(256) void k_forward(FwdArgs a) {
    forward_body<G, VEC, MODE, UNR, 4>(a, blockIdx.x, gridDim.x);
}

// In-LDS exclusive scan of a tile's nb bin counts into the packed form (count << 16) | start.
// Thread t owns bins t, t + 1024, ... (bank-conflict free) and those bins are CONSECUTIVE in the
// tile's sorted order (order index of bin q * 1024 + t is t * per + q): equal ids stay contiguous,
// which is all the reduce and the sweep need - the order of distinct ids inside a tile is free.
// wtot: 16 ints of scratch.  All 1024 threads call; ends with a barrier.  nb <= 16384.
__device__ __forceinline__ void tile_scan_pack(int32_t* cnt, int nb, int32_t* wtot) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (nb + 1023) / 1024;                  // <= 16
    int32_t sum = 0;
    for (int q = 0; q < per; ++q) { const int b = q * 1024 + tid; if (b < nb) sum += cnt[b]; }
    int32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int32_t t2 = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t2;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int32_t run = incl - sum;
    for (int w = 0; w < wave; ++w) run += wtot[w];
    for (int q = 0; q < per; ++q) {
        const int b = q * 1024 + tid;
        if (b < nb) { const int32_t c2 = cnt[b]; cnt[b] = (c2 << 16) | run; run += c2; }
    }
    __syncthreads();
}

// Small-table training: the forward and the counting sort's rank pass do not depend on each other,
// so they share one launch - blocks [0, nfwd) run the forward (16 waves each), the rest rank one
// (tile, column) each.  In fused-gather mode the rank blocks read the store records themselves and
// publish the gathered ids for the later passes.
template <int G, int VEC>
__global__ __launch_bounds__(1024) void k_front(FrontArgs fa) {
    if ((int)blockIdx.x < fa.nfwd) {
        forward_body<G, VEC, MODE_TRAIN, 2, 16>(fa.f, blockIdx.x, fa.nfwd);
        return;
    }
    extern __shared__ int32_t cnt[];
    const CSortArgs& a = fa.c;
    const int rb = blockIdx.x - fa.nfwd;
    const int col = rb / a.ntiles, tile = rb % a.ntiles, tid = threadIdx.x;
    const int nb = a.nbins[col];
    for (int b = tid; b < nb; b += CSORT_TILE) cnt[b] = 0;
    __syncthreads();
    const int64_t k = (int64_t)tile * CSORT_TILE + tid;
    const bool valid = k < a.B;
    int32_t key = 0, fullkey = 0;
    if (valid) {
        if (fa.f.ids) {
            int64_t id = fa.f.ids[k];
            if ((uint64_t)id >= (uint64_t)fa.f.N) id = 0;          // flagged by the forward blocks
            const int4 rec = fa.f.store[id];
            fullkey = col == 0 ? rec.x : rec.y;
            fa.key_out[col][k] = fullkey;
        } else {
            fullkey = a.keys[col][k];
        }
        key = fullkey & (nb - 1);
    }
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
            if (below == 0) cnt[key] = base + group_size;
        }
        __syncthreads();
    }
    if (!fa.tile_local) {
        if (valid) a.lrank[col][k] = base + rank_in_wave;
        for (int b = tid; b < nb; b += CSORT_TILE) a.hist[col][(size_t)tile * nb + b] = cnt[b];
        return;
    }
    // tile-local sort: the tile's entries are put in key order right here (no global scan / scatter
    // launches).  hist[tile][bin] = (entries of the tile with that key) << 16 | (where they start inside
    // the tile's sorted list); the sweep finds a row's per-tile runs through this table.
    __shared__ int32_t wtot[256];
    tile_scan_pack(cnt, nb, wtot);
    for (int b = tid; b < nb; b += CSORT_TILE) a.hist[col][(size_t)tile * nb + b] = cnt[b];     // packed lookup table
    if (valid) {
        const int64_t dst = (int64_t)tile * CSORT_TILE + (cnt[key] & 0xffff) + base + rank_in_wave;
        a.ks[col][dst] = fullkey;
        a.ps[col][dst] = (int32_t)k;
    }
}

// ------------------------------------------------------------------------------------
// K1+K2+K3 for small tables in one launch.  grid = (ntiles * G, 2): block (tile, slice, side).
// Every block of a tile repeats the tile's gather and its stable counting sort by the side's id
// column (1024 entries, all in LDS: ballot ranks, ordered wave turns, in-LDS scan - cheap next to a
// launch and three dependent memory round trips), then owns slice = 1024/G consecutive entries of
// the tile's sorted list: one lane group per entry loads P[u], Q[i] once, forms the logit and
// g = dcost/dlogit exactly as K1 does (both sides compute the same g from the same registers), its
// contribution goes to LDS and the run heads add their runs in entry order (K3).  Piece sums land at
// the tile-sorted position; slice 0 publishes the tile's packed lookup table for k_dense_tiles.
// The item side also writes the logits and the per-block {loss, reg, sum g}.
template <int G, int VEC>
__global__ __launch_bounds__(1024) void k_tile_step(TileStepArgs a) {
    constexpr int EPB = 1024 / G;
    extern __shared__ int32_t dyn[];                     // sort: cnt[nbins]; then the contributions
    __shared__ int32_t rec_u[1024], rec_i[1024], srt_key[1024], srt_pos[1024];
    __shared__ float rec_r[1024];
    __shared__ float lds_gb[2 * EPB];
    __shared__ int32_t lds_key[EPB];
    __shared__ int32_t wtot[256];
    const int side = blockIdx.y;                         // 0: user rows, 1: item rows
    long long* pb = a.probe ? a.probe + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
    if (pb && threadIdx.x == 0) pb[0] = wall_clock64();
    const int tile = blockIdx.x / G, slice = blockIdx.x % G;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t tile0 = (int64_t)tile * 1024;
    const int nvalid = (a.B - tile0 < 1024) ? (int)(a.B - tile0) : 1024;
    if (slice * EPB >= nvalid) {                         // a short last tile: nothing in this slice
        if (side == 1 && tid < 4) a.partials[(size_t)blockIdx.x * 4 + tid] = 0.f;
        return;
    }
    // ---- gather the tile (dataio.py:115-117 when fused) and range-check the ids
    const int nb = a.nbins[side];
    int32_t* cnt = dyn;
    for (int b = tid; b < nb; b += 1024) cnt[b] = 0;
    const bool valid = tid < nvalid;
    int32_t u = 0, it = 0;
    float r = 0.f;
    if (valid) {
        const int64_t k = tile0 + tid;
        bool oob = false;
        if (a.ids) {
            int64_t id = a.ids[k];
            if ((uint64_t)id >= (uint64_t)a.N) { atomicOr(a.err, 2); id = 0; }
            const int4 rec = a.store[id];
            u = rec.x; it = rec.y; r = __int_as_float(rec.z);
        } else {
            u = a.u[k]; it = a.it[k]; r = a.r[k];
        }
        if ((uint64_t)(int64_t)u >= (uint64_t)a.U) { oob = true; u = 0; }
        if ((uint64_t)(int64_t)it >= (uint64_t)a.I) { oob = true; it = 0; }
        if (oob) atomicOr(a.err, 1);
    }
    rec_u[tid] = u; rec_i[tid] = it; rec_r[tid] = r;
    __syncthreads();
    if (pb && tid == 0) pb[1] = wall_clock64();
    // ---- stable counting sort of the tile by this side's id
    const int32_t key = side == 0 ? u : it;
    unsigned long long mask = __ballot(valid);
    for (int bit = 1; bit < nb; bit <<= 1) {
        const unsigned long long mm = __ballot((key & bit) != 0);
        mask &= (key & bit) ? mm : ~mm;
    }
    const unsigned long long below = mask & ((1ull << lane) - 1ull);
    const int rank_in_wave = __popcll(below);
    const int group_size = __popcll(mask);
    int base = 0;
    for (int w = 0; w < 16; ++w) {
        if (wave == w && valid) {
            base = cnt[key];
            if (below == 0) cnt[key] = base + group_size;
        }
        __syncthreads();
    }
    if (pb && tid == 0) pb[2] = wall_clock64();
    tile_scan_pack(cnt, nb, wtot);
    {   // the tile's packed lookup table, written once: its bins are dealt round the tile's active slices
        const int nact = (nvalid + EPB - 1) / EPB;
        for (int b = slice * 1024 + tid; b < nb; b += nact * 1024) a.tab[side][(size_t)tile * nb + b] = cnt[b];
    }
    if (valid) {
        const int dst = (cnt[key] & 0xffff) + base + rank_in_wave;
        srt_key[dst] = key;
        srt_pos[dst] = tid;
    }
    __syncthreads();                                     // cnt is dead from here: its memory takes the contributions
    if (pb && tid == 0) pb[3] = wall_clock64();
    float* lds_t = reinterpret_cast<float*>(dyn);

    // ---- this slice's entries: forward + contribution
    const int grp = tid / G, gl = tid % G, d0 = gl * VEC;
    const int D = a.D;
    const int jl = slice * EPB + grp;
    const bool ev = jl < nvalid;
    int32_t row = -1, prev = -2, pl = 0;
    if (ev) {
        row = srt_key[jl];
        prev = (jl > 0) ? srt_key[jl - 1] : -2;
        pl = srt_pos[jl];
    }
    const bool head = ev && prev != row;                 // jl == 0: the sorted order restarts with the tile
    const bool pstart = ev && (head || grp == 0);
    Frag<VEC> t;
#pragma unroll
    for (int q = 0; q < VEC; ++q) t.v[q] = 0.f;
    float tb = 0.f;
    float facc[3] = {0.f, 0.f, 0.f};
    if (ev) {
        const int32_t uu = rec_u[pl], ii = rec_i[pl];
        const float rr = rec_r[pl];
        const Frag<VEC> p = load_frag<VEC>(a.P + (size_t)uu * D, d0, D);
        const Frag<VEC> q = load_frag<VEC>(a.Q + (size_t)ii * D, d0, D);
        const float bu_ = a.bu[uu], bi_ = a.bi[ii];
        float sdot = 0.f, sq = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float qv = q.v[e];
            sdot = fmaf(p.v[e], a.item_abs ? fabsf(qv) : qv, sdot);
            sq = fmaf(p.v[e], p.v[e], fmaf(qv, qv, sq));
        }
        sdot = group_sum<G>(sdot);
        const float logit = ((sdot + *a.mu) + bu_) + bi_;          // ops.py:45-47 order
        float gk, l;
        if (a.loss == 0) {                               // ops.py:124
            gk = logit - rr;
            l = 0.5f * gk * gk;
        } else {                                         // ops.py:125-126
            gk = sigmoidf_(logit) - rr;
            l = fmaxf(logit, 0.f) - logit * rr + log1pf(__expf(-fabsf(logit)));
        }
        if (side == 1) {
            if (gl == 0) {
                if (a.logits) a.logits[tile0 + pl] = logit;
                facc[0] = l;
                facc[2] = gk;
                if (a.reg_bias) sq = fmaf(bu_, bu_, fmaf(bi_, bi_, sq));
            }
            facc[1] = 0.5f * sq;                         // tf.nn.l2_loss = sum(x^2)/2
        }
        const float ob = side == 0 ? bu_ : bi_;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float ov = side == 0 ? p.v[e] : q.v[e];
            float xv = side == 0 ? q.v[e] : p.v[e];
            if (side == 0) { if (a.item_abs) xv = fabsf(xv); }
            else if (a.item_abs) xv = xv * ((ov > 0.f) ? 1.f : ((ov < 0.f) ? -1.f : 0.f));
            t.v[e] = gk * xv + a.lam * ov;
        }
        tb = a.reg_bias ? (gk + a.lam * ob) : gk;
    }
    // ---- K3 inside the slice: suffix sums within runs by doubling (log2 EPB rounds, ping-pong in
    //      LDS).  After round d a group holds the sum of its next 2d entries of the same run; the fixed
    //      tree order keeps results bit-identical run to run however long the runs are.
    float* bufv[2] = {lds_t, lds_t + 1024 * VEC};
    float* bufb[2] = {lds_gb, lds_gb + EPB};
#pragma unroll
    for (int q = 0; q < VEC; ++q) bufv[0][(grp * G + gl) * VEC + q] = t.v[q];
    if (gl == 0) { bufb[0][grp] = tb; lds_key[grp] = row; }
    if (side == 1) block_sum_store<3, 16>(facc, a.partials + (size_t)blockIdx.x * 4);   // has the barrier
    else __syncthreads();
    if (pb && tid == 0) pb[4] = wall_clock64();
    Frag<VEC> acc = t;
    float gb = tb;
    int cur = 0;
#pragma unroll
    for (int d = 1; d < EPB; d <<= 1) {
        const int e2 = grp + d;
        const bool take = ev && e2 < EPB && lds_key[e2] == row;
        if (take) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc.v[q] += bufv[cur][(e2 * G + gl) * VEC + q];
            gb += bufb[cur][e2];
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) bufv[cur ^ 1][(grp * G + gl) * VEC + q] = acc.v[q];
        if (gl == 0) bufb[cur ^ 1][grp] = gb;
        __syncthreads();
        cur ^= 1;
    }
    if (!pstart) return;
    const int64_t j = tile0 + jl;
    store_frag<VEC>(a.grad_rows[side] + (size_t)j * D, d0, D, acc);
    if (gl == 0) a.grad_bias[side][j] = gb;
    if (pb && tid == 0) pb[5] = wall_clock64();
}

// ------------------------------------------------------------------------------------
// K0  triple gather from the HBM-resident (user,item,rate) store: what
//     ShuffleIterator.next does on the host (dataio.py:115-117), ids drawn by the host.
//     (the training path fuses this into K1; this kernel serves forward_resident)
__global__ __launch_bounds__(256) void k_gather_triples(GatherArgs a) {
    bool oob = false;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < a.B;
         k += (int64_t)gridDim.x * blockDim.x) {
        int64_t id = a.ids ? a.ids[k] : a.lo + k;
        if ((uint64_t)id >= (uint64_t)a.N) { oob = true; id = 0; }
        const int4 rec = a.store[id];
        a.u[k] = rec.x;
        a.it[k] = rec.y;
        a.r[k] = __int_as_float(rec.z);
    }
    if (oob) atomicOr(a.err, 2);
}

// row fetch for a peer shard: rows_out[j] = table[ids[j]], bias_out[j] = bias[ids[j]]
template <int G, int VEC>
__global__ __launch_bounds__(256) void k_gather_rows(GatherRowsArgs a) {
    constexpr int GPB = 256 / G;
    const int gl = threadIdx.x % G;
    const int d0 = gl * VEC;
    bool oob = false;
    for (int64_t j = (int64_t)blockIdx.x * GPB + threadIdx.x / G; j < a.n; j += (int64_t)gridDim.x * GPB) {
        int32_t id = a.ids[j];
        if ((uint64_t)(int64_t)id >= (uint64_t)a.rows) { oob = true; id = 0; }
        const Frag<VEC> f = load_frag<VEC>(a.table + (size_t)id * a.D, d0, a.D);
        store_frag<VEC>(a.rows_out + (size_t)j * a.D, d0, a.D, f);
        if (gl == 0) a.bias_out[j] = a.bias[id];
    }
    if (oob) atomicOr(a.err, 1);
}

// columns -> 16-byte records of the resident rating store
__global__ __launch_bounds__(256) void k_pack_triples(const int32_t* u, const int32_t* it, const float* r,
                                                     int4* store, int64_t n) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
         k += (int64_t)gridDim.x * blockDim.x)
        store[k] = make_int4(u[k], it[k], __float_as_int(r[k]), 0);
}

// ------------------------------------------------------------------------------------
// Optimiser arithmetic, written in the operation order of the TF kernels they restate.
struct AdamC { float alpha, b1, b2, eps, omb1, omb2; };

// AdamOptimizer._apply_sparse_shared [TF1-lib]: m*b1 + g*(1-b1); v*b2 + g*g*(1-b2);
// var - alpha*m/(sqrt(v)+eps)
__device__ __forceinline__ void adam_sparse(float& w, float& m, float& v, float g, const AdamC& c) {
    m = m * c.b1 + g * c.omb1;
    v = v * c.b2 + (g * g) * c.omb2;
    w = w - c.alpha * m / (sqrtf(v) + c.eps);
}

// ------------------------------------------------------------------------------------
// K3  deterministic segmented reduce over a table's rows (backward of embedding_lookup:
//     IndexedSlices -> unique + unsorted_segment_sum [TF1-lib], ops.py:143-149).
//     Input: batch positions stably sorted by row id (ks = sorted ids, ps = positions).
//     One 1024-thread block owns EPB = 1024/G consecutive sorted entries (a PIECE), one
//     lane group per entry:
//       phase 1 (all groups in parallel, no walk): the entry's contribution
//          user side: t = g_k * Qt[i_k] + lam * P[u]          Qt = |Q| if item_abs
//          item side: t = g_k * P[u_k] * s + lam * Q[i]       s = sign(Q[i]) if item_abs
//          bias     : t = g_k (+ lam * b[row] if reg_bias)          
//        goes to LDS;
//       phase 2: the group at each piece start (run head, or first entry of the block) adds
//        its run's contributions from LDS in entry order = batch order.
//     A run's total is its pieces added in piece order: a fixed order, so results are
//     run-to-run bit-identical, and the work is balanced however skewed the ids are.
//     RMODE_SCRATCH : every piece sum -> scratch[piece start] (+ map[row] = head+1 for tf1)
//     RMODE_ADAM/SGD: a run that lies inside one block is applied in place at once (fused lazy
//                     Adam / SGD); split runs go to scratch and k_apply_rows finishes them.
//     Both sides run fused: the item side goes first and copies each entry's pre-update Q row
//     to own_copy_out[pos]; the user side then takes its partner rows from that copy
//     (partner_by_pos), so neither side sees a row the other has already moved.
template <int G, int VEC, int RMODE, bool FWD = false>
__global__ __launch_bounds__(1024) void k_seg_reduce(RedPair pr) {
    constexpr int EPB = 1024 / G;
    __shared__ float lds_t[EPB * G * VEC];
    __shared__ float lds_gb[EPB];
    __shared__ int32_t lds_key[EPB];
    const RedArgs& a = pr.a[blockIdx.y];
    const int32_t err = *a.err;
    const int grp = threadIdx.x / G;
    const int gl = threadIdx.x % G;
    const int d0 = gl * VEC;
    const int D = a.D;
    const int64_t blk0 = (int64_t)blockIdx.x * EPB;
    const int64_t j = blk0 + grp;
    const bool valid = j < a.B;
    int32_t row = -1, prev = -2, pos = 0;
    if (valid) {
        row = a.ks[j];
        prev = (j > 0) ? a.ks[j - 1] : -2;
        pos = a.ps[j];
    }
    if (err || blk0 >= a.B) return;          // an out-of-range id voids the whole step (block-uniform)
    // tile mode: the sorted order is per 1024-entry tile, so a run also starts at every tile start
    const bool head = valid && (prev != row || (a.tile && (j % a.tile) == 0));
    const bool pstart = valid && (head || grp == 0);
    const AdamC c = {a.alpha, a.b1, a.b2, a.eps, 1.f - a.b1, 1.f - a.b2};
    const size_t roff = (size_t)(valid ? row : 0) * D;

    Frag<VEC> o, t, mrow, vrow;
    float ob = 0.f, tb = 0.f, mb = 0.f, vb = 0.f;
    float facc[3] = {0.f, 0.f, 0.f};                     // FWD: this lane's {loss, reg, g} share
#pragma unroll
    for (int q = 0; q < VEC; ++q) { o.v[q] = 0.f; t.v[q] = 0.f; mrow.v[q] = 0.f; vrow.v[q] = 0.f; }
    if (valid && a.rows_in) {
        // owner side of the sharded step: the contribution is a gradient row already reduced by
        // a peer (it includes that peer's lam * Q[i] terms); just add them up in arrival order
        t = load_frag<VEC>(a.rows_in + (size_t)pos * D, d0, D);
        tb = a.bias_in[pos];
        o = load_frag<VEC>(a.own + roff, d0, D);
        ob = a.own_bias[row];
        if constexpr (RMODE == RMODE_ADAM) {
            if (head) {
                mrow = load_frag<VEC>(a.m + roff, d0, D);
                vrow = load_frag<VEC>(a.v + roff, d0, D);
                mb = a.bias_m[row];
                vb = a.bias_v[row];
            }
        }
    } else if (valid) {
        float gk = 0.f;
        if constexpr (!FWD) gk = a.g[pos];
        const int32_t pid = a.other[pos];
        const float lam_e = a.lam_arr ? a.lam_arr[pos] : a.lam;      // FM: lam - g x^2 per non-zero
        const Frag<VEC> x = a.partner_by_pos ? load_frag<VEC>(a.partner_by_pos + (size_t)pos * D, d0, D)
                                             : load_frag<VEC>(a.partner + (size_t)pid * D, d0, D);
        o = load_frag<VEC>(a.own + roff, d0, D);
        ob = a.own_bias[row];
        if constexpr (RMODE == RMODE_ADAM) {
            if (head) {
                mrow = load_frag<VEC>(a.m + roff, d0, D);
                vrow = load_frag<VEC>(a.v + roff, d0, D);
                mb = a.bias_m[row];
                vb = a.bias_v[row];
            }
        }
        // this side updates its table in place before the other side runs: leave the other side the
        // pre-update row it needs, per entry
        if (a.own_copy_out) store_frag<VEC>(a.own_copy_out + (size_t)pos * D, d0, D, o);
        if constexpr (FWD) {
            // K1 on the rows already in registers (item side: partner = P[u], own = Q[i]); same
            // arithmetic and order as forward_body
            float sdot = 0.f, sq = 0.f;
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const float qv = o.v[q];
                sdot = fmaf(x.v[q], a.item_abs ? fabsf(qv) : qv, sdot);
                sq = fmaf(x.v[q], x.v[q], fmaf(qv, qv, sq));
            }
            sdot = group_sum<G>(sdot);
            const float pb = a.partner_bias[pid];
            const float logit = ((sdot + *a.mu) + pb) + ob;
            const float r = a.r[pos];
            float l;
            if (a.loss == 0) {
                gk = logit - r;
                l = 0.5f * gk * gk;
            } else {
                gk = sigmoidf_(logit) - r;
                l = fmaxf(logit, 0.f) - logit * r + log1pf(__expf(-fabsf(logit)));
            }
            if (gl == 0) {
                a.g_out[pos] = gk;
                if (a.logits_out) a.logits_out[pos] = logit;
                facc[0] = l;
                facc[2] = gk;
                if (a.reg_bias) sq = fmaf(pb, pb, fmaf(ob, ob, sq));
            }
            facc[1] = 0.5f * sq;
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            float xv = x.v[q];
            if (a.side == 0) { if (a.item_abs) xv = fabsf(xv); }
            else if (a.item_abs) xv = xv * ((o.v[q] > 0.f) ? 1.f : ((o.v[q] < 0.f) ? -1.f : 0.f));
            t.v[q] = gk * xv + lam_e * o.v[q];
        }
        tb = a.reg_bias ? (gk + a.lam * ob) : gk;
    }
    // contributions to LDS ([entry][G*VEC], a lane's VEC floats contiguous)
#pragma unroll
    for (int q = 0; q < VEC; ++q) lds_t[(grp * G + gl) * VEC + q] = t.v[q];
    if (gl == 0) { lds_gb[grp] = tb; lds_key[grp] = row; }
    if constexpr (FWD) block_sum_store<3, 16>(facc, a.partials + (size_t)blockIdx.x * 4);   // has the barrier
    else __syncthreads();
    if (!pstart) return;

    Frag<VEC> acc = t;
    float gb = tb;
    int e = grp + 1;
    for (;;) {                                           // four LDS entries per round trip, added in order
        bool same[4];
        float xv[4][VEC], xb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ee = (e + q < EPB) ? e + q : grp;   // in-bounds address; masked by same[]
            same[q] = (e + q < EPB) && (lds_key[ee] == row);
#pragma unroll
            for (int c2 = 0; c2 < VEC; ++c2) xv[q][c2] = lds_t[(ee * G + gl) * VEC + c2];
            xb[q] = lds_gb[ee];
        }
        bool go = true;
        int taken = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            go = go && same[q];
            if (go) {
#pragma unroll
                for (int c2 = 0; c2 < VEC; ++c2) acc.v[c2] += xv[q][c2];
                gb += xb[q];
                ++taken;
            }
        }
        e += taken;
        if (!go) break;
    }
    bool cont = false;                                   // does the run continue in the next block?
    if (e == EPB && blk0 + EPB < a.B && !(a.tile && ((blk0 + EPB) % a.tile) == 0)) cont = (a.ks[blk0 + EPB] == row);
    const bool whole = head && !cont;
    if (RMODE == RMODE_SCRATCH && whole && a.dense_rows) {
        // dense-gradient form (TF1 Adam sweep / data parallel): a run that lies in one block goes
        // straight to its row of the dense buffer - the consumer streams it with no indirection
        store_frag<VEC>(a.dense_rows + roff, d0, D, acc);
        if (gl == 0) a.dense_bias[row] = gb;
    } else if (RMODE == RMODE_SCRATCH || !whole) {
        store_frag<VEC>(a.grad_rows + (size_t)j * D, d0, D, acc);
        if (gl == 0) {
            a.grad_bias[j] = gb;
            if (head && a.map) a.map[row] = (int32_t)j + 1;      // split run: where its pieces start
        }
    } else if constexpr (RMODE == RMODE_ADAM) {
        if (!a.frozen_rows) {
            Frag<VEC> w = o;
#pragma unroll
            for (int q = 0; q < VEC; ++q) adam_sparse(w.v[q], mrow.v[q], vrow.v[q], acc.v[q], c);
            store_frag<VEC>(a.own_w + roff, d0, D, w);
            store_frag<VEC>(a.m + roff, d0, D, mrow);
            store_frag<VEC>(a.v + roff, d0, D, vrow);
        }
        if (gl == 0 && !a.frozen_bias) {
            float w = ob;
            adam_sparse(w, mb, vb, gb, c);
            a.bias_w[row] = w;
            a.bias_m[row] = mb;
            a.bias_v[row] = vb;
        }
    } else if constexpr (RMODE == RMODE_SGD) {
        if (!a.frozen_rows) {
            Frag<VEC> w = o;
#pragma unroll
            for (int q = 0; q < VEC; ++q) w.v[q] = w.v[q] - a.lr * acc.v[q];
            store_frag<VEC>(a.own_w + roff, d0, D, w);
        }
        if (gl == 0 && !a.frozen_bias) a.bias_w[row] = ob - a.lr * gb;
    }
}

// a run's reduced gradient = its pieces added in piece order (head piece first); pieces
// start at the run head and at every multiple of PIECE (= the reduce kernel's block span).
// Continuation pieces are probed and loaded eight at a time (independent loads), so a hot row
// split into a handful of pieces costs two memory round trips, not one per piece.
template <int VEC, int PIECE>
__device__ __forceinline__ Frag<VEC> run_total(const float* __restrict__ grad_rows,
                                               const float* __restrict__ grad_bias,
                                               const int32_t* __restrict__ ks, int64_t B, int64_t j,
                                               int32_t row, int d0, int D, float& gb) {
    Frag<VEC> t = load_frag<VEC>(grad_rows + (size_t)j * D, d0, D);
    gb = grad_bias[j];
    int64_t p = (j / PIECE + 1) * PIECE;
    if (p >= B || ks[p] != row) return t;                // the common case: a single piece
    for (;;) {
        bool same[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int64_t pp = p + (int64_t)q * PIECE;
            same[q] = (pp < B) && (ks[pp] == row);
        }
        Frag<VEC> x[8];
        float xb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int64_t pp = (p + (int64_t)q * PIECE < B) ? p + (int64_t)q * PIECE : j;   // safe address
            x[q] = load_frag<VEC>(grad_rows + (size_t)pp * D, d0, D);
            xb[q] = grad_bias[pp];
        }
        bool go = true;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            go = go && same[q];                          // pieces are contiguous: stop at the first miss
            if (go) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) t.v[e] += x[q].v[e];
                gb += xb[q];
            }
        }
        if (!go) break;
        p += 8 * (int64_t)PIECE;
    }
    return t;
}

// ------------------------------------------------------------------------------------
// K5a  apply reduced gradients held in scratch to the touched rows only: lazy Adam or SGD
//      (ops.py:143-149).  One lane group per sorted entry; only run heads work.
//      only_split: finish the runs k_seg_reduce could not apply in place (cut in >1 piece).
template <int G, int VEC, int OPT>
__global__ __launch_bounds__(256) void k_apply_rows(ApplyPair pr) {
    constexpr int PIECE = 1024 / G;
    const ApplyArgs& a = pr.a[blockIdx.y];
    const int32_t err = *a.err;
    constexpr int GPB = 256 / G;
    const int gl = threadIdx.x % G;
    const int d0 = gl * VEC;
    const int D = a.D;
    int64_t j = (int64_t)blockIdx.x * GPB + threadIdx.x / G;
    int32_t row;
    if (a.only_split) {
        // one lane group per piece boundary p: a run is split iff it crosses one.  The first
        // boundary a run crosses owns it; its head then lies in the PIECE entries before p.
        const int64_t p = (j + 1) * PIECE;
        if (err || p >= a.B) return;
        row = a.ks[p];
        if (a.ks[p - 1] != row) return;                  // no run crosses this boundary
        const int64_t lo = p - PIECE;
        if (lo > 0 && a.ks[lo - 1] == row) return;       // crossed an earlier boundary: handled there
        const int sh = ((threadIdx.x % 64) / G) * G;     // this group's lanes within the wave
        j = p - 1;
        for (int k = 0; k < PIECE; k += G) {             // sorted keys: the matches are a suffix of [lo, p)
            const bool hit = (k + gl < PIECE) && a.ks[lo + k + gl] == row;
            unsigned long long bits = __ballot(hit) >> sh;
            if constexpr (G < 64) bits &= (1ull << G) - 1ull;
            if (bits) { j = lo + k + (__ffsll((long long)bits) - 1); break; }
        }
    } else {
        if (j >= a.B) return;
        row = a.ks[j];
        const int32_t prev = (j > 0) ? a.ks[j - 1] : -2;
        if (err || prev == row) return;                  // voided step / not a run head
    }
    const AdamC c = {a.alpha, a.b1, a.b2, a.eps, 1.f - a.b1, 1.f - a.b2};
    const size_t roff = (size_t)row * D;
    Frag<VEC> w, mrow, vrow;
#pragma unroll
    for (int q = 0; q < VEC; ++q) { w.v[q] = 0.f; mrow.v[q] = 0.f; vrow.v[q] = 0.f; }
    if constexpr (OPT == 2) {                            // sharded step: emit the reduced gradient row
        float gb2;
        const Frag<VEC> tot = run_total<VEC, PIECE>(a.grad_rows, a.grad_bias, a.ks, a.B, j, row, d0, D, gb2);
        store_frag<VEC>(a.w + roff, d0, D, tot);
        if (gl == 0) a.bias_w[row] = gb2;
        return;
    }
    if (!a.frozen_rows) {                                // issued before the piece walk
        w = load_frag<VEC>(a.w + roff, d0, D);
        if constexpr (OPT == 0) {
            mrow = load_frag<VEC>(a.m + roff, d0, D);
            vrow = load_frag<VEC>(a.v + roff, d0, D);
        }
    }
    float gb;
    const Frag<VEC> gr = run_total<VEC, PIECE>(a.grad_rows, a.grad_bias, a.ks, a.B, j, row, d0, D, gb);
    if (!a.frozen_rows) {
        if constexpr (OPT == 0) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) adam_sparse(w.v[q], mrow.v[q], vrow.v[q], gr.v[q], c);
            store_frag<VEC>(a.m + roff, d0, D, mrow);
            store_frag<VEC>(a.v + roff, d0, D, vrow);
        } else {
#pragma unroll
            for (int q = 0; q < VEC; ++q) w.v[q] = w.v[q] - a.lr * gr.v[q];
        }
        store_frag<VEC>(a.w + roff, d0, D, w);
    }
    if (gl == 0 && !a.frozen_bias) {
        float bw = a.bias_w[row];
        if constexpr (OPT == 0) {
            float mb = a.bias_m[row], vb = a.bias_v[row];
            adam_sparse(bw, mb, vb, gb, c);
            a.bias_m[row] = mb;
            a.bias_v[row] = vb;
        } else {
            bw = bw - a.lr * gb;
        }
        a.bias_w[row] = bw;
    }
}

// ------------------------------------------------------------------------------------
// K5b  TF1 "sparse" Adam = dense sweep: every row decays m, v and moves;
//      touched rows (map[row] = head+1) add their reduced gradient.  One lane group per
//      row, consecutive groups on consecutive rows -> fully coalesced streaming.  The map
//      entry is cleared by the group that consumed it.
template <int G, int VEC>
__global__ __launch_bounds__(256) void k_adam_dense(DensePair pr) {
    constexpr int PIECE = 1024 / G;
    if (blockIdx.y == 2) {                               // optional: the step's K4 rides in this launch
        if (blockIdx.x == 0) finalize_body(pr.f);
        return;
    }
    const DenseArgs& a = pr.a[blockIdx.y];
    if (*a.err) return;
    constexpr int GPB = 256 / G;
    const int gl = threadIdx.x % G;
    const int d0 = gl * VEC;
    const int D = a.D;
    const AdamC c = {a.alpha, a.b1, a.b2, a.eps, 1.f - a.b1, 1.f - a.b2};
    for (int64_t row = (int64_t)blockIdx.x * GPB + threadIdx.x / G; row < a.rows;
         row += (int64_t)gridDim.x * GPB) {
        const int32_t slot = a.map ? a.map[row] : 0;
        const size_t roff = (size_t)row * D;
        Frag<VEC> w, mrow, vrow;
#pragma unroll
        for (int q = 0; q < VEC; ++q) { w.v[q] = 0.f; mrow.v[q] = 0.f; vrow.v[q] = 0.f; }
        if (!a.frozen_rows) {                            // independent of the map: issue first
            w = load_frag<VEC>(a.w + roff, d0, D);
            if (a.opt == 0) {
                mrow = load_frag<VEC>(a.m + roff, d0, D);
                vrow = load_frag<VEC>(a.v + roff, d0, D);
            }
        }
        float bw = 0.f, mb = 0.f, vb = 0.f;
        if (gl == 0 && !a.frozen_bias) {
            bw = a.bias_w[row];
            if (a.opt == 0) { mb = a.bias_m[row]; vb = a.bias_v[row]; }
        }
        Frag<VEC> gr;
        float gb = 0.f;
        if (a.dense_grad) {                              // dense gradients (whole runs land here directly)
            gr = load_frag<VEC>(a.dense_grad + roff, d0, D);
            gb = a.dense_gbias[row];
            if (slot) {                                  // a run split over several reduce blocks
                gr = run_total<VEC, PIECE>(a.grad_rows, a.grad_bias, a.ks, a.B, (int64_t)slot - 1, (int32_t)row, d0, D, gb);
                if (gl == 0) a.map[row] = 0;
            }
            Frag<VEC> z;
#pragma unroll
            for (int q = 0; q < VEC; ++q) z.v[q] = 0.f;
            store_frag<VEC>(a.dense_grad + roff, d0, D, z);      // consumed: clean for the next step
            if (gl == 0) a.dense_gbias[row] = 0.f;
        } else if (slot) {
            gr = run_total<VEC, PIECE>(a.grad_rows, a.grad_bias, a.ks, a.B, (int64_t)slot - 1, (int32_t)row, d0, D, gb);
            if (gl == 0) a.map[row] = 0;                 // consumed (after the read above)
        } else {
#pragma unroll
            for (int q = 0; q < VEC; ++q) gr.v[q] = 0.f;
        }
        if (!a.frozen_rows) {
            if (a.opt == 0) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) adam_sparse(w.v[q], mrow.v[q], vrow.v[q], gr.v[q], c);
                store_frag<VEC>(a.m + roff, d0, D, mrow);
                store_frag<VEC>(a.v + roff, d0, D, vrow);
            } else {
#pragma unroll
                for (int q = 0; q < VEC; ++q) w.v[q] = w.v[q] - a.lr * gr.v[q];
            }
            store_frag<VEC>(a.w + roff, d0, D, w);
        }
        if (gl == 0 && !a.frozen_bias) {
            if (a.opt == 0) {
                adam_sparse(bw, mb, vb, gb, c);
                a.bias_m[row] = mb;
                a.bias_v[row] = vb;
            } else {
                bw = bw - a.lr * gb;
            }
            a.bias_w[row] = bw;
        }
    }
}

// ------------------------------------------------------------------------------------
// K5c  small-table sweep over per-tile partials.  With the tile-local sort (k_front) a row's
//      gradient is spread over the <= 16 tiles of the batch: tab[t][row] packs how many entries tile t
//      has for the row (<< 16) and where its run starts in the tile's sorted list; the run's piece
//      sums (k_seg_reduce) sit at that position and at the following multiples of EPB.  One lane
//      group per row adds them in tile order = batch order (all addresses come from the two lookup
//      tables, so the loads are independent) and then either applies the optimiser (TF1 Adam: every
//      row; lazy Adam / SGD: touched rows) or - data parallel - writes the row into the dense
//      gradient buffer.  blockIdx.y == 2 runs the step's finalize (K4).
template <int G, int VEC, bool WRITE>
__global__ __launch_bounds__(256) void k_dense_tiles(TileDenseLaunch L) {
    if (blockIdx.y == 2) {
        if (blockIdx.x == 0) finalize_body(L.f);
        return;
    }
    const TileDenseArgs& a = L.a[blockIdx.y];
    if (*a.err) return;
    constexpr int GPB = 256 / G;
    constexpr int EPB = 1024 / G;
    const int gl = threadIdx.x % G;
    const int d0 = gl * VEC;
    const int D = a.D;
    const AdamC c = {a.alpha, a.b1, a.b2, a.eps, 1.f - a.b1, 1.f - a.b2};
    for (int64_t row = (int64_t)blockIdx.x * GPB + threadIdx.x / G; row < a.rows;
         row += (int64_t)gridDim.x * GPB) {
        const size_t roff = (size_t)row * D;
        Frag<VEC> w, mrow, vrow;
#pragma unroll
        for (int q = 0; q < VEC; ++q) { w.v[q] = 0.f; mrow.v[q] = 0.f; vrow.v[q] = 0.f; }
        float bw = 0.f, mb = 0.f, vb = 0.f;
        if constexpr (!WRITE) {                          // independent of the lookups: issue first
            if (!a.frozen_rows) {
                w = load_frag<VEC>(a.w + roff, d0, D);
                if (a.opt == 0) {
                    mrow = load_frag<VEC>(a.m + roff, d0, D);
                    vrow = load_frag<VEC>(a.v + roff, d0, D);
                }
            }
            if (gl == 0 && !a.frozen_bias) {
                bw = a.bias_w[row];
                if (a.opt == 0) { mb = a.bias_m[row]; vb = a.bias_v[row]; }
            }
        }
        Frag<VEC> tot;
#pragma unroll
        for (int q = 0; q < VEC; ++q) tot.v[q] = 0.f;
        float gb = 0.f;
        bool touched = false;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half * 8 < a.ntiles) {                   // uniform
                int32_t cn[8], of[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int tt = half * 8 + t;
                    const int32_t e = (tt < a.ntiles) ? a.tab[(size_t)tt * a.nbins + row] : 0;
                    cn[t] = e >> 16;                      // entries of tile tt for this row (<= 1024)
                    of[t] = e & 0xffff;                   // where its run starts in the tile's sorted list
                }
                Frag<VEC> x[8];
                float xb[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {            // unconditional loads (safe address when absent)
                    const int64_t j = cn[t] ? (int64_t)(half * 8 + t) * 1024 + of[t] : 0;
                    x[t] = load_frag<VEC>(a.grad_rows + (size_t)j * D, d0, D);
                    xb[t] = a.grad_bias[j];
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    if (cn[t]) {
                        touched = true;
#pragma unroll
                        for (int q = 0; q < VEC; ++q) tot.v[q] += x[t].v[q];
                        gb += xb[t];
                        const int64_t j = (int64_t)(half * 8 + t) * 1024 + of[t];
                        const int64_t end = j + cn[t];
                        for (int64_t p = (j / EPB + 1) * EPB; p < end; p += EPB) {     // further pieces (hot rows)
                            const Frag<VEC> y = load_frag<VEC>(a.grad_rows + (size_t)p * D, d0, D);
#pragma unroll
                            for (int q = 0; q < VEC; ++q) tot.v[q] += y.v[q];
                            gb += a.grad_bias[p];
                        }
                    }
                }
            }
        }
        if constexpr (WRITE) {
            if (touched) {
                store_frag<VEC>(a.out_rows + roff, d0, D, tot);
                if (gl == 0) a.out_bias[row] = gb;
            }
        } else {
            if (!touched && a.skip_untouched) continue;
            if (!a.frozen_rows) {
                if (a.opt == 0) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) adam_sparse(w.v[q], mrow.v[q], vrow.v[q], tot.v[q], c);
                    store_frag<VEC>(a.m + roff, d0, D, mrow);
                    store_frag<VEC>(a.v + roff, d0, D, vrow);
                } else {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) w.v[q] = w.v[q] - a.lr * tot.v[q];
                }
                store_frag<VEC>(a.w + roff, d0, D, w);
            }
            if (gl == 0 && !a.frozen_bias) {
                if (a.opt == 0) {
                    adam_sparse(bw, mb, vb, gb, c);
                    a.bias_m[row] = mb;
                    a.bias_v[row] = vb;
                } else {
                    bw = bw - a.lr * gb;
                }
                a.bias_w[row] = bw;
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// K4  finalize (body in finalize.inc.h; the small-table path runs it inside the csort scan launch)
__global__ __launch_bounds__(256) void k_finalize(FinArgs a) { finalize_body(a); }

// ------------------------------------------------------------------------------------
// Variable initialisers (ops.py:8-12,29-32): truncated normal = N(0, s) resampled until
// |x| <= 2 s [TF1-lib].  Counter-based (splitmix64 of seed and element index), so the
// values do not depend on the launch geometry.  TF's Philox stream is not reproducible
// here: initial values are not a parity target.
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void k_init_trunc_normal(float* p, int64_t n, float stddev, uint64_t seed) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
         k += (int64_t)gridDim.x * blockDim.x) {
        float z = 0.f;
        for (int attempt = 0; attempt < 16; ++attempt) {
            const uint64_t h = splitmix64(seed ^ splitmix64((uint64_t)k * 16 + attempt));
            const float u1 = ((float)(uint32_t)(h >> 40) + 0.5f) * (1.f / 16777216.f);   // (0,1)
            const float u2 = ((float)(uint32_t)((h >> 16) & 0xFFFFFF) + 0.5f) * (1.f / 16777216.f);
            z = sqrtf(-2.f * __logf(u1)) * __cosf(6.28318530718f * u2);
            if (fabsf(z) <= 2.f) break;
            z = 0.f;
        }
        p[k] = z * stddev;
    }
}

__global__ void k_init_uniform_scalar(float* p, float lo, float hi, uint64_t seed) {
    const uint64_t h = splitmix64(seed ^ 0xABCDEF12345ull);
    const float u = ((float)(uint32_t)(h >> 40) + 0.5f) * (1.f / 16777216.f);
    p[0] = lo + (hi - lo) * u;
}

// ------------------------------------------------------------------------------------
// launch helpers
template <int MODE, int UNR>
static void launch_forward_mode(const FwdArgs& a, int G, int VEC, int grid, hipStream_t s) {
#define TFR_FWD_CASE(g, v) \
    if (G == g && VEC == v) { hipLaunchKernelGGL((k_forward<g, v, MODE, UNR>), dim3(grid), dim3(256), 0, s, a); return; }
    TFR_FWD_CASE(4, 4) TFR_FWD_CASE(8, 4) TFR_FWD_CASE(16, 4) TFR_FWD_CASE(32, 4) TFR_FWD_CASE(64, 4)
    TFR_FWD_CASE(4, 1) TFR_FWD_CASE(8, 1) TFR_FWD_CASE(16, 1) TFR_FWD_CASE(32, 1) TFR_FWD_CASE(64, 1)
#undef TFR_FWD_CASE
}

int forward_grid(int64_t B, int G, int mode) {
    const int64_t per_block = 4 * (64 / G) * 4;   // waves * SPW * UNR
    int64_t nb = (B + per_block - 1) / per_block;
    const int64_t cap = (mode == MODE_INFER) ? 8192 : 2048;   // TRAIN/EVAL: fewer partials for K4
    if (nb > cap) nb = cap;                              // then grid-stride
    if (nb < 1) nb = 1;
    return (int)nb;
}

int front_forward_blocks(int64_t B, int G) {
    const int64_t per_block = 16 * (64 / G) * 2;        // waves * SPW * UNR of k_front's forward part
    int64_t nb = (B + per_block - 1) / per_block;
    if (nb > 512) nb = 512;
    if (nb < 1) nb = 1;
    return (int)nb;
}

void launch_front(const FrontArgs& fa, int G, int VEC, hipStream_t s) {
    const int nbmax = fa.c.nbins[0] > fa.c.nbins[1] ? fa.c.nbins[0] : fa.c.nbins[1];
    const dim3 grid(fa.nfwd + 2 * fa.c.ntiles);
#define TFR_FRONT_CASE(g, v)                                                                          \
    if (G == g && VEC == v) {                                                                         \
        static bool attr = false;                                                                     \
        if (!attr) {                                                                                  \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_front<g, v>),                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, CSORT_MAX_BINS * 4); \
            attr = true;                                                                              \
        }                                                                                             \
        hipLaunchKernelGGL((k_front<g, v>), grid, dim3(1024), (size_t)nbmax * 4, s, fa);              \
        return;                                                                                       \
    }
    TFR_FRONT_CASE(4, 4) TFR_FRONT_CASE(8, 4) TFR_FRONT_CASE(16, 4) TFR_FRONT_CASE(32, 4) TFR_FRONT_CASE(64, 4)
    TFR_FRONT_CASE(4, 1) TFR_FRONT_CASE(8, 1) TFR_FRONT_CASE(16, 1) TFR_FRONT_CASE(32, 1) TFR_FRONT_CASE(64, 1)
#undef TFR_FRONT_CASE
}

void launch_tile_step(const TileStepArgs& a, int G, int VEC, hipStream_t s) {
    const int nbmax = a.nbins[0] > a.nbins[1] ? a.nbins[0] : a.nbins[1];
    size_t dyn = (size_t)nbmax * 4;                       // bins during the sort, contributions afterwards
    if (dyn < (size_t)2 * 1024 * VEC * 4) dyn = (size_t)2 * 1024 * VEC * 4;     // ping-pong buffers of the in-slice reduce
    const dim3 grid(a.ntiles * G, 2);
#define TFR_TS_CASE(g, v)                                                                             \
    if (G == g && VEC == v) {                                                                         \
        static bool attr = false;                                                                     \
        if (!attr) {                                                                                  \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_step<g, v>),               \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, CSORT_MAX_BINS * 4); \
            attr = true;                                                                              \
        }                                                                                             \
        hipLaunchKernelGGL((k_tile_step<g, v>), grid, dim3(1024), dyn, s, a);                         \
        return;                                                                                       \
    }
    TFR_TS_CASE(4, 4) TFR_TS_CASE(8, 4) TFR_TS_CASE(16, 4) TFR_TS_CASE(32, 4) TFR_TS_CASE(64, 4)
    TFR_TS_CASE(4, 1) TFR_TS_CASE(8, 1) TFR_TS_CASE(16, 1) TFR_TS_CASE(32, 1) TFR_TS_CASE(64, 1)
#undef TFR_TS_CASE
}

void launch_forward(const FwdArgs& a, int mode, int G, int VEC, int grid, hipStream_t s) {
    if (mode == MODE_INFER) launch_forward_mode<MODE_INFER, 4>(a, G, VEC, grid, s);
    else if (mode == MODE_TRAIN) launch_forward_mode<MODE_TRAIN, 4>(a, G, VEC, grid, s);
    else launch_forward_mode<MODE_EVAL, 4>(a, G, VEC, grid, s);
}

static int entry_grid(int64_t B, int G) {
    const int gpb = 256 / G;
    int64_t nb = (B + gpb - 1) / gpb;
    if (nb < 1) nb = 1;
    return (int)nb;
}

void launch_seg_reduce(const RedPair& p, int n, int rmode, int G, int VEC, hipStream_t s, bool fwd) {
    int64_t B = p.a[0].B;
    if (n > 1 && p.a[1].B > B) B = p.a[1].B;
    const int epb = 1024 / G;
    int64_t nb = (B + epb - 1) / epb;
    if (nb < 1) nb = 1;
    const dim3 grid((int)nb, n);
#define TFR_RED_CASE(g, v)                                                                             \
    if (G == g && VEC == v) {                                                                          \
        if (fwd && rmode == RMODE_ADAM) hipLaunchKernelGGL((k_seg_reduce<g, v, RMODE_ADAM, true>), grid, dim3(1024), 0, s, p); \
        else if (fwd) hipLaunchKernelGGL((k_seg_reduce<g, v, RMODE_SGD, true>), grid, dim3(1024), 0, s, p); \
        else if (rmode == RMODE_SCRATCH) hipLaunchKernelGGL((k_seg_reduce<g, v, RMODE_SCRATCH>), grid, dim3(1024), 0, s, p); \
        else if (rmode == RMODE_ADAM) hipLaunchKernelGGL((k_seg_reduce<g, v, RMODE_ADAM>), grid, dim3(1024), 0, s, p);  \
        else hipLaunchKernelGGL((k_seg_reduce<g, v, RMODE_SGD>), grid, dim3(1024), 0, s, p);           \
        return;                                                                                        \
    }
    TFR_RED_CASE(4, 4) TFR_RED_CASE(8, 4) TFR_RED_CASE(16, 4) TFR_RED_CASE(32, 4) TFR_RED_CASE(64, 4)
    TFR_RED_CASE(4, 1) TFR_RED_CASE(8, 1) TFR_RED_CASE(16, 1) TFR_RED_CASE(32, 1) TFR_RED_CASE(64, 1)
#undef TFR_RED_CASE
}

void launch_apply_rows(const ApplyPair& p, int n, int opt, int G, int VEC, hipStream_t s) {
    int64_t B = p.a[0].B;
    if (n > 1 && p.a[1].B > B) B = p.a[1].B;
    bool all_split = p.a[0].only_split != 0;
    if (n > 1 && !p.a[1].only_split) all_split = false;
    if (all_split) B = (B + 1024 / G - 1) / (1024 / G);      // one lane group per piece boundary
    const dim3 grid(entry_grid(B, G), n);
#define TFR_APP_CASE(g, v)                                                                  \
    if (G == g && VEC == v) {                                                               \
        if (opt == 0) hipLaunchKernelGGL((k_apply_rows<g, v, 0>), grid, dim3(256), 0, s, p); \
        else if (opt == 1) hipLaunchKernelGGL((k_apply_rows<g, v, 1>), grid, dim3(256), 0, s, p); \
        else hipLaunchKernelGGL((k_apply_rows<g, v, 2>), grid, dim3(256), 0, s, p);         \
        return;                                                                             \
    }
    TFR_APP_CASE(4, 4) TFR_APP_CASE(8, 4) TFR_APP_CASE(16, 4) TFR_APP_CASE(32, 4) TFR_APP_CASE(64, 4)
    TFR_APP_CASE(4, 1) TFR_APP_CASE(8, 1) TFR_APP_CASE(16, 1) TFR_APP_CASE(32, 1) TFR_APP_CASE(64, 1)
#undef TFR_APP_CASE
}

void launch_dense_tiles(const TileDenseLaunch& L, bool write, bool with_fin, int G, int VEC, hipStream_t s) {
    const int gpb = 256 / G;
    int64_t rows = L.a[0].rows > L.a[1].rows ? L.a[0].rows : L.a[1].rows;
    int64_t nb = (rows + gpb - 1) / gpb;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    const dim3 grid((int)nb, with_fin ? 3 : 2);
#define TFR_DT_CASE(g, v)                                                                         \
    if (G == g && VEC == v) {                                                                     \
        if (write) hipLaunchKernelGGL((k_dense_tiles<g, v, true>), grid, dim3(256), 0, s, L);     \
        else hipLaunchKernelGGL((k_dense_tiles<g, v, false>), grid, dim3(256), 0, s, L);          \
        return;                                                                                   \
    }
    TFR_DT_CASE(4, 4) TFR_DT_CASE(8, 4) TFR_DT_CASE(16, 4) TFR_DT_CASE(32, 4) TFR_DT_CASE(64, 4)
    TFR_DT_CASE(4, 1) TFR_DT_CASE(8, 1) TFR_DT_CASE(16, 1) TFR_DT_CASE(32, 1) TFR_DT_CASE(64, 1)
#undef TFR_DT_CASE
}

void launch_adam_dense(DensePair& p, int n, int G, int VEC, hipStream_t s, const FinArgs* fin) {
    if (fin) { p.f = *fin; n = 3; }
    const int gpb = 256 / G;
    int64_t rows = p.a[0].rows;
    if (n > 1 && p.a[1].rows > rows) rows = p.a[1].rows;
    int64_t nb = (rows + gpb - 1) / gpb;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    const dim3 grid((int)nb, n);
#define TFR_DEN_CASE(g, v) \
    if (G == g && VEC == v) { hipLaunchKernelGGL((k_adam_dense<g, v>), grid, dim3(256), 0, s, p); return; }
    TFR_DEN_CASE(4, 4) TFR_DEN_CASE(8, 4) TFR_DEN_CASE(16, 4) TFR_DEN_CASE(32, 4) TFR_DEN_CASE(64, 4)
    TFR_DEN_CASE(4, 1) TFR_DEN_CASE(8, 1) TFR_DEN_CASE(16, 1) TFR_DEN_CASE(32, 1) TFR_DEN_CASE(64, 1)
#undef TFR_DEN_CASE
}

static int flat_grid(int64_t n) {
    int64_t nb = (n + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    return (int)nb;
}

void launch_gather_rows(const GatherRowsArgs& a, int G, int VEC, hipStream_t s) {
    const int gpb = 256 / G;
    int64_t nb = (a.n + gpb - 1) / gpb;
    if (nb > 8192) nb = 8192;
    if (nb < 1) nb = 1;
#define TFR_GR_CASE(g, v) \
    if (G == g && VEC == v) { hipLaunchKernelGGL((k_gather_rows<g, v>), dim3((int)nb), dim3(256), 0, s, a); return; }
    TFR_GR_CASE(4, 4) TFR_GR_CASE(8, 4) TFR_GR_CASE(16, 4) TFR_GR_CASE(32, 4) TFR_GR_CASE(64, 4)
    TFR_GR_CASE(4, 1) TFR_GR_CASE(8, 1) TFR_GR_CASE(16, 1) TFR_GR_CASE(32, 1) TFR_GR_CASE(64, 1)
#undef TFR_GR_CASE
}

void launch_gather(const GatherArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_triples, dim3(flat_grid(a.B)), dim3(256), 0, s, a);
}

void launch_pack_triples(const int32_t* u, const int32_t* it, const float* r, void* store, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_triples, dim3(flat_grid(n)), dim3(256), 0, s, u, it, r, reinterpret_cast<int4*>(store), n);
}

void launch_init_trunc_normal(float* p, int64_t n, float stddev, uint64_t seed, hipStream_t s) {
    hipLaunchKernelGGL(k_init_trunc_normal, dim3(flat_grid(n)), dim3(256), 0, s, p, n, stddev, seed);
}

void launch_init_uniform_scalar(float* p, float lo, float hi, uint64_t seed, hipStream_t s) {
    hipLaunchKernelGGL(k_init_uniform_scalar, dim3(1), dim3(1), 0, s, p, lo, hi, seed);
}

void launch_finalize(const FinArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, s, a);
}

}  // namespace tfr
