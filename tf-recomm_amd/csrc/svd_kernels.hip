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
#include <cstdlib>
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

// a lane's VEC floats of a row known to be G * VEC wide: no `d0 < D` guard, so the loaded registers are not merged with a
// zero-initialised copy afterwards (that merge is a register move behind a full wait for the load)
template <int VEC, bool NT>
__device__ __forceinline__ Frag<VEC> load_full(const float* __restrict__ p) {
    Frag<VEC> f;
    if constexpr (VEC == 4) {
        floatx4 t;
        if constexpr (NT) t = __builtin_nontemporal_load(reinterpret_cast<const floatx4*>(p));
        else t = *reinterpret_cast<const floatx4*>(p);
        f.v[0] = t.x; f.v[1] = t.y; f.v[2] = t.z; f.v[3] = t.w;
    } else {
        if constexpr (NT) f.v[0] = __builtin_nontemporal_load(p);
        else f.v[0] = *p;
    }
    return f;
}

// cache-policy experiments on the big-table step (RedArgs::nt bits; A/B by TFR_NT): runtime-selected hints
template <int VEC>
__device__ __forceinline__ Frag<VEC> load_frag_h(const float* __restrict__ row, int d0, int D, bool nt) {
    return nt ? load_frag<VEC, true>(row, d0, D) : load_frag<VEC, false>(row, d0, D);
}
template <int VEC>
__device__ __forceinline__ void store_frag_h(float* __restrict__ row, int d0, int D, const Frag<VEC>& f, bool nt) {
    if (!nt) { store_frag<VEC>(row, d0, D, f); return; }
    if (d0 < D) {
        if constexpr (VEC == 4) {
            floatx4 t; t.x = f.v[0]; t.y = f.v[1]; t.z = f.v[2]; t.w = f.v[3];
            __builtin_nontemporal_store(t, reinterpret_cast<floatx4*>(row + d0));
        } else {
            __builtin_nontemporal_store(f.v[0], row + d0);
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

// per-piece {loss, reg, sum g} for k_tile_step without a cross-lane reduction in every wave: lane
// group leaders park the entry's loss and g, every lane its share of the regulariser, in LDS;
// after the barrier wave (h, c) alone adds piece h's values of kind c in a fixed order.
// stage: EPG * (2 * EPB + 1024) floats.  16 waves; has one barrier.
template <int G, int EPG>
__device__ __forceinline__ void block_sum_pieces(const float (&val)[3 * EPG], float* stage, float* out) {
    constexpr int EPB = 1024 / G;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = tid / G, gl = tid % G;
    float* sl = stage;                                   // [EPG][EPB]   loss per entry
    float* sg = stage + EPG * EPB;                       // [EPG][EPB]   g per entry
    float* sr = stage + 2 * EPG * EPB;                   // [EPG][1024]  regulariser share per lane
#pragma unroll
    for (int h = 0; h < EPG; ++h) {
        if (gl == 0) { sl[h * EPB + grp] = val[3 * h + 0]; sg[h * EPB + grp] = val[3 * h + 2]; }
        sr[h * 1024 + tid] = val[3 * h + 1];
    }
    __syncthreads();
    if (wave < 3 * EPG) {
        const int h = wave / 3, c = wave % 3;
        float x = 0.f;
        if (c == 1) {
#pragma unroll
            for (int k = 0; k < 16; ++k) x += sr[h * 1024 + k * 64 + lane];
        } else {
            const float* src = (c == 0 ? sl : sg) + h * EPB;
            for (int k = lane; k < EPB; k += 64) x += src[k];
        }
        x = wave_sum(x);
        if (lane == 0) out[h * 4 + c] = x;
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
// Bias gathers take the SCALAR path where a wave holds few ratings (SPW <= 4, i.e. D >= 64): the ids come
// out of the lanes by v_readlane and the 4-byte loads go through the scalar cache, so they do not queue
// behind the row gathers in the vector memory pipeline.  Measured on the north-star shape
// (tools/probes/fwd_shape.hip): per-group vector loads 45.6 us, scalar 42.9 us per 262144 ratings.  The
// arrays are read-only for the whole launch (constant address space = s_load).  Narrow rows (SPW >= 8) use
// one wave-wide vector load per table instead: lane j fetches the bias of the wave's j-th rating.
typedef const float __attribute__((address_space(4))) * cfloat_p;

template <int G, int VEC, int MODE, int UNR, int NW, bool PNT = false>
__device__ __forceinline__ void forward_body(const FwdArgs& a, int block, int nblocks) {
    constexpr int SPW = 64 / G;        // ratings per wave per pass; UNR passes in flight
    constexpr bool SBIAS = SPW <= 4;
    constexpr int SPI = SPW * UNR;     // ratings per wave-iteration
    const int lane = threadIdx.x & 63;
    const int sub = lane / G;
    const int gl = lane % G;
    const int d0 = gl * VEC;
    const int D = a.D;
    // batch positions fit 32 bits (the workspace caps a batch at 2^30 ratings)
    const int Bn = a.dB ? *a.dB : (int)a.B;             // row-sharded step: the local batch size lives on the device
    const size_t qs = a.qstride ? (size_t)a.qstride : (size_t)a.D;      // item rows may sit in a packed exchange buffer
    const int bis = a.bistride ? a.bistride : 1;
    const int wave_id = block * NW + (threadIdx.x >> 6);
    const int stride = nblocks * NW * SPI;
    const float mu = *a.mu;

    float acc[3] = {0.f, 0.f, 0.f};    // TRAIN: loss, reg, sum g | EVAL: sse, n_equal, -
    bool oob = false, oob_store = false;

    // TRAIN (few, fat blocks: k_front, the sharded step): the ids of iteration n+1 are fetched while the rows
    // of iteration n are in flight, so the only exposed round trip per iteration is the row gather itself.
    // INFER / EVAL fetch them at the top of the iteration instead: without the second id set the kernel fits
    // 64 VGPRs = 8 waves per SIMD, and the extra waves hide that round trip better than the prefetch did
    // (north-star shape, one 262144-rating batch per launch: 51.0 -> 43.4 us together with the scalar-path
    // bias loads and the non-temporal user rows; holding the kernel to 80 SGPRs for an eighth wave per SIMD
    // changed nothing - gpurun_out/ns_s80_*.json; tools/probes/fwd_shape.hip).
    constexpr bool PF = MODE == MODE_TRAIN;
    int32_t u_n[UNR], it_n[UNR];
    float rr_n[UNR];
    auto fetch_ids = [&](int base) {
        if (a.ids) {
            // fused ShuffleIterator gather (dataio.py:115-117): id -> (user, item, rate) from the
            // HBM-resident store
            int64_t id[UNR];
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
                const int k = base + j * SPW + sub;
                id[j] = (k < Bn) ? a.ids[k] : 0;
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
                const int k = base + j * SPW + sub;
                const bool ok = k < Bn;
                u_n[j] = ok ? a.u[k] : 0;
                it_n[j] = ok ? a.it[k] : 0;
                rr_n[j] = (MODE != MODE_INFER && ok) ? a.r[k] : 0.f;
            }
        }
    };

    int base = wave_id * SPI;
    if (PF && base < Bn) fetch_ids(base);
    for (; base < Bn; base += stride) {
        if (!PF) fetch_ids(base);
        int k[UNR];
        int32_t u[UNR], it[UNR];
        bool ok[UNR];
        float rr[UNR];
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            k[j] = base + j * SPW + sub;
            ok[j] = k[j] < Bn;
            u[j] = u_n[j]; it[j] = it_n[j]; rr[j] = rr_n[j];
            if (a.ids && a.u_out && gl == 0 && ok[j]) { a.u_out[k[j]] = u[j]; a.it_out[k[j]] = it[j]; }   // kept for the backward
            if ((uint64_t)(int64_t)u[j] >= (uint64_t)a.U) { oob = true; u[j] = 0; }
            if ((uint64_t)(int64_t)it[j] >= (uint64_t)a.I) { oob = true; it[j] = 0; }
        }
        Frag<VEC> p[UNR], q[UNR];
        float bu_[UNR], bi_[UNR];
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            p[j] = load_frag<VEC, PNT>(a.P + (size_t)u[j] * D, d0, D);
            q[j] = load_frag<VEC>(a.Q + (size_t)it[j] * qs, d0, D);
        }
        if constexpr (SBIAS) {
            const cfloat_p cbu = (cfloat_p)(uintptr_t)a.bu;
            const cfloat_p cbi = (cfloat_p)(uintptr_t)a.bi;
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
                float xs[SPW], ys[SPW];
#pragma unroll
                for (int s2 = 0; s2 < SPW; ++s2) {
                    xs[s2] = cbu[__builtin_amdgcn_readlane(u[j], s2 * G)];
                    ys[s2] = cbi[(size_t)__builtin_amdgcn_readlane(it[j], s2 * G) * bis];
                }
                bu_[j] = xs[0]; bi_[j] = ys[0];
#pragma unroll
                for (int s2 = 1; s2 < SPW; ++s2) { if (sub == s2) { bu_[j] = xs[s2]; bi_[j] = ys[s2]; } }
            }
        } else {
            // lane l = j * SPW + s holds rating (j, s) of this iteration (SPI <= 64 ratings): one load per table
            int32_t mu_ = 0, mi_ = 0;
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
                const int src = ((lane - j * SPW) & (SPW - 1)) * G;
                const int32_t su = __shfl(u[j], src, 64), si = __shfl(it[j], src, 64);
                if (lane / SPW == j) { mu_ = su; mi_ = si; }
            }
            float wx = 0.f, wy = 0.f;
            if (lane < SPI) { wx = a.bu[mu_]; wy = a.bi[(size_t)mi_ * bis]; }
#pragma unroll
            for (int j = 0; j < UNR; ++j) { bu_[j] = __shfl(wx, j * SPW + sub, 64); bi_[j] = __shfl(wy, j * SPW + sub, 64); }
        }
        if (PF && base + stride < Bn) fetch_ids(base + stride);      // next iteration's ids, behind the rows
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
                    // svd_train_val.py:94,170-178: the epoch line's mean NLL (ops.py:125-126 on the fed logits)
                    if (a.loss != 0) acc[2] += fmaxf(logit, 0.f) - logit * r + log1pf(__expf(-fabsf(logit)));
                    if (a.logits) a.logits[k[j]] = logit;                // kept for the AUC (rank sum over the sorted logits)
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

// PNT: user rows by non-temporal loads - set when the user table cannot stay in the 256 MB Infinity Cache
// anyway, so that it does not evict the item rows and the bias arrays, which can
// (tools/probes/fwd_policy.hip: 48.3 -> 46.0 us per 262144 ratings at 10M x 1M rows)
template <int G, int VEC, int MODE, int UNR, bool PNT>
__global__ __launch_bounds__(256) void k_forward(FwdArgs a) {
    warm_args(a);
    forward_body<G, VEC, MODE, UNR, 4, PNT>(a, blockIdx.x, gridDim.x);
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
    warm_args(fa);
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
// K1+K2+K3 for small tables in one launch (k_tile_step).
//
// tile_sort: one 1024-thread block gathers a 1024-entry tile of a batch (dataio.py:115-117 when the
// ids index the resident store), range-checks the ids and sorts the tile by one id column with a
// stable counting sort held entirely in LDS (ballot ranks, ordered wave turns, in-LDS scan).  On
// return cnt[] holds the packed lookup table, srt_key / srt_pos the sorted order, rec_* the tile's
// records by batch position.  Ends with a barrier.
struct TileLds {
    int32_t* cnt; int32_t* rec_u; int32_t* rec_i; float* rec_r; int32_t* srt_key; int32_t* srt_pos; int32_t* wtot;
};
__device__ __forceinline__ void tile_sort(int32_t* cnt_, int32_t* rec_u_, int32_t* rec_i_, float* rec_r_, int32_t* srt_key_,
                                          int32_t* srt_pos_, int32_t* wtot_, const int64_t* ids, const int4* store, const int4* recs,
                                          const int32_t* bu_, const int32_t* bi_, const float* br_, int64_t tile0, int nvalid, int side,
                                          int nb, int64_t N, int64_t U, int64_t I, int32_t* err) {
    TileLds L;
    L.cnt = cnt_; L.rec_u = rec_u_; L.rec_i = rec_i_; L.rec_r = rec_r_; L.srt_key = srt_key_; L.srt_pos = srt_pos_; L.wtot = wtot_;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int b = tid; b < nb; b += 1024) L.cnt[b] = 0;
    const bool valid = tid < nvalid;
    int32_t u = 0, it = 0;
    float r = 0.f;
    if (valid) {
        const int64_t k = tile0 + tid;
        bool oob = false;
        if (recs) {                                      // the batch's store records, left beside the ids by the stream that drew
            const int4 rec = recs[k];                    // them (api.hip RecBuf): one round trip instead of ids -> store
            u = rec.x; it = rec.y; r = __int_as_float(rec.z);
        } else if (ids) {
            int64_t id = ids[k];
            if ((uint64_t)id >= (uint64_t)N) { atomicOr(err, 2); id = 0; }
            const int4 rec = store[id];
            u = rec.x; it = rec.y; r = __int_as_float(rec.z);
        } else {
            u = bu_[k]; it = bi_[k]; r = br_[k];
        }
        if ((uint64_t)(int64_t)u >= (uint64_t)U) { oob = true; u = 0; }
        if ((uint64_t)(int64_t)it >= (uint64_t)I) { oob = true; it = 0; }
        if (oob) atomicOr(err, 1);
    }
    L.rec_u[tid] = u; L.rec_i[tid] = it; L.rec_r[tid] = r;
    __syncthreads();
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
            base = L.cnt[key];
            if (below == 0) L.cnt[key] = base + group_size;
        }
        __syncthreads();
    }
    tile_scan_pack(L.cnt, nb, L.wtot);
    if (valid) {
        const int dst = (L.cnt[key] & 0xffff) + base + rank_in_wave;
        L.srt_key[dst] = key;
        L.srt_pos[dst] = tid;
    }
    __syncthreads();
}

// grid = (nsort + ntiles * G / EPG, 2).
//  * blocks [0, nsort), y == 0: look-ahead - tile_sort one (side, tile) of the NEXT batch and publish its
//    packed table and its sorted records; that work has no dependence on the tables, so it rides
//    here, in the shadow of this step's reduce, instead of heading the next step's critical path.
//  * the other blocks: block (tile, slice, side) owns EPG pieces of 1024/G consecutive entries of the
//    tile's sorted list - read from the records the previous launch published, or, without look-ahead,
//    after repeating the tile's sort itself (every block of a tile does: cheap next to a launch and
//    two more dependent memory round trips).  One lane group per entry loads P[u], Q[i] once, forms
//    the logit and g = dcost/dlogit exactly as K1 does (both sides compute the same g from the same
//    registers), and the runs inside each piece are summed in two levels - inside a wave in registers,
//    across the piece's 16 waves through LDS - always by doubling, i.e. in a fixed tree order (K3).
//    Piece sums land at the tile-sorted position.  The item side also writes the logits and the
//    per-piece {loss, reg, sum g}.  EPG is chosen so that the grid stays within one block per CU.
template <int G, int VEC, int EPG>
__global__ __launch_bounds__(1024) void k_tile_step(TileStepArgs a, int nsort) {
    warm_args(a);
    constexpr int EPB = 1024 / G;                        // entries per piece = lane groups per block
    constexpr int EPS = EPB * EPG;                       // entries per block
    constexpr int NSL = 1024 / EPS;                      // blocks per tile
    extern __shared__ int32_t dyn[];                     // sort: cnt[nbins]; then the wave-level sums (ping-pong)
    __shared__ int32_t rec_u[1024], rec_i[1024], srt_key[1024], srt_pos[1024];
    __shared__ float rec_r[1024];
    __shared__ float lds_gb[2 * EPG * 16];
    __shared__ int32_t lds_key[EPG * 16];
    __shared__ float lds_stage[EPG * (2 * EPB + 1024)];
    __shared__ int32_t wtot[16];
    static_assert(sizeof(rec_u) + sizeof(rec_i) + sizeof(srt_key) + sizeof(srt_pos) + sizeof(rec_r) + sizeof(lds_gb) + sizeof(lds_key) +
                  sizeof(lds_stage) + sizeof(wtot) == tile_step_static_lds(G, EPG), "tile_step_static_lds() is out of date");
    static_assert(tile_step_static_lds(G, EPG) + 64 * 1024 <= LDS_PER_CU, "static LDS leaves no room for 16384 sort bins");
    struct Stamp {                                       // diagnostic (TFR_TILE_DEBUG): when did this block start, pass its phases, end?
        unsigned long long* p; unsigned long long t0;
        __device__ void mark(int k) const { if (p && threadIdx.x == 0) p[k] = __builtin_amdgcn_s_memrealtime(); }
        __device__ ~Stamp() { if (p && threadIdx.x == 0) { p[0] = t0; p[1] = __builtin_amdgcn_s_memrealtime(); } }
    } stamp = {a.dbg ? a.dbg + 8 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) : nullptr, a.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull};
    const int tid = threadIdx.x;
    const bool ahead = (int)blockIdx.x < nsort;          // look-ahead block: sort (side, tile) of the next batch
    if (ahead && blockIdx.y) return;
    const int bx = ahead ? 0 : (int)blockIdx.x - nsort;
    const int side = ahead ? (int)blockIdx.x / a.next_ntiles : (int)blockIdx.y;      // 0: user rows, 1: item rows
    const int tile = ahead ? (int)blockIdx.x % a.next_ntiles : bx / NSL;
    const int slice = bx % NSL;
    const int64_t tile0 = (int64_t)tile * 1024;
    const int64_t Bt = ahead ? a.next_B : a.B;
    const int nvalid = (Bt - tile0 < 1024) ? (int)(Bt - tile0) : 1024;
    if (!ahead && slice * EPS >= nvalid) {               // a short last tile: nothing in this slice
        if (side == 1 && tid < 4 * EPG) a.partials[(size_t)bx * EPG * 4 + tid] = 0.f;
        return;
    }
    const bool presorted = a.srt[0] != nullptr;
    const int nb = a.nbins[side];
    if (ahead || !presorted) {
        tile_sort(dyn, rec_u, rec_i, rec_r, srt_key, srt_pos, wtot, ahead ? a.next_ids : a.ids, a.store, ahead ? a.next_recs : a.recs,
                  a.u, a.it, a.r, tile0, nvalid, side, nb, a.N, a.U, a.I, a.err);
        if (ahead) {                                     // publish the packed table and the sorted records
            for (int b = tid; b < nb; b += 1024) a.next_tab[side][(size_t)tile * nb + b] = dyn[b];
            if (tid < nvalid) {
                const int pl = srt_pos[tid];
                a.next_srt[side][tile0 + tid] = make_int4(rec_u[pl], rec_i[pl], __float_as_int(rec_r[pl]), pl);
            }
            return;
        }
        // the tile's packed lookup table, written once: its bins are dealt round the tile's active blocks
        const int nact = (nvalid + EPS - 1) / EPS;
        for (int b = slice * 1024 + tid; b < nb; b += nact * 1024) a.tab[side][(size_t)tile * nb + b] = dyn[b];
        __syncthreads();                                 // cnt is dead from here: its memory takes the contributions
    }

    // ---- this block's entries (EPG per lane group, one from each piece): forward + contribution
    const int grp = tid / G, gl = tid % G, d0 = gl * VEC;
    const int D = a.D;
    const float mu = *a.mu;
    int jl[EPG], bpos[EPG];
    int32_t row[EPG], uu[EPG], ii[EPG];
    float rr[EPG];
    bool ev[EPG], pstart[EPG];
    int4 rec_[EPG];
    int32_t prev_[EPG];
#pragma unroll
    for (int h = 0; h < EPG; ++h) {
        jl[h] = slice * EPS + h * EPB + grp;
        ev[h] = jl[h] < nvalid;
        rec_[h] = make_int4(0, 0, 0, 0); prev_[h] = -2;
    }
    if (presorted) {                                     // the published records of all EPG entries in one round trip (see the
#pragma unroll                                           // row gathers below for why they are not written `if (ev) rec = *sr`)
        for (int h = 0; h < EPG; ++h) {
            const int js = ev[h] ? jl[h] : 0;            // slot 0 of the tile always exists
            const int4* sr = a.srt[side] + tile0 + js;
            rec_[h] = *sr;
            prev_[h] = reinterpret_cast<const int32_t*>(sr - (js > 0 ? 1 : 0))[side];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int h = 0; h < EPG; ++h) {
        row[h] = -1; uu[h] = 0; ii[h] = 0; rr[h] = 0.f; bpos[h] = 0;
        int32_t prev = -2;
        if (ev[h]) {
            if (presorted) {
                const int4 rec = rec_[h];
                uu[h] = rec.x; ii[h] = rec.y; rr[h] = __int_as_float(rec.z); bpos[h] = rec.w;
                row[h] = side == 0 ? rec.x : rec.y;
                if (jl[h] > 0) prev = prev_[h];
            } else {
                row[h] = srt_key[jl[h]];
                if (jl[h] > 0) prev = srt_key[jl[h] - 1];
                bpos[h] = srt_pos[jl[h]];
                uu[h] = rec_u[bpos[h]]; ii[h] = rec_i[bpos[h]]; rr[h] = rec_r[bpos[h]];
            }
        }
        pstart[h] = ev[h] && (prev != row[h] || grp == 0);       // run head, or first entry of the piece
    }
    if (a.dbg) { if (row[0] == -12345) return; stamp.mark(2); }       // records have arrived
    // All row and bias gathers of the block's entries in flight together: unconditional loads from addresses that are
    // always valid (entry slots past the tile's end read row 0, lanes past the row's end its first words), a scheduling
    // barrier, and only then the zeroing of what did not count.  Written as `if (ev) p = load_frag(...)` the compiler
    // merged the loaded registers with the zeros of the other path by copies behind an s_waitcnt vmcnt(0), so the P row
    // of an entry had to arrive before its Q row was requested, and entry 0 before entry 1: three memory round trips
    // instead of one (ISA of k_tile_step<16,4,2>).
    Frag<VEC> p[EPG], q[EPG];
    float bu_[EPG], bi_[EPG];
    {
        const bool in = d0 < D;
        const int dd = in ? d0 : 0;
#pragma unroll
        for (int h = 0; h < EPG; ++h) {
            const float* pr = a.P + (size_t)(ev[h] ? uu[h] : 0) * D + dd;
            const float* qr = a.Q + (size_t)(ev[h] ? ii[h] : 0) * D + dd;
            if constexpr (VEC == 4) {
                const floatx4 tp = *reinterpret_cast<const floatx4*>(pr), tq = *reinterpret_cast<const floatx4*>(qr);
                p[h].v[0] = tp.x; p[h].v[1] = tp.y; p[h].v[2] = tp.z; p[h].v[3] = tp.w;
                q[h].v[0] = tq.x; q[h].v[1] = tq.y; q[h].v[2] = tq.z; q[h].v[3] = tq.w;
            } else {
                p[h].v[0] = *pr; q[h].v[0] = *qr;
            }
            bu_[h] = a.bu[ev[h] ? uu[h] : 0];
            bi_[h] = a.bi[ev[h] ? ii[h] : 0];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < EPG; ++h) {
            const bool keep = ev[h] && in;
#pragma unroll
            for (int e = 0; e < VEC; ++e) { p[h].v[e] = keep ? p[h].v[e] : 0.f; q[h].v[e] = keep ? q[h].v[e] : 0.f; }
            bu_[h] = ev[h] ? bu_[h] : 0.f;
            bi_[h] = ev[h] ? bi_[h] : 0.f;
        }
    }
    Frag<VEC> acc[EPG];
    float gb[EPG];
    float facc[3 * EPG];                                 // per piece {loss, reg, sum g}: results do not depend on EPG
#pragma unroll
    for (int h = 0; h < 3 * EPG; ++h) facc[h] = 0.f;
#pragma unroll
    for (int h = 0; h < EPG; ++h) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[h].v[e] = 0.f;
        gb[h] = 0.f;
        float sdot = 0.f, sq = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float qv = q[h].v[e];
            sdot = fmaf(p[h].v[e], a.item_abs ? fabsf(qv) : qv, sdot);
            sq = fmaf(p[h].v[e], p[h].v[e], fmaf(qv, qv, sq));
        }
        sdot = group_sum<G>(sdot);
        if (ev[h]) {
            const float logit = ((sdot + mu) + bu_[h]) + bi_[h];   // ops.py:45-47 order
            float gk, l;
            if (a.loss == 0) {                           // ops.py:124
                gk = logit - rr[h];
                l = 0.5f * gk * gk;
            } else {                                     // ops.py:125-126
                gk = sigmoidf_(logit) - rr[h];
                l = fmaxf(logit, 0.f) - logit * rr[h] + log1pf(__expf(-fabsf(logit)));
            }
            if (side == 1) {
                if (gl == 0) {
                    if (a.logits) a.logits[tile0 + bpos[h]] = logit;
                    facc[3 * h + 0] = l;
                    facc[3 * h + 2] = gk;
                    if (a.reg_bias) sq = fmaf(bu_[h], bu_[h], fmaf(bi_[h], bi_[h], sq));
                }
                facc[3 * h + 1] = 0.5f * sq;             // tf.nn.l2_loss = sum(x^2)/2
            }
            const float ob = side == 0 ? bu_[h] : bi_[h];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float ov = side == 0 ? p[h].v[e] : q[h].v[e];
                float xv = side == 0 ? q[h].v[e] : p[h].v[e];
                if (side == 0) { if (a.item_abs) xv = fabsf(xv); }
                else if (a.item_abs) xv = xv * ((ov > 0.f) ? 1.f : ((ov < 0.f) ? -1.f : 0.f));
                acc[h].v[e] = fmaf(gk, xv, a.lam * ov);
            }
            gb[h] = a.reg_bias ? (gk + a.lam * ob) : gk;
        }
    }
    if (a.dbg) { if (gb[0] == -12345.f) return; stamp.mark(3); }      // rows have arrived, contributions formed
    // ---- K3 inside each piece, in two levels, every sum in a fixed tree order (bit-identical run to
    //      run however long the runs are):
    //      1. a wave holds GPW consecutive entries of the piece: suffix sums within runs by doubling,
    //         in registers (cross-lane moves, no LDS memory, no barrier);
    //      2. the 16 waves' leading-run sums: the same doubling over 16 values per piece through LDS
    //         (a value whose run ends within reach is final and drops out);
    //      3. an entry whose run reaches the end of its wave adds the next wave's total.
    constexpr int GPW = 64 / G;
    const int wv = tid >> 6, gw = grp % GPW;
#pragma unroll
    for (int d = 1; d < GPW; d <<= 1) {
#pragma unroll
        for (int h = 0; h < EPG; ++h) {
            const int32_t krow = __shfl_down(row[h], d * G, 64);
            float xs[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) xs[e] = __shfl_down(acc[h].v[e], d * G, 64);
            const float xg = __shfl_down(gb[h], d * G, 64);
            if (ev[h] && gw + d < GPW && krow == row[h]) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[h].v[e] += xs[e];
                gb[h] += xg;
            }
        }
    }
    float* wbuf[2] = {reinterpret_cast<float*>(dyn), reinterpret_cast<float*>(dyn) + EPG * 16 * G * VEC};
    float* wgb[2] = {lds_gb, lds_gb + EPG * 16};
    bool reach[EPG], live[EPG];
    Frag<VEC> tacc[EPG];
    float tgb[EPG];
#pragma unroll
    for (int h = 0; h < EPG; ++h) {
        reach[h] = ev[h] && __shfl(row[h], (GPW - 1) * G + gl, 64) == row[h];     // run reaches the wave's last entry
        tacc[h] = acc[h];
        tgb[h] = gb[h];
        live[h] = gw == 0;                               // one lane group per wave carries the wave's value
        if (gw == 0) {
            const int w0 = h * 16 + wv;
#pragma unroll
            for (int e = 0; e < VEC; ++e) wbuf[0][(w0 * G + gl) * VEC + e] = tacc[h].v[e];
            if (gl == 0) { wgb[0][w0] = tgb[h]; lds_key[w0] = row[h]; }
        }
    }
    stamp.mark(4);                                       // wave-level sums staged
    if (side == 1) block_sum_pieces<G, EPG>(facc, lds_stage, a.partials + (size_t)bx * EPG * 4);   // has the barrier
    else __syncthreads();
    stamp.mark(5);
    int cur = 0;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
#pragma unroll
        for (int h = 0; h < EPG; ++h) {
            if (live[h]) {
                const int w0 = h * 16 + wv;
                const bool take = row[h] >= 0 && wv + d < 16 && lds_key[w0 + d] == row[h];
                if (take) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) tacc[h].v[e] += wbuf[cur][((w0 + d) * G + gl) * VEC + e];
                    tgb[h] += wgb[cur][w0 + d];
                } else {
                    live[h] = false;
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) wbuf[cur ^ 1][(w0 * G + gl) * VEC + e] = tacc[h].v[e];
                if (gl == 0) wgb[cur ^ 1][w0] = tgb[h];
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    stamp.mark(6);                                       // the four doubling rounds over the waves
#pragma unroll
    for (int h = 0; h < EPG; ++h) {
        const int w1 = h * 16 + wv + 1;
        if (reach[h] && wv + 1 < 16 && lds_key[w1] == row[h]) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[h].v[e] += wbuf[cur][(w1 * G + gl) * VEC + e];
            gb[h] += wgb[cur][w1];
        }
    }
#pragma unroll
    for (int h = 0; h < EPG; ++h) {
        if (pstart[h]) {
            const int64_t j = tile0 + jl[h];
            store_frag<VEC>(a.grad_rows[side] + (size_t)j * D, d0, D, acc[h]);
            if (gl == 0) a.grad_bias[side][j] = gb[h];
        }
    }
}

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
    m = fmaf(m, c.b1, g * c.omb1);                       // which product is fused is pinned here: left to the compiler it
    v = fmaf(v, c.b2, (g * g) * c.omb2);                 // differed between instantiations of the same kernel
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
//     FAST (full-width rows; the fused big-table step, the row-sharded step's sides and owners' apply, the FM backward -
//     launch_seg_reduce checks the conditions): the same arithmetic with the loads issued in three dependent rounds instead
//     of eight - {sorted id, neighbour, position}, then the words the row id / position alone address (partner id, table
//     selector, rating or g, biases) with the rows the row id alone addresses behind them (m / v, the user side's own row),
//     then the partner row and the item side's own row - so a block spends one HBM latency on its rows, not three in a row.
template <int G, int VEC, int RMODE, bool FWD = false, bool LEAN = true, bool FAST = false>
__global__ __launch_bounds__(1024) void k_seg_reduce(RedPair pr) {
    // (no warm_args here: thousands of workgroups per launch, the lines are in the scalar cache after the first ones - it cost
    // the user side 172 -> 180 us and, at 64 VGPRs, the item side a block per CU)
    constexpr bool MV_BRANCH = FWD;                      // FAST item side: m / v rows loaded under `if (head)` only, never merged with zeros
    constexpr int EPB = 1024 / G;
    __shared__ float lds_t[EPB * G * VEC];
    __shared__ float lds_gb[EPB];
    __shared__ int32_t lds_key[EPB];
    __shared__ float lds_stage[FWD ? 2 * EPB + 1024 : 1];
    // FAST item side with the forward inside: the own row waits in LDS for the update after the walk (it cannot stay in
    // registers - four of them decide whether two blocks fit a CU - and re-reading it from memory is a round trip in every
    // block's tail and, the first read being a streaming one, 118 MB of traffic per launch at C3's shape)
    constexpr bool OWN_LDS = FAST && FWD && LEAN && RMODE == RMODE_ADAM;
    __shared__ float lds_o[OWN_LDS ? EPB * G * VEC : 1];
    static_assert(sizeof(lds_t) + sizeof(lds_gb) + sizeof(lds_key) + sizeof(lds_stage) + (FWD ? 16 * 3 * 4 : 0) + (OWN_LDS ? sizeof(lds_o) : 0)
                      <= seg_reduce_static_lds(G, VEC, FWD), "seg_reduce_static_lds() is out of date");
    const RedArgs& a = pr.a[blockIdx.y];
    const int32_t err = *a.err;
    const int grp = threadIdx.x / G;
    const int gl = threadIdx.x % G;
    const int d0 = gl * VEC;
    const int D = a.D;
    const int64_t blk0 = (int64_t)blockIdx.x * EPB;
    const int64_t j = blk0 + grp;
    const int64_t Bn = a.dB ? (int64_t)*a.dB : a.B;      // row-sharded step: the entry count lives on the device
    const bool valid = j < Bn;
    int32_t row = -1, prev = -2, pos = 0;
    if (valid) {                                         // three independent loads, one round trip
        row = a.ks[j];
        pos = a.ps[j];
        prev = a.ks[j > 0 ? j - 1 : 0];
        if (j == 0) prev = -2;
    }
    if (err || blk0 >= Bn) {                 // an out-of-range id voids the whole step (block-uniform)
        // a grid sized for a capacity (row-sharded step: the entry count lives on the device): the blocks past the
        // entries still owe K4 their - empty - partial sums
        if constexpr (FWD) { if (!err && a.dB && threadIdx.x < 4) a.partials[(size_t)blockIdx.x * 4 + threadIdx.x] = 0.f; }
        return;
    }
    // the key the next block starts with (does this block's last run continue there?): block-uniform, asked for now and not
    // after the walk, where it was one more dependent round trip at the end of every block
    const int32_t nextkey = (blk0 + EPB < Bn) ? a.ks[blk0 + EPB] : -3;
    // tile mode: the sorted order is per 1024-entry tile, so a run also starts at every tile start
    const bool head = valid && (prev != row || (a.tile && (j % a.tile) == 0));
    const bool pstart = valid && (head || grp == 0);
    const AdamC c = {a.alpha, a.b1, a.b2, a.eps, 1.f - a.b1, 1.f - a.b2};
    const size_t roff = (size_t)(valid ? row : 0) * D;
    const size_t ooff = a.ostride ? (size_t)(valid ? row : 0) * a.ostride : roff;     // own rows read from a packed exchange buffer
    const size_t oboff = a.obstride ? (size_t)(valid ? row : 0) * a.obstride : (size_t)(valid ? row : 0);

    Frag<VEC> o, t, mrow, vrow;
    float ob = 0.f, tb = 0.f, mb = 0.f, vb = 0.f;
    int32_t cur = 0;                                     // two-table form: the table this entry's own row is in
    float facc[3] = {0.f, 0.f, 0.f};                     // FWD: this lane's {loss, reg, g} share
#pragma unroll
    for (int q = 0; q < VEC; ++q) { o.v[q] = 0.f; t.v[q] = 0.f; }
    if constexpr (!(FAST && MV_BRANCH)) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) { mrow.v[q] = 0.f; vrow.v[q] = 0.f; }
    }
    bool owner_side = false;
    if constexpr (!FAST) owner_side = valid && a.rows_in;
    // FAST: the loads, branch-free (lanes past the entries and non-head lanes read row / position 0 and drop the result) and in
    // three rounds.  `cond ? a.p : a.q` on two kernel-argument pointers would be compiled into a VECTOR load of the pointer from
    // the argument block at a selected offset - a dependent round trip of its own; the table is chosen by an offset instead.
    Frag<VEC> xf;
    // (lam through a scalar-register read: `ent ? record.z : a.lam` would otherwise become ONE vector load from a selected
    // address - the argument block or the record - with a round trip of its own)
    float pbf = 0.f, rvf = 0.f, gkf = 0.f, lamf = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a.lam)));
    int32_t pidf = 0;
    bool fast_owner = false;
    if constexpr (FAST && !FWD) fast_owner = a.rows_in != nullptr;
    if (fast_owner) {
        // owner side of the row-sharded step: the contributions are gradient rows already reduced by the peers (by arrival
        // position); everything is addressed by the row id / position alone - one round
        const int32_t rowh = head ? row : 0;
        const int rstr = a.rstride ? a.rstride : D;
        t = load_full<VEC, true>(a.rows_in + (size_t)pos * rstr + d0);
        tb = a.bias_in[(size_t)pos * (a.rbstride ? a.rbstride : 1)];
        o = load_full<VEC, true>(a.own + roff + d0);
        ob = a.own_bias[valid ? row : 0];
        if constexpr (RMODE == RMODE_ADAM) {
            const size_t mvoff = (head ? roff : 0) + d0;
            mrow = load_full<VEC, true>(a.m + mvoff);
            vrow = load_full<VEC, true>(a.v + mvoff);
            mb = a.bias_m[rowh];
            vb = a.bias_v[rowh];
        }
        if (!valid) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) t.v[q] = 0.f;
            tb = 0.f;
        }
    } else if constexpr (FAST) {
        const int32_t rowc = valid ? row : 0;
        const int32_t rowh = head ? row : 0;
        // round 2: the words the row id / position alone address (ids, selector, rating or g, biases: caches)
        const bool two_p = a.osel_in != nullptr;         // partner ids carry the table bit (two-table form, user side)
        const int32_t* idp = two_p ? a.osel_in : a.other;
        int32_t pidw;
        if constexpr (FWD) {
            pidw = idp[pos];
            rvf = a.r[pos];
        } else if (a.ent) {                              // FM: partner row, g x and the entry's coefficient in one 16-byte record
            const int4 e = a.ent[pos];
            pidw = e.x; gkf = __int_as_float(e.y); lamf = __int_as_float(e.z);
        } else {
            pidw = idp[pos];
            gkf = a.g[pos];
        }
        const int32_t* selp = a.sel ? a.sel : a.ks;      // no selector: a word that is there anyway, dropped
        int32_t curw = selp[a.sel ? rowc : 0];
        if (!a.sel) curw = 0;
        ob = a.own_bias[(size_t)rowc * (a.obstride ? a.obstride : 1)];
        if constexpr (RMODE == RMODE_ADAM) {
            mb = a.bias_m[rowh];
            vb = a.bias_v[rowh];
        }
        // the rows.  What needs the row id only goes out now, BEHIND the words above in issue order (compiler barriers; the
        // counter of outstanding loads retires in order, so the wait for the partner id and the selector leaves these in
        // flight): the side without a selector (user side, FM) asks for its own row and the m / v rows, the item side for the
        // m / v rows of run heads (a fifth of its entries repeat a row and skip them: 190 -> 184 us, Zipf 146 -> 136; the user
        // side - hardly any repeats - reads row 0 for them and stays branch-free, 175.5 against 177.3 us).
        pidf = two_p ? (pidw & 0x7fffffff) : pidw;
        const int ostr = a.ostride ? a.ostride : D, pstr = a.pstride ? a.pstride : D;
        constexpr bool EARLY = !FWD;
        asm volatile("" ::: "memory");
        if constexpr (EARLY) {
            o = load_full<VEC, true>(a.own + ((size_t)rowc * ostr + d0));
            if constexpr (RMODE == RMODE_ADAM) {
                const size_t mvoff = (head ? roff : 0) + d0;
                mrow = load_full<VEC, true>(a.m + mvoff);
                vrow = load_full<VEC, true>(a.v + mvoff);
            }
        } else if constexpr (RMODE == RMODE_ADAM) {
            if (head) {
                mrow = load_full<VEC, true>(a.m + roff + d0);
                vrow = load_full<VEC, true>(a.v + roff + d0);
            }
        }
        asm volatile("" ::: "memory");
        // last round: the partner row (and the item side's own row, whose table the selector names).  `tok` is 0, but the
        // compiler only learns that from an instruction that reads the partner id and the selector: these addresses carry it,
        // so the loads cannot be scheduled ahead of the wait for those two words
        int32_t tok;
        asm volatile("v_mov_b32 %0, 0" : "=v"(tok) : "v"(pidw), "v"(curw));
        cur = EARLY ? 0 : curw;
        const int d0t = d0 + tok;
        const ptrdiff_t oalt = a.own_alt ? a.own_alt - a.own : 0;
        const ptrdiff_t palt = two_p ? a.partner_alt - a.partner : 0;
        xf = load_full<VEC, true>(a.partner + ((ptrdiff_t)((size_t)pidf * pstr + d0t) + ((pidw < 0) ? palt : 0)));
        // (the item side's own rows are the user side's partner rows a moment later: default policy, so that they stay cached)
        if constexpr (!EARLY) o = load_full<VEC, !OWN_LDS>(a.own + ((ptrdiff_t)((size_t)rowc * ostr + d0t) + (cur ? oalt : 0)));
        if constexpr (FWD) pbf = a.partner_bias[pidf];
    }
    if (fast_owner) {
        // loaded above
    } else if (owner_side) {
        // owner side of the sharded step: the contribution is a gradient row already reduced by
        // a peer (it includes that peer's lam * Q[i] terms); just add them up in arrival order
        t = load_frag<VEC>(a.rows_in + (size_t)pos * (a.rstride ? a.rstride : D), d0, D);
        tb = a.bias_in[(size_t)pos * (a.rbstride ? a.rbstride : 1)];
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
        int32_t pid = 0;
        float lam_e;
        const float* ptab = a.partner;
        bool have = false;
        float pb = 0.f, rv = 0.f;                        // FWD: partner bias, rating
        Frag<VEC> x;
      if constexpr (FAST) {
        x = xf; pb = pbf; rv = rvf; gk = gkf; pid = pidf;
        lam_e = lamf;
        if constexpr (FWD) { if (gl == 0 && a.osel_out) a.osel_out[pos] = row | (cur << 31); }
      } else {
        if constexpr (!FWD) {
            if (a.ent) {                                             // FM: one 16-byte record per entry
                const int4 e = a.ent[pos];
                pid = e.x; gk = __int_as_float(e.y); lam_e = __int_as_float(e.z);
                have = true;
            }
        }
        if (!have) {
            if constexpr (!FWD) gk = a.g[pos];
            pid = a.partner_by_pos ? 0 : (a.osel_in ? a.osel_in[pos] : a.other[pos]);   // not needed when the partner row comes by position
            if (a.osel_in) { if (pid < 0) ptab = a.partner_alt; pid &= 0x7fffffff; }   // the table that still holds the pre-update row
            lam_e = a.lam;
        }
        x = a.partner_by_pos ? load_frag_h<VEC>(a.partner_by_pos + (size_t)pos * D, d0, D, a.nt & 8)
                             : load_frag_h<VEC>(ptab + (size_t)pid * (a.pstride ? a.pstride : D), d0, D, a.nt & 1);
        if (a.sel) {                                                 // two-table form: which table holds this row now
            cur = a.sel[row];
            if (gl == 0) a.osel_out[pos] = row | (cur << 31);
        }
        o = load_frag_h<VEC>((cur ? a.own_alt : a.own) + ooff, d0, D, a.nt & 2);
        ob = a.own_bias[oboff];
        if constexpr (RMODE == RMODE_ADAM) {
            if (head) {
                mrow = load_frag_h<VEC>(a.m + roff, d0, D, a.nt & 2);
                vrow = load_frag_h<VEC>(a.v + roff, d0, D, a.nt & 2);
                mb = a.bias_m[row];
                vb = a.bias_v[row];
            }
        }
        // this side updates its table in place before the other side runs: leave the other side the
        // pre-update row it needs, per entry
        if (a.own_copy_out) store_frag_h<VEC>(a.own_copy_out + (size_t)pos * D, d0, D, o, a.nt & 16);
        if constexpr (FWD) { pb = a.partner_bias[pid]; rv = a.r[pos]; }
      }
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
            const float logit = ((sdot + *a.mu) + pb) + ob;
            const float r = rv;
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
            t.v[q] = fmaf(gk, xv, lam_e * o.v[q]);        // pinned: left to the compiler, the fused product differed between instantiations
        }
        tb = a.reg_bias ? (gk + a.lam * ob) : gk;
    }
    // contributions to LDS ([entry][G*VEC], a lane's VEC floats contiguous)
#pragma unroll
    for (int q = 0; q < VEC; ++q) lds_t[(grp * G + gl) * VEC + q] = t.v[q];
    if constexpr (OWN_LDS) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) lds_o[(grp * G + gl) * VEC + q] = o.v[q];
    }
    if (gl == 0) { lds_gb[grp] = tb; lds_key[grp] = row; }
    if constexpr (FWD) {                                 // has the barrier
        if (a.stage_sum) block_sum_pieces<G, 1>(facc, lds_stage, a.partials + (size_t)blockIdx.x * 4);
        else block_sum_store<3, 16>(facc, a.partials + (size_t)blockIdx.x * 4);
    } else __syncthreads();
    if (!pstart) return;

    // where the row is (two-table form: the table sel[row] named) and where its new value goes (the other table)
    const ptrdiff_t walt = a.sel ? a.own_w_alt - a.own_w : 0;
    const float* const wsrc = a.own_w + ((ptrdiff_t)roff + (cur ? walt : 0));
    float* const wdst = a.own_w + ((ptrdiff_t)roff + (cur ? 0 : walt));
    // FWD variant: the own row is re-read (an L2 hit) rather than kept live across the forward and the walk - four registers
    // that decide whether two blocks fit a CU; FAST asks for it before the walk, not after it
    Frag<VEC> wre;
    if constexpr (OWN_LDS) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) wre.v[q] = lds_o[(grp * G + gl) * VEC + q];
    }

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
    if (e == EPB && blk0 + EPB < Bn && !(a.tile && ((blk0 + EPB) % a.tile) == 0)) cont = (nextkey == row);
    const bool whole = head && !cont;
    if (RMODE == RMODE_SCRATCH && whole && a.dense_rows) {
        // dense-gradient form (TF1 Adam sweep / data parallel): a run that lies in one block goes
        // straight to its row of the dense buffer - the consumer streams it with no indirection
        store_frag<VEC>(a.dense_rows + (a.dstride ? (size_t)row * a.dstride : roff), d0, D, acc);
        if (gl == 0) a.dense_bias[a.dbstride ? (size_t)row * a.dbstride : (size_t)row] = gb;
    } else if (RMODE == RMODE_SCRATCH || !whole) {
        store_frag<VEC>(a.grad_rows + (size_t)j * D, d0, D, acc);
        if (gl == 0) {
            a.grad_bias[j] = gb;
            if (head && a.map) a.map[row] = (int32_t)j + 1;      // split run: where its pieces start
        }
    } else if constexpr (RMODE == RMODE_ADAM) {
        if (!a.frozen_rows) {
            Frag<VEC> w;
            if constexpr (OWN_LDS) w = wre;
            else w = (FWD && LEAN) ? load_frag<VEC>(wsrc, d0, D) : o;
#pragma unroll
            for (int q = 0; q < VEC; ++q) adam_sparse(w.v[q], mrow.v[q], vrow.v[q], acc.v[q], c);
            // two-table form: the new row goes to the other table (the user side still reads the old one), then the row flips
            const bool nts = FAST || (a.nt & 4);
            store_frag_h<VEC>(wdst, d0, D, w, nts);
            if (a.sel && gl == 0) a.sel[row] = cur ^ 1;
            store_frag_h<VEC>(a.m + roff, d0, D, mrow, nts);
            store_frag_h<VEC>(a.v + roff, d0, D, vrow, nts);
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
            store_frag<VEC>(wdst, d0, D, w);
            if (a.sel && gl == 0) a.sel[row] = cur ^ 1;
        }
        if (gl == 0 && !a.frozen_bias) a.bias_w[row] = ob - a.lr * gb;
    }
}

// a run's reduced gradient = its pieces added in piece order (head piece first); pieces
// start at the run head and at every multiple of PIECE (= the reduce kernel's block span).
// Continuation pieces are probed and loaded N at a time (independent loads, added in the order walked), so a hot row
// split into a handful of pieces costs two memory round trips, not one per piece.  `step` = distance between the
// pieces this caller walks (PIECE for all of them, a multiple when several lane groups share a run).
template <int VEC, int N>
__device__ __forceinline__ bool add_pieces(Frag<VEC>& t, float& gb, const float* __restrict__ grad_rows,
                                           const float* __restrict__ grad_bias, const int32_t* __restrict__ ks,
                                           int64_t B, int64_t safe, int64_t p, int64_t step, int32_t row, int d0, int D) {
    int32_t key[N];                                      // unconditional loads from in-bounds addresses: a load behind
#pragma unroll                                           // `pp < B &&` is a branch and a full wait per probe
    for (int q = 0; q < N; ++q) {
        const int64_t pp = p + (int64_t)q * step;
        key[q] = ks[pp < B ? pp : safe];
    }
    Frag<VEC> x[N];
    float xb[N];
#pragma unroll
    for (int q = 0; q < N; ++q) {
        const int64_t pp = (p + (int64_t)q * step < B) ? p + (int64_t)q * step : safe;
        x[q] = load_frag<VEC>(grad_rows + (size_t)pp * D, d0, D);
        xb[q] = grad_bias[pp];
    }
    bool go = true;
#pragma unroll
    for (int q = 0; q < N; ++q) {
        go = go && (p + (int64_t)q * step < B) && key[q] == row;     // pieces are contiguous: stop at the first miss
        if (go) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) t.v[e] += x[q].v[e];
            gb += xb[q];
        }
    }
    return go;
}

// the narrow walk (k_adam_dense, where registers decide the sweep's occupancy): eight pieces per round trip
template <int VEC, int PIECE>
__device__ __forceinline__ Frag<VEC> run_total(const float* __restrict__ grad_rows,
                                               const float* __restrict__ grad_bias,
                                               const int32_t* __restrict__ ks, int64_t B, int64_t j,
                                               int32_t row, int d0, int D, float& gb) {
    Frag<VEC> t = load_frag<VEC>(grad_rows + (size_t)j * D, d0, D);
    gb = grad_bias[j];
    int64_t p = (j / PIECE + 1) * PIECE;
    if (p >= B || ks[p] != row) return t;                // the common case: a single piece
    while (add_pieces<VEC, 8>(t, gb, grad_rows, grad_bias, ks, B, j, p, PIECE, row, d0, D)) p += 8 * (int64_t)PIECE;
    return t;
}

// ------------------------------------------------------------------------------------
// K5a  apply reduced gradients held in scratch to the touched rows only: lazy Adam or SGD
//      (ops.py:143-149).  One lane group per sorted entry; only run heads work.
//      only_split: finish the runs k_seg_reduce could not apply in place (cut in >1 piece).
//      A hot row (popularity-skewed ids: under Zipf(1.05) items the top row of a 262144 batch is cut into ~800
//      pieces) is shared by the block: the owner adds the head piece and the next eight as run_total does; if all
//      eight belonged to the run, lane group g of the block adds pieces 9+g, 9+g+GPB, ... (16 in flight each) and the
//      owner adds the GPB partial sums in group order - a fixed order, so results stay bit-identical run to run.
template <int G, int VEC, int OPT>
__global__ __launch_bounds__(256) void k_apply_rows(ApplyPair pr) {
    warm_args(pr.a[blockIdx.y]);
    constexpr int PIECE = 1024 / G;
    constexpr int GPB = 256 / G;
    constexpr int CW = 16;                               // pieces in flight per lane group in the shared walk
    if (pr.with_fin && blockIdx.x == gridDim.x - 1) {    // the extra block column: K4 (block-uniform branch)
        if (blockIdx.y == 0) finalize_body(pr.f);
        return;
    }
    __shared__ float s_part[256 * VEC];
    __shared__ float s_partb[GPB];
    __shared__ long long s_next[GPB];                    // a long run's next piece position, or -1
    __shared__ int32_t s_row[GPB];
    const ApplyArgs& a = pr.a[blockIdx.y];
    const int32_t err = *a.err;
    const int gl = threadIdx.x % G, grp = threadIdx.x / G;
    const int d0 = gl * VEC;
    const int D = a.D;
    const int64_t Bn = a.dB ? (int64_t)*a.dB : a.B;      // row-sharded step: the entry count lives on the device
    int64_t j = (int64_t)blockIdx.x * GPB + grp;
    int32_t row = 0;
    bool own;                                            // this lane group finishes a run (no early exits: barriers below)
    if (a.only_split) {
        // one lane group per piece boundary p: a run is split iff it crosses one.  The first
        // boundary a run crosses owns it; its head then lies in the PIECE entries before p.
        const int64_t p = (j + 1) * PIECE;
        own = !err && p < Bn;
        if (own) {
            row = a.ks[p];
            const int64_t lo = p - PIECE;
            own = a.ks[p - 1] == row && !(lo > 0 && a.ks[lo - 1] == row);   // crossed here, and not an earlier boundary
            if (own) {
                const int sh = ((threadIdx.x % 64) / G) * G;     // this group's lanes within the wave
                j = p - 1;
                for (int k = 0; k < PIECE; k += G) {             // sorted keys: the matches are a suffix of [lo, p)
                    const bool hit = (k + gl < PIECE) && a.ks[lo + k + gl] == row;
                    unsigned long long bits = __ballot(hit) >> sh;
                    if constexpr (G < 64) bits &= (1ull << G) - 1ull;
                    if (bits) { j = lo + k + (__ffsll((long long)bits) - 1); break; }
                }
            }
        }
    } else {
        own = !err && j < Bn;
        if (own) {
            row = a.ks[j];
            own = ((j > 0) ? a.ks[j - 1] : -2) != row;           // run heads only
        }
    }
    const size_t roff = (size_t)row * D;
    Frag<VEC> w, mrow, vrow, gr;
#pragma unroll
    for (int q = 0; q < VEC; ++q) { w.v[q] = 0.f; mrow.v[q] = 0.f; vrow.v[q] = 0.f; gr.v[q] = 0.f; }
    float gb = 0.f;
    long long next = -1;
    float* wtab = a.w;
    if (own) {
        if (a.sel && a.sel[row]) wtab = a.w_alt;          // two-table form: finish the row where it lives
        if (OPT != 2 && !a.frozen_rows) {                // issued before the piece walk
            w = load_frag<VEC>(wtab + roff, d0, D);
            if constexpr (OPT == 0) {
                mrow = load_frag<VEC>(a.m + roff, d0, D);
                vrow = load_frag<VEC>(a.v + roff, d0, D);
            }
        }
        gr = load_frag<VEC>(a.grad_rows + (size_t)j * D, d0, D);
        gb = a.grad_bias[j];
        const int64_t p = (j / PIECE + 1) * PIECE;
        if (p < Bn && a.ks[p] == row &&
            add_pieces<VEC, 8>(gr, gb, a.grad_rows, a.grad_bias, a.ks, Bn, j, p, PIECE, row, d0, D))
            next = p + 8 * (int64_t)PIECE;
    }
    if (gl == 0) { s_next[grp] = next; s_row[grp] = row; }
    if (__syncthreads_or(next >= 0)) {                   // some run of this block is long: share it
        for (int sl = 0; sl < GPB; ++sl) {
            const long long ps = s_next[sl];             // block-uniform
            if (ps < 0) continue;
            const int32_t rs = s_row[sl];
            Frag<VEC> part;
#pragma unroll
            for (int q = 0; q < VEC; ++q) part.v[q] = 0.f;
            float pb = 0.f;
            int64_t pp = ps + (int64_t)grp * PIECE;
            while (add_pieces<VEC, CW>(part, pb, a.grad_rows, a.grad_bias, a.ks, Bn, 0, pp, (int64_t)GPB * PIECE, rs, d0, D))
                pp += (int64_t)CW * GPB * PIECE;
#pragma unroll
            for (int q = 0; q < VEC; ++q) s_part[threadIdx.x * VEC + q] = part.v[q];
            if (gl == 0) s_partb[grp] = pb;
            __syncthreads();
            if (grp == sl) {
                for (int g2 = 0; g2 < GPB; ++g2) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) gr.v[q] += s_part[(g2 * G + gl) * VEC + q];
                    gb += s_partb[g2];
                }
            }
            __syncthreads();
        }
    }
    if (!own) return;
    const AdamC c = {a.alpha, a.b1, a.b2, a.eps, 1.f - a.b1, 1.f - a.b2};
    if constexpr (OPT == 2) {                            // sharded step: emit the reduced gradient row
        store_frag<VEC>(a.w + (a.wstride ? (size_t)row * a.wstride : roff), d0, D, gr);
        if (gl == 0) a.bias_w[a.wbstride ? (size_t)row * a.wbstride : (size_t)row] = gb;
        return;
    }
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
        store_frag<VEC>(wtab + roff, d0, D, w);
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
    warm_args(pr.a[blockIdx.y]);                         // (after the y == 2 exit: pr.a has two entries)
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
template <int G, int VEC, bool WRITE, int NT>
__global__ __launch_bounds__(256) void k_dense_tiles(TileDenseLaunch L) {
    warm_args(L.a[blockIdx.y]);
    if (blockIdx.x == gridDim.x - 1) {                   // the extra block column: K4 (optional), nothing else
        if (blockIdx.y == 0 && L.with_fin) finalize_body(L.f);
        return;
    }
    const TileDenseArgs& a = L.a[blockIdx.y];
    const int32_t err = *a.err;                          // looked at behind the first round of loads (they need no lookup table):
    constexpr int GPB = 256 / G;                         // its round trip is then not a round of its own
    constexpr int EPB = 1024 / G;
    const int gl = threadIdx.x % G;
    const int d0 = gl * VEC;
    const int D = a.D;
    const AdamC c = {a.alpha, a.b1, a.b2, a.eps, 1.f - a.b1, 1.f - a.b2};
    for (int64_t row = (int64_t)blockIdx.x * GPB + threadIdx.x / G; row < a.rows;
         row += (int64_t)(gridDim.x - 1) * GPB) {
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
        // one round trip for all NT table entries, then the piece heads in batches of NB whose
        // addresses are all known up front (NT <= 12: a single batch)
        constexpr int NB = (NT > 12) ? 8 : NT;
        int32_t ent[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) ent[t] = (t < a.ntiles) ? a.tab[(size_t)t * a.nbins + row] : 0;
        if (err) return;                                 // a voided step: the tables' contents are not to be trusted as addresses
#pragma unroll
        for (int t0 = 0; t0 < NT; t0 += NB) {
            Frag<VEC> x[NB];
            float xb[NB];
#pragma unroll
            for (int t = 0; t < NB; ++t) {               // predicated: a dummy address would be one hot L2 line
                const int32_t e = ent[t0 + t];
#pragma unroll
                for (int q = 0; q < VEC; ++q) x[t].v[q] = 0.f;
                xb[t] = 0.f;
                if (e >> 16) {
                    const int64_t j = (int64_t)(t0 + t) * 1024 + (e & 0xffff);
                    x[t] = load_frag<VEC>(a.grad_rows + (size_t)j * D, d0, D);
                    xb[t] = a.grad_bias[j];
                }
            }
#pragma unroll
            for (int t = 0; t < NB; ++t) {               // head pieces, added in tile order
                if (ent[t0 + t] >> 16) {
                    touched = true;
#pragma unroll
                    for (int q = 0; q < VEC; ++q) tot.v[q] += x[t].v[q];
                    gb += xb[t];
                }
            }
        }
        // hot rows: runs cut into several pieces.  Round k adds the k-th continuation piece of every
        // tile (loads of a round are independent), so the chain is as long as the most-split run of
        // one tile, not the sum over tiles.  Fixed order: round-major, tile-minor.
        for (int k = 1;; ++k) {
            bool more = false;
#pragma unroll
            for (int t0 = 0; t0 < NT; t0 += NB) {
                Frag<VEC> y[NB];
                float yb[NB];
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    const int32_t e = ent[t0 + t];
                    const int64_t j = (int64_t)(t0 + t) * 1024 + (e & 0xffff);
                    const int64_t pp = (j / EPB + k) * EPB;
                    const bool has = (e >> 16) && pp < j + (e >> 16);
#pragma unroll
                    for (int q = 0; q < VEC; ++q) y[t].v[q] = 0.f;
                    yb[t] = 0.f;
                    if (has) {
                        y[t] = load_frag<VEC>(a.grad_rows + (size_t)pp * D, d0, D);
                        yb[t] = a.grad_bias[pp];
                        more = true;
                    }
                }
#pragma unroll
                for (int t = 0; t < NB; ++t) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) tot.v[q] += y[t].v[q];
                    gb += yb[t];
                }
            }
            if (!more) break;
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

__global__ __launch_bounds__(256) void k_finalize(FinArgs a) { warm_args(a); finalize_body(a); }

// ------------------------------------------------------------------------------------
// AUC on the device (svd_train_val.py:97,173: sklearn's roc_auc_score(rates, sigmoid(logits)) per batch is the host
// bottleneck SURVEY 8f #1 names).  AUC = (sum of the positives' mid-ranks - n_pos (n_pos + 1) / 2) / (n_pos n_neg)
// (Mann-Whitney; equal scores share the mean of their ranks, which is what the ROC trapezoids give).  Scores are
// mapped to 32-bit keys in score order, sorted by the library's radix sort, and every positive finds how many keys
// are smaller / not larger than its own by two binary searches of the sorted keys; twice the mid-rank is the
// integer lo + hi + 1, so the rank sum is exact integer arithmetic (64-bit integer adds: order-independent).
__global__ __launch_bounds__(256) void k_auc_keys(const float* __restrict__ score, int32_t* __restrict__ keys, int64_t n) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        float x = score[k];
        if (x == 0.f) x = 0.f;                           // -0 and +0 are equal scores
        const uint32_t b = __float_as_uint(x);
        keys[k] = (int32_t)((b & 0x80000000u) ? ~b : (b | 0x80000000u));
    }
}

__global__ __launch_bounds__(256) void k_auc_ranksum(const int32_t* __restrict__ ks, const int32_t* __restrict__ ps,
                                                     const float* __restrict__ label, int64_t n, unsigned long long* out /* {2 x rank sum, n_pos} */) {
    unsigned long long s2 = 0, np = 0;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        if (label[ps[j]] > 0.5f) {
            const uint32_t key = (uint32_t)ks[j];
            int64_t lo = 0, hi = n;                      // first index with a key >= mine
            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((uint32_t)ks[mid] < key) lo = mid + 1; else hi = mid; }
            const int64_t first = lo;
            hi = n;                                      // first index with a key > mine
            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((uint32_t)ks[mid] <= key) lo = mid + 1; else hi = mid; }
            s2 += (unsigned long long)(first + lo + 1);  // 1-based ranks first+1 .. lo: twice their mean
            np += 1;
        }
    }
    for (int o = 32; o >= 1; o >>= 1) { s2 += __shfl_down(s2, o, 64); np += __shfl_down(np, o, 64); }
    if ((threadIdx.x & 63) == 0 && np) { atomicAdd(&out[0], s2); atomicAdd(&out[1], np); }
}

void launch_auc_keys(const float* score, int32_t* keys, int64_t n, hipStream_t s) {
    int64_t nb = (n + 255) / 256; if (nb > 2048) nb = 2048; if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_auc_keys, dim3((int)nb), dim3(256), 0, s, score, keys, n);
}
void launch_auc_ranksum(const int32_t* ks, const int32_t* ps, const float* label, int64_t n, unsigned long long* out, hipStream_t s) {
    int64_t nb = (n + 255) / 256; if (nb > 2048) nb = 2048; if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_auc_ranksum, dim3((int)nb), dim3(256), 0, s, ks, ps, label, n, out);
}

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
    const bool pnt = MODE != MODE_TRAIN && (size_t)a.U * a.D * 4 > ((size_t)256 << 20);
#define TFR_FWD_CASE(g, v) \
    if (G == g && VEC == v) {                                                                                  \
        if (pnt) hipLaunchKernelGGL((k_forward<g, v, MODE, UNR, true>), dim3(grid), dim3(256), 0, s, a);       \
        else hipLaunchKernelGGL((k_forward<g, v, MODE, UNR, false>), dim3(grid), dim3(256), 0, s, a);          \
        return;                                                                                                \
    }
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

int tile_step_epg(int ntiles, int G, int VEC) {
    // smallest EPG in {1, 2, 4} that brings the grid (two sides) down to one block per CU (256 CUs), as far
    // as LDS allows: the wave-level ping-pong buffers (2 * EPG * 16 * G * VEC floats) share the 64 KB dynamic
    // region with the sort's bins; static arrays take another ~40 KB of the CU's 160 KB
    auto fits = [&](int e) {
        return (size_t)2 * e * 16 * G * VEC * 4 <= (size_t)64 * 1024 && tile_step_static_lds(G, e) + (size_t)64 * 1024 <= LDS_PER_CU;
    };
    static int forced = -1;                              // TFR_EPG=1|2|4: A/B override
    if (forced < 0) { const char* e = getenv("TFR_EPG"); forced = e ? atoi(e) : 0; }
    if (forced == 1 || ((forced == 2 || forced == 4) && forced <= G && fits(forced))) return forced;
    int epg = 1;
    while (epg < 4 && epg < G && ntiles * (G / epg) * 2 > 256 && fits(2 * epg)) epg *= 2;
    return epg;
}

void launch_tile_step(const TileStepArgs& a, int G, int VEC, hipStream_t s) {
    const int nbmax = a.nbins[0] > a.nbins[1] ? a.nbins[0] : a.nbins[1];
    const int epg = tile_step_epg(a.ntiles, G, VEC);
    size_t dyn = (size_t)nbmax * 4;                       // bins during the sort, contributions afterwards
    if (dyn < (size_t)2 * epg * 16 * G * VEC * 4) dyn = (size_t)2 * epg * 16 * G * VEC * 4;   // wave-level ping-pong buffers of the reduce
    if (tile_step_static_lds(G, epg) + dyn > LDS_PER_CU) return;  // cannot happen for a shape tile_step_epg() admits (tests/test_lds_budget.py)
    const int nsort = a.next_ids ? 2 * a.next_ntiles : 0;         // look-ahead sort blocks come first
    const dim3 grid(nsort + a.ntiles * (G / epg), 2);
#define TFR_TS_LAUNCH(g, v, e)                                                                        \
    {                                                                                                 \
        static size_t attr = 0;              /* static + dynamic LDS must stay within 160 KB: ask for what is needed */ \
        if (dyn > attr) {                                                                             \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_step<g, v, e>),              \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) \
                return;                      /* leaves the error for the caller's hipGetLastError() */ \
            attr = dyn;                                                                               \
        }                                                                                             \
        hipLaunchKernelGGL((k_tile_step<g, v, e>), grid, dim3(1024), dyn, s, a, nsort);               \
    }
#define TFR_TS_CASE(g, v)                                                                             \
    if (G == g && VEC == v) {                                                                         \
        if (epg == 1) TFR_TS_LAUNCH(g, v, 1) else if (epg == 2) TFR_TS_LAUNCH(g, v, 2) else TFR_TS_LAUNCH(g, v, 4) \
        return;                                                                                       \
    }
    TFR_TS_CASE(4, 4) TFR_TS_CASE(8, 4) TFR_TS_CASE(16, 4) TFR_TS_CASE(32, 4) TFR_TS_CASE(64, 4)
    TFR_TS_CASE(4, 1) TFR_TS_CASE(8, 1) TFR_TS_CASE(16, 1) TFR_TS_CASE(32, 1) TFR_TS_CASE(64, 1)
#undef TFR_TS_CASE
#undef TFR_TS_LAUNCH
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
    static int lean = -1;                                // TFR_LEAN=0: A/B switch (own row kept in registers)
    if (lean < 0) { const char* e = getenv("TFR_LEAN"); lean = (e && e[0] == '0') ? 0 : 1; }
    // FAST: full-width rows (D = G * VEC), ids and rows from tables or packed exchange buffers (strides), with or without the
    // two-table item form - the fused big-table step (api.hip run_train_step, `dual`) and the row-sharded step's item / user
    // sides and its owners' apply (pre-reduced rows), the FM backward (16-byte entry records); not the per-position copies of
    // the one-table form or tile mode.
    // Its row loads are non-temporal whatever RedArgs::nt says.
    static int fast_en = -1;                             // TFR_FAST=0: A/B switch
    if (fast_en < 0) { const char* e = getenv("TFR_FAST"); fast_en = (e && e[0] == '0') ? 0 : 1; }
    const RedArgs& a0 = p.a[0];
    const bool fast_own = fast_en && lean && n == 1 && a0.rows_in && a0.bias_in && !fwd && !a0.tile && !a0.sel && !a0.ostride && !a0.obstride &&
                          a0.D == G * VEC && a0.B > 0 && (rmode == RMODE_SCRATCH || a0.own == a0.own_w);
    const bool fast = fast_own ||
                      (fast_en && lean && n == 1 && !a0.rows_in && !a0.partner_by_pos && !a0.own_copy_out && !a0.tile &&
                      a0.D == G * VEC && a0.B > 0 && (a0.other || a0.ent) && (!a0.ent || (!fwd && !a0.osel_in)) && (!a0.osel_in || a0.partner_alt) && (!a0.sel || (a0.own_alt && a0.own_alt == a0.own_w_alt)) &&
                      (rmode == RMODE_SCRATCH || (a0.own == a0.own_w && !a0.ostride)) &&
                      (fwd ? (a0.r && a0.partner_bias && !a0.osel_in) : ((a0.g || a0.ent) && !a0.sel)));
#define TFR_RED_CASE(g, v)                                                                             \
    if (G == g && VEC == v) {                                                                          \
        if (fast && fwd && rmode == RMODE_ADAM) hipLaunchKernelGGL((k_seg_reduce<g, v, RMODE_ADAM, true, true, true>), grid, dim3(1024), 0, s, p); \
        else if (fast && fwd && rmode == RMODE_SCRATCH) hipLaunchKernelGGL((k_seg_reduce<g, v, RMODE_SCRATCH, true, true, true>), grid, dim3(1024), 0, s, p); \
        else if (fast && fwd) hipLaunchKernelGGL((k_seg_reduce<g, v, RMODE_SGD, true, true, true>), grid, dim3(1024), 0, s, p); \
        else if (fast && rmode == RMODE_SCRATCH) hipLaunchKernelGGL((k_seg_reduce<g, v, RMODE_SCRATCH, false, true, true>), grid, dim3(1024), 0, s, p); \
        else if (fast && rmode == RMODE_ADAM) hipLaunchKernelGGL((k_seg_reduce<g, v, RMODE_ADAM, false, true, true>), grid, dim3(1024), 0, s, p); \
        else if (fast) hipLaunchKernelGGL((k_seg_reduce<g, v, RMODE_SGD, false, true, true>), grid, dim3(1024), 0, s, p); \
        else if (fwd && rmode == RMODE_ADAM && !lean) hipLaunchKernelGGL((k_seg_reduce<g, v, RMODE_ADAM, true, false>), grid, dim3(1024), 0, s, p); \
        else if (fwd && rmode == RMODE_ADAM) hipLaunchKernelGGL((k_seg_reduce<g, v, RMODE_ADAM, true>), grid, dim3(1024), 0, s, p); \
        else if (fwd && rmode == RMODE_SCRATCH) hipLaunchKernelGGL((k_seg_reduce<g, v, RMODE_SCRATCH, true>), grid, dim3(1024), 0, s, p); \
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
    const dim3 grid(entry_grid(B, G) + (p.with_fin ? 1 : 0), n);
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
    const dim3 grid((int)nb + 1, 2);                     // + one block column for K4
    TileDenseLaunch LL = L;
    LL.with_fin = with_fin ? 1 : 0;
#define TFR_DT_LAUNCH(g, v, wr, nt) hipLaunchKernelGGL((k_dense_tiles<g, v, wr, nt>), grid, dim3(256), 0, s, LL)
#define TFR_DT_CASE(g, v)                                                                         \
    if (G == g && VEC == v) {                                                                     \
        const int nt = L.a[0].ntiles;                                                             \
        if (write) {                                                                              \
            if (nt <= 4) TFR_DT_LAUNCH(g, v, true, 4); else if (nt <= 8) TFR_DT_LAUNCH(g, v, true, 8);      \
            else if (nt <= 10) TFR_DT_LAUNCH(g, v, true, 10);                                               \
            else if (nt <= 12) TFR_DT_LAUNCH(g, v, true, 12); else TFR_DT_LAUNCH(g, v, true, 16);           \
        } else {                                                                                  \
            if (nt <= 4) TFR_DT_LAUNCH(g, v, false, 4); else if (nt <= 8) TFR_DT_LAUNCH(g, v, false, 8);    \
            else if (nt <= 10) TFR_DT_LAUNCH(g, v, false, 10);                                              \
            else if (nt <= 12) TFR_DT_LAUNCH(g, v, false, 12); else TFR_DT_LAUNCH(g, v, false, 16);         \
        }                                                                                         \
        return;                                                                                   \
    }
    TFR_DT_CASE(4, 4) TFR_DT_CASE(8, 4) TFR_DT_CASE(16, 4) TFR_DT_CASE(32, 4) TFR_DT_CASE(64, 4)
    TFR_DT_CASE(4, 1) TFR_DT_CASE(8, 1) TFR_DT_CASE(16, 1) TFR_DT_CASE(32, 1) TFR_DT_CASE(64, 1)
#undef TFR_DT_LAUNCH
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

// the store records of a run of drawn ids, left beside the ids (api.hip RecBuf; runs on the stream that drew them)
__global__ __launch_bounds__(256) void k_gather_recs(const int64_t* __restrict__ ids, const int4* __restrict__ store, int4* __restrict__ recs,
                                                     int64_t n, int64_t N) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        int64_t id = ids[k];
        if ((uint64_t)id >= (uint64_t)N) id = 0;         // (drawn ids are in range by construction)
        recs[k] = store[id];
    }
}

void launch_gather_recs(const int64_t* ids, const void* store, void* recs, int64_t n, int64_t N, hipStream_t s) {
    int64_t nb = (n + 255) / 256;
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_gather_recs, dim3((unsigned)nb), dim3(256), 0, s, ids, reinterpret_cast<const int4*>(store), reinterpret_cast<int4*>(recs), n, N);
}

// two-table form of the big-table step: bring every row that currently lives in the alternate table back to the main
// one (one wave per row; run before anything but the fused step looks at the item table)
__global__ __launch_bounds__(256) void k_settle_alt(float* __restrict__ main_t, const float* __restrict__ alt_t,
                                                    int32_t* __restrict__ sel, int64_t rows, int D) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        if (!sel[r]) continue;                            // wave-uniform
        for (int d = lane; d < D; d += 64) main_t[(size_t)r * D + d] = alt_t[(size_t)r * D + d];
        if (lane == 0) sel[r] = 0;
    }
}

void launch_settle_alt(float* main_t, const float* alt_t, int32_t* sel, int64_t rows, int D, hipStream_t s) {
    int64_t nb = (rows + 3) / 4;
    if (nb > 16384) nb = 16384;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_settle_alt, dim3((int)nb), dim3(256), 0, s, main_t, alt_t, sel, rows, D);
}

void launch_finalize(const FinArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, s, a);
}

}  // namespace tfr
