// shard.hip - integer routing of the row-sharded step, on the device, with this library's own scan /
// sort kernels: no host round trip, no torch.unique / nonzero / bincount.
//
// Every rank sees the same global batch.  A rank keeps the samples whose USER row it owns (owner = id / per_u, block
// partition) in batch order, finds the distinct ITEM ids among them (stable radix sort by global item id: sorted ids
// are grouped by owner because owner = id / per_i is monotone) and lays the requests out as [world][cap]: slot
// w * cap + k holds the k-th distinct item this rank needs from owner w, as the owner's local row id; unused slots
// hold -1.  The same slot numbers address the rows that come back (and the gradient rows that return), so all three
// exchanges are equal-split all-to-alls of fixed size - nothing variable-length ever has to be known on the host.
// Capacity overflow (more local samples than Bcap, more distinct items for one owner than cap) raises the device
// error flag: the step is void, like an out-of-range id.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "svd_kernels.h"

namespace tfr {

// block-level exclusive rank of `flag` among the block's 1024 threads (in thread order) + the block's total
__device__ __forceinline__ int block_rank_1024(bool flag, int* total) {
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long bal = __ballot(flag);
    const int r = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int c = wsum[w]; if (w < wave) before += c; tot += c; }
    __syncthreads();                                     // wsum may be reused by the caller's next call
    *total = tot;
    return before + r;
}

// sample k of the global batch: columns u / it / r, or (ids != NULL) row ids[k] of the rank's copy of the rating store -
// the fused ShuffleIterator gather, as in the single-GPU step: only the samples a rank owns ever leave the store as columns
__device__ __forceinline__ bool route_sample(const RouteArgs& a, int64_t k, int32_t* u, int32_t* it, float* r, bool* bad, bool* bad_id) {
    if (a.recs) {
        const int4 rec = a.recs[k];
        if (rec.x < 0) return false;                     // unused slot of the exchange buffer
        *u = rec.x; *it = rec.y; *r = __int_as_float(rec.z);
    } else if (a.ids) {
        int64_t id = a.ids[k];
        if ((uint64_t)id >= (uint64_t)a.N) { *bad_id = true; return false; }
        const int4 rec = a.store[id];
        *u = rec.x; *it = rec.y; *r = __int_as_float(rec.z);
    } else {
        *u = a.u[k]; *it = a.it[k]; *r = a.r[k];
    }
    if ((uint64_t)(int64_t)*u >= (uint64_t)a.U || (uint64_t)(int64_t)*it >= (uint64_t)a.I) { *bad = true; return false; }
    return true;
}
__device__ __forceinline__ bool route_mine(const RouteArgs& a, int64_t k, bool* bad, bool* bad_id, int32_t* u, int32_t* it, float* r) {
    if (k >= a.Bg) return false;
    if (!route_sample(a, k, u, it, r, bad, bad_id)) return false;
    if (a.recs && (int64_t)*u / a.per_u != a.rank) { *bad = true; return false; }      // a peer sent a sample this rank does not own
    return (int64_t)*u / a.per_u == a.rank;
}

// ---- pre-split batches: bucket this rank's own samples by the owner of their user row --------------------------------------
__device__ __forceinline__ int bucket_owner(const BucketArgs& a, int64_t k, int4* rec, bool* bad, bool* bad_id) {
    if (k >= a.B) return -1;
    int64_t id = a.ids[k];
    if ((uint64_t)id >= (uint64_t)a.N) { *bad_id = true; return -1; }
    *rec = a.store[id];
    if ((uint64_t)(int64_t)rec->x >= (uint64_t)a.U) { *bad = true; return -1; }
    return (int)((int64_t)rec->x / a.per_u);
}

// per 1024-sample block and owner: how many samples (integer LDS counters: order does not matter)
__global__ __launch_bounds__(1024) void k_bucket_count(BucketArgs a) {
    warm_args(a);
    extern __shared__ int32_t ocnt[];                    // [world]
    for (int w = threadIdx.x; w < a.world; w += 1024) ocnt[w] = 0;
    __syncthreads();
    bool bad = false, bad_id = false;
    int4 rec;
    const int w = bucket_owner(a, (int64_t)blockIdx.x * 1024 + threadIdx.x, &rec, &bad, &bad_id);
    if (w >= 0) atomicAdd(&ocnt[w], 1);
    __syncthreads();
    for (int q = threadIdx.x; q < a.world; q += 1024) a.blk[(size_t)blockIdx.x * a.world + q] = ocnt[q];
    if (bad) atomicOr(a.err, 1);
    if (bad_id) atomicOr(a.err, 2);
}

// per owner: exclusive scan of its counts over the blocks (one thread per owner; a few hundred blocks)
__global__ __launch_bounds__(256) void k_bucket_scan(int32_t* blk, int nblocks, int world, int32_t cap, int32_t* err) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= world) return;
    int run = 0;
    for (int b = 0; b < nblocks; ++b) { const int c = blk[(size_t)b * world + w]; blk[(size_t)b * world + w] = run; run += c; }
    if (run > cap) atomicOr(err, 4);
}

// the records, in batch order inside every owner's group
__global__ __launch_bounds__(1024) void k_bucket_scatter(BucketArgs a) {
    warm_args(a);
    bool bad = false, bad_id = false;
    int4 rec = make_int4(-1, -1, 0, -1);
    const int64_t k = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const int w = bucket_owner(a, k, &rec, &bad, &bad_id);
    // stable rank among the block's samples of the same owner: one ordered ballot count per owner present in the world
    int myrank = 0;
    for (int q = 0; q < a.world; ++q) {                  // block-uniform loop; two barriers per owner
        int tot;
        const int r = block_rank_1024(w == q, &tot);
        if (w == q) myrank = r;
    }
    if (w >= 0) {
        const int dst = a.blk[(size_t)blockIdx.x * a.world + w] + myrank;
        if (dst < a.cap) a.send[(size_t)w * a.cap + dst] = make_int4(rec.x, rec.y, rec.z, (int32_t)k);
    }
}

void launch_bucket(const BucketArgs& a, hipStream_t s) {
    const int nb = (int)((a.B + 1023) / 1024 > 0 ? (a.B + 1023) / 1024 : 1);
    hipLaunchKernelGGL(k_bucket_count, dim3(nb), dim3(1024), (size_t)a.world * 4, s, a);
    hipLaunchKernelGGL(k_bucket_scan, dim3((a.world + 255) / 256), dim3(256), 0, s, a.blk, nb, a.world, a.cap, a.err);
    hipLaunchKernelGGL(k_bucket_scatter, dim3(nb), dim3(1024), 0, s, a);
}

// phase 1a: per 1024-sample block, how many samples are this rank's (and the global range check)
__global__ __launch_bounds__(1024) void k_route_count(RouteArgs a) {
    warm_args(a);
    bool bad = false, bad_id = false;
    int32_t u, it; float r;
    const bool f = route_mine(a, (int64_t)blockIdx.x * 1024 + threadIdx.x, &bad, &bad_id, &u, &it, &r);
    int tot;
    (void)block_rank_1024(f, &tot);
    if (threadIdx.x == 0) a.blk[blockIdx.x] = tot;
    if (bad) atomicOr(a.err, 1);
    if (bad_id) atomicOr(a.err, 2);
}

// exclusive scan of blk[0..n) in place by ONE block (n <= a few thousand); total -> *out (clamped to limit, flagging overflow)
__global__ __launch_bounds__(1024) void k_route_scan(int32_t* blk, int n, int32_t* out, int32_t limit, int32_t* err, int32_t* zero, int nzero) {
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n + 1023) / 1024;
    int loc = 0;
    for (int q = 0; q < per; ++q) { const int b = tid * per + q; if (b < n) loc += blk[b]; }
    int incl = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int run = incl - loc;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    for (int q = 0; q < per; ++q) { const int b = tid * per + q; if (b < n) { const int c = blk[b]; blk[b] = run; run += c; } }
    if (tid == 1023) {
        int tot = run;
        if (tot > limit) { atomicOr(err, 4); tot = limit; }
        *out = tot;
    }
    if (tid < nzero) zero[tid] = 0;
}

// phase 1b: the rank's samples, compacted in batch order; unused sample slots get keys that sort last
__global__ __launch_bounds__(1024) void k_route_scatter(RouteArgs a) {
    warm_args(a);
    const int64_t k = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    bool bad = false, bad_id = false;
    int32_t su = 0, sit = 0; float sr = 0.f;
    const bool f = route_mine(a, k, &bad, &bad_id, &su, &sit, &sr);
    int tot;
    const int r = block_rank_1024(f, &tot);
    const int dst = a.blk[blockIdx.x] + r;
    if (f && dst < a.Bcap) {
        a.mine[dst] = (int32_t)k;
        a.u_local[dst] = (int32_t)(su - a.u_lo);
        a.it_glob[dst] = sit;
        a.r_loc[dst] = sr;
    }
    // the tail [n_local, Bcap): every block pads a stripe of it
    const int n_local = a.counts[0];
    for (int64_t q = (int64_t)n_local + (int64_t)blockIdx.x * 1024 + threadIdx.x; q < a.Bcap; q += (int64_t)gridDim.x * 1024) {
        a.mine[q] = -1; a.u_local[q] = a.u_pad; a.it_glob[q] = a.i_pad; a.r_loc[q] = 0.f;
    }
}

__device__ __forceinline__ bool route_head(const RouteArgs& a, int64_t j, int n_local) {
    return j < n_local && (j == 0 || a.ks[j] != a.ks[j - 1]);
}

// phase 2a: run heads (= distinct item ids) per block of the sorted order; distinct items per owner
__global__ __launch_bounds__(1024) void k_route_heads(RouteArgs a) {
    warm_args(a);
    const int64_t j = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const int n_local = a.counts[0];
    const bool h = route_head(a, j, n_local);
    int tot;
    (void)block_rank_1024(h, &tot);
    if (threadIdx.x == 0) a.blk[blockIdx.x] = tot;
    // distinct items per owner: counted per block in LDS first (a block of the sorted order spans one or two owners, and a
    // quarter of a million atomics on one global word take milliseconds), then one global add per owner and block.
    // Integer counters: the result does not depend on the order.
    __shared__ int ocnt[64];
    if (threadIdx.x < 64) ocnt[threadIdx.x] = 0;
    __syncthreads();
    if (h) {
        const int w = (int)(a.ks[j] / a.per_i);
        if (w < 64) atomicAdd(&ocnt[w], 1); else atomicAdd(&a.counts[2 + w], 1);
    }
    __syncthreads();
    if ((int)threadIdx.x < a.world && threadIdx.x < 64 && ocnt[threadIdx.x]) atomicAdd(&a.counts[2 + threadIdx.x], ocnt[threadIdx.x]);
}

// phase 2b: slot of every local sample, the request list
__global__ __launch_bounds__(1024) void k_route_slots(RouteArgs a) {
    warm_args(a);
    const int64_t j = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const int n_local = a.counts[0];
    const bool h = route_head(a, j, n_local);
    int tot;
    const int r = block_rank_1024(h, &tot);
    if (j >= a.Bcap) return;
    if (j >= n_local) { a.slot[a.ps[j]] = a.world * a.cap; return; }      // unused sample slot: a key that sorts last
    const int uidx = a.blk[blockIdx.x] + r + (h ? 0 : -1);               // index of this entry's run among the distinct ids
    const int32_t id = a.ks[j];
    const int w = (int)(id / a.per_i);
    int start = 0;
    for (int q = 0; q < w; ++q) start += a.counts[2 + q];
    const int kk = uidx - start;
    if (kk >= a.cap) { atomicOr(a.err, 4); a.slot[a.ps[j]] = a.world * a.cap; return; }
    const int sl = w * a.cap + kk;
    a.slot[a.ps[j]] = sl;
    if (h) a.req[sl] = (int32_t)(id - (int64_t)w * a.per_i);
}

void launch_route_compact(const RouteArgs& a, hipStream_t s) {
    const int nb = (int)((a.Bg + 1023) / 1024);
    hipLaunchKernelGGL(k_route_count, dim3(nb), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(k_route_scan, dim3(1), dim3(1024), 0, s, a.blk, nb, a.counts, a.Bcap, a.err, a.counts + 1, a.world + 1);
    hipLaunchKernelGGL(k_route_scatter, dim3(nb), dim3(1024), 0, s, a);
}

void launch_route_slots(const RouteArgs& a, hipStream_t s) {
    const int nb = (a.Bcap + 1023) / 1024;
    hipLaunchKernelGGL(k_route_heads, dim3(nb), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(k_route_scan, dim3(1), dim3(1024), 0, s, a.blk, nb, a.counts + 1, (int32_t)0x7fffffff, a.err, (int32_t*)nullptr, 0);
    hipLaunchKernelGGL(k_route_slots, dim3(nb), dim3(1024), 0, s, a);
}

// owner side: rows_out[j] = [table[ids[j]] | bias[ids[j]] | flag | pad] (row stride `stride` floats); unused slots (-1) give
// zero features.  `flag` = this rank's device error word as the launch found it (bucket / route errors of this step or an
// earlier, not yet reported one): it travels to every requester with the rows, so that a step one rank must void is void
// on all of them (k_adopt_peer_err on the receiving side) - no extra collective.
template <int G, int VEC>
__global__ __launch_bounds__(256) void k_gather_packed(GatherPackedArgs a) {
    warm_args(a);
    constexpr int GPB = 256 / G;
    constexpr int UN = 4;                                // requests in flight per lane group: ids in one round, rows in the next
    const int gl = threadIdx.x % G, d0 = gl * VEC;
    bool oob = false;
    const float flag = (float)(*a.err != 0);
    const bool full = a.D == G * VEC;                    // full-width rows: unguarded 16-byte loads (block-uniform)
    for (int64_t j0 = (int64_t)blockIdx.x * GPB * UN + threadIdx.x / G; j0 < a.n; j0 += (int64_t)gridDim.x * GPB * UN) {
        int32_t id[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t j = j0 + (int64_t)u * GPB;
            id[u] = a.ids[j < a.n ? j : j0];
        }
        float v[UN][VEC], b[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const bool bad = id[u] >= 0 && (int64_t)id[u] >= a.rows;
            oob |= bad && (j0 + (int64_t)u * GPB < a.n);
            const int32_t idc = (id[u] >= 0 && !bad) ? id[u] : 0;     // an address that is there; the value is dropped below
            const float* src = a.table + (size_t)idc * a.D;
            if constexpr (VEC == 4) {
                if (full) {
                    typedef float f4 __attribute__((ext_vector_type(4)));
                    const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(src + d0));
                    v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
                } else {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) v[u][q] = (d0 + q < a.D) ? src[d0 + q] : 0.f;
                }
            } else {
                v[u][0] = (d0 < a.D) ? src[d0] : 0.f;
            }
            b[u] = a.bias[idc];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t j = j0 + (int64_t)u * GPB;
            if (j >= a.n) break;
            const bool keep = id[u] >= 0 && (int64_t)id[u] < a.rows;
            float* dst = a.out + (size_t)j * a.stride;
            if (d0 < a.D) {
                if constexpr (VEC == 4) *reinterpret_cast<float4*>(dst + d0) = keep ? make_float4(v[u][0], v[u][1], v[u][2], v[u][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
                else dst[d0] = keep ? v[u][0] : 0.f;
            }
            if (gl == 0) {
                const float bb = keep ? b[u] : 0.f;
                if constexpr (VEC == 4) *reinterpret_cast<float4*>(dst + a.D) = make_float4(bb, flag, 0.f, 0.f);
                else { dst[a.D] = bb; dst[a.D + 1] = flag; }
            }
        }
    }
    if (oob) atomicOr(a.err, 1);
}

// requester side: chunk w of the received rows came from owner w; any owner's flag set -> this rank's step is void too
__global__ __launch_bounds__(64) void k_adopt_peer_err(const float* rows, int64_t chunk_floats, int32_t world, int32_t D, int32_t* err) {
    bool any = false;
    for (int w = threadIdx.x; w < world; w += 64) any |= rows[(size_t)w * chunk_floats + D + 1] != 0.f;
    if (__ballot(any) && threadIdx.x == 0) atomicOr(err, 8);
}

void launch_adopt_peer_err(const float* rows, int64_t chunk_floats, int32_t world, int32_t D, int32_t* err, hipStream_t s) {
    hipLaunchKernelGGL(k_adopt_peer_err, dim3(1), dim3(64), 0, s, rows, chunk_floats, world, D, err);
}

void launch_gather_packed(const GatherPackedArgs& a, int G, int VEC, hipStream_t s) {
    const int gpb = 4 * (256 / G);                       // UN requests per lane group and round
    int64_t nb = (a.n + gpb - 1) / gpb;
    if (nb > 8192) nb = 8192;
    if (nb < 1) nb = 1;
#define TFR_GP_CASE(g, v) \
    if (G == g && VEC == v) { hipLaunchKernelGGL((k_gather_packed<g, v>), dim3((int)nb), dim3(256), 0, s, a); return; }
    TFR_GP_CASE(4, 4) TFR_GP_CASE(8, 4) TFR_GP_CASE(16, 4) TFR_GP_CASE(32, 4) TFR_GP_CASE(64, 4)
    TFR_GP_CASE(4, 1) TFR_GP_CASE(8, 1) TFR_GP_CASE(16, 1) TFR_GP_CASE(32, 1) TFR_GP_CASE(64, 1)
#undef TFR_GP_CASE
}

__global__ __launch_bounds__(256) void k_pad_keys(const int32_t* in, int32_t* out, int64_t n, int32_t pad_key, int32_t* count) {
    int c = 0;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const int32_t id = in[k];
        out[k] = id < 0 ? pad_key : id;
        c += id >= 0;
    }
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_down(c, o, 64);
    __shared__ int wc[4];
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {                              // one integer add per block (5120 adds on one word, one per wave, took 40 us)
        const int t = (wc[0] + wc[1]) + (wc[2] + wc[3]);
        if (t) atomicAdd(count, t);
    }
}

void launch_pad_keys(const int32_t* ids_in, int32_t* keys_out, int64_t n, int32_t pad_key, int32_t* count, hipStream_t s) {
    (void)hipMemsetAsync(count, 0, 4, s);
    int64_t nb = (n + 255) / 256;
    if (nb > 512) nb = 512;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_pad_keys, dim3((int)nb), dim3(256), 0, s, ids_in, keys_out, n, pad_key, count);
}

}  // namespace tfr
