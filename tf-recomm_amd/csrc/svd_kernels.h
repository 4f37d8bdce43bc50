// svd_kernels.h - argument blocks and launchers shared by svd_kernels.hip, sort.hip and api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tfr {

enum { MODE_INFER = 0, MODE_TRAIN = 1, MODE_EVAL = 2 };
enum { RMODE_SCRATCH = 0, RMODE_ADAM = 1, RMODE_SGD = 2 };

// Runs of equal row id in the sorted order are cut into pieces: piece starts = run heads and
// multiples of PIECE = 1024/G (the span of one k_seg_reduce block), so the work per block is
// fixed however skewed the ids are.

struct FwdArgs {
    const float* P; const float* Q; const float* bu; const float* bi; const float* mu;
    const int32_t* u; const int32_t* it; const float* r;
    // optional fused gather from the resident store (ids != NULL): u/it/r above are then
    // ignored and the gathered ids are written to u_out / it_out for the backward
    const int64_t* ids; const int4* store;
    int32_t* u_out; int32_t* it_out;
    float* logits; float* g; float* partials; int32_t* err;
    int64_t B, U, I, N;
    int32_t D, loss, item_abs, reg_bias, lds_reduce;
    // row-sharded step: the batch size lives on the device (dB, <= B), item rows come from the packed exchange
    // buffer (row stride qstride floats, bias inside the row: bi = Q + D, stride bistride).  0 = D / 1.
    const int32_t* dB; int32_t qstride, bistride;
};

struct GatherArgs {
    const int64_t* ids;          // NULL -> contiguous range starting at lo
    int64_t lo, B, N;
    const int4* store;
    int32_t* u; int32_t* it; float* r; int32_t* err;
};

// Touch every 64-byte line of a kernel-argument struct with scalar loads that are all in flight together.  The compiler places
// each argument's s_load where the argument is first used, so a latency-bound kernel pays the lines' first-touch misses one
// after the other along its dependent chain; after this they hit the scalar cache.
// `args` must lie inside the kernel-argument block (an array element: check the index first - reading past the block's end is
// a page fault when the block ends a page).
template <typename T>
__device__ __forceinline__ void warm_args(const T& args) {
    const char* base = reinterpret_cast<const char*>(&args);
    constexpr int NL = (int)((sizeof(T) + 63) / 64);
    int32_t w[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) w[k] = *reinterpret_cast<const int32_t*>(base + (k * 64 < (int)sizeof(T) - 4 ? k * 64 : (int)sizeof(T) - 4));
#pragma unroll
    for (int k = 0; k < NL; ++k) asm volatile("" :: "s"(w[k]));
}


struct RedArgs {
    const int32_t* ks; const int32_t* ps; const int32_t* other; const float* g;
    const float* own; const float* partner; const float* own_bias;
    float* own_w; float* m; float* v; float* bias_w; float* bias_m; float* bias_v;
    float* grad_rows; float* grad_bias; int32_t* map;
    float* dense_rows; float* dense_bias;          // optional dense [rows,D] / [rows] gradient buffers
    int32_t dstride, dbstride;                     // their row / bias strides in floats (0 = D / 1): the packed exchange layout
    const float* rows_in; const float* bias_in;    // sharded owner side: pre-reduced gradient rows
    const int4* ent;                               // FM: {partner row, g x, lam - g x^2, -} per entry in one 16-byte record (one
                                                   // request instead of the partner-id, g and coefficient gathers)
    float* own_copy_out;                           // optional [B,D]: the entry's pre-update own row, by batch position
    const float* partner_by_pos;                   // optional [B,D]: read the partner row from such a copy instead
    // two-table form of the fused big-table step (no pre-update copy): an item row lives in `own` or in `own_alt`, sel[row] says
    // which.  Item side: reads the current one, writes the updated row into the OTHER table and flips sel[row] (whole runs; runs
    // cut by a block boundary stay where they are, k_apply_rows finishes them in place), and leaves osel_out[pos] = row | cur << 31.
    // User side: osel_in[pos] gives the partner row id and the table that still holds its pre-update value.
    const float* own_alt; float* own_w_alt; int32_t* sel; int32_t* osel_out;
    const int32_t* osel_in; const float* partner_alt;
    // forward fused into this side (FWD kernels): logits, g and the per-block {loss, reg, sum g}
    // are produced here from the rows the reduce loads anyway
    const float* partner_bias; const float* mu; const float* r;
    float* g_out; float* logits_out; float* partials; int32_t loss;
    int32_t stage_sum;                             // FWD: per-block sums staged through LDS (A/B switch)
    const int32_t* err;
    int64_t B;
    int32_t D, side, item_abs, reg_bias, frozen_rows, frozen_bias;
    int32_t tile;                                  // >0: the sorted order restarts every `tile` entries
    // row-sharded step (0 = the plain layout): device-resident entry count, strides of packed exchange buffers
    const int32_t* dB;                             // entries actually present (<= B)
    int32_t ostride, obstride;                     // own rows / own bias read-only source (item rows fetched from their owners)
    int32_t pstride;                               // partner rows
    int32_t rstride, rbstride;                     // rows_in / bias_in
    int32_t nt;                                    // cache-policy hints (bits: 1 partner rows, 2 own/m/v loads, 4 w/m/v stores,
                                                   // 8 loads of the pre-update copies, 16 stores of them): non-temporal
    float lam, alpha, b1, b2, eps, lr;
};
struct RedPair { RedArgs a[2]; };

struct GatherRowsArgs {
    const int32_t* ids; const float* table; const float* bias;
    float* rows_out; float* bias_out; int32_t* err;
    int64_t n, rows;
    int32_t D;
};

struct ApplyArgs {
    const int32_t* ks; const float* grad_rows; const float* grad_bias;
    float* w; float* m; float* v; float* bias_w; float* bias_m; float* bias_v;
    const int32_t* err;
    int64_t B;
    int32_t D, frozen_rows, frozen_bias, only_split;   // only_split: runs cut into >1 piece only
    float alpha, b1, b2, eps, lr;
    const int32_t* dB;                             // row-sharded step: entries actually present (<= B)
    int32_t wstride, wbstride;                     // OPT 2 (emit reduced rows): output row / bias stride (0 = D / 1)
    float* w_alt; const int32_t* sel;              // two-table form: the row is updated in place in the table sel[row] names
};
struct FinArgs {
    const float* partials; int32_t nblk;
    float* scalars; float* out;
    float* mu; float* mu_m; float* mu_v; const int32_t* err;
    int32_t update_mu, opt, clear_partials;        // clear_partials: zero partials[0..3] after use
    int32_t out_err;                               // out[3] = the step's error flag (as a float)
    float alpha, b1, b2, eps, lr;
};

// with_fin: one more block column carries K4 (the step's scalars + bias_global) in the same launch
struct ApplyPair { ApplyArgs a[2]; FinArgs f = {}; int32_t with_fin = 0; };

struct DenseArgs {
    int32_t* map; const int32_t* ks; const float* grad_rows; const float* grad_bias;
    float* dense_grad; float* dense_gbias;       // data-parallel: dense gradients [rows,D] / [rows]
    float* w; float* m; float* v; float* bias_w; float* bias_m; float* bias_v;
    const int32_t* err;
    int64_t rows, B;
    int32_t D, frozen_rows, frozen_bias, opt;    // opt: 0 Adam, 1 SGD
    float alpha, b1, b2, eps, lr;
};

struct DensePair { DenseArgs a[2]; FinArgs f; };


// small-table sweep over per-tile partial gradients (k_dense_tiles)
struct TileDenseArgs {
    const int32_t* tab;                            // [ntiles * nbins]: (count << 16) | offset, from the tile-local sort
    const float* grad_rows; const float* grad_bias;   // piece sums by tile-sorted position
    float* w; float* m; float* v; float* bias_w; float* bias_m; float* bias_v;
    float* out_rows; float* out_bias;              // WRITE mode: dense gradient buffer (data parallel)
    const int32_t* err;
    int64_t rows;
    int32_t D, nbins, ntiles, frozen_rows, frozen_bias, opt, skip_untouched;
    float alpha, b1, b2, eps, lr;
};
struct TileDenseLaunch { TileDenseArgs a[2]; FinArgs f; int32_t with_fin; };
void launch_dense_tiles(const TileDenseLaunch& L, bool write, bool with_fin, int G, int VEC, hipStream_t s);

// one-pass stable counting sort of both id columns (small tables: all bins fit in LDS)
struct CSortArgs {
    const int32_t* keys[2];      // [B] each
    int32_t* ks[2]; int32_t* ps[2];
    int32_t* lrank[2];           // [B] rank of the entry among equal keys inside its tile
    int32_t* hist[2];            // [ntiles * nbins], tile-major: per-tile key histogram
    int32_t* offs[2];            // [ntiles * nbins]: entries with this key in earlier tiles
    int32_t* binbase[2];         // [nbins]: entries with a smaller key inside the bin's 1024-bin block
    int32_t* blocktot[2];        // [nbins/1024]: entries per 1024-bin block
    int32_t nbins[2];            // power of two
    int32_t ntiles;
    int64_t B;
};

// second-order FM forward on CSR rows (fm_kernels.hip); training extras are NULL for inference
struct FmArgs {
    const float* V; const float* W; const float* mu;
    const int64_t* indptr; const int32_t* indices; const float* data;
    float* out; int32_t* err;
    const float* y; float* s_rows; int4* ent; float* partials;   // ent[nnz]: {row, g x, lam - g x^2, -} per non-zero (training)
    int64_t n_rows, F;
    int32_t D, loss;
    float lam;
    int32_t variant;                               // bit 1: non-temporal V rows (inference; launch_fm)
};
int fm_grid(int64_t n_rows, int G, bool train);
void launch_fm(const FmArgs& a, bool train, int G, int VEC, int grid, hipStream_t s);

// forward + csort rank pass in one launch (small tables)
struct FrontArgs {
    FwdArgs f; CSortArgs c;
    int32_t* key_out[2];         // fused-gather mode: where the rank blocks publish the gathered ids
    int32_t nfwd;                // number of forward blocks (1024 threads each)
    int32_t tile_local;          // 1: sort each tile locally (hist / offs become per-tile lookup tables)
};

// small tables, whole step before the optimiser in ONE launch (k_tile_step): gather + tile-local
// sort + forward + per-tile segmented reduce, for both id columns
struct TileStepArgs {
    const float* P; const float* Q; const float* bu; const float* bi; const float* mu;
    const int32_t* u; const int32_t* it; const float* r;    // the batch, or (ids != NULL) rows of the store
    const int64_t* ids; const int4* store;
    const int4* recs; const int4* next_recs;                 // optional: the (next) batch's store records, contiguous (instead of ids -> store)
    const int4* srt[2];                                      // non-NULL: this batch's tile-sorted records {u, i, r, pos},
                                                             // published by the previous step's launch (then u/it/r/ids unused)
    float* logits;                                           // optional [B]
    float* partials;                                         // [ntiles * G / EPG][4]: {loss, reg, sum g, -}
    int32_t* err;
    int32_t* tab[2];                                         // [ntiles * nbins[c]] packed (count << 16) | offset
    float* grad_rows[2]; float* grad_bias[2];                // piece sums by tile-sorted position; 0 = user side
    int64_t B, U, I, N;
    int32_t D, loss, item_abs, reg_bias, ntiles, nbins[2];
    float lam;
    // look-ahead: sort the NEXT batch (rows next_ids[0..next_B) of the store) in spare blocks of this launch
    const int64_t* next_ids; int64_t next_B; int32_t next_ntiles;
    int32_t* next_tab[2]; int4* next_srt[2];
    unsigned long long* dbg;                                 // TFR_TILE_DEBUG: {start, end} of every block, 100 MHz ticks
};
void launch_tile_step(const TileStepArgs& a, int G, int VEC, hipStream_t s);
void launch_gather_recs(const int64_t* ids, const void* store, void* recs, int64_t n, int64_t N, hipStream_t s);
// LDS of k_tile_step<G, VEC, EPG>: the static arrays (the kernel static_asserts this sum against its own
// declarations) and the dynamic request (sort bins first, then the wave-level ping-pong buffers).  A CU has 160 KB;
// tfr_lds_bytes / tests/test_lds_budget.py enumerate every shape the dispatcher can select.
constexpr size_t LDS_PER_CU = 160 * 1024;
constexpr size_t tile_step_static_lds(int G, int EPG) {
    return (size_t)5 * 1024 * 4                        /* rec_u, rec_i, rec_r, srt_key, srt_pos */
           + (size_t)2 * EPG * 16 * 4 + (size_t)EPG * 16 * 4                 /* lds_gb, lds_key */
           + (size_t)EPG * (2 * (1024 / G) + 1024) * 4                       /* lds_stage */
           + 16 * 4;                                                         /* wtot */
}
constexpr size_t tile_step_dyn_lds(int G, int VEC, int EPG, int nbmax) {
    return (size_t)nbmax * 4 > (size_t)2 * EPG * 16 * G * VEC * 4 ? (size_t)nbmax * 4 : (size_t)2 * EPG * 16 * G * VEC * 4;
}
constexpr size_t seg_reduce_static_lds(int G, int VEC, bool fwd) {
    // (an upper bound over the instantiations: the three-round item side with the forward inside also parks the own rows in LDS)
    return (size_t)1024 * VEC * 4 + (size_t)2 * (1024 / G) * 4 + (fwd ? (size_t)(2 * (1024 / G) + 1024) * 4 + 16 * 3 * 4 + (size_t)1024 * VEC * 4 : 4);
}
int tile_step_epg(int ntiles, int G, int VEC);       // pieces per block k_tile_step will use (grid = ntiles * G / epg per side)

// row geometry for a dim: returns false if unsupported
inline bool geometry(int D, int* G, int* VEC) {
    if (D < 1) return false;
    int vec = (D % 4 == 0) ? 4 : 1;
    int lanes = (D + vec - 1) / vec;
    int g = 4;
    while (g < lanes) g <<= 1;
    if (g > 64) return false;
    *G = g; *VEC = vec;
    return true;
}

int forward_grid(int64_t B, int G, int mode);
void launch_forward(const FwdArgs& a, int mode, int G, int VEC, int grid, hipStream_t s);
int front_forward_blocks(int64_t B, int G);
void launch_front(const FrontArgs& fa, int G, int VEC, hipStream_t s);
void launch_csort_tail(const CSortArgs& a, const FinArgs* fin, hipStream_t s);   // scan (+K4) and scatter only
void launch_seg_reduce(const RedPair& p, int n, int rmode, int G, int VEC, hipStream_t s, bool fwd = false);
void launch_apply_rows(const ApplyPair& p, int n, int opt, int G, int VEC, hipStream_t s);
void launch_adam_dense(DensePair& p, int n, int G, int VEC, hipStream_t s, const FinArgs* fin = nullptr);
void launch_gather(const GatherArgs& a, hipStream_t s);
void launch_gather_rows(const GatherRowsArgs& a, int G, int VEC, hipStream_t s);
void launch_pack_triples(const int32_t* u, const int32_t* it, const float* r, void* store, int64_t n, hipStream_t s);
void launch_finalize(const FinArgs& a, hipStream_t s);
void launch_auc_keys(const float* score, int32_t* keys, int64_t n, hipStream_t s);
void launch_auc_ranksum(const int32_t* ks, const int32_t* ps, const float* label, int64_t n, unsigned long long* out, hipStream_t s);
void launch_init_trunc_normal(float* p, int64_t n, float stddev, uint64_t seed, hipStream_t s);
void launch_init_uniform_scalar(float* p, float lo, float hi, uint64_t seed, hipStream_t s);

// shard.hip: device-side routing of the row-sharded step - which samples of the global batch this rank owns,
// the distinct item rows they need, grouped by owner into a fixed-capacity request layout [world][cap]
struct RouteArgs {
    const int32_t* u; const int32_t* it; const float* r;     // the GLOBAL batch [Bg], identical on every rank
    const int64_t* ids; const int4* store; int64_t N;        // or (ids != NULL): rows ids[0..Bg) of the rank's copy of the rating store
    const int4* recs;                                        // or (recs != NULL): records {u, i, r bits, origin} received from the peers
                                                             // (pre-split batches, launch_bucket); u < 0 = unused slot
    int64_t Bg, U, I;                                        // global row counts (range check)
    int64_t per_u, per_i, u_lo;                              // block partition: owner = id / per
    int32_t rank, world, Bcap, cap;                          // capacities: local samples, request slots per owner
    int32_t u_pad, i_pad;                                    // keys of unused sample slots (one past the last row: they sort last)
    int32_t* mine; int32_t* u_local; int32_t* it_glob; float* r_loc; int32_t* slot;   // [Bcap] outputs
    int32_t* req;                                            // [world * cap] local item ids at their owner, -1 = unused slot
    int32_t* counts;                                         // [0] local samples, [1] distinct items, [2 + w] distinct items owned by w
    int32_t* blk;                                            // scratch [>= ceil(max(Bg, Bcap) / 1024) + 1]
    const int32_t* ks; const int32_t* ps;                    // local samples sorted by global item id (radix sort between the phases)
    int32_t* err;                                            // |= 1 id out of range, |= 4 capacity exceeded
};
void launch_route_compact(const RouteArgs& a, hipStream_t s);    // mine / u_local / it_glob / r_loc, counts[0]
// pre-split batches (SURVEY 8e's other variant): this rank's own B samples, as rows ids[] of its store copy, go into a
// fixed-capacity send buffer [world][cap] of 16-byte records grouped by the owner of their user row (stable: batch order
// inside a group); unused slots keep u = -1 (the caller presets the buffer to 0xff)
struct BucketArgs {
    const int64_t* ids; const int4* store; int64_t N, B, U, per_u;
    int32_t world, cap;
    int4* send; int32_t* blk;                                // blk: [nblocks * world] counts, then offsets (scratch)
    int32_t* err;                                            // |= 2 id out of range, |= 1 user id out of range, |= 4 capacity exceeded
};
void launch_bucket(const BucketArgs& a, hipStream_t s);
void launch_route_slots(const RouteArgs& a, hipStream_t s);      // after the sort: slot, req, counts[1..]
struct GatherPackedArgs {
    const int32_t* ids; const float* table; const float* bias; float* out; int32_t* err;
    int64_t n, rows; int32_t D, stride;
};
void launch_gather_packed(const GatherPackedArgs& a, int G, int VEC, hipStream_t s);
// the error flags the owners packed beside their rows (float D + 1 of a row; one chunk of `chunk_floats` per owner) -> err |= 8
void launch_adopt_peer_err(const float* rows, int64_t chunk_floats, int32_t world, int32_t D, int32_t* err, hipStream_t s);
// pads (-1) of a received request list -> `pad_key` (one past the last row, sorts last); counts the real ones
void launch_pad_keys(const int32_t* ids_in, int32_t* keys_out, int64_t n, int32_t pad_key, int32_t* count, hipStream_t s);

// rng.hip: NumPy's legacy randint(0, rng + 1, (need,)) from the MT19937 state {key[624], pos} at d_state
// ws (optional): scratch of the wide form - raw[cap_blocks * 624] words, counts[cap_blocks], hdr[4]; draws of MT_WIDE_MIN ids
// and more then run as k_mt_blocks + k_mt_count + k_mt_emit (+ a k_mt_draw launch that normally has nothing left to do)
struct MtScratch { uint32_t* raw; int32_t* counts; int32_t* hdr; int64_t cap_blocks; };
constexpr int64_t MT_WIDE_MIN = 32768;                   // ids per draw from which the wide form pays for its three extra launches
constexpr int64_t MT_WIDE_MAX = (int64_t)1 << 22;        // ids per wide pass (scratch: 34 MB at the worst acceptance rate)
constexpr int64_t MT_WIDE_BLOCKS = 13700;                // 2^22 ids at acceptance 1/2 + 6 sigma, in 624-word blocks
void launch_mt_draw(uint32_t* d_state, int64_t* d_out, int64_t need, uint32_t rng, uint32_t mask, hipStream_t s,
                    unsigned long long* d_dbg = nullptr, const MtScratch* ws = nullptr);     // d_dbg: optional {shader cycles, 100 MHz ticks} of the launch

// sort.hip
constexpr int CSORT_TILE = 1024;
constexpr int CSORT_MAX_BINS = 16384;          // 64 KB of LDS counters
void launch_settle_alt(float* main_t, const float* alt_t, int32_t* sel, int64_t rows, int D, hipStream_t s);
bool csort_eligible(int64_t B, int bits_u, int bits_i);
void launch_csort(const CSortArgs& a, const FinArgs* fin, hipStream_t s);   // fin: run K4 in the scan launch
// LSD radix sort pass (8-bit digit at `shift`) over up to two key columns
struct RSortArgs {
    const int32_t* keys_in[2]; const int32_t* vals_in[2];    // vals_in NULL -> the batch position
    int32_t* keys_out[2]; int32_t* vals_out[2];
    int32_t* lrank[2]; int32_t* hist[2]; int32_t* offs[2];   // hist/offs: [256 * ntiles], bin-major
    int32_t* blocktot[2];                                    // per scan block (chunk entries) totals
    int32_t shift, ntiles, chunk;                            // chunk: multiple of 1024, <= 16384
    int64_t B;
    int32_t limit[2]; int32_t* err;                          // err != NULL: pass 0 flags keys outside [0, limit)
    // pass 0 may gather the batch itself (dataio.py:115-117 on the resident store): key = store[ids[k]].x / .y; the column-0
    // blocks also write the gathered (user, item, rate) to keys_in[0], keys_in[1] (non-const alias) and r_out for the later passes
    const int64_t* ids; const int4* store; int64_t N; float* r_out; int32_t* u_out; int32_t* i_out;
};
void launch_rsort_pass(const RSortArgs& a, int ncols, hipStream_t s);
// the same pass with 4096-key tiles whose workgroups rank, stage in LDS and write whole digit runs (>= 2^20 keys, no fused gather)
bool rsortw_eligible(int64_t B);
void launch_rsortw_pass(RSortArgs a, int ncols, hipStream_t s);


}  // namespace tfr
