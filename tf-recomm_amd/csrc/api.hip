// api.hip - the C-ABI of include/tfrecomm.h over the kernels in svd_kernels.hip.
// Host-side orchestration of one minibatch (svd_train_val.py:66-72):
//   K1 forward+loss+g  ->  key sorts (user ids, item ids)  ->  item-side segmented reduce
//   -> user-side segmented reduce (+ fused lazy Adam / SGD)  ->  item apply  ->  finalize
// or, in TF1 Adam mode, both reduces to scratch followed by the dense sweeps.
#include <hip/hip_runtime.h>
#include <utility>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <new>
#include <chrono>
#include <vector>
#include "tfrecomm.h"
#include "svd_kernels.h"

using namespace tfr;

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(e_ == hipErrorOutOfMemory ? TFR_ERR_NOMEM : TFR_ERR_HIP, "%s: %s",   \
                        #expr, hipGetErrorString(e_));                                       \
    } while (0)

struct ProfEvent { hipEvent_t a, b; int kid; };

struct tfr_model {
    int64_t U = 0, I = 0;
    int32_t D = 0, G = 0, VEC = 0;
    int bits_u = 1, bits_i = 1;
    tfr_opts o;
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // tables: index TFR_MU..TFR_Q; slots m, v
    float* w[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    float* m[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    float* v[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int64_t n[5] = {0, 0, 0, 0, 0};
    uint32_t frozen = 0;
    int64_t step = 0;
    float b1p = 0.f, b2p = 0.f;
    // batch workspace
    int64_t cap = 0;
    int32_t *d_u = nullptr, *d_i = nullptr;
    float *d_r = nullptr, *d_logits = nullptr, *d_g = nullptr;
    int32_t *ks_u = nullptr, *ps_u = nullptr, *ks_i = nullptr, *ps_i = nullptr;
    int32_t *ks2_u = nullptr, *ps2_u = nullptr, *ks2_i = nullptr, *ps2_i = nullptr;   // rsort ping-pong
    int32_t *lrank_u = nullptr, *lrank_i = nullptr, *hist_u = nullptr, *hist_i = nullptr;   // csort
    int32_t *offs_u = nullptr, *offs_i = nullptr, *binbase_u = nullptr, *binbase_i = nullptr;
    int32_t *blocktot_u = nullptr, *blocktot_i = nullptr;
    // two-table form of the fused big-table step (RedArgs::sel): the alternate item table, the per-row "which table" word,
    // the per-entry {partner row | old table} words of the current batch; q_dirty = some row may live in q_alt
    float* q_alt = nullptr; int32_t* q_sel = nullptr; int32_t* osel = nullptr; bool q_dirty = false;
    bool csort_ok = false;
    float *gq = nullptr, *gp = nullptr, *gbq = nullptr, *gbp = nullptr;
    int32_t *map_u = nullptr, *map_i = nullptr;
    float *dg_p = nullptr, *dg_q = nullptr, *dg_bu = nullptr, *dg_bi = nullptr;   // tf1: dense per-row gradients
    float* partials = nullptr;
    float* scalars = nullptr;         // {loss, reg, sum_g, -}
    float* step_out = nullptr;        // per-step {loss, reg, sum_g} ring for multi-step calls
    int64_t step_out_cap = 0;
    int32_t* d_err = nullptr;
    unsigned long long* d_auc = nullptr;   // {2 x rank sum of the positives, number of positives}
    const float* last_r = nullptr; int64_t last_B = 0;     // rates of the last host-fed training batch whose logits were kept
    // host-fed calls (tfr_train_step / tfr_forward): one pinned staging buffer each way, so a step is one
    // H2D copy, the kernels and one D2H copy instead of five pageable transfers
    int32_t* d_in = nullptr; int32_t* h_in = nullptr; float* h_out = nullptr; int64_t stage_cap = 0;
    int32_t* h_err = nullptr;                                // pinned landing place of the device error flag
    // look-ahead of the small-table step: the next batch's tile sort, published by the previous launch
    int4* srt[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [parity][side] sorted records {u, i, r, pos}
    const int64_t* pf_ids = nullptr; int64_t pf_B = 0; int pf_par = 0; bool pf_valid = false;
    const int64_t* dp_next_ids = nullptr;                        // tfr_dp_hint_next: batch of the next tfr_dp_local_grads
    // big-table look-ahead: the next batch is gathered and sorted on a second stream into the alternate
    // set of batch buffers while this step's bandwidth-bound kernels run
    struct SortSet { int32_t *d_u = nullptr, *d_i = nullptr; float* d_r = nullptr;
                     int32_t *ks_u = nullptr, *ps_u = nullptr, *ks_i = nullptr, *ps_i = nullptr; } alt;
    int64_t alt_cap = 0;
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_sorted[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr}, ev_first = nullptr;
    // tfr_draw_ids_dev / tfr_join_draws / tfr_join_draw: one event per issued draw (ring; draws complete in issue order)
    static const int DRAW_RING = 8;
    hipEvent_t draw_evs[DRAW_RING] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int64_t draw_count = 0, draw_joined = 0;
    MtScratch rng_ws = {nullptr, nullptr, nullptr, 0};     // wide form of the id draw (rng.hip), allocated with the generator state
    hipEvent_t ev_mid = nullptr; bool ev_mid_on = false;   // recorded between the item-side and the user-side kernel of a big-table step
    // resident store
    int4* store = nullptr;            // {user, item, rate bits, -} per rating
    int64_t N = 0;
    int64_t* d_ids = nullptr;
    int64_t n_ids = 0;
    int64_t d_ids_cap = 0;
    // device id draw (rng.hip): NumPy's MT19937 state {key[624], pos}; draws run on their own stream, ahead
    // of the steps that consume them; one event per drawn chunk
    uint32_t* d_rng = nullptr;
    bool rng_set = false;
    // run-ahead between calls: after a drawn call the generator goes on into the alternate id buffer, so the next
    // call with the same batch size starts on ids that are already there; anything else that looks at the generator
    // first puts it back to the snapshot taken where the consumed ids end
    int64_t* d_ids_alt = nullptr; int64_t d_ids_alt_cap = 0;
    // store records of drawn / staged ids, left beside an id buffer by the stream that filled it (one round trip instead of
    // ids -> store for the small-table step's sorts).  Keyed by the id buffer's address, so the buffers' swaps need no care;
    // `n` = how many leading ids of that buffer have (or will have, in stream order before their chunk's event) their records.
    struct RecBuf { const int64_t* ids_base = nullptr; int4* recs = nullptr; int64_t cap = 0; int64_t n = 0; } rb[2];
    uint32_t* d_rng_snap = nullptr;
    bool spec_valid = false; int64_t spec_B = 0, spec_N = 0, spec_steps = 0;
    hipEvent_t spec_ev = nullptr;
    hipStream_t stream3 = nullptr;
    std::vector<hipEvent_t> chunk_ev;
    hipEvent_t ev_ids_free = nullptr;
    // host-drawn ids, one step at a time, without a host sync: pinned ring + device ring
    static const int HRING = 4;
    int64_t* h_ring = nullptr; int64_t* d_ring = nullptr; int64_t ring_cap = 0; int ring_pos = 0;
    hipEvent_t ring_ev[HRING] = {nullptr, nullptr, nullptr, nullptr};
    // row-sharded step: the routed local batch (tfr_shard_route)
    // Two sets, so that a caller can route (and pre-sort) batch s+1 on a side stream while step s still reads its own:
    // tfr_shard_select picks the set the shard calls fill and consume.
    struct RouteSet {
        int32_t *mine = nullptr, *u = nullptr, *it = nullptr, *slot = nullptr, *counts = nullptr;
        float* r = nullptr;
        int64_t cap = 0, B = 0, slots = 0;
        int32_t cap_world = 0, world = 0;
        // sorted orders: of the routed samples by local user row / by request slot (forward + reduce), made by tfr_shard_presort
        // ahead of time or by the step itself; of the requests received as an owner (apply_items)
        int32_t *ks_u = nullptr, *ps_u = nullptr, *ks_i = nullptr, *ps_i = nullptr;
        int32_t *akeys = nullptr, *aks = nullptr, *aps = nullptr; int64_t acap = 0;
        bool sorted_fwd = false; const int32_t* sorted_req = nullptr; int64_t sorted_req_n = 0;
        const int32_t *fks_u = nullptr, *fps_u = nullptr, *fks_i = nullptr, *fps_i = nullptr;   // where the forward's sorted columns are
    } rt[2];
    int rt_sel = 0;
    // resident validation set (svd_train_val.py:33-38: the whole set is one batch)
    int32_t *ev_u = nullptr, *ev_i = nullptr;
    float* ev_r = nullptr;
    int64_t ev_n = 0;
    // profiling
    bool prof = false;
    std::vector<ProfEvent> events;
    double prof_ms[TFR_K_COUNT];
    int64_t prof_n[TFR_K_COUNT];
    float prof_overhead_ms = 0.f;     // elapsed time of an empty event pair on this stream
};

// ---------------------------------------------------------------------------------------
static int bits_for(int64_t rows) {
    int b = 1;
    while (((int64_t)1 << b) < rows && b < 31) ++b;
    return b;
}

template <typename T>
static int dmalloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    HIPCHK(hipMalloc((void**)p, count * sizeof(T)));
    return TFR_OK;
}

static void dfree(void* p) {
    if (p) (void)hipFree(p);
}

struct Prof {
    tfr_model* m;
    int kid;
    hipEvent_t a = nullptr, b = nullptr;
    Prof(tfr_model* m_, int kid_) : m(m_), kid(kid_) {
        if (m->prof) {
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
            (void)hipEventRecord(a, m->stream);
        }
    }
    ~Prof() {
        if (a && b) {
            (void)hipEventRecord(b, m->stream);
            m->events.push_back({a, b, kid});
        }
    }
};

static int drain_profile(tfr_model* m) {
    if (m->events.empty()) return TFR_OK;
    HIPCHK(hipStreamSynchronize(m->stream));
    for (auto& e : m->events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            ms -= m->prof_overhead_ms;                   // what an empty start/stop pair measures
            if (ms < 0.f) ms = 0.f;
            m->prof_ms[e.kid] += ms;
            m->prof_n[e.kid] += 1;
        }
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    m->events.clear();
    return TFR_OK;
}

static void free_workspace(tfr_model* m) {
    dfree(m->d_u); dfree(m->d_i); dfree(m->d_r); dfree(m->d_logits); dfree(m->d_g);
    dfree(m->ks_u); dfree(m->ps_u); dfree(m->ks_i); dfree(m->ps_i);
    dfree(m->ks2_u); dfree(m->ps2_u); dfree(m->ks2_i); dfree(m->ps2_i); dfree(m->gq); dfree(m->gp); dfree(m->gbq); dfree(m->gbp);
    dfree(m->partials); dfree(m->lrank_u); dfree(m->lrank_i); dfree(m->hist_u); dfree(m->hist_i);
    dfree(m->offs_u); dfree(m->offs_i); dfree(m->binbase_u); dfree(m->binbase_i);
    dfree(m->blocktot_u); dfree(m->blocktot_i);
    dfree(m->osel); m->osel = nullptr;
    for (int pz = 0; pz < 2; ++pz) for (int sd = 0; sd < 2; ++sd) { dfree(m->srt[pz][sd]); m->srt[pz][sd] = nullptr; }
    dfree(m->d_in); m->d_in = nullptr;
    if (m->h_in) (void)hipHostFree(m->h_in);
    if (m->h_out) (void)hipHostFree(m->h_out);
    if (m->h_err) (void)hipHostFree(m->h_err);
    m->h_err = nullptr;
    m->h_in = nullptr; m->h_out = nullptr; m->stage_cap = 0;
    dfree(m->alt.d_u); dfree(m->alt.d_i); dfree(m->alt.d_r); dfree(m->alt.ks_u); dfree(m->alt.ps_u); dfree(m->alt.ks_i); dfree(m->alt.ps_i);
    m->alt = tfr_model::SortSet(); m->alt_cap = 0;
    m->pf_valid = false;
    m->blocktot_u = m->blocktot_i = nullptr;
    m->lrank_u = m->lrank_i = m->hist_u = m->hist_i = nullptr;
    m->offs_u = m->offs_i = m->binbase_u = m->binbase_i = nullptr;
    m->d_u = m->d_i = nullptr; m->d_r = m->d_logits = m->d_g = nullptr;
    m->ks_u = m->ps_u = m->ks_i = m->ps_i = nullptr;
    m->ks2_u = m->ps2_u = m->ks2_i = m->ps2_i = nullptr; m->gq = m->gp = m->gbq = m->gbp = nullptr; m->partials = nullptr;
    m->cap = 0;
}

static int ensure_capacity(tfr_model* m, int64_t B) {
    if (B <= m->cap) return TFR_OK;
    HIPCHK(hipStreamSynchronize(m->stream));
    free_workspace(m);
    int64_t cap = 1024;
    while (cap < B) cap <<= 1;
    if (cap > (int64_t)1 << 30) return fail(TFR_ERR_ARG, "batch %lld too large", (long long)B);
    int rc;
    if ((rc = dmalloc(&m->d_u, cap))) return rc;
    if ((rc = dmalloc(&m->d_i, cap))) return rc;
    if ((rc = dmalloc(&m->d_r, cap))) return rc;
    if ((rc = dmalloc(&m->d_logits, cap + 4))) return rc;          // + {loss, reg, sum g, error flag} behind the logits
    if ((rc = dmalloc(&m->d_g, cap))) return rc;
    if ((rc = dmalloc(&m->ks_u, cap))) return rc;
    if ((rc = dmalloc(&m->ps_u, cap))) return rc;
    if ((rc = dmalloc(&m->ks_i, cap))) return rc;
    if ((rc = dmalloc(&m->ps_i, cap))) return rc;
    if ((rc = dmalloc(&m->ks2_u, cap))) return rc;
    if ((rc = dmalloc(&m->ps2_u, cap))) return rc;
    if ((rc = dmalloc(&m->ks2_i, cap))) return rc;
    if ((rc = dmalloc(&m->ps2_i, cap))) return rc;
    const bool tf1_ws = m->o.optimizer == TFR_OPT_ADAM && m->o.adam_mode == TFR_ADAM_TF1;
    // non-tf1: second third of gq parks the pieces of split user runs, the last third the
    // per-entry copies of pre-update item rows the fused user side reads
    if ((rc = dmalloc(&m->gq, (size_t)cap * m->D * (tf1_ws ? 1 : 3)))) return rc;
    if ((rc = dmalloc(&m->gbq, cap))) return rc;
    for (int pz = 0; pz < 2; ++pz)
        for (int sd = 0; sd < 2; ++sd)
            if ((rc = dmalloc(&m->srt[pz][sd], cap))) return rc;
    m->pf_valid = false;
    if ((rc = dmalloc(&m->gbp, cap))) return rc;
    if (tf1_ws)
        if ((rc = dmalloc(&m->gp, (size_t)cap * m->D))) return rc;
    {   // per-block {loss, reg, sum g}: the forward launches <= 8192 blocks, the reduce with the
        // forward fused in one block per 1024/G sorted entries
        const size_t nb = (size_t)(cap + 1024 / m->G - 1) / (1024 / m->G);
        if ((rc = dmalloc(&m->partials, (nb > 8192 ? nb : 8192) * 4))) return rc;
    }
    {
        const size_t ntiles = (size_t)(cap + CSORT_TILE - 1) / CSORT_TILE;
        const bool small = (1 << (m->bits_u > m->bits_i ? m->bits_u : m->bits_i)) <= CSORT_MAX_BINS;
        const size_t hu = (small ? ((size_t)1 << m->bits_u) : 256) * ntiles;
        const size_t hi = (small ? ((size_t)1 << m->bits_i) : 256) * ntiles;
        if ((rc = dmalloc(&m->osel, cap))) return rc;
        if ((rc = dmalloc(&m->lrank_u, cap))) return rc;
        if ((rc = dmalloc(&m->lrank_i, cap))) return rc;
        if ((rc = dmalloc(&m->hist_u, hu > 256 * ntiles ? hu : 256 * ntiles))) return rc;
        if ((rc = dmalloc(&m->hist_i, hi > 256 * ntiles ? hi : 256 * ntiles))) return rc;
        if ((rc = dmalloc(&m->offs_u, hu > 256 * ntiles ? hu : 256 * ntiles))) return rc;
        if ((rc = dmalloc(&m->offs_i, hi > 256 * ntiles ? hi : 256 * ntiles))) return rc;
        if ((rc = dmalloc(&m->binbase_u, small ? (size_t)1 << m->bits_u : 1))) return rc;
        if ((rc = dmalloc(&m->binbase_i, small ? (size_t)1 << m->bits_i : 1))) return rc;
        if ((rc = dmalloc(&m->blocktot_u, 64 + 256 * ntiles / 4096))) return rc;
        if ((rc = dmalloc(&m->blocktot_i, 64 + 256 * ntiles / 4096))) return rc;
        m->csort_ok = small;
    }
    m->cap = cap;
    return TFR_OK;
}

static int cancel_run_ahead(tfr_model* m);

static int ensure_step_out(tfr_model* m, int64_t nsteps) {
    if (nsteps <= m->step_out_cap) return TFR_OK;
    HIPCHK(hipStreamSynchronize(m->stream));
    dfree(m->step_out);
    m->step_out = nullptr;
    m->step_out_cap = 0;
    int rc;
    if ((rc = dmalloc(&m->step_out, (size_t)nsteps * 4))) return rc;
    m->step_out_cap = nsteps;
    return TFR_OK;
}

// read + clear the device error flag (stream must be idle or this call synchronises)
static int check_device_error(tfr_model* m) {
    // the flag lands in pinned memory: a pageable destination turns the 4-byte copy into a staged, blocking one
    if (!m->h_err) HIPCHK(hipHostMalloc((void**)&m->h_err, 64, hipHostMallocDefault));
    *m->h_err = 0;
    HIPCHK(hipMemcpyAsync(m->h_err, m->d_err, sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    const int32_t e = *m->h_err;
    if (e) {
        HIPCHK(hipMemsetAsync(m->d_err, 0, sizeof(int32_t), m->stream));
        HIPCHK(hipStreamSynchronize(m->stream));
        if (e & 1) return fail(TFR_ERR_OOB, "user/item id out of range [0,%lld) / [0,%lld)",
                               (long long)m->U, (long long)m->I);
        if (e == 8) return fail(TFR_ERR_OOB, "row-sharded step: another rank voided the step (capacity exceeded or id out of range there) - "
                                             "it was void on every rank; that rank's sync names the cause");
        if (e & 4) return fail(TFR_ERR_OOB, "row-sharded step: more local samples or distinct items per owner than the fixed capacities "
                                            "(sample_cap / slot_cap) hold - the step was void; raise the slack");
        return fail(TFR_ERR_OOB, "store index out of range [0,%lld)", (long long)m->N);
    }
    return TFR_OK;
}

#define MODEL_ENTER(m)                                                \
    if (!(m)) return fail(TFR_ERR_ARG, "null model");                 \
    HIPCHK(hipSetDevice((m)->device));

// TFR_CALL_TRACE=1: host-side timestamps (us since the call's entry) of a multi-step call's phases on stderr
struct CallTrace {
    bool on; std::chrono::steady_clock::time_point t0; const char* name;
    explicit CallTrace(const char* n) : name(n) {
        static int en = -1;
        if (en < 0) { const char* e = getenv("TFR_CALL_TRACE"); en = (e && e[0] == '1') ? 1 : 0; }
        on = en == 1;
        if (on) t0 = std::chrono::steady_clock::now();
    }
    void mark(const char* what) const {
        if (on) fprintf(stderr, "[%s] %-28s %8.1f us\n", name, what,
                        std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
};

// ---------------------------------------------------------------------------------------
extern "C" {

void tfr_default_opts(tfr_opts* o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->loss = TFR_LOSS_MSE;
    o->optimizer = TFR_OPT_ADAM;
    o->adam_mode = TFR_ADAM_TF1;
    o->lr = 1e-3f;
    o->reg = 0.05f;
    o->beta1 = 0.9f;
    o->beta2 = 0.999f;
    o->eps = 1e-8f;
}

int tfr_version(void) { return TFR_ABI_VERSION; }

const char* tfr_last_error(void) { return g_err; }

int tfr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- measurement yardstick: what a plain float4 read+write copy reaches on this device ---------------------------
// (MI355X_MICROARCH.md quotes 6.29 TB/s for exactly this shape of kernel; bench.py prints both)
// the copy form that is fastest on this part (tools/probes/copy_bw.hip: one 16-byte element per thread, short-lived workgroups in
// address order, streaming loads and stores - 6.2-6.5 TB/s where a grid-stride loop over the same bytes gives 4.8)
typedef float copy_f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_copy_f4(const copy_f4* __restrict__ src, copy_f4* __restrict__ dst, int64_t n4) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n4) __builtin_nontemporal_store(__builtin_nontemporal_load(&src[k]), &dst[k]);
}

int tfr_device_copy_rate(int32_t device, int64_t bytes, int32_t reps, double* best_gbs, double* mean_gbs) {
    if (bytes < (1 << 20) || reps < 1 || reps > 1000 || !best_gbs) return fail(TFR_ERR_ARG, "tfr_device_copy_rate: bad argument");
    HIPCHK(hipSetDevice(device));
    const int64_t n4 = bytes / 16;
    copy_f4 *a = nullptr, *b = nullptr;
    HIPCHK(hipMalloc(&a, n4 * 16));
    if (hipMalloc(&b, n4 * 16) != hipSuccess) { (void)hipFree(a); return fail(TFR_ERR_NOMEM, "tfr_device_copy_rate: out of memory"); }
    hipStream_t st;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    (void)hipMemsetAsync(a, 1, n4 * 16, st);
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((n4 + 255) / 256);
    double best = 0.0, sum = 0.0;
    for (int r = -2; r < reps; ++r) {
        (void)hipEventRecord(e0, st);
        k_copy_f4<<<grid, 256, 0, st>>>(a, b, n4);
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r < 0 || ms <= 0.f) continue;
        const double g = 2.0 * (double)(n4 * 16) / (ms * 1e-3) / 1e9;
        sum += g;
        if (g > best) best = g;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipStreamDestroy(st);
    (void)hipFree(a);
    (void)hipFree(b);
    *best_gbs = best;
    if (mean_gbs) *mean_gbs = sum / reps;
    return TFR_OK;
}

int tfr_destroy(tfr_model* m) {
    if (!m) return TFR_OK;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    for (auto& e : m->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    free_workspace(m);
    for (int t = 0; t < 5; ++t) { dfree(m->w[t]); dfree(m->m[t]); dfree(m->v[t]); }
    dfree(m->q_alt); dfree(m->q_sel);
    dfree(m->map_u); dfree(m->map_i); dfree(m->dg_p); dfree(m->dg_q); dfree(m->dg_bu); dfree(m->dg_bi); dfree(m->scalars); dfree(m->step_out); dfree(m->d_err);
    dfree(m->store); dfree(m->d_auc);
    dfree(m->d_ids); dfree(m->ev_u); dfree(m->ev_i); dfree(m->ev_r);
    if (m->stream2) { (void)hipStreamSynchronize(m->stream2); (void)hipStreamDestroy(m->stream2); }
    if (m->stream3) { (void)hipStreamSynchronize(m->stream3); (void)hipStreamDestroy(m->stream3); }
    for (auto e : m->chunk_ev) (void)hipEventDestroy(e);
    if (m->ev_ids_free) (void)hipEventDestroy(m->ev_ids_free);
    dfree(m->d_rng); dfree(m->d_ring); dfree(m->d_ids_alt); dfree(m->d_rng_snap);
    dfree(m->rb[0].recs); dfree(m->rb[1].recs);
    dfree(m->rng_ws.raw); dfree(m->rng_ws.counts); dfree(m->rng_ws.hdr);
    for (auto& R : m->rt) {
        dfree(R.mine); dfree(R.u); dfree(R.it); dfree(R.r); dfree(R.slot); dfree(R.counts);
        dfree(R.ks_u); dfree(R.ps_u); dfree(R.ks_i); dfree(R.ps_i); dfree(R.akeys); dfree(R.aks); dfree(R.aps);
    }
    if (m->spec_ev) (void)hipEventDestroy(m->spec_ev);
    if (m->h_ring) (void)hipHostFree(m->h_ring);
    for (int z = 0; z < tfr_model::HRING; ++z) if (m->ring_ev[z]) (void)hipEventDestroy(m->ring_ev[z]);
    for (int z = 0; z < 2; ++z) { if (m->ev_sorted[z]) (void)hipEventDestroy(m->ev_sorted[z]); if (m->ev_free[z]) (void)hipEventDestroy(m->ev_free[z]); }
    if (m->ev_first) (void)hipEventDestroy(m->ev_first);
    if (m->ev_mid) (void)hipEventDestroy(m->ev_mid);
    for (auto& e : m->draw_evs) if (e) (void)hipEventDestroy(e);
    if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
    delete m;
    return TFR_OK;
}

int tfr_create(tfr_model** out, int64_t U, int64_t I, int32_t D, const tfr_opts* opts) {
    if (!out) return fail(TFR_ERR_ARG, "out is null");
    *out = nullptr;
    if (!opts) return fail(TFR_ERR_ARG, "opts is null");
    if (U < 1 || I < 1 || U > 0x7fffffffLL || I > 0x7fffffffLL)
        return fail(TFR_ERR_ARG, "user_num/item_num must be in [1, 2^31)");
    int G, VEC;
    if (!geometry(D, &G, &VEC))
        return fail(TFR_ERR_ARG, "unsupported dim %d (need dim %% 4 == 0 and dim <= 256, or dim <= 64)", D);
    if (opts->loss != TFR_LOSS_MSE && opts->loss != TFR_LOSS_NLL) return fail(TFR_ERR_ARG, "bad loss");
    if (opts->optimizer != TFR_OPT_ADAM && opts->optimizer != TFR_OPT_SGD) return fail(TFR_ERR_ARG, "bad optimizer");
    if (opts->adam_mode != TFR_ADAM_TF1 && opts->adam_mode != TFR_ADAM_LAZY) return fail(TFR_ERR_ARG, "bad adam_mode");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return fail(TFR_ERR_HIP, "no HIP device available (%s) - this library has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (opts->device < 0 || opts->device >= ndev) return fail(TFR_ERR_ARG, "device %d not in [0,%d)", opts->device, ndev);
    HIPCHK(hipSetDevice(opts->device));
    tfr_model* m = new (std::nothrow) tfr_model();
    if (!m) return fail(TFR_ERR_NOMEM, "host allocation failed");
    m->U = U; m->I = I; m->D = D; m->G = G; m->VEC = VEC;
    m->o = *opts;
    m->device = opts->device;
    m->bits_u = bits_for(U);
    m->bits_i = bits_for(I);
    m->n[TFR_MU] = 1; m->n[TFR_BU] = U; m->n[TFR_BI] = I; m->n[TFR_P] = U * D; m->n[TFR_Q] = I * D;
    m->b1p = opts->beta1;
    m->b2p = opts->beta2;
    memset(m->prof_ms, 0, sizeof(m->prof_ms));
    memset(m->prof_n, 0, sizeof(m->prof_n));
    int rc = TFR_OK;
    do {
        if (hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking) != hipSuccess) {
            rc = fail(TFR_ERR_HIP, "hipStreamCreate failed");
            break;
        }
        m->stream = m->own_stream;
        const bool adam = opts->optimizer == TFR_OPT_ADAM;
        for (int t = 0; t < 5 && rc == TFR_OK; ++t) {
            rc = dmalloc(&m->w[t], (size_t)m->n[t]);
            if (rc == TFR_OK && hipMemsetAsync(m->w[t], 0, (size_t)m->n[t] * 4, m->stream) != hipSuccess)
                rc = fail(TFR_ERR_HIP, "memset failed");
            if (rc == TFR_OK && adam) {
                rc = dmalloc(&m->m[t], (size_t)m->n[t]);
                if (rc == TFR_OK) rc = dmalloc(&m->v[t], (size_t)m->n[t]);
                if (rc == TFR_OK && (hipMemsetAsync(m->m[t], 0, (size_t)m->n[t] * 4, m->stream) != hipSuccess ||
                                     hipMemsetAsync(m->v[t], 0, (size_t)m->n[t] * 4, m->stream) != hipSuccess))
                    rc = fail(TFR_ERR_HIP, "memset failed");
            }
        }
        if (rc) break;
        if (adam && opts->adam_mode == TFR_ADAM_TF1) {
            if ((rc = dmalloc(&m->map_u, (size_t)U))) break;
            if ((rc = dmalloc(&m->map_i, (size_t)I))) break;
            if (hipMemsetAsync(m->map_u, 0, (size_t)U * 4, m->stream) != hipSuccess ||
                hipMemsetAsync(m->map_i, 0, (size_t)I * 4, m->stream) != hipSuccess) {
                rc = fail(TFR_ERR_HIP, "memset failed");
                break;
            }
            // dense per-row gradient buffers for the TF1 sweep, while they stay small (<= 256 MB)
            if ((size_t)(U + I) * (D + 1) * 4 <= ((size_t)256 << 20)) {
                if ((rc = dmalloc(&m->dg_p, (size_t)U * D))) break;
                if ((rc = dmalloc(&m->dg_q, (size_t)I * D))) break;
                if ((rc = dmalloc(&m->dg_bu, (size_t)U))) break;
                if ((rc = dmalloc(&m->dg_bi, (size_t)I))) break;
                if (hipMemsetAsync(m->dg_p, 0, (size_t)U * D * 4, m->stream) != hipSuccess ||
                    hipMemsetAsync(m->dg_q, 0, (size_t)I * D * 4, m->stream) != hipSuccess ||
                    hipMemsetAsync(m->dg_bu, 0, (size_t)U * 4, m->stream) != hipSuccess ||
                    hipMemsetAsync(m->dg_bi, 0, (size_t)I * 4, m->stream) != hipSuccess) {
                    rc = fail(TFR_ERR_HIP, "memset failed");
                    break;
                }
            }
        }
        if ((rc = dmalloc(&m->scalars, 4))) break;
        if ((rc = dmalloc(&m->d_err, 1))) break;
        if (hipMemsetAsync(m->scalars, 0, 16, m->stream) != hipSuccess ||
            hipMemsetAsync(m->d_err, 0, 4, m->stream) != hipSuccess ||
            hipStreamSynchronize(m->stream) != hipSuccess) {
            rc = fail(TFR_ERR_HIP, "init failed: %s", hipGetErrorString(hipGetLastError()));
            break;
        }
    } while (0);
    if (rc) {
        char keep[512];
        strncpy(keep, g_err, sizeof(keep));
        tfr_destroy(m);
        strncpy(g_err, keep, sizeof(g_err));
        return rc;
    }
    *out = m;
    return TFR_OK;
}

// ---- variables -------------------------------------------------------------------------
// two-table form: every row that lives in the alternate item table goes back to the main one.  Called by everything that
// looks at item_features other than the fused big-table step itself (forward / eval, get / set, the other step paths).
static int settle_q(tfr_model* m) {
    if (!m->q_dirty) return TFR_OK;
    launch_settle_alt(m->w[TFR_Q], m->q_alt, m->q_sel, m->I, m->D, m->stream);
    HIPCHK(hipGetLastError());
    m->q_dirty = false;
    return TFR_OK;
}

static int table_ptr(tfr_model* m, int32_t which, float** p, int64_t* n) {
    const int t = which & 7;
    if (t > TFR_Q || (which & ~(7 | TFR_SLOT_M | TFR_SLOT_V)) || ((which & TFR_SLOT_M) && (which & TFR_SLOT_V)))
        return fail(TFR_ERR_ARG, "bad table id %d", which);
    if (t == TFR_Q && settle_q(m)) return TFR_ERR_HIP;
    float* q = (which & TFR_SLOT_M) ? m->m[t] : (which & TFR_SLOT_V) ? m->v[t] : m->w[t];
    if (!q) return fail(TFR_ERR_STATE, "table %d has no such slot (optimizer is not Adam)", which);
    *p = q;
    *n = m->n[t];
    return TFR_OK;
}

int tfr_set_table(tfr_model* m, int32_t which, const float* host, int64_t n) {
    MODEL_ENTER(m);
    float* p; int64_t cnt;
    int rc = table_ptr(m, which, &p, &cnt);
    if (rc) return rc;
    if (!host || n != cnt) return fail(TFR_ERR_ARG, "table %d expects %lld floats, got %lld", which, (long long)cnt, (long long)n);
    HIPCHK(hipMemcpyAsync(p, host, (size_t)n * 4, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return TFR_OK;
}

int tfr_get_table(tfr_model* m, int32_t which, float* host, int64_t n) {
    MODEL_ENTER(m);
    float* p; int64_t cnt;
    int rc = table_ptr(m, which, &p, &cnt);
    if (rc) return rc;
    if (!host || n != cnt) return fail(TFR_ERR_ARG, "table %d holds %lld floats, asked %lld", which, (long long)cnt, (long long)n);
    HIPCHK(hipMemcpyAsync(host, p, (size_t)n * 4, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return TFR_OK;
}

int tfr_table_devptr(tfr_model* m, int32_t which, void** ptr, int64_t* n) {
    MODEL_ENTER(m);
    float* p; int64_t cnt;
    int rc = table_ptr(m, which, &p, &cnt);
    if (rc) return rc;
    if (ptr) *ptr = p;
    if (n) *n = cnt;
    return TFR_OK;
}

int tfr_scalars_devptr(tfr_model* m, void** ptr) {
    MODEL_ENTER(m);
    if (ptr) *ptr = m->scalars;
    return TFR_OK;
}

int tfr_set_frozen(tfr_model* m, uint32_t mask) {
    MODEL_ENTER(m);
    if (mask >> 5) return fail(TFR_ERR_ARG, "frozen mask has bits beyond the 5 tables");
    m->frozen = mask;
    return TFR_OK;
}

int tfr_get_step(tfr_model* m, int64_t* step, float* b1p, float* b2p) {
    MODEL_ENTER(m);
    if (step) *step = m->step;
    if (b1p) *b1p = m->b1p;
    if (b2p) *b2p = m->b2p;
    return TFR_OK;
}

int tfr_set_step(tfr_model* m, int64_t step, float b1p, float b2p) {
    MODEL_ENTER(m);
    if (step < 0) return fail(TFR_ERR_ARG, "negative step");
    m->step = step;
    m->b1p = b1p;
    m->b2p = b2p;
    return TFR_OK;
}

int tfr_set_hyper(tfr_model* m, float lr, float reg) {
    MODEL_ENTER(m);
    m->o.lr = lr;
    m->o.reg = reg;
    return TFR_OK;
}

int tfr_set_stream(tfr_model* m, void* s) {
    MODEL_ENTER(m);
    HIPCHK(hipStreamSynchronize(m->stream));
    m->stream = s ? (hipStream_t)s : m->own_stream;
    m->draw_joined = 0;                                  // joins were made on the old stream
    return TFR_OK;
}

// the same without draining the old stream: for a caller that alternates between two streams and orders them itself (events)
int tfr_switch_stream(tfr_model* m, void* s) {
    MODEL_ENTER(m);
    m->stream = s ? (hipStream_t)s : m->own_stream;
    m->draw_joined = 0;                                  // joins were made on the old stream
    return TFR_OK;
}

int tfr_get_stream(tfr_model* m, void** s) {
    MODEL_ENTER(m);
    if (s) *s = (void*)m->stream;
    return TFR_OK;
}

int tfr_sync(tfr_model* m) {
    MODEL_ENTER(m);
    CallTrace tr("sync");
    const int rc = check_device_error(m);
    tr.mark("main stream drained");
    return rc;
}

int tfr_profile(tfr_model* m, int32_t enable) {
    MODEL_ENTER(m);
    int rc = drain_profile(m);
    if (rc) return rc;
    if (enable) {
        memset(m->prof_ms, 0, sizeof(m->prof_ms));
        memset(m->prof_n, 0, sizeof(m->prof_n));
        // calibrate: median-of-9 elapsed time of back-to-back event pairs with nothing in between
        float v[9];
        int n = 0;
        for (int k = 0; k < 9; ++k) {
            hipEvent_t e0, e1;
            if (hipEventCreate(&e0) != hipSuccess) break;
            if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); break; }
            (void)hipEventRecord(e0, m->stream);
            (void)hipEventRecord(e1, m->stream);
            float ms = 0.f;
            if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) v[n++] = ms;
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
        }
        for (int x = 1; x < n; ++x) for (int y = x; y > 0 && v[y] < v[y - 1]; --y) { float t = v[y]; v[y] = v[y - 1]; v[y - 1] = t; }
        m->prof_overhead_ms = n ? v[n / 2] : 0.f;
    }
    m->prof = enable != 0;
    return TFR_OK;
}

int tfr_profile_read(tfr_model* m, int32_t kernel, double* total_ms, int64_t* launches) {
    MODEL_ENTER(m);
    if (kernel < 0 || kernel >= TFR_K_COUNT) return fail(TFR_ERR_ARG, "bad kernel id");
    int rc = drain_profile(m);
    if (rc) return rc;
    if (total_ms) *total_ms = m->prof_ms[kernel];
    if (launches) *launches = m->prof_n[kernel];
    return TFR_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------
// forward on device-resident ids
static int run_forward(tfr_model* m, int mode, const int32_t* du, const int32_t* di, const float* dr,
                       int64_t B, float* d_logits, float* d_g, int* nblk_out,
                       const int64_t* d_store_ids = nullptr) {
    { const int rcq = settle_q(m); if (rcq) return rcq; }
    FwdArgs a;
    memset(&a, 0, sizeof(a));
    a.P = m->w[TFR_P]; a.Q = m->w[TFR_Q]; a.bu = m->w[TFR_BU]; a.bi = m->w[TFR_BI]; a.mu = m->w[TFR_MU];
    a.u = du; a.it = di; a.r = dr;
    a.logits = d_logits; a.g = d_g; a.partials = m->partials; a.err = m->d_err;
    a.B = B; a.U = m->U; a.I = m->I; a.N = m->N;
    if (d_store_ids) {                 // fused gather from the resident store
        a.ids = d_store_ids; a.store = m->store;
        a.u_out = m->d_u; a.it_out = m->d_i;
    }
    { static int lr = -1; if (lr < 0) { const char* e = getenv("TFR_LDS_REDUCE"); lr = (e && e[0] == '1') ? 1 : 0; } a.lds_reduce = lr; }
    a.D = m->D; a.loss = m->o.loss; a.item_abs = m->o.item_abs; a.reg_bias = m->o.reg_bias;
    const int grid = forward_grid(B, m->G, mode);
    if (nblk_out) *nblk_out = grid;
    {
        Prof p(m, TFR_K_FORWARD);
        launch_forward(a, mode, m->G, m->VEC, grid, m->stream);
    }
    HIPCHK(hipGetLastError());
    return TFR_OK;
}

// hand-written LSD radix sort of one or two key columns: column c = (keys[c], bits[c]) ->
// sorted keys in ks_out[c], original positions in ps_out[c].  ceil(maxbits/8) passes, 3 launches each.
static int radix_sort_columns(tfr_model* m, int ncols, const int32_t* const* keys, const int* bits,
                              int32_t* const* ks_out, int32_t* const* ps_out, int64_t B,
                              const int64_t* limits = nullptr, const int64_t* store_ids = nullptr) {
    int maxbits = bits[0];
    if (ncols > 1 && bits[1] > maxbits) maxbits = bits[1];
    const int passes = (maxbits + 7) / 8;
    int32_t* tmpk[2] = {m->ks2_u, m->ks2_i};
    int32_t* tmpv[2] = {m->ps2_u, m->ps2_i};
    RSortArgs r;
    memset(&r, 0, sizeof(r));
    r.B = B;
    r.ntiles = (int32_t)((B + CSORT_TILE - 1) / CSORT_TILE);
    r.lrank[0] = m->lrank_u; r.lrank[1] = m->lrank_i;
    r.hist[0] = m->hist_u; r.hist[1] = m->hist_i;
    r.offs[0] = m->offs_u; r.offs[1] = m->offs_i;
    r.blocktot[0] = m->blocktot_u; r.blocktot[1] = m->blocktot_i;
    r.chunk = 4096;                                      // entries per scan block (k_rsort_scan)
    for (int p = 0; p < passes; ++p) {
        const bool to_final = ((passes - 1 - p) % 2) == 0;      // last pass lands in ks_out / ps_out
        for (int c = 0; c < ncols; ++c) {
            if (p == 0) { r.keys_in[c] = keys[c]; r.vals_in[c] = nullptr; }
            else { r.keys_in[c] = r.keys_out[c]; r.vals_in[c] = r.vals_out[c]; }
        }
        for (int c = 0; c < ncols; ++c) {
            r.keys_out[c] = to_final ? ks_out[c] : tmpk[c];
            r.vals_out[c] = to_final ? ps_out[c] : tmpv[c];
        }
        r.shift = 8 * p;
        r.ids = nullptr;
        if (p == 0 && store_ids) {                               // the first pass gathers the batch from the resident store itself
            r.ids = store_ids; r.store = m->store; r.N = m->N;
            r.u_out = m->d_u; r.i_out = m->d_i; r.r_out = m->d_r;
        }
        r.err = (p == 0 && limits) ? m->d_err : nullptr;         // ids outside the tables void the step
        if (limits) for (int c = 0; c < ncols; ++c) r.limit[c] = (int32_t)limits[c];
        static int wide = -1;                                    // TFR_RSORT_WIDE=0: A/B switch
        if (wide < 0) { const char* e = getenv("TFR_RSORT_WIDE"); wide = (e && e[0] == '0') ? 0 : 1; }
        if (wide && rsortw_eligible(B)) launch_rsortw_pass(r, ncols, m->stream);   // from 65536 keys: 4096-key tiles, LDS-staged scatter
        else launch_rsort_pass(r, ncols, m->stream);
    }
    HIPCHK(hipGetLastError());
    return TFR_OK;
}

// stable sort of batch positions by user id and by item id
static int sort_columns(tfr_model* m, const int32_t* du, const int32_t* di, int64_t B,
                        const FinArgs* fin = nullptr, bool* fin_done = nullptr, bool validate = false,
                        const int64_t* store_ids = nullptr) {
    Prof p(m, TFR_K_SORT);
    if (m->csort_ok && csort_eligible(B, m->bits_u, m->bits_i)) {
        CSortArgs c;
        c.keys[0] = du; c.keys[1] = di;
        c.ks[0] = m->ks_u; c.ks[1] = m->ks_i; c.ps[0] = m->ps_u; c.ps[1] = m->ps_i;
        c.lrank[0] = m->lrank_u; c.lrank[1] = m->lrank_i; c.hist[0] = m->hist_u; c.hist[1] = m->hist_i;
        c.offs[0] = m->offs_u; c.offs[1] = m->offs_i; c.binbase[0] = m->binbase_u; c.binbase[1] = m->binbase_i;
        c.blocktot[0] = m->blocktot_u; c.blocktot[1] = m->blocktot_i;
        c.nbins[0] = 1 << m->bits_u; c.nbins[1] = 1 << m->bits_i;
        c.ntiles = (int32_t)((B + CSORT_TILE - 1) / CSORT_TILE);
        c.B = B;
        launch_csort(c, fin, m->stream);
        if (fin && fin_done) *fin_done = true;
        HIPCHK(hipGetLastError());
        return TFR_OK;
    }
    const int32_t* keys[2] = {du, di};
    const int bits[2] = {m->bits_u, m->bits_i};
    int32_t* ks[2] = {m->ks_u, m->ks_i};
    int32_t* ps[2] = {m->ps_u, m->ps_i};
    const int64_t limits[2] = {m->U, m->I};
    return radix_sort_columns(m, 2, keys, bits, ks, ps, B, validate ? limits : nullptr, store_ids);
}

// big tables with a touched-rows optimiser: the forward is computed inside the item-side reduce,
// which needs the sorted order first - so the step starts with (gather +) sort
static bool fwd_in_reduce(const tfr_model* m, int64_t B) {
    const bool tf1 = m->o.optimizer == TFR_OPT_ADAM && m->o.adam_mode == TFR_ADAM_TF1;
    return B > 0 && !tf1 && !(m->csort_ok && csort_eligible(B, m->bits_u, m->bits_i));
}

static int gather_batch(tfr_model* m, const int64_t* d_ids, int64_t lo, int64_t B);

// forward (+ fused store gather) and the stable sort of both id columns for one minibatch; K4
// (`f`) rides in the csort scan launch when that path is taken (fin_done).  du/di are updated
// to where the batch ids live afterwards.
static bool tiles_eligible(const tfr_model* m, int64_t B) {
    return B > 0 && m->csort_ok && csort_eligible(B, m->bits_u, m->bits_i) && (B + CSORT_TILE - 1) / CSORT_TILE <= 16;
}

static int front_and_sort(tfr_model* m, const int32_t*& du, const int32_t*& di, const float*& dr, int64_t B,
                          float* d_logits, const int64_t* d_store_ids, FinArgs& f, int& nblk, bool& fin_done,
                          bool tiles = false, bool sort_only = false) {
    m->pf_valid = false;                                 // the sort scratch doubles as the published tables of the tile step
    const tfr_opts& o = m->o;
    hipStream_t s = m->stream;
    int rc;
    if (sort_only) {
        const bool radix = !(m->csort_ok && csort_eligible(B, m->bits_u, m->bits_i));
        static int fuse = -1;                            // TFR_FUSE_GATHER=0: A/B switch (separate k_gather_triples launch)
        if (fuse < 0) { const char* e = getenv("TFR_FUSE_GATHER"); fuse = (e && e[0] == '0') ? 0 : 1; }
        if (d_store_ids && radix && fuse) {                      // the radix sort's first pass gathers the batch itself (one launch less)
            du = m->d_u; di = m->d_i; dr = m->d_r;
            return sort_columns(m, du, di, B, nullptr, nullptr, true, d_store_ids);
        }
        if (d_store_ids) {
            if ((rc = gather_batch(m, d_store_ids, 0, B))) return rc;
            du = m->d_u; di = m->d_i; dr = m->d_r;
        }
        return sort_columns(m, du, di, B, nullptr, nullptr, true);       // + id range check
    }
    if (d_store_ids && B >= 32768) {
        // big batches are bandwidth-bound: a separate gather keeps the forward's dependent chain
        // at ids -> rows; small batches are launch-bound and gather inside the forward instead
        if ((rc = gather_batch(m, d_store_ids, 0, B))) return rc;
        d_store_ids = nullptr;
        du = m->d_u; di = m->d_i; dr = m->d_r;
    }
    if (m->csort_ok && csort_eligible(B, m->bits_u, m->bits_i)) {
        // small tables: forward and the counting sort's rank pass share one launch, then
        // scan (+K4) and scatter
        FrontArgs fa;
        memset(&fa, 0, sizeof(fa));
        FwdArgs& fw = fa.f;
        fw.P = m->w[TFR_P]; fw.Q = m->w[TFR_Q]; fw.bu = m->w[TFR_BU]; fw.bi = m->w[TFR_BI]; fw.mu = m->w[TFR_MU];
        fw.u = du; fw.it = di; fw.r = dr;
        fw.logits = d_logits; fw.g = m->d_g; fw.partials = m->partials; fw.err = m->d_err;
        fw.B = B; fw.U = m->U; fw.I = m->I; fw.N = m->N;
        fw.D = m->D; fw.loss = o.loss; fw.item_abs = o.item_abs; fw.reg_bias = o.reg_bias;
        if (d_store_ids) { fw.ids = d_store_ids; fw.store = m->store; }     // rank blocks publish the ids
        CSortArgs& c = fa.c;
        c.keys[0] = d_store_ids ? m->d_u : du; c.keys[1] = d_store_ids ? m->d_i : di;
        c.ks[0] = m->ks_u; c.ks[1] = m->ks_i; c.ps[0] = m->ps_u; c.ps[1] = m->ps_i;
        c.lrank[0] = m->lrank_u; c.lrank[1] = m->lrank_i; c.hist[0] = m->hist_u; c.hist[1] = m->hist_i;
        c.offs[0] = m->offs_u; c.offs[1] = m->offs_i; c.binbase[0] = m->binbase_u; c.binbase[1] = m->binbase_i;
        c.blocktot[0] = m->blocktot_u; c.blocktot[1] = m->blocktot_i;
        c.nbins[0] = 1 << m->bits_u; c.nbins[1] = 1 << m->bits_i;
        c.ntiles = (int32_t)((B + CSORT_TILE - 1) / CSORT_TILE);
        c.B = B;
        fa.key_out[0] = m->d_u; fa.key_out[1] = m->d_i;
        fa.nfwd = front_forward_blocks(B, m->G);
        fa.tile_local = tiles ? 1 : 0;
        nblk = fa.nfwd;
        {
            Prof p(m, TFR_K_FORWARD);
            launch_front(fa, m->G, m->VEC, s);
        }
        HIPCHK(hipGetLastError());
        if (d_store_ids) { du = m->d_u; di = m->d_i; }
        f.nblk = nblk;
        if (tiles) return TFR_OK;      // tile-local order is final; K4 rides in the sweep launch
        {
            Prof p(m, TFR_K_SORT);
            launch_csort_tail(c, &f, s);
        }
        HIPCHK(hipGetLastError());
        fin_done = true;
    } else {
        rc = run_forward(m, MODE_TRAIN, du, di, dr, B, d_logits, m->d_g, &nblk, d_store_ids);
        if (rc) return rc;
        if (d_store_ids) { du = m->d_u; di = m->d_i; }
        f.nblk = nblk;
        if ((rc = sort_columns(m, du, di, B, &f, &fin_done))) return rc;
    }
    return TFR_OK;
}

// K1+K2+K3 of the small-table step in one launch (k_tile_step).  Look-ahead (multi-step calls on the
// resident store): was this batch's tile sort published by the previous launch?  Is there a next
// batch to sort in this one?  The packed tables and the sorted records are double-buffered by step
// parity (hist_* / offs_* serve as the two tables).  *par_out = which table set k_dense_tiles reads.
// ---- record buffers (tfr_model::RecBuf) ----------------------------------------------------------------
static bool recs_on() {                                  // TFR_RECS=0: A/B switch
    static int on = -1;
    if (on < 0) { const char* e = getenv("TFR_RECS"); on = (e && e[0] == '0') ? 0 : 1; }
    return on == 1;
}
static void recs_forget(tfr_model* m) { m->rb[0].n = 0; m->rb[1].n = 0; }      // the store changed / an id buffer was rewritten elsewhere
// the record buffer that goes with the id buffer at `base` (capacity `cap` ids); a slot whose id buffer is gone is reused
static tfr_model::RecBuf* recs_slot(tfr_model* m, const int64_t* base, int64_t cap) {
    if (!recs_on() || !base) return nullptr;
    tfr_model::RecBuf* r = nullptr;
    for (int k = 0; k < 2; ++k) if (m->rb[k].ids_base == base) r = &m->rb[k];
    if (!r) {
        for (int k = 0; k < 2 && !r; ++k)
            if (m->rb[k].ids_base != m->d_ids && m->rb[k].ids_base != m->d_ids_alt) r = &m->rb[k];
        if (!r) return nullptr;
        r->ids_base = base; r->n = 0;
    }
    if (r->cap < cap) {                                  // the old records' readers: the id buffer they went with was freed after a sync
        dfree(r->recs);
        r->recs = nullptr; r->cap = 0; r->n = 0;
        if (hipMalloc((void**)&r->recs, (size_t)cap * sizeof(int4)) != hipSuccess) { (void)hipGetLastError(); r->ids_base = nullptr; return nullptr; }
        r->cap = cap;
    }
    return r;
}
// after a run of ids [off, off + count) of the buffer at `base` was written on stream `st`: gather their records behind it
static void recs_follow(tfr_model* m, const int64_t* base, int64_t cap, int64_t off, int64_t count, hipStream_t st) {
    tfr_model::RecBuf* r = recs_slot(m, base, cap);
    if (!r) return;
    if (off > r->n) { r->n = 0; return; }                // a gap: nothing before `off` is known to have records
    launch_gather_recs(base + off, m->store, r->recs + off, count, m->N, st);
    r->n = off + count;
}
static const int4* recs_for(tfr_model* m, const int64_t* ids, int64_t B) {
    if (!ids || !recs_on()) return nullptr;
    for (int k = 0; k < 2; ++k) {
        const tfr_model::RecBuf& r = m->rb[k];
        if (r.recs && r.ids_base && ids >= r.ids_base && (ids - r.ids_base) + B <= r.n) return r.recs + (ids - r.ids_base);
    }
    return nullptr;
}

static int tile_step_launch(tfr_model* m, const int32_t* du, const int32_t* di, const float* dr, int64_t B,
                            float* d_logits, const int64_t* d_store_ids, const int64_t* next_store_ids,
                            float* gp_rows, int* par_out, int* nblk_out) {
    const tfr_opts& o = m->o;
    TileStepArgs ts;
    memset(&ts, 0, sizeof(ts));
    ts.P = m->w[TFR_P]; ts.Q = m->w[TFR_Q]; ts.bu = m->w[TFR_BU]; ts.bi = m->w[TFR_BI]; ts.mu = m->w[TFR_MU];
    ts.u = du; ts.it = di; ts.r = dr;
    if (d_store_ids) { ts.ids = d_store_ids; ts.store = m->store; ts.recs = recs_for(m, d_store_ids, B); }
    ts.logits = d_logits; ts.partials = m->partials; ts.err = m->d_err;
    const bool presorted = m->pf_valid && d_store_ids && m->pf_ids == d_store_ids && m->pf_B == B;
    const int par = presorted ? m->pf_par : 0;
    int32_t* tabs[2][2] = {{m->hist_u, m->hist_i}, {m->offs_u, m->offs_i}};
    ts.tab[0] = tabs[par][0]; ts.tab[1] = tabs[par][1];
    if (presorted) { ts.srt[0] = m->srt[par][0]; ts.srt[1] = m->srt[par][1]; }
    m->pf_valid = false;
    if (next_store_ids && d_store_ids) {
        ts.store = m->store;
        ts.next_ids = next_store_ids; ts.next_B = B; ts.next_ntiles = (int32_t)((B + CSORT_TILE - 1) / CSORT_TILE);
        ts.next_recs = recs_for(m, next_store_ids, B);
        ts.next_tab[0] = tabs[par ^ 1][0]; ts.next_tab[1] = tabs[par ^ 1][1];
        ts.next_srt[0] = m->srt[par ^ 1][0]; ts.next_srt[1] = m->srt[par ^ 1][1];
        m->pf_valid = true; m->pf_ids = next_store_ids; m->pf_B = B; m->pf_par = par ^ 1;
    }
    ts.grad_rows[0] = gp_rows; ts.grad_rows[1] = m->gq;
    ts.grad_bias[0] = m->gbp; ts.grad_bias[1] = m->gbq;
    ts.B = B; ts.U = m->U; ts.I = m->I; ts.N = m->N;
    ts.D = m->D; ts.loss = o.loss; ts.item_abs = o.item_abs; ts.reg_bias = o.reg_bias;
    ts.ntiles = (int32_t)((B + CSORT_TILE - 1) / CSORT_TILE);
    ts.nbins[0] = 1 << m->bits_u; ts.nbins[1] = 1 << m->bits_i;
    ts.lam = o.reg;
    *par_out = par;
    *nblk_out = ts.ntiles * m->G;            // one {loss, reg, sum g} slot per piece
    static int dbg_on = -1;                  // TFR_TILE_DEBUG=1: per-block start / end stamps of every launch on stderr (synchronises)
    if (dbg_on < 0) { const char* e = getenv("TFR_TILE_DEBUG"); dbg_on = (e && e[0] == '1') ? 1 : 0; }
    static unsigned long long* d_dbg = nullptr;
    if (dbg_on && !d_dbg) (void)hipMalloc((void**)&d_dbg, 2048 * 64);
    if (dbg_on && d_dbg && 2 * (ts.next_ntiles + ts.ntiles * m->G) <= 2048) { (void)hipMemsetAsync(d_dbg, 0, 2048 * 64, m->stream); ts.dbg = d_dbg; }
    {
        Prof p(m, TFR_K_REDUCE_ITEM);
        launch_tile_step(ts, m->G, m->VEC, m->stream);
    }
    HIPCHK(hipGetLastError());
    if (ts.dbg) {
        std::vector<unsigned long long> h(2048 * 8);
        (void)hipStreamSynchronize(m->stream);
        (void)hipMemcpy(h.data(), d_dbg, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0;
        for (size_t k = 0; k < h.size(); k += 8) if (h[k]) { if (h[k] < t0) t0 = h[k]; if (h[k + 1] > t1) t1 = h[k + 1]; }
        const int nsort = ts.next_ids ? 2 * ts.next_ntiles : 0;
        const size_t gridx = (size_t)nsort + (size_t)ts.ntiles * (m->G / tile_step_epg(ts.ntiles, m->G, m->VEC));
        double ahead_end = 0, ahead_dur = 0, comp_end = 0, comp_dur = 0, ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, phl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int longest_y = -1; size_t longest_k = 0;
        int nblocks = 0, ncomp = 0;
        for (size_t k = 0; k < h.size() / 8; ++k) {
            const unsigned long long* b = &h[8 * k];
            if (!b[0]) continue;
            ++nblocks;
            const double st = (b[0] - t0) / 100.0, en = (b[1] - t0) / 100.0;
            const bool ah = k < gridx && (int)k < nsort;                // blockIdx.y == 0 and blockIdx.x < nsort
            if (ah) { if (en > ahead_end) ahead_end = en; if (en - st > ahead_dur) ahead_dur = en - st; }
            else {
                if (en > comp_end) comp_end = en;
                if (en - st > comp_dur) {
                    comp_dur = en - st; longest_k = k; longest_y = (int)(k / gridx);
                    if (b[2] && b[6]) { for (int q = 2; q <= 6; ++q) phl[q] = (b[q] - b[0]) / 100.0; phl[7] = en - st; }
                }
                if (b[2] && b[6]) { ++ncomp; for (int q = 2; q <= 6; ++q) ph[q] += (b[q] - b[0]) / 100.0; ph[7] += en - st; }
            }
        }
        fprintf(stderr, "[k_tile_step] %d blocks (%d look-ahead): span %.2f us; look-ahead blocks end by %.2f (longest %.2f), step blocks end by "
                        "%.2f (longest %.2f); mean step block, us since its start: records %.2f, rows+contributions %.2f, wave sums staged %.2f, "
                        "partials %.2f, wave rounds %.2f, end %.2f\n", nblocks, nsort, (t1 - t0) / 100.0, ahead_end, ahead_dur, comp_end, comp_dur,
                ph[2] / (ncomp ? ncomp : 1), ph[3] / (ncomp ? ncomp : 1), ph[4] / (ncomp ? ncomp : 1), ph[5] / (ncomp ? ncomp : 1),
                ph[6] / (ncomp ? ncomp : 1), ph[7] / (ncomp ? ncomp : 1));
        fprintf(stderr, "[k_tile_step] longest step block: side %d, block %zu of its side: records %.2f, rows+contributions %.2f, wave sums staged %.2f, "
                        "partials %.2f, wave rounds %.2f, end %.2f\n", longest_y, longest_k % gridx, phl[2], phl[3], phl[4], phl[5], phl[6], phl[7]);
    }
    return TFR_OK;
}

// one minibatch on device-resident (u, i, r) - or, with d_store_ids, on rows of the resident
// store gathered inside the forward kernel; out3 = optional device {loss, reg, sum_g} slot
static int run_train_step(tfr_model* m, const int32_t* du, const int32_t* di, const float* dr, int64_t B,
                          float* d_logits, float* out3, const int64_t* d_store_ids = nullptr,
                          const int64_t* next_store_ids = nullptr, bool presorted_big = false, bool out_err = false) {
    const tfr_opts& o = m->o;
    const bool adam = o.optimizer == TFR_OPT_ADAM;
    const bool tf1 = adam && o.adam_mode == TFR_ADAM_TF1;
    // lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t), float32 like the TF graph [TF1-lib]
    const float alpha = adam ? o.lr * sqrtf(1.f - m->b2p) / (1.f - m->b1p) : 0.f;
    int nblk = 0;
    hipStream_t s = m->stream;
    bool fin_done = false, tiles = false;
    FinArgs f;
    memset(&f, 0, sizeof(f));
    f.partials = m->partials; f.scalars = m->scalars; f.out = out3; f.out_err = out_err ? 1 : 0;
    f.mu = m->w[TFR_MU]; f.mu_m = m->m[TFR_MU]; f.mu_v = m->v[TFR_MU]; f.err = m->d_err;
    f.update_mu = !((m->frozen >> TFR_MU) & 1); f.opt = adam ? 0 : 1;
    f.alpha = alpha; f.b1 = o.beta1; f.b2 = o.beta2; f.eps = o.eps; f.lr = o.lr;
    if (B > 0) {
        int rc;
        tiles = tiles_eligible(m, B);
        const bool fwd_fused = fwd_in_reduce(m, B);
        // two-table form of the fused big-table step (no per-entry copy of the pre-update item rows): TFR_DUALQ=0 restores the copy
        static int dualq = -1;
        if (dualq < 0) { const char* e = getenv("TFR_DUALQ"); dualq = (e && e[0] == '0') ? 0 : 1; }
        const bool dual = fwd_fused && dualq;
        if (dual && !m->q_alt) {
            if ((rc = dmalloc(&m->q_alt, (size_t)m->n[TFR_Q])) || (rc = dmalloc(&m->q_sel, (size_t)m->I))) return rc;
            HIPCHK(hipMemsetAsync(m->q_sel, 0, (size_t)m->I * 4, s));
        }
        if (!dual && (rc = settle_q(m))) return rc;
        static int split_tiles = -1;   // TFR_TILE_SPLIT=1: the three-launch form (k_front + k_seg_reduce), kept for A/B
        if (split_tiles < 0) { const char* e = getenv("TFR_TILE_SPLIT"); split_tiles = (e && e[0] == '1') ? 1 : 0; }
        const bool one_launch = tiles && !split_tiles;
        if (!one_launch && !(presorted_big && fwd_fused))     // presorted_big: gathered + sorted ahead, on the second stream
            if ((rc = front_and_sort(m, du, di, dr, B, d_logits, d_store_ids, f, nblk, fin_done, tiles, fwd_fused))) return rc;
        if (tiles) {
            // small tables: per-tile sorted order -> piece sums per tile -> one sweep that combines a
            // row's per-tile partials, applies the optimiser to both tables and runs K4
            float* gp_rows = m->gp ? m->gp : m->gq + (size_t)m->cap * m->D;
            int par = 0;
            if (one_launch) {
                // gather + tile-local sort + forward + per-tile reduce of both sides: one launch
                if ((rc = tile_step_launch(m, du, di, dr, B, d_logits, d_store_ids, next_store_ids, gp_rows, &par, &nblk))) return rc;
                f.nblk = nblk;
            }
            RedArgs r;
            memset(&r, 0, sizeof(r));
            r.g = m->d_g; r.err = m->d_err; r.B = B; r.D = m->D; r.tile = CSORT_TILE;
            r.item_abs = o.item_abs; r.reg_bias = o.reg_bias; r.lam = o.reg;
            RedPair pr;
            pr.a[0] = r;
            pr.a[0].side = 1; pr.a[0].ks = m->ks_i; pr.a[0].ps = m->ps_i; pr.a[0].other = du;
            pr.a[0].own = m->w[TFR_Q]; pr.a[0].partner = m->w[TFR_P]; pr.a[0].own_bias = m->w[TFR_BI];
            pr.a[0].grad_rows = m->gq; pr.a[0].grad_bias = m->gbq;
            pr.a[1] = r;
            pr.a[1].side = 0; pr.a[1].ks = m->ks_u; pr.a[1].ps = m->ps_u; pr.a[1].other = di;
            pr.a[1].own = m->w[TFR_P]; pr.a[1].partner = m->w[TFR_Q]; pr.a[1].own_bias = m->w[TFR_BU];
            pr.a[1].grad_rows = gp_rows; pr.a[1].grad_bias = m->gbp;
            if (!one_launch) {
                Prof p(m, TFR_K_REDUCE_ITEM);
                launch_seg_reduce(pr, 2, RMODE_SCRATCH, m->G, m->VEC, s);
            }
            HIPCHK(hipGetLastError());
            TileDenseLaunch L;
            memset(&L, 0, sizeof(L));
            TileDenseArgs d;
            memset(&d, 0, sizeof(d));
            d.err = m->d_err; d.D = m->D; d.ntiles = (int32_t)((B + CSORT_TILE - 1) / CSORT_TILE);
            d.opt = adam ? 0 : 1; d.skip_untouched = tf1 ? 0 : 1;
            d.alpha = alpha; d.b1 = o.beta1; d.b2 = o.beta2; d.eps = o.eps; d.lr = o.lr;
            L.a[0] = d;                // items
            L.a[0].tab = par ? m->offs_i : m->hist_i; L.a[0].nbins = 1 << m->bits_i; L.a[0].rows = m->I;
            L.a[0].grad_rows = m->gq; L.a[0].grad_bias = m->gbq;
            L.a[0].w = m->w[TFR_Q]; L.a[0].m = m->m[TFR_Q]; L.a[0].v = m->v[TFR_Q];
            L.a[0].bias_w = m->w[TFR_BI]; L.a[0].bias_m = m->m[TFR_BI]; L.a[0].bias_v = m->v[TFR_BI];
            L.a[0].frozen_rows = (m->frozen >> TFR_Q) & 1; L.a[0].frozen_bias = (m->frozen >> TFR_BI) & 1;
            L.a[1] = d;                // users
            L.a[1].tab = par ? m->offs_u : m->hist_u; L.a[1].nbins = 1 << m->bits_u; L.a[1].rows = m->U;
            L.a[1].grad_rows = pr.a[1].grad_rows; L.a[1].grad_bias = m->gbp;
            L.a[1].w = m->w[TFR_P]; L.a[1].m = m->m[TFR_P]; L.a[1].v = m->v[TFR_P];
            L.a[1].bias_w = m->w[TFR_BU]; L.a[1].bias_m = m->m[TFR_BU]; L.a[1].bias_v = m->v[TFR_BU];
            L.a[1].frozen_rows = (m->frozen >> TFR_P) & 1; L.a[1].frozen_bias = (m->frozen >> TFR_BU) & 1;
            L.f = f;
            {
                Prof p(m, TFR_K_APPLY);
                launch_dense_tiles(L, false, true, m->G, m->VEC, s);
            }
            HIPCHK(hipGetLastError());
            fin_done = true;
        } else {
        RedArgs r;
        memset(&r, 0, sizeof(r));
        r.g = m->d_g; r.err = m->d_err; r.B = B; r.D = m->D;
        r.item_abs = o.item_abs; r.reg_bias = o.reg_bias;
        r.lam = o.reg; r.alpha = alpha; r.b1 = o.beta1; r.b2 = o.beta2; r.eps = o.eps; r.lr = o.lr;
        // big tables: every row of this step is touched once and cannot stay cached - non-temporal loads / stores for the
        // rows, default policy only for reading back the pre-update copies (A/B in one gpurun call, TFR_NT=<bits>:
        // 523-548 us/step with 0, 498 with 23; bits in svd_kernels.h RedArgs::nt)
        { static int nt = -1; if (nt < 0) { const char* e = getenv("TFR_NT"); nt = e ? atoi(e) : 23; } r.nt = fwd_fused ? nt : 0; }
        // item side -> scratch (reads the pre-update user rows)
        RedArgs ri = r;
        ri.side = 1;
        ri.ks = m->ks_i; ri.ps = m->ps_i; ri.other = du;
        ri.own = m->w[TFR_Q]; ri.partner = m->w[TFR_P]; ri.own_bias = m->w[TFR_BI];
        ri.grad_rows = m->gq; ri.grad_bias = m->gbq; ri.map = tf1 ? m->map_i : nullptr;
        // user side: fused update (lazy Adam / SGD) or scratch (TF1 Adam)
        RedArgs ru = r;
        ru.side = 0;
        ru.ks = m->ks_u; ru.ps = m->ps_u; ru.other = di;
        ru.own = m->w[TFR_P]; ru.partner = m->w[TFR_Q]; ru.own_bias = m->w[TFR_BU];
        ru.own_w = m->w[TFR_P]; ru.m = m->m[TFR_P]; ru.v = m->v[TFR_P];
        ru.bias_w = m->w[TFR_BU]; ru.bias_m = m->m[TFR_BU]; ru.bias_v = m->v[TFR_BU];
        ru.grad_bias = m->gbp; ru.map = tf1 ? m->map_u : nullptr;
        ru.frozen_rows = (m->frozen >> TFR_P) & 1; ru.frozen_bias = (m->frozen >> TFR_BU) & 1;
        if (tf1) {
            // both sides only read the tables: one launch
            ru.grad_rows = m->gp;
            if (m->dg_p) {
                ri.dense_rows = m->dg_q; ri.dense_bias = m->dg_bi;
                ru.dense_rows = m->dg_p; ru.dense_bias = m->dg_bu;
            }
            RedPair pr;
            pr.a[0] = ri; pr.a[1] = ru;
            Prof p(m, TFR_K_REDUCE_ITEM);
            launch_seg_reduce(pr, 2, RMODE_SCRATCH, m->G, m->VEC, s);
        } else {
            // both sides fused (reduce + lazy Adam / SGD in place).  The item side goes first and
            // leaves a per-entry copy of the pre-update Q rows (last third of gq) for the user
            // side, which then updates P; pieces of runs cut by a block boundary are parked in
            // the first (items) and second (users) third of gq and finished by k_apply_rows.
            float* qcopy = m->gq + 2 * (size_t)m->cap * m->D;
            ri.own_w = m->w[TFR_Q]; ri.m = m->m[TFR_Q]; ri.v = m->v[TFR_Q];
            ri.bias_w = m->w[TFR_BI]; ri.bias_m = m->m[TFR_BI]; ri.bias_v = m->v[TFR_BI];
            ri.frozen_rows = (m->frozen >> TFR_Q) & 1; ri.frozen_bias = (m->frozen >> TFR_BI) & 1;
            if (dual) {
                // the updated item row goes to the table the row is NOT in, so the user side still finds the pre-update row where
                // it was: no copy written (4D per rating) and none read
                ri.own_alt = m->q_alt; ri.own_w_alt = m->q_alt; ri.sel = m->q_sel; ri.osel_out = m->osel;
                ru.osel_in = m->osel; ru.partner_alt = m->q_alt;
                m->q_dirty = true;
            } else {
                ri.own_copy_out = qcopy;
                ru.partner_by_pos = qcopy;
            }
            ru.grad_rows = m->gq + (size_t)m->cap * m->D;
            if (fwd_fused) {           // K1 inside the item side: logits, g, per-block {loss, reg, sum g}
                ri.partner_bias = m->w[TFR_BU]; ri.mu = m->w[TFR_MU]; ri.r = dr; ri.loss = o.loss;
                ri.g_out = m->d_g; ri.logits_out = d_logits; ri.partials = m->partials;
                { static int st = -1; if (st < 0) { const char* e = getenv("TFR_STAGE_SUM"); st = (e && e[0] == '0') ? 0 : 1; } ri.stage_sum = st; }
                const int epb = 1024 / m->G;
                nblk = (int)((B + epb - 1) / epb);
            }
            RedPair pr;
            pr.a[0] = ri;
            {
                Prof p(m, TFR_K_REDUCE_ITEM);
                launch_seg_reduce(pr, 1, adam ? RMODE_ADAM : RMODE_SGD, m->G, m->VEC, s, fwd_fused);
            }
            HIPCHK(hipGetLastError());
            if (m->ev_mid_on) HIPCHK(hipEventRecord(m->ev_mid, s));
            pr.a[0] = ru;
            {
                Prof p(m, TFR_K_REDUCE_USER);
                launch_seg_reduce(pr, 1, adam ? RMODE_ADAM : RMODE_SGD, m->G, m->VEC, s);
            }
            HIPCHK(hipGetLastError());
            ApplyArgs ap;
            memset(&ap, 0, sizeof(ap));
            ap.err = m->d_err; ap.B = B; ap.D = m->D; ap.only_split = 1;
            ap.alpha = alpha; ap.b1 = o.beta1; ap.b2 = o.beta2; ap.eps = o.eps; ap.lr = o.lr;
            ApplyPair app;
            app.a[0] = ap;
            app.a[0].ks = m->ks_i; app.a[0].grad_rows = m->gq; app.a[0].grad_bias = m->gbq;
            app.a[0].w = m->w[TFR_Q]; app.a[0].m = m->m[TFR_Q]; app.a[0].v = m->v[TFR_Q];
            app.a[0].bias_w = m->w[TFR_BI]; app.a[0].bias_m = m->m[TFR_BI]; app.a[0].bias_v = m->v[TFR_BI];
            app.a[0].frozen_rows = ri.frozen_rows; app.a[0].frozen_bias = ri.frozen_bias;
            if (dual) { app.a[0].w_alt = m->q_alt; app.a[0].sel = m->q_sel; }
            app.a[1] = ap;
            app.a[1].ks = m->ks_u; app.a[1].grad_rows = ru.grad_rows; app.a[1].grad_bias = m->gbp;
            app.a[1].w = m->w[TFR_P]; app.a[1].m = m->m[TFR_P]; app.a[1].v = m->v[TFR_P];
            app.a[1].bias_w = m->w[TFR_BU]; app.a[1].bias_m = m->m[TFR_BU]; app.a[1].bias_v = m->v[TFR_BU];
            app.a[1].frozen_rows = ru.frozen_rows; app.a[1].frozen_bias = ru.frozen_bias;
            if (!fin_done) {           // K4 rides in the same launch (one launch and ~6 us fewer per big-table step)
                f.nblk = nblk;
                app.f = f; app.with_fin = 1;
                fin_done = true;
            }
            {
                Prof p(m, TFR_K_APPLY);
                launch_apply_rows(app, 2, adam ? 0 : 1, m->G, m->VEC, s);
            }
        }
        HIPCHK(hipGetLastError());
        }
    }
    if (tf1 && !tiles) {
        // dense sweeps: every row of every unfrozen table moves; one launch
        Prof p(m, TFR_K_APPLY);
        DenseArgs d;
        memset(&d, 0, sizeof(d));
        d.err = m->d_err; d.D = m->D; d.B = B;
        d.alpha = alpha; d.b1 = o.beta1; d.b2 = o.beta2; d.eps = o.eps;
        DensePair dp;
        dp.a[0] = d;
        dp.a[0].map = m->map_u; dp.a[0].ks = m->ks_u; dp.a[0].grad_rows = m->gp; dp.a[0].grad_bias = m->gbp;
        dp.a[0].dense_grad = m->dg_p; dp.a[0].dense_gbias = m->dg_bu;
        dp.a[0].rows = m->U;
        dp.a[0].w = m->w[TFR_P]; dp.a[0].m = m->m[TFR_P]; dp.a[0].v = m->v[TFR_P];
        dp.a[0].bias_w = m->w[TFR_BU]; dp.a[0].bias_m = m->m[TFR_BU]; dp.a[0].bias_v = m->v[TFR_BU];
        dp.a[0].frozen_rows = (m->frozen >> TFR_P) & 1; dp.a[0].frozen_bias = (m->frozen >> TFR_BU) & 1;
        dp.a[1] = d;
        dp.a[1].map = m->map_i; dp.a[1].ks = m->ks_i; dp.a[1].grad_rows = m->gq; dp.a[1].grad_bias = m->gbq;
        dp.a[1].dense_grad = m->dg_q; dp.a[1].dense_gbias = m->dg_bi;
        dp.a[1].rows = m->I;
        dp.a[1].w = m->w[TFR_Q]; dp.a[1].m = m->m[TFR_Q]; dp.a[1].v = m->v[TFR_Q];
        dp.a[1].bias_w = m->w[TFR_BI]; dp.a[1].bias_m = m->m[TFR_BI]; dp.a[1].bias_v = m->v[TFR_BI];
        dp.a[1].frozen_rows = (m->frozen >> TFR_Q) & 1; dp.a[1].frozen_bias = (m->frozen >> TFR_BI) & 1;
        // the sweep also consumes (clears) the row->slot maps, so it always runs on both tables
        launch_adam_dense(dp, 2, m->G, m->VEC, s);
        HIPCHK(hipGetLastError());
    }
    if (!fin_done) {
        f.nblk = nblk;
        Prof p(m, TFR_K_FINALIZE);
        launch_finalize(f, s);
    }
    HIPCHK(hipGetLastError());
    if (adam) {                       // beta-power accumulators advance after the applies [TF1-lib]
        m->b1p *= o.beta1;
        m->b2p *= o.beta2;
    }
    m->step += 1;
    return TFR_OK;
}

static void rollback_step(tfr_model* m, int64_t step0, float b1p0, float b2p0) {
    m->step = step0;
    m->b1p = b1p0;
    m->b2p = b2p0;
}

// pinned staging for host-fed batches up to 1M ratings (larger ones take the plain copies)
static const int64_t STAGE_MAX = 1 << 20;
static int ensure_staging(tfr_model* m, int64_t B) {
    if (B <= m->stage_cap) return TFR_OK;
    HIPCHK(hipStreamSynchronize(m->stream));
    dfree(m->d_in); m->d_in = nullptr;
    if (m->h_in) (void)hipHostFree(m->h_in);
    if (m->h_out) (void)hipHostFree(m->h_out);
    m->h_in = nullptr; m->h_out = nullptr; m->stage_cap = 0;
    int64_t cap = 1024;
    while (cap < B) cap *= 2;
    int rc;
    if ((rc = dmalloc(&m->d_in, (size_t)3 * cap))) return rc;
    HIPCHK(hipHostMalloc((void**)&m->h_in, (size_t)3 * cap * 4, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&m->h_out, (size_t)(cap + 4) * 4, hipHostMallocDefault));
    m->stage_cap = cap;
    return TFR_OK;
}

static int report_device_error(tfr_model* m, int32_t e) {
    HIPCHK(hipMemsetAsync(m->d_err, 0, sizeof(int32_t), m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    if (e & 1) return fail(TFR_ERR_OOB, "user/item id out of range [0,%lld) / [0,%lld)", (long long)m->U, (long long)m->I);
    return fail(TFR_ERR_OOB, "store index out of range [0,%lld)", (long long)m->N);
}

static int check_batch(const void* u, const void* i, int64_t B) {
    if (B < 0) return fail(TFR_ERR_ARG, "negative batch");
    if (B > 0 && (!u || !i)) return fail(TFR_ERR_ARG, "null id pointer");
    return TFR_OK;
}

extern "C" {

// ---- forward ---------------------------------------------------------------------------
int tfr_forward_dev(tfr_model* m, const int32_t* du, const int32_t* di, int64_t B, float* d_logits) {
    MODEL_ENTER(m);
    int rc = check_batch(du, di, B);
    if (rc) return rc;
    if (B == 0) return TFR_OK;
    if (!d_logits) return fail(TFR_ERR_ARG, "null logits pointer");
    if ((rc = ensure_capacity(m, 1))) return rc;
    return run_forward(m, MODE_INFER, du, di, nullptr, B, d_logits, nullptr, nullptr);
}

int tfr_forward(tfr_model* m, const int32_t* u, const int32_t* i, int64_t B, float* logits_out) {
    MODEL_ENTER(m);
    int rc = check_batch(u, i, B);
    if (rc) return rc;
    if (B == 0) return TFR_OK;
    if (!logits_out) return fail(TFR_ERR_ARG, "null logits pointer");
    if ((rc = ensure_capacity(m, B))) return rc;
    if (B <= STAGE_MAX) {                                // pinned staging: one copy in, one out (+ the error flag)
        if ((rc = ensure_staging(m, B))) return rc;
        memcpy(m->h_in, u, (size_t)B * 4);
        memcpy(m->h_in + B, i, (size_t)B * 4);
        HIPCHK(hipMemcpyAsync(m->d_in, m->h_in, (size_t)2 * B * 4, hipMemcpyHostToDevice, m->stream));
        if ((rc = run_forward(m, MODE_INFER, m->d_in, m->d_in + B, nullptr, B, m->d_logits, nullptr, nullptr))) return rc;
        HIPCHK(hipMemcpyAsync(m->h_out, m->d_logits, (size_t)B * 4, hipMemcpyDeviceToHost, m->stream));
        if ((rc = check_device_error(m))) return rc;
        memcpy(logits_out, m->h_out, (size_t)B * 4);
        return TFR_OK;
    }
    HIPCHK(hipMemcpyAsync(m->d_u, u, (size_t)B * 4, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(m->d_i, i, (size_t)B * 4, hipMemcpyHostToDevice, m->stream));
    if ((rc = run_forward(m, MODE_INFER, m->d_u, m->d_i, nullptr, B, m->d_logits, nullptr, nullptr))) return rc;
    HIPCHK(hipMemcpyAsync(logits_out, m->d_logits, (size_t)B * 4, hipMemcpyDeviceToHost, m->stream));
    return check_device_error(m);
}

// rank-sum AUC of n device scores against device labels (> 0.5 = positive); NaN when a class is empty
static int auc_device(tfr_model* m, const float* d_score, const float* d_label, int64_t n, double* auc_out) {
    int rc;
    if ((rc = ensure_capacity(m, n))) return rc;
    if (!m->d_auc) { if ((rc = dmalloc(&m->d_auc, 2))) return rc; }
    hipStream_t s = m->stream;
    HIPCHK(hipMemsetAsync(m->d_auc, 0, 16, s));
    launch_auc_keys(d_score, m->d_i, n, s);              // d_i: batch-sized int scratch
    HIPCHK(hipGetLastError());
    const int32_t* keys[2] = {m->d_i, nullptr};
    const int bits[2] = {32, 0};
    int32_t* ks[2] = {m->ks_i, nullptr};
    int32_t* ps[2] = {m->ps_i, nullptr};
    if ((rc = radix_sort_columns(m, 1, keys, bits, ks, ps, n))) return rc;
    launch_auc_ranksum(m->ks_i, m->ps_i, d_label, n, m->d_auc, s);
    HIPCHK(hipGetLastError());
    unsigned long long h[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(h, m->d_auc, 16, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const double np = (double)h[1], nn = (double)n - np;
    if (auc_out) *auc_out = (np > 0 && nn > 0) ? (0.5 * (double)h[0] - 0.5 * np * (np + 1.0)) / (np * nn) : NAN;
    return TFR_OK;
}

static int eval_device(tfr_model* m, const int32_t* du, const int32_t* di, const float* dr, int64_t B,
                       double* sse_out, int64_t* neq_out, double* nll_out = nullptr, double* auc_out = nullptr) {
    int nblk = 0, rc;
    if (auc_out && (rc = ensure_capacity(m, B))) return rc;
    if ((rc = run_forward(m, MODE_EVAL, du, di, dr, B, auc_out ? m->d_logits : nullptr, nullptr, &nblk))) return rc;
    std::vector<float> part((size_t)nblk * 4);
    HIPCHK(hipMemcpyAsync(part.data(), m->partials, part.size() * 4, hipMemcpyDeviceToHost, m->stream));
    if ((rc = check_device_error(m))) return rc;
    double sse = 0.0, nll = 0.0;
    int64_t neq = 0;
    for (int b = 0; b < nblk; ++b) {
        sse += (double)part[(size_t)b * 4 + 0];
        neq += (int64_t)llround((double)part[(size_t)b * 4 + 1]);
        nll += (double)part[(size_t)b * 4 + 2];
    }
    if (sse_out) *sse_out = sse;
    if (neq_out) *neq_out = neq;
    if (nll_out) *nll_out = nll;
    if (auc_out) return auc_device(m, m->d_logits, dr, B, auc_out);
    return TFR_OK;
}

int tfr_eval(tfr_model* m, const int32_t* u, const int32_t* i, const float* r, int64_t B,
             double* sse_out, int64_t* neq_out) {
    MODEL_ENTER(m);
    int rc = check_batch(u, i, B);
    if (rc) return rc;
    if (sse_out) *sse_out = 0.0;
    if (neq_out) *neq_out = 0;
    if (B == 0) return TFR_OK;
    if (!r) return fail(TFR_ERR_ARG, "null rate pointer");
    if ((rc = ensure_capacity(m, B))) return rc;
    HIPCHK(hipMemcpyAsync(m->d_u, u, (size_t)B * 4, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(m->d_i, i, (size_t)B * 4, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(m->d_r, r, (size_t)B * 4, hipMemcpyHostToDevice, m->stream));
    return eval_device(m, m->d_u, m->d_i, m->d_r, B, sse_out, neq_out);
}

int tfr_upload_eval_triples(tfr_model* m, const int32_t* u, const int32_t* i, const float* r, int64_t N) {
    MODEL_ENTER(m);
    if (N < 1 || !u || !i || !r) return fail(TFR_ERR_ARG, "upload_eval_triples: need n >= 1 and non-null columns");
    HIPCHK(hipStreamSynchronize(m->stream));
    dfree(m->ev_u); dfree(m->ev_i); dfree(m->ev_r);
    m->ev_u = m->ev_i = nullptr; m->ev_r = nullptr; m->ev_n = 0;
    int rc;
    if ((rc = dmalloc(&m->ev_u, (size_t)N))) return rc;
    if ((rc = dmalloc(&m->ev_i, (size_t)N))) return rc;
    if ((rc = dmalloc(&m->ev_r, (size_t)N))) return rc;
    HIPCHK(hipMemcpyAsync(m->ev_u, u, (size_t)N * 4, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(m->ev_i, i, (size_t)N * 4, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(m->ev_r, r, (size_t)N * 4, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    m->ev_n = N;
    if ((rc = ensure_capacity(m, 1))) return rc;
    return TFR_OK;
}

/* the fork's epoch line on the device (svd_train_val.py:94-98,170-178): accuracy count, summed sigmoid
 * cross-entropy and AUC of the given batch / the resident validation set */
int tfr_eval_binary(tfr_model* m, const int32_t* u, const int32_t* i, const float* r, int64_t B,
                    int64_t* neq_out, double* nll_sum_out, double* auc_out) {
    MODEL_ENTER(m);
    int rc = check_batch(u, i, B);
    if (rc) return rc;
    if (neq_out) *neq_out = 0;
    if (nll_sum_out) *nll_sum_out = 0.0;
    if (auc_out) *auc_out = NAN;
    if (B == 0) return TFR_OK;
    if (!r) return fail(TFR_ERR_ARG, "null rate pointer");
    if (m->o.loss != TFR_LOSS_NLL) return fail(TFR_ERR_STATE, "eval_binary needs the binary-outcome model (loss = nll)");
    if ((rc = ensure_capacity(m, B))) return rc;
    HIPCHK(hipMemcpyAsync(m->d_u, u, (size_t)B * 4, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(m->d_i, i, (size_t)B * 4, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(m->d_r, r, (size_t)B * 4, hipMemcpyHostToDevice, m->stream));
    // (the AUC reuses d_i as key scratch - after the forward, in stream order, has read the ids)
    return eval_device(m, m->d_u, m->d_i, m->d_r, B, nullptr, neq_out, nll_sum_out, auc_out);
}

int tfr_eval_binary_resident(tfr_model* m, int64_t* neq_out, double* nll_sum_out, double* auc_out, int64_t* n_out) {
    MODEL_ENTER(m);
    if (!m->ev_n) return fail(TFR_ERR_STATE, "no resident validation set: call tfr_upload_eval_triples first");
    if (m->o.loss != TFR_LOSS_NLL) return fail(TFR_ERR_STATE, "eval_binary needs the binary-outcome model (loss = nll)");
    if (n_out) *n_out = m->ev_n;
    return eval_device(m, m->ev_u, m->ev_i, m->ev_r, m->ev_n, nullptr, neq_out, nll_sum_out, auc_out);
}

/* AUC of arbitrary device scores / labels (label > 0.5 = positive): roc_auc_score on the device */
int tfr_auc_dev(tfr_model* m, const float* d_score, const float* d_label, int64_t n, double* auc_out) {
    MODEL_ENTER(m);
    if (n < 1 || !d_score || !d_label || !auc_out) return fail(TFR_ERR_ARG, "auc_dev: bad arguments");
    return auc_device(m, d_score, d_label, n, auc_out);
}

int tfr_eval_resident(tfr_model* m, double* sse_out, int64_t* neq_out, int64_t* n_out) {
    MODEL_ENTER(m);
    if (!m->ev_n) return fail(TFR_ERR_STATE, "no resident validation set: call tfr_upload_eval_triples first");
    if (n_out) *n_out = m->ev_n;
    return eval_device(m, m->ev_u, m->ev_i, m->ev_r, m->ev_n, sse_out, neq_out);
}

// ---- one minibatch ---------------------------------------------------------------------
int tfr_train_step_dev(tfr_model* m, const int32_t* du, const int32_t* di, const float* dr, int64_t B,
                       float* d_logits) {
    MODEL_ENTER(m);
    int rc = check_batch(du, di, B);
    if (rc) return rc;
    if (B > 0 && !dr) return fail(TFR_ERR_ARG, "null rate pointer");
    if ((rc = ensure_capacity(m, B > 0 ? B : 1))) return rc;
    return run_train_step(m, du, di, dr, B, d_logits, nullptr);
}

int tfr_train_step(tfr_model* m, const int32_t* u, const int32_t* i, const float* r, int64_t B,
                   float* logits_out, float* loss_out, float* reg_out) {
    MODEL_ENTER(m);
    int rc = check_batch(u, i, B);
    if (rc) return rc;
    if (B > 0 && !r) return fail(TFR_ERR_ARG, "null rate pointer");
    if ((rc = ensure_capacity(m, B > 0 ? B : 1))) return rc;
    const int64_t step0 = m->step;
    const float b1p0 = m->b1p, b2p0 = m->b2p;
    float sc[4] = {0.f, 0.f, 0.f, 0.f};
    m->last_r = nullptr;
    if (B > 0 && B <= STAGE_MAX) {
        // one pinned H2D copy in, the kernels, one D2H copy out: [logits | loss, reg, sum g, error flag]
        if ((rc = ensure_staging(m, B))) return rc;
        memcpy(m->h_in, u, (size_t)B * 4);
        memcpy(m->h_in + B, i, (size_t)B * 4);
        memcpy(m->h_in + 2 * B, r, (size_t)B * 4);
        HIPCHK(hipMemcpyAsync(m->d_in, m->h_in, (size_t)3 * B * 4, hipMemcpyHostToDevice, m->stream));
        const int64_t nl = logits_out ? B : 0;
        float* out4 = m->d_logits + nl;
        if ((rc = run_train_step(m, m->d_in, m->d_in + B, reinterpret_cast<const float*>(m->d_in + 2 * B), B,
                                 logits_out ? m->d_logits : nullptr, out4, nullptr, nullptr, false, true))) return rc;
        HIPCHK(hipMemcpyAsync(m->h_out, m->d_logits, (size_t)(nl + 4) * 4, hipMemcpyDeviceToHost, m->stream));
        HIPCHK(hipStreamSynchronize(m->stream));
        const int32_t e = (int32_t)m->h_out[nl + 3];
        if (e) {                                         // a bad batch never advances the step
            rollback_step(m, step0, b1p0, b2p0);
            return report_device_error(m, e);
        }
        if (logits_out) memcpy(logits_out, m->h_out, (size_t)B * 4);
        sc[0] = m->h_out[nl]; sc[1] = m->h_out[nl + 1];
        m->last_r = logits_out ? reinterpret_cast<const float*>(m->d_in + 2 * B) : nullptr; m->last_B = B;
    } else {
        if (B > 0) {
            HIPCHK(hipMemcpyAsync(m->d_u, u, (size_t)B * 4, hipMemcpyHostToDevice, m->stream));
            HIPCHK(hipMemcpyAsync(m->d_i, i, (size_t)B * 4, hipMemcpyHostToDevice, m->stream));
            HIPCHK(hipMemcpyAsync(m->d_r, r, (size_t)B * 4, hipMemcpyHostToDevice, m->stream));
        }
        if ((rc = run_train_step(m, m->d_u, m->d_i, m->d_r, B, logits_out ? m->d_logits : nullptr, nullptr))) return rc;
        if (logits_out && B > 0)
            HIPCHK(hipMemcpyAsync(logits_out, m->d_logits, (size_t)B * 4, hipMemcpyDeviceToHost, m->stream));
        HIPCHK(hipMemcpyAsync(sc, m->scalars, 16, hipMemcpyDeviceToHost, m->stream));
        // synchronous entry point: always validate so a bad batch never advances the step
        if ((rc = check_device_error(m))) {
            rollback_step(m, step0, b1p0, b2p0);
            return rc;
        }
    }
    if (loss_out) *loss_out = sc[0];
    if (reg_out) *reg_out = sc[1];
    return TFR_OK;
}

int tfr_train_steps_repeat(tfr_model* m, const int32_t* u, const int32_t* i, const float* r, int64_t B, int32_t nsteps,
                           float* logits_out, float* loss_out) {
    MODEL_ENTER(m);
    int rc = check_batch(u, i, B);
    if (rc) return rc;
    if (B < 1 || !r) return fail(TFR_ERR_ARG, "train_steps_repeat: need a batch of at least one rating");
    if (nsteps < 0) return fail(TFR_ERR_ARG, "bad nsteps");
    if (nsteps == 0) return TFR_OK;
    if ((rc = ensure_capacity(m, B))) return rc;
    if (loss_out && (rc = ensure_step_out(m, nsteps))) return rc;
    const int64_t step0 = m->step;
    const float b1p0 = m->b1p, b2p0 = m->b2p;
    m->last_r = nullptr;
    HIPCHK(hipMemcpyAsync(m->d_u, u, (size_t)B * 4, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(m->d_i, i, (size_t)B * 4, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(m->d_r, r, (size_t)B * 4, hipMemcpyHostToDevice, m->stream));
    for (int32_t s = 0; s < nsteps; ++s) {
        const bool last = s + 1 == nsteps;
        if ((rc = run_train_step(m, m->d_u, m->d_i, m->d_r, B, (last && logits_out) ? m->d_logits : nullptr,
                                 loss_out ? m->step_out + (size_t)s * 4 : nullptr))) {
            (void)hipStreamSynchronize(m->stream);
            rollback_step(m, step0, b1p0, b2p0);
            return rc;
        }
    }
    std::vector<float> tmp;
    if (loss_out) {
        tmp.resize((size_t)nsteps * 4);
        HIPCHK(hipMemcpyAsync(tmp.data(), m->step_out, tmp.size() * 4, hipMemcpyDeviceToHost, m->stream));
    }
    if (logits_out) HIPCHK(hipMemcpyAsync(logits_out, m->d_logits, (size_t)B * 4, hipMemcpyDeviceToHost, m->stream));
    if ((rc = check_device_error(m))) {                    // a bad id voids every step of the call
        rollback_step(m, step0, b1p0, b2p0);
        return rc;
    }
    for (int32_t s = 0; loss_out && s < nsteps; ++s) loss_out[s] = tmp[(size_t)s * 4];
    if (logits_out) { m->last_r = m->d_r; m->last_B = B; }
    return TFR_OK;
}

/* roc_auc_score(rates, sigmoid(logits)) of the batch the last tfr_train_step ran on (its pre-update logits, the ones
 * the caller was handed): svd_train_val.py:97, without sklearn on the host.  Needs that call to have asked for logits. */
int tfr_last_batch_auc(tfr_model* m, double* auc_out) {
    MODEL_ENTER(m);
    if (!auc_out) return fail(TFR_ERR_ARG, "null output");
    if (!m->last_r || m->last_B < 1) return fail(TFR_ERR_STATE, "no kept batch: call tfr_train_step with logits_out first (batches up to 2^20)");
    return auc_device(m, m->d_logits, m->last_r, m->last_B, auc_out);
}

// ---- resident store --------------------------------------------------------------------
// pack three columns (host or device) into the 16-byte-record store, in chunks
static int build_store(tfr_model* m, const int32_t* u, const int32_t* i, const float* r, int64_t N, bool on_device) {
    HIPCHK(hipStreamSynchronize(m->stream));
    if (m->spec_valid) { int rc0 = cancel_run_ahead(m); if (rc0) return rc0; }     // ids drawn ahead were for the old store size
    dfree(m->store);
    m->store = nullptr; m->N = 0; m->pf_valid = false;
    recs_forget(m);
    int rc;
    if ((rc = dmalloc(&m->store, (size_t)N))) return rc;
    if (on_device) {
        launch_pack_triples(u, i, r, m->store, N, m->stream);
        HIPCHK(hipGetLastError());
    } else {
        const int64_t chunk = (int64_t)1 << 24;
        int32_t *tu = nullptr, *ti = nullptr;
        float* tr = nullptr;
        const int64_t c0 = N < chunk ? N : chunk;
        if ((rc = dmalloc(&tu, (size_t)c0)) || (rc = dmalloc(&ti, (size_t)c0)) || (rc = dmalloc(&tr, (size_t)c0))) {
            dfree(tu); dfree(ti); dfree(tr);
            return rc;
        }
        hipError_t e = hipSuccess;
        for (int64_t off = 0; off < N && e == hipSuccess; off += chunk) {
            const int64_t n = (N - off < chunk) ? N - off : chunk;
            e = hipMemcpyAsync(tu, u + off, (size_t)n * 4, hipMemcpyHostToDevice, m->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(ti, i + off, (size_t)n * 4, hipMemcpyHostToDevice, m->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(tr, r + off, (size_t)n * 4, hipMemcpyHostToDevice, m->stream);
            if (e == hipSuccess) {
                launch_pack_triples(tu, ti, tr, m->store + off, n, m->stream);
                e = hipStreamSynchronize(m->stream);
            }
        }
        dfree(tu); dfree(ti); dfree(tr);
        if (e != hipSuccess) return fail(TFR_ERR_HIP, "upload_triples: %s", hipGetErrorString(e));
    }
    HIPCHK(hipStreamSynchronize(m->stream));
    m->N = N;
    return TFR_OK;
}

int tfr_upload_triples(tfr_model* m, const int32_t* u, const int32_t* i, const float* r, int64_t N) {
    MODEL_ENTER(m);
    if (N < 1 || !u || !i || !r) return fail(TFR_ERR_ARG, "upload_triples: need n >= 1 and non-null columns");
    return build_store(m, u, i, r, N, false);
}

int tfr_set_triples_dev(tfr_model* m, const int32_t* du, const int32_t* di, const float* dr, int64_t N) {
    MODEL_ENTER(m);
    if (N < 1 || !du || !di || !dr) return fail(TFR_ERR_ARG, "set_triples_dev: need n >= 1 and non-null columns");
    return build_store(m, du, di, dr, N, true);
}

int tfr_init_tables(tfr_model* m, uint64_t seed, float fstd, float bstd) {
    MODEL_ENTER(m);
    hipStream_t s = m->stream;
    if (m->q_dirty) {                                    // fresh tables: every row lives in the main table again
        HIPCHK(hipMemsetAsync(m->q_sel, 0, (size_t)m->I * 4, s));
        m->q_dirty = false;
    }
    launch_init_trunc_normal(m->w[TFR_P], m->n[TFR_P], fstd, seed * 4 + 0, s);
    launch_init_trunc_normal(m->w[TFR_Q], m->n[TFR_Q], fstd, seed * 4 + 1, s);
    launch_init_trunc_normal(m->w[TFR_BU], m->n[TFR_BU], bstd, seed * 4 + 2, s);
    launch_init_trunc_normal(m->w[TFR_BI], m->n[TFR_BI], bstd, seed * 4 + 3, s);
    launch_init_uniform_scalar(m->w[TFR_MU], -1.7320508f, 1.7320508f, seed, s);
    HIPCHK(hipGetLastError());
    for (int t = 0; t < 5; ++t) {
        if (m->m[t]) HIPCHK(hipMemsetAsync(m->m[t], 0, (size_t)m->n[t] * 4, s));
        if (m->v[t]) HIPCHK(hipMemsetAsync(m->v[t], 0, (size_t)m->n[t] * 4, s));
    }
    m->step = 0;
    m->b1p = m->o.beta1;
    m->b2p = m->o.beta2;
    HIPCHK(hipStreamSynchronize(s));
    return TFR_OK;
}

static const int64_t IDS_MIN_CAP = (int64_t)1 << 24;   // 128 MB: a 900-step call at batch 10000 fits without reallocating
// room for n staged ids (contents undefined afterwards); the stream must be idle
static int ensure_ids(tfr_model* m, int64_t n) {
    m->n_ids = 0; m->pf_valid = false;
    if (n > m->d_ids_cap) {
        if (m->stream3) HIPCHK(hipStreamSynchronize(m->stream3));
        dfree(m->d_ids);
        m->d_ids = nullptr; m->d_ids_cap = 0;
        int rc;
        const int64_t want = n < IDS_MIN_CAP ? IDS_MIN_CAP : n;      // allocations are slow: never size for a short call only
        if ((rc = dmalloc(&m->d_ids, (size_t)want))) return rc;
        m->d_ids_cap = want;
    }
    for (int k = 0; k < 2; ++k) if (m->rb[k].ids_base == m->d_ids) m->rb[k].n = 0;      // (a new buffer may sit at an old one's address)
    return TFR_OK;
}

int tfr_stage_ids(tfr_model* m, const int64_t* ids, int64_t n) {
    MODEL_ENTER(m);
    if (n < 1 || !ids) return fail(TFR_ERR_ARG, "stage_ids: need n >= 1 and non-null ids");
    HIPCHK(hipStreamSynchronize(m->stream));
    int rc;
    if ((rc = ensure_ids(m, n))) return rc;
    HIPCHK(hipMemcpyAsync(m->d_ids, ids, (size_t)n * 8, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    for (int k = 0; k < 2; ++k) if (m->rb[k].ids_base == m->d_ids) m->rb[k].n = 0;
    m->n_ids = n;
    return TFR_OK;
}

static int gather_batch(tfr_model* m, const int64_t* d_ids, int64_t lo, int64_t B) {
    GatherArgs g;
    g.ids = d_ids; g.lo = lo; g.B = B; g.N = m->N;
    g.store = m->store;
    g.u = m->d_u; g.it = m->d_i; g.r = m->d_r; g.err = m->d_err;
    {
        Prof p(m, TFR_K_GATHER);
        launch_gather(g, m->stream);
    }
    HIPCHK(hipGetLastError());
    return TFR_OK;
}

static void swap_sortset(tfr_model* m) {
    std::swap(m->d_u, m->alt.d_u); std::swap(m->d_i, m->alt.d_i); std::swap(m->d_r, m->alt.d_r);
    std::swap(m->ks_u, m->alt.ks_u); std::swap(m->ps_u, m->alt.ps_u);
    std::swap(m->ks_i, m->alt.ks_i); std::swap(m->ps_i, m->alt.ps_i);
}

static int ensure_lookahead(tfr_model* m) {
    int rc;
    if (m->alt_cap < m->cap) {
        HIPCHK(hipStreamSynchronize(m->stream));
        dfree(m->alt.d_u); dfree(m->alt.d_i); dfree(m->alt.d_r); dfree(m->alt.ks_u); dfree(m->alt.ps_u); dfree(m->alt.ks_i); dfree(m->alt.ps_i);
        m->alt = tfr_model::SortSet(); m->alt_cap = 0;
        if ((rc = dmalloc(&m->alt.d_u, m->cap)) || (rc = dmalloc(&m->alt.d_i, m->cap)) || (rc = dmalloc(&m->alt.d_r, m->cap)) ||
            (rc = dmalloc(&m->alt.ks_u, m->cap)) || (rc = dmalloc(&m->alt.ps_u, m->cap)) ||
            (rc = dmalloc(&m->alt.ks_i, m->cap)) || (rc = dmalloc(&m->alt.ps_i, m->cap))) return rc;
        m->alt_cap = m->cap;
    }
    if (!m->stream2) {
        // (tried, one gpurun call each, C3 step: a high-priority look-ahead stream - no change; the look-ahead stream
        // confined to 8 / 16 / 32 / 64 CUs by hipExtStreamCreateWithCUMask - 1070 / 717 / 584 / 487 us against 497-504)
        HIPCHK(hipStreamCreateWithFlags(&m->stream2, hipStreamNonBlocking));
        for (int z = 0; z < 2; ++z) {
            HIPCHK(hipEventCreateWithFlags(&m->ev_sorted[z], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&m->ev_free[z], hipEventDisableTiming));
        }
        HIPCHK(hipEventCreateWithFlags(&m->ev_first, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&m->ev_mid, hipEventDisableTiming));
    }
    return TFR_OK;
}

// Big tables, touched-rows optimiser: gather + radix sort (+ id range check) of batch s+1 do not depend
// on the tables, are a few small latency-bound launches, and would otherwise head every step; they run
// on a second stream into the alternate buffer set while step s's HBM-bound kernels own the CUs.
// mask of legacy randint's rejection loop: smallest 2^k - 1 >= rng
static uint32_t mask_for(uint32_t rng) {
    uint32_t mk = rng;
    mk |= mk >> 1; mk |= mk >> 2; mk |= mk >> 4; mk |= mk >> 8; mk |= mk >> 16;
    return mk;
}

// Steps whose ids are drawn on the side stream (tfr_train_steps_drawn).  The draws are cut into chunks of whole steps -
// small ones first, so the first steps start after two batches' worth of draws, then doubling while the generator's
// lead over the steps allows it - and a chunk is ENQUEUED when the step loop needs it (need) or one at a time after a step's
// own launches (feed): the host never spends a stretch feeding the draw stream while the main stream sits empty (a 20-step
// call: the first step used to start 36 us into the call, behind six draw launches).
// need(step, stream) makes `stream` wait for the chunk that holds that step's ids.  NULL = ids staged by the host.
static int enqueue_run_ahead(tfr_model* m, int64_t B, int64_t nsteps, uint32_t rng);

struct IdsReady {
    tfr_model* m = nullptr;
    int64_t B = 0, nsteps = 0;
    uint32_t rng = 0;
    bool ahead_done = false;
    std::vector<int64_t> first;                            // first[c] = first step of chunk c; first[nchunks] = nsteps
    int enq = 0;                                           // chunks enqueued so far
    int waited[2] = {-1, -1};                              // highest chunk waited for: [0] main stream, [1] stream2
    int64_t pre = 0;                                       // pre: steps whose ids the previous call drew ahead (chunk 0, event spec_ev)
    int chunk_of(int64_t step) const {
        int c = 0;
        while (c + 1 < (int)first.size() - 1 && first[c + 1] <= step) ++c;
        return c;
    }
    void plan(int64_t cap_steps, int64_t pre_steps) {
        first.clear();
        pre = pre_steps;
        int64_t s = 0, n = pre ? pre : (nsteps < 2 ? nsteps : 2), done = 0;
        enq = pre ? 1 : 0;
        while (s < nsteps) {
            first.push_back(s);
            s += n; done += n;
            // next size: 1 while little has been drawn, then roughly half of what is already behind (the generator is at
            // most ~1.5x faster than a small-table step, so its lead grows by about a third of a step per step)
            n = done / 3;
            if (n < 1) n = 1;
            if (n > cap_steps) n = cap_steps;
            if (s + n > nsteps) n = nsteps - s;
        }
        first.push_back(nsteps);
    }
    // every chunk that starts at or before `step`, and at most `extra` more
    int enqueue_through(int64_t step, int extra = 0) {
        const int nch = (int)first.size() - 1;
        while (enq < nch && (first[enq] <= step || extra-- > 0)) {
            const int c = enq++;
            while ((int)m->chunk_ev.size() <= c) {
                hipEvent_t e;
                if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(TFR_ERR_HIP, "hipEventCreate failed");
                m->chunk_ev.push_back(e);
            }
            const int64_t s0 = first[c], s1 = first[c + 1];
            hipEvent_t pa = nullptr, pb = nullptr;
            if (m->prof && hipEventCreate(&pa) == hipSuccess && hipEventCreate(&pb) == hipSuccess) (void)hipEventRecord(pa, m->stream3);
            if (rng != 0) {
                launch_mt_draw(m->d_rng, m->d_ids + s0 * B, (s1 - s0) * B, rng, mask_for(rng), m->stream3, nullptr, &m->rng_ws);
                recs_follow(m, m->d_ids, m->d_ids_cap, s0 * B, (s1 - s0) * B, m->stream3);
            }
            if (pa && pb) { (void)hipEventRecord(pb, m->stream3); m->events.push_back({pa, pb, TFR_K_DRAW}); }
            if (hipGetLastError() != hipSuccess || hipEventRecord(m->chunk_ev[c], m->stream3) != hipSuccess)
                return fail(TFR_ERR_HIP, "draw launch failed");
        }
        return TFR_OK;
    }
    // after a step's launches: one more chunk for the draw stream (never before them - with ids drawn ahead by the previous
    // call the first step must not queue behind draw launches it does not need; one per step keeps the host ahead of a
    // 20-us step and the generator busy from the start, so the run-ahead draw at the end of the call fits inside it)
    int feed(int64_t step) { return enqueue_through(step, 1); }
    int run_ahead() {
        if (ahead_done) return TFR_OK;
        ahead_done = true;
        return enqueue_run_ahead(m, B, nsteps, rng);
    }
    int need(int64_t step, hipStream_t st, int which) {
        if (step >= nsteps) step = nsteps - 1;
        int rc = enqueue_through(step);
        if (rc) return rc;
        const int c = chunk_of(step);
        if (c > waited[which]) {                           // chunks complete in order on the draw stream
            if (hipStreamWaitEvent(st, (c == 0 && pre) ? m->spec_ev : m->chunk_ev[c], 0) != hipSuccess)
                return fail(TFR_ERR_HIP, "hipStreamWaitEvent failed");
            waited[which] = c;                             // (skipping the wait when hipEventQuery says "done": no gain, A/B)
        }
        return TFR_OK;
    }
};

static int staged_steps_lookahead(tfr_model* m, int64_t first_step, int64_t B, int32_t nsteps, float* loss_out, IdsReady* ready) {
    int rc;
    if ((rc = ensure_lookahead(m))) return rc;
    hipStream_t main_s = m->stream;
    FinArgs fdummy;
    memset(&fdummy, 0, sizeof(fdummy));
    int nb0 = 0;
    bool fd0 = false;
    auto sort_batch = [&](int64_t step) -> int {           // into the buffer set the model currently points at
        const int32_t* du = m->d_u; const int32_t* di = m->d_i; const float* dr = m->d_r;
        if (ready) { const int e = ready->need(first_step + step, m->stream, m->stream == main_s ? 0 : 1); if (e) return e; }
        return front_and_sort(m, du, di, dr, B, nullptr, m->d_ids + (first_step + step) * B, fdummy, nb0, fd0, false, true);
    };
    if ((rc = sort_batch(0))) return rc;
    HIPCHK(hipEventRecord(m->ev_first, main_s));
    // the next batch's sort chain starts beside the user-side kernel of this step (after the item-side one, which runs the
    // forward and suffers more from company): three A/B pairs in one gpurun call, 494/519/537 -> 489/512/523 us per step.
    // TFR_SORT_LATE=0: start it beside the item-side kernel (the round-1 order), kept for A/B
    static int late = -1;
    if (late < 0) { const char* e = getenv("TFR_SORT_LATE"); late = (e && e[0] == '0') ? 0 : 1; }
    if (late) {
        m->ev_mid_on = true;
        for (int32_t s = 0; s < nsteps && !rc; ++s) {
            const int z = s & 1;
            if (s > 0) HIPCHK(hipStreamWaitEvent(main_s, m->ev_sorted[z], 0));
            if ((rc = run_train_step(m, m->d_u, m->d_i, m->d_r, B, nullptr, loss_out ? m->step_out + (size_t)s * 4 : nullptr,
                                     m->d_ids + (first_step + s) * B, nullptr, true)))
                break;
            HIPCHK(hipEventRecord(m->ev_free[z], main_s));
            if (ready && (rc = ready->feed(first_step + s))) break;
            if (s + 1 < nsteps) {
                swap_sortset(m);                           // the next step's set: free since step s-1, which the main stream has passed
                HIPCHK(hipStreamWaitEvent(m->stream2, m->ev_mid, 0));
                m->stream = m->stream2;
                rc = sort_batch(s + 1);
                m->stream = main_s;
                if (!rc) rc = hipEventRecord(m->ev_sorted[z ^ 1], m->stream2) == hipSuccess ? TFR_OK : fail(TFR_ERR_HIP, "event record");
            }
        }
        m->ev_mid_on = false;
        HIPCHK(hipStreamSynchronize(m->stream2));
        return rc;
    }
    for (int32_t s = 0; s < nsteps; ++s) {
        const int z = s & 1;
        if (s + 1 < nsteps) {
            swap_sortset(m);                               // the other set: target of the look-ahead sort
            if (s == 0) HIPCHK(hipStreamWaitEvent(m->stream2, m->ev_first, 0));      // sort scratch is shared
            else HIPCHK(hipStreamWaitEvent(m->stream2, m->ev_free[z ^ 1], 0));       // step s-1 has finished with this set
            m->stream = m->stream2;
            rc = sort_batch(s + 1);
            m->stream = main_s;
            if (!rc) rc = hipEventRecord(m->ev_sorted[z ^ 1], m->stream2) == hipSuccess ? TFR_OK : fail(TFR_ERR_HIP, "event record");
            swap_sortset(m);
            if (rc) return rc;
        }
        if (s > 0) HIPCHK(hipStreamWaitEvent(main_s, m->ev_sorted[z], 0));
        if ((rc = run_train_step(m, m->d_u, m->d_i, m->d_r, B, nullptr, loss_out ? m->step_out + (size_t)s * 4 : nullptr,
                                 m->d_ids + (first_step + s) * B, nullptr, true)))
            return rc;
        HIPCHK(hipEventRecord(m->ev_free[z], main_s));
        if (ready && (rc = ready->feed(first_step + s))) return rc;
        if (s + 1 < nsteps) swap_sortset(m);               // the next step's batch lives in the other set
    }
    HIPCHK(hipStreamSynchronize(m->stream2));              // nothing of ours is left in flight on the side stream
    return TFR_OK;
}

static int staged_steps(tfr_model* m, int64_t first_step, int64_t B, int32_t nsteps, float* loss_out, IdsReady* ready = nullptr) {
    int rc;
    if ((rc = ensure_capacity(m, B))) return rc;
    if (loss_out && (rc = ensure_step_out(m, nsteps))) return rc;
    const int64_t step0 = m->step;
    const float b1p0 = m->b1p, b2p0 = m->b2p;
    static int no_ahead = -1;                              // TFR_NO_LOOKAHEAD=1: A/B switch
    if (no_ahead < 0) { const char* e = getenv("TFR_NO_LOOKAHEAD"); no_ahead = (e && e[0] == '1') ? 1 : 0; }
    if (nsteps > 1 && fwd_in_reduce(m, B) && !m->prof && !no_ahead) {
        if ((rc = staged_steps_lookahead(m, first_step, B, nsteps, loss_out, ready))) {
            (void)hipStreamSynchronize(m->stream2);
            return rc;
        }
    } else {
    for (int32_t s = 0; s < nsteps; ++s) {
        if (ready && (rc = ready->need(first_step + s + ((first_step + s + 2) * B <= m->n_ids ? 1 : 0), m->stream, 0))) return rc;
        // look ahead past the end of this call too when more staged batches follow: the
        // next call then starts presorted (the sort is free, hidden in this launch)
        const int64_t* nxt = (first_step + s + 2) * B <= m->n_ids ? m->d_ids + (first_step + s + 1) * B : nullptr;
        if ((rc = run_train_step(m, m->d_u, m->d_i, m->d_r, B, nullptr,
                                 loss_out ? m->step_out + (size_t)s * 4 : nullptr,
                                 m->d_ids + (first_step + s) * B, nxt))) {
            m->pf_valid = false;
            return rc;
        }
        if (ready && (rc = ready->feed(first_step + s))) return rc;
    }
    }
    if (loss_out) {
        std::vector<float> tmp((size_t)nsteps * 4);
        HIPCHK(hipMemcpyAsync(tmp.data(), m->step_out, tmp.size() * 4, hipMemcpyDeviceToHost, m->stream));
        if ((rc = check_device_error(m))) {
            rollback_step(m, step0, b1p0, b2p0);
            m->pf_valid = false;
            return rc;
        }
        for (int32_t s = 0; s < nsteps; ++s) loss_out[s] = tmp[(size_t)s * 4];
    }
    return TFR_OK;
}

int tfr_train_steps_staged(tfr_model* m, int64_t first_step, int64_t B, int32_t nsteps, float* loss_out) {
    MODEL_ENTER(m);
    if (!m->N) return fail(TFR_ERR_STATE, "no resident triples: call tfr_upload_triples first");
    if (!m->d_ids) return fail(TFR_ERR_STATE, "no staged ids: call tfr_stage_ids first");
    if (B < 1 || nsteps < 0 || first_step < 0) return fail(TFR_ERR_ARG, "bad batch/nsteps/first_step");
    if ((first_step + nsteps) * B > m->n_ids)
        return fail(TFR_ERR_ARG, "steps [%lld,%lld) x batch %lld exceed the %lld staged ids", (long long)first_step,
                    (long long)(first_step + nsteps), (long long)B, (long long)m->n_ids);
    CallTrace tr("steps_staged");
    const int rc = staged_steps(m, first_step, B, nsteps, loss_out);
    tr.mark("all steps enqueued");
    return rc;
}

int tfr_train_steps_resident(tfr_model* m, const int64_t* ids, int64_t B, int32_t nsteps, float* loss_out) {
    MODEL_ENTER(m);
    if (!m->N) return fail(TFR_ERR_STATE, "no resident triples: call tfr_upload_triples first");
    if (B < 1 || nsteps < 0 || !ids) return fail(TFR_ERR_ARG, "bad batch/nsteps/ids");
    if (nsteps == 0) return TFR_OK;
    int rc = tfr_stage_ids(m, ids, B * nsteps);
    if (rc) return rc;
    return staged_steps(m, 0, B, nsteps, loss_out);
}

// ---- device id draw (rng.hip) ------------------------------------------------------------
static int ensure_rng(tfr_model* m) {
    if (!m->d_rng) {
        int rc;
        if ((rc = dmalloc(&m->d_rng, 625))) return rc;
        if ((rc = dmalloc(&m->d_rng_snap, 625))) return rc;
        // scratch of the wide draw; TFR_RNG_WIDE=0 keeps every draw on the one-workgroup kernel (A/B)
        const char* e = getenv("TFR_RNG_WIDE");
        if (!(e && e[0] == '0')) {
            if ((rc = dmalloc(&m->rng_ws.raw, (size_t)MT_WIDE_BLOCKS * 624))) return rc;
            if ((rc = dmalloc(&m->rng_ws.counts, (size_t)MT_WIDE_BLOCKS))) return rc;
            if ((rc = dmalloc(&m->rng_ws.hdr, 4))) return rc;
            m->rng_ws.cap_blocks = MT_WIDE_BLOCKS;
        }
    }
    if (!m->stream3) {
        HIPCHK(hipStreamCreateWithFlags(&m->stream3, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&m->ev_ids_free, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&m->spec_ev, hipEventDisableTiming));
    }
    return TFR_OK;
}

// forget the ids drawn ahead for a call that is not coming: the generator returns to where the consumed ids end
static int cancel_run_ahead(tfr_model* m) {
    if (m->spec_valid) {
        m->spec_valid = false;
        HIPCHK(hipMemcpyAsync(m->d_rng, m->d_rng_snap, 625 * 4, hipMemcpyDeviceToDevice, m->stream3));
    }
    return TFR_OK;
}

int tfr_rng_set_state(tfr_model* m, const uint32_t* key, int32_t pos) {
    MODEL_ENTER(m);
    if (!key || pos < 0 || pos > 624) return fail(TFR_ERR_ARG, "rng_set_state: need key[624] and pos in [0, 624]");
    int rc;
    if ((rc = ensure_rng(m))) return rc;
    m->spec_valid = false;                                 // the new state replaces whatever was drawn ahead
    HIPCHK(hipStreamSynchronize(m->stream3));
    uint32_t h[625];
    memcpy(h, key, 624 * 4);
    h[624] = (uint32_t)pos;
    HIPCHK(hipMemcpy(m->d_rng, h, sizeof(h), hipMemcpyHostToDevice));
    m->rng_set = true;
    return TFR_OK;
}

int tfr_rng_seed(tfr_model* m, uint32_t seed) {
    // init_genrand of MT19937 = what np.random.seed(int) does [NumPy-lib: _legacy_seeding -> mt19937_seed]
    uint32_t key[624];
    key[0] = seed;
    for (int i = 1; i < 624; ++i) key[i] = 1812433253u * (key[i - 1] ^ (key[i - 1] >> 30)) + (uint32_t)i;
    return tfr_rng_set_state(m, key, 624);
}

int tfr_rng_get_state(tfr_model* m, uint32_t* key, int32_t* pos) {
    MODEL_ENTER(m);
    if (!m->rng_set) return fail(TFR_ERR_STATE, "no generator state: call tfr_rng_seed / tfr_rng_set_state first");
    int rc;
    if ((rc = cancel_run_ahead(m))) return rc;
    HIPCHK(hipStreamSynchronize(m->stream3));
    uint32_t h[625];
    HIPCHK(hipMemcpy(h, m->d_rng, sizeof(h), hipMemcpyDeviceToHost));
    if (key) memcpy(key, h, 624 * 4);
    if (pos) *pos = (int32_t)h[624];
    return TFR_OK;
}

static int check_high(int64_t high) {
    if (high < 1 || high > ((int64_t)1 << 32))
        return fail(TFR_ERR_ARG, "randint(0, high): high must be in [1, 2^32] (NumPy draws 64-bit words beyond that)");
    return TFR_OK;
}

int tfr_draw_ids(tfr_model* m, int64_t high, int64_t count, int64_t* ids_out) {
    MODEL_ENTER(m);
    int rc;
    if ((rc = check_high(high))) return rc;
    if (count < 0 || (count > 0 && !ids_out)) return fail(TFR_ERR_ARG, "draw_ids: bad count / null output");
    if (!m->rng_set) return fail(TFR_ERR_STATE, "no generator state: call tfr_rng_seed / tfr_rng_set_state first");
    if (count == 0) return TFR_OK;
    if (high == 1) { memset(ids_out, 0, (size_t)count * 8); return TFR_OK; }    // rng == 0: no draw is consumed
    if ((rc = cancel_run_ahead(m))) return rc;
    int64_t* d = nullptr;
    if ((rc = dmalloc(&d, (size_t)count))) return rc;
    const uint32_t rng = (uint32_t)(high - 1);
    unsigned long long* dbg = nullptr;
    if (getenv("TFR_RNG_DEBUG")) (void)hipMalloc((void**)&dbg, 16);           // diagnostic: in-kernel clock of the generator
    launch_mt_draw(m->d_rng, d, count, rng, mask_for(rng), m->stream3, dbg, &m->rng_ws);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(ids_out, d, (size_t)count * 8, hipMemcpyDeviceToHost, m->stream3);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream3);
    if (dbg && e == hipSuccess) {
        unsigned long long h[2] = {0, 0};
        (void)hipMemcpy(h, dbg, 16, hipMemcpyDeviceToHost);
        fprintf(stderr, "k_mt_draw: %lld ids, %llu shader cycles, %.1f us, %.0f MHz, %.1f cycles per 624-word block (mask %u, rng %u)\n",
                (long long)count, h[0], h[1] / 100.0, h[1] ? h[0] * 100.0 / h[1] : 0.0,
                h[0] / ((double)count * ((double)mask_for(rng) + 1.0) / ((double)rng + 1.0) / 624.0), mask_for(rng), rng);
    }
    dfree(dbg);
    dfree(d);
    if (e != hipSuccess) return fail(TFR_ERR_HIP, "draw_ids: %s", hipGetErrorString(e));
    return TFR_OK;
}

int tfr_draw_ids_dev(tfr_model* m, int64_t high, int64_t count, int64_t* d_ids_out) {
    MODEL_ENTER(m);
    int rc;
    if ((rc = check_high(high))) return rc;
    if (count < 0 || (count > 0 && !d_ids_out)) return fail(TFR_ERR_ARG, "draw_ids_dev: bad count / null output");
    if (!m->rng_set) return fail(TFR_ERR_STATE, "no generator state: call tfr_rng_seed / tfr_rng_set_state first");
    if (count == 0) return TFR_OK;
    if ((rc = cancel_run_ahead(m))) return rc;
    hipEvent_t& dev = m->draw_evs[m->draw_count % tfr_model::DRAW_RING];
    if (!dev) HIPCHK(hipEventCreateWithFlags(&dev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(m->ev_ids_free, m->stream));     // the buffer's readers queued so far
    HIPCHK(hipStreamWaitEvent(m->stream3, m->ev_ids_free, 0));
    if (high == 1) {                                       // rng == 0: no draw is consumed
        HIPCHK(hipMemsetAsync(d_ids_out, 0, (size_t)count * 8, m->stream3));
    } else {
        const uint32_t rng = (uint32_t)(high - 1);
        launch_mt_draw(m->d_rng, d_ids_out, count, rng, mask_for(rng), m->stream3, nullptr, &m->rng_ws);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(dev, m->stream3));
    m->draw_count += 1;
    return TFR_OK;
}

// the model's stream waits for the first `ordinal` draws issued by tfr_draw_ids_dev (1-based count; draws complete in issue
// order).  A draw whose event slot has been reused by a later draw is covered by waiting for that later one.
static int join_draw_upto(tfr_model* m, int64_t ordinal) {
    if (ordinal > m->draw_count) ordinal = m->draw_count;
    if (ordinal <= m->draw_joined) return TFR_OK;
    int64_t k = ordinal - 1;                                   // index of the draw to wait for
    while (k + tfr_model::DRAW_RING < m->draw_count) k += tfr_model::DRAW_RING;   // its slot now holds a later draw of the same residue
    HIPCHK(hipStreamWaitEvent(m->stream, m->draw_evs[k % tfr_model::DRAW_RING], 0));
    m->draw_joined = k + 1 > ordinal ? ordinal : k + 1;
    if (k + 1 > m->draw_joined) m->draw_joined = k + 1;
    return TFR_OK;
}

int tfr_join_draws(tfr_model* m) {
    MODEL_ENTER(m);
    return join_draw_upto(m, m->draw_count);
}

int tfr_join_draw(tfr_model* m, int64_t ordinal) {
    MODEL_ENTER(m);
    if (ordinal < 1) return fail(TFR_ERR_ARG, "join_draw: ordinal counts issued draws from 1");
    return join_draw_upto(m, ordinal);
}

// run ahead: the next call's first batches, drawn into the other id buffer (last read by the call before the current one)
static int enqueue_run_ahead(tfr_model* m, int64_t B, int64_t nsteps, uint32_t rng) {
    int rc;
    if (rng != 0) {
        int64_t spec = 131072 / B;
        if (spec > 8) spec = 8;
        if (spec > nsteps) spec = nsteps;
        if (spec < 1) spec = 1;
        if (spec * B > m->d_ids_alt_cap || m->d_ids_alt_cap < m->d_ids_cap) {   // keep both buffers the same size: a repeat of this call then fits
            HIPCHK(hipStreamSynchronize(m->stream));       // the old alternate buffer may still be read by queued steps
            HIPCHK(hipStreamSynchronize(m->stream3));
            dfree(m->d_ids_alt);
            m->d_ids_alt = nullptr; m->d_ids_alt_cap = 0;
            int64_t want = spec * B > m->d_ids_cap ? spec * B : m->d_ids_cap;
            if (want < IDS_MIN_CAP) want = IDS_MIN_CAP;
            if ((rc = dmalloc(&m->d_ids_alt, (size_t)want))) return rc;
            m->d_ids_alt_cap = want;
        }
        HIPCHK(hipStreamWaitEvent(m->stream3, m->ev_ids_free, 0));  // the alternate buffer's last readers (before this call) are done
        HIPCHK(hipMemcpyAsync(m->d_rng_snap, m->d_rng, 625 * 4, hipMemcpyDeviceToDevice, m->stream3));
        launch_mt_draw(m->d_rng, m->d_ids_alt, spec * B, rng, mask_for(rng), m->stream3, nullptr, &m->rng_ws);
        recs_follow(m, m->d_ids_alt, m->d_ids_alt_cap, 0, spec * B, m->stream3);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(m->spec_ev, m->stream3));
        m->spec_valid = true; m->spec_B = B; m->spec_N = m->N; m->spec_steps = spec;
    }
    return TFR_OK;
}

int tfr_train_steps_drawn(tfr_model* m, int64_t B, int32_t nsteps, float* loss_out) {
    MODEL_ENTER(m);
    if (!m->N) return fail(TFR_ERR_STATE, "no resident triples: call tfr_upload_triples first");
    if (!m->rng_set) return fail(TFR_ERR_STATE, "no generator state: call tfr_rng_seed / tfr_rng_set_state first");
    if (B < 1 || nsteps < 0) return fail(TFR_ERR_ARG, "bad batch/nsteps");
    if (nsteps == 0) return TFR_OK;
    int rc;
    if ((rc = check_high(m->N))) return rc;
    CallTrace tr("steps_drawn");
    const int64_t total = B * (int64_t)nsteps;
    const uint32_t rng = (uint32_t)(m->N - 1);
    int64_t pre = 0;
    if (m->spec_valid && m->spec_B == B && m->spec_N == m->N && nsteps >= m->spec_steps) {
        // the previous call left the first spec_steps batches of this one in the alternate buffer
        pre = m->spec_steps;
        m->spec_valid = false;
        std::swap(m->d_ids, m->d_ids_alt);
        std::swap(m->d_ids_cap, m->d_ids_alt_cap);
        if (total > m->d_ids_cap) {                        // grow, keeping the head (rare: a longer call than ever before)
            HIPCHK(hipStreamSynchronize(m->stream3));
            int64_t* bigger = nullptr;
            if ((rc = dmalloc(&bigger, (size_t)total))) return rc;         // total > capacity >= IDS_MIN_CAP
            HIPCHK(hipMemcpy(bigger, m->d_ids, (size_t)pre * B * 8, hipMemcpyDeviceToDevice));
            dfree(m->d_ids);
            m->d_ids = bigger; m->d_ids_cap = total;
            recs_follow(m, m->d_ids, m->d_ids_cap, 0, pre * B, m->stream3);
        }
        HIPCHK(hipEventRecord(m->ev_ids_free, m->stream));          // everything queued so far: the earlier calls' steps
    } else {
        if ((rc = cancel_run_ahead(m))) return rc;
        if (total > m->d_ids_cap) {                        // (re)allocation: nothing may still read the old buffer
            HIPCHK(hipStreamSynchronize(m->stream));
            if ((rc = ensure_ids(m, total))) return rc;
        }
        // the draws overwrite the id buffer: they may start once every step already queued has read it
        HIPCHK(hipEventRecord(m->ev_ids_free, m->stream));
        HIPCHK(hipStreamWaitEvent(m->stream3, m->ev_ids_free, 0));
        for (int k = 0; k < 2; ++k) if (m->rb[k].ids_base == m->d_ids) m->rb[k].n = 0;
    }
    m->pf_valid = false;                                   // the buffer's contents change: no published look-ahead sort survives
    m->n_ids = total;
    IdsReady ready;
    ready.m = m; ready.B = B; ready.nsteps = nsteps; ready.rng = rng;
    ready.plan(B >= 65536 ? 1 : (65536 / B < 16 ? 65536 / B : 16), pre);     // a chunk holds at most ~64K ids / 16 steps
    if (rng == 0) {                                        // one-rating store: no draw consumed
        HIPCHK(hipMemsetAsync(m->d_ids, 0, (size_t)total * 8, m->stream3));
        for (int k = 0; k < 2; ++k) if (m->rb[k].ids_base == m->d_ids) m->rb[k].n = 0;
    }
    tr.mark(pre ? "ids drawn ahead taken" : "no ids drawn ahead");
    if ((rc = staged_steps(m, 0, B, nsteps, loss_out, &ready))) return rc;
    tr.mark("all steps enqueued");
    if ((rc = ready.enqueue_through(nsteps))) return rc;   // (every chunk is out by now; this is a no-op kept for clarity)
    if (rng != 0 && (rc = ready.run_ahead())) return rc;
    tr.mark("run-ahead draw enqueued");
    return TFR_OK;
}

// which kernels (rocprof's demangled spelling of the template arguments) one training step of this model
// launches at this batch size - so that bench.py can name what it timed and look the same kernels up in
// the committed rocprofv3 summaries
int tfr_kernel_plan(tfr_model* m, int64_t B, char* buf, int64_t buflen) {
    MODEL_ENTER(m);
    if (!buf || buflen < 64 || B < 1) return fail(TFR_ERR_ARG, "kernel_plan: need a buffer of >= 64 bytes and batch >= 1");
    const tfr_opts& o = m->o;
    const bool adam = o.optimizer == TFR_OPT_ADAM;
    const bool tf1 = adam && o.adam_mode == TFR_ADAM_TF1;
    const int G = m->G, V = m->VEC;
    const bool small = (1 << (m->bits_u > m->bits_i ? m->bits_u : m->bits_i)) <= CSORT_MAX_BINS;
    const bool csort = small && csort_eligible(B, m->bits_u, m->bits_i);
    const int64_t ntiles = (B + CSORT_TILE - 1) / CSORT_TILE;
    char tmp[1024];
    if (csort && ntiles <= 16) {
        const int nt = ntiles <= 4 ? 4 : ntiles <= 8 ? 8 : ntiles <= 10 ? 10 : ntiles <= 12 ? 12 : 16;
        snprintf(tmp, sizeof(tmp), "reduce_item=k_tile_step<%d, %d, %d>;apply=k_dense_tiles<%d, %d, false, %d>", G, V,
                 tile_step_epg((int)ntiles, G, V), G, V, nt);
    } else if (!tf1 && !csort) {
        const int rm = adam ? RMODE_ADAM : RMODE_SGD;
        // the sixth argument: the three-round load form (launch_seg_reduce picks it for the two-table step on full-width rows)
        const char* e1 = getenv("TFR_DUALQ"); const char* e2 = getenv("TFR_FAST"); const char* e3 = getenv("TFR_LEAN");
        const bool fast = !(e1 && e1[0] == '0') && !(e2 && e2[0] == '0') && !(e3 && e3[0] == '0') && m->D == G * V;
        snprintf(tmp, sizeof(tmp), "sort=%s x%d passes (the first gathers the batch);reduce_item=k_seg_reduce<%d, %d, %d, true, true, %s>;"
                 "reduce_user=k_seg_reduce<%d, %d, %d, false, true, %s>;apply=k_apply_rows<%d, %d, %d>",   // K4 rides in the apply launch
                 rsortw_eligible(B) ? "k_rsortw_hist/k_rsort_scan/k_rsortw_scatter" : "k_rsort_rank/scan/scatter",
                 ((m->bits_u > m->bits_i ? m->bits_u : m->bits_i) + 7) / 8, G, V, rm, fast ? "true" : "false", G, V, rm, fast ? "true" : "false", G, V, adam ? 0 : 1);
    } else if (csort) {
        snprintf(tmp, sizeof(tmp), "forward=k_front<%d, %d>;sort=k_csort_scan/scatter;reduce_item=k_seg_reduce<%d, %d, %d, false, true, false>;apply=%s",
                 G, V, G, V, tf1 ? RMODE_SCRATCH : (adam ? RMODE_ADAM : RMODE_SGD), tf1 ? "k_adam_dense" : "k_apply_rows");
    } else {
        snprintf(tmp, sizeof(tmp), "forward=k_forward<%d, %d, 1, 4, false>;sort=k_rsort_rank/scan/scatter;reduce_item=k_seg_reduce<%d, %d, 0, false, true, false>;apply=k_adam_dense<%d, %d>;finalize=k_finalize",
                 G, V, G, V, G, V);
    }
    snprintf(buf, (size_t)buflen, "%s", tmp);
    return TFR_OK;
}

// static + dynamic LDS bytes per workgroup of the shape-dependent kernels, computed exactly as the launchers do.
// Needs no device (pure host arithmetic): tests/test_lds_budget.py enumerates every selectable combination.
int tfr_lds_bytes(int32_t kernel, int32_t dim, int64_t batch, int64_t user_num, int64_t item_num,
                  int64_t* static_bytes, int64_t* dynamic_bytes) {
    int G, VEC;
    if (!geometry(dim, &G, &VEC)) return fail(TFR_ERR_ARG, "unsupported dim %d", dim);
    if (batch < 1 || user_num < 1 || item_num < 1) return fail(TFR_ERR_ARG, "lds_bytes: batch and row counts must be >= 1");
    const int bu = bits_for(user_num), bi = bits_for(item_num);
    const int nbmax = 1 << (bu > bi ? bu : bi);
    int64_t st = 0, dy = 0;
    if (kernel == 0 || kernel == 2) {                    // k_tile_step / k_front: small tables only
        if (nbmax > CSORT_MAX_BINS || !csort_eligible(batch, bu, bi)) return fail(TFR_ERR_ARG, "shape does not take the counting-sort path");
        const int64_t ntiles = (batch + CSORT_TILE - 1) / CSORT_TILE;
        if (kernel == 0) {
            if (ntiles > 16) return fail(TFR_ERR_ARG, "batch beyond 16 tiles does not take k_tile_step");
            const int epg = tile_step_epg((int)ntiles, G, VEC);
            st = (int64_t)tile_step_static_lds(G, epg);
            dy = (int64_t)tile_step_dyn_lds(G, VEC, epg, nbmax);
        } else {
            st = 256 * 4 + 16 * 64 * 4 + 16 * 3 * 4;     // wtot, the forward's LDS-staged reduce slot, block_sum_store
            dy = (int64_t)nbmax * 4;
        }
    } else if (kernel == 1 || kernel == 4) {             // k_seg_reduce with / without the forward inside
        st = (int64_t)seg_reduce_static_lds(G, VEC, kernel == 1);
    } else if (kernel == 3) {                            // k_mt_draw: two generator blocks + the wave counts
        st = 2 * 624 * 4 + 2 * 10 * 4;
    } else {
        return fail(TFR_ERR_ARG, "lds_bytes: kernel must be 0 (k_tile_step), 1 (k_seg_reduce fwd), 2 (k_front), 3 (k_mt_draw), 4 (k_seg_reduce)");
    }
    if (static_bytes) *static_bytes = st;
    if (dynamic_bytes) *dynamic_bytes = dy;
    return TFR_OK;
}

static int ensure_ring(tfr_model* m, int64_t B) {
    if (B <= m->ring_cap) return TFR_OK;
    HIPCHK(hipStreamSynchronize(m->stream));
    dfree(m->d_ring); m->d_ring = nullptr;
    if (m->h_ring) (void)hipHostFree(m->h_ring);
    m->h_ring = nullptr; m->ring_cap = 0;
    int64_t cap = 1024;
    while (cap < B) cap *= 2;
    int rc;
    if ((rc = dmalloc(&m->d_ring, (size_t)cap * tfr_model::HRING))) return rc;
    HIPCHK(hipHostMalloc((void**)&m->h_ring, (size_t)cap * tfr_model::HRING * 8, hipHostMallocDefault));
    for (int z = 0; z < tfr_model::HRING; ++z)
        if (!m->ring_ev[z]) HIPCHK(hipEventCreateWithFlags(&m->ring_ev[z], hipEventDisableTiming));
    m->ring_cap = cap;
    return TFR_OK;
}

int tfr_train_step_ids(tfr_model* m, const int64_t* ids, int64_t B) {
    MODEL_ENTER(m);
    if (!m->N) return fail(TFR_ERR_STATE, "no resident triples: call tfr_upload_triples first");
    if (B < 1 || !ids) return fail(TFR_ERR_ARG, "bad batch / null ids");
    int rc;
    if ((rc = ensure_capacity(m, B))) return rc;
    if ((rc = ensure_ring(m, B))) return rc;
    const int z = m->ring_pos;
    m->ring_pos = (z + 1) % tfr_model::HRING;
    HIPCHK(hipEventSynchronize(m->ring_ev[z]));            // the copy that last used this pinned slot has left it
    int64_t* h = m->h_ring + (size_t)z * m->ring_cap;
    int64_t* d = m->d_ring + (size_t)z * m->ring_cap;
    memcpy(h, ids, (size_t)B * 8);
    HIPCHK(hipMemcpyAsync(d, h, (size_t)B * 8, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipEventRecord(m->ring_ev[z], m->stream));
    m->pf_valid = false;
    return run_train_step(m, m->d_u, m->d_i, m->d_r, B, nullptr, nullptr, d, nullptr);
}

int tfr_forward_resident(tfr_model* m, int64_t lo, int64_t hi, float* logits_out) {
    MODEL_ENTER(m);
    if (!m->N) return fail(TFR_ERR_STATE, "no resident triples: call tfr_upload_triples first");
    if (lo < 0 || hi < lo || hi > m->N) return fail(TFR_ERR_ARG, "bad store range [%lld,%lld)", (long long)lo, (long long)hi);
    const int64_t B = hi - lo;
    if (B == 0) return TFR_OK;
    int rc;
    if ((rc = ensure_capacity(m, B))) return rc;
    if ((rc = gather_batch(m, nullptr, lo, B))) return rc;
    if ((rc = run_forward(m, MODE_INFER, m->d_u, m->d_i, nullptr, B, m->d_logits, nullptr, nullptr))) return rc;
    if (logits_out) {
        HIPCHK(hipMemcpyAsync(logits_out, m->d_logits, (size_t)B * 4, hipMemcpyDeviceToHost, m->stream));
        return check_device_error(m);
    }
    return TFR_OK;
}

int tfr_sort_segments(tfr_model* m, int32_t side, const int32_t* ids, int64_t B, int32_t* ks_out, int32_t* ps_out) {
    MODEL_ENTER(m);
    if (side != 0 && side != 1) return fail(TFR_ERR_ARG, "side must be 0 (user) or 1 (item)");
    if (B < 0 || (B > 0 && (!ids || !ks_out || !ps_out))) return fail(TFR_ERR_ARG, "bad batch / null pointer");
    if (B == 0) return TFR_OK;
    int rc;
    if ((rc = ensure_capacity(m, B))) return rc;
    HIPCHK(hipMemcpyAsync(m->d_u, ids, (size_t)B * 4, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(m->d_i, ids, (size_t)B * 4, hipMemcpyHostToDevice, m->stream));
    if ((rc = sort_columns(m, m->d_u, m->d_i, B))) return rc;      // the training step's own sort path
    HIPCHK(hipMemcpyAsync(ks_out, side == 0 ? m->ks_u : m->ks_i, (size_t)B * 4, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipMemcpyAsync(ps_out, side == 0 ? m->ps_u : m->ps_i, (size_t)B * 4, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return TFR_OK;
}

// ---- row-sharded building blocks --------------------------------------------
// The model handle holds this rank's shard: user rows [U_local], item rows [I_local].  User
// rows of a sample are always local (samples are routed to the owner of their user row);
// item rows are fetched from and their gradients returned to their owners through fixed-capacity
// exchange buffers laid out [world][slot_cap] rows of tfr_shard_row_stride() floats: D features, the bias, padding.

// floats per exchanged row: features, bias, the sender's error flag, padding to 16 bytes where rows are read as float4
static int shard_stride(const tfr_model* m) { return m->VEC == 4 ? m->D + 4 : m->D + 2; }

int32_t tfr_shard_row_stride(tfr_model* m) { return m ? shard_stride(m) : 0; }

static int shard_route_core(tfr_model* m, const int32_t* d_user, const int32_t* d_item, const float* d_rate, const int64_t* d_ids,
                            int64_t Bg, int32_t rank, int32_t world, int64_t U_global, int64_t I_global, int32_t sample_cap,
                            int32_t slot_cap, int32_t* d_req, const int4* d_recs = nullptr) {
    if (Bg < 0 || world < 1 || rank < 0 || rank >= world || sample_cap < 1 || slot_cap < 1 || !d_req || U_global < 1 || I_global < 1 ||
        I_global >= 0x7fffffffLL || U_global >= 0x7fffffffLL || (Bg > 0 && !d_ids && !d_recs && (!d_user || !d_item || !d_rate)))
        return fail(TFR_ERR_ARG, "shard_route: bad arguments");
    if ((int64_t)world * slot_cap >= 0x7fffffffLL) return fail(TFR_ERR_ARG, "shard_route: world * slot_cap too large");
    int rc;
    const int64_t need = sample_cap > (int64_t)world * slot_cap ? sample_cap : (int64_t)world * slot_cap;
    if ((rc = ensure_capacity(m, need > Bg ? need : (Bg > 0 ? Bg : 1)))) return rc;
    tfr_model::RouteSet& R = m->rt[m->rt_sel];
    if (sample_cap > R.cap || world > R.cap_world) {
        HIPCHK(hipStreamSynchronize(m->stream));
        dfree(R.mine); dfree(R.u); dfree(R.it); dfree(R.r); dfree(R.slot); dfree(R.counts);
        dfree(R.ks_u); dfree(R.ps_u); dfree(R.ks_i); dfree(R.ps_i);
        R.mine = R.u = R.it = R.slot = R.counts = R.ks_u = R.ps_u = R.ks_i = R.ps_i = nullptr; R.r = nullptr; R.cap = 0;
        if ((rc = dmalloc(&R.mine, (size_t)sample_cap)) || (rc = dmalloc(&R.u, (size_t)sample_cap)) ||
            (rc = dmalloc(&R.it, (size_t)sample_cap)) || (rc = dmalloc(&R.r, (size_t)sample_cap)) ||
            (rc = dmalloc(&R.slot, (size_t)sample_cap)) || (rc = dmalloc(&R.counts, (size_t)world + 4)) ||
            (rc = dmalloc(&R.ks_u, (size_t)sample_cap)) || (rc = dmalloc(&R.ps_u, (size_t)sample_cap)) ||
            (rc = dmalloc(&R.ks_i, (size_t)sample_cap)) || (rc = dmalloc(&R.ps_i, (size_t)sample_cap))) return rc;
        R.cap = sample_cap; R.cap_world = world;
    }
    R.B = sample_cap; R.slots = world * slot_cap; R.world = world;
    R.sorted_fwd = false; R.sorted_req = nullptr;
    hipStream_t s = m->stream;
    RouteArgs a;
    memset(&a, 0, sizeof(a));
    a.u = d_user; a.it = d_item; a.r = d_rate; a.Bg = Bg; a.U = U_global; a.I = I_global;
    if (d_ids) { a.ids = d_ids; a.store = m->store; a.N = m->N; }
    a.recs = d_recs;
    a.per_u = (U_global + world - 1) / world; a.per_i = (I_global + world - 1) / world; a.u_lo = a.per_u * rank;
    a.rank = rank; a.world = world; a.Bcap = sample_cap; a.cap = slot_cap;
    a.u_pad = (int32_t)m->U; a.i_pad = (int32_t)I_global;
    a.mine = R.mine; a.u_local = R.u; a.it_glob = R.it; a.r_loc = R.r; a.slot = R.slot;
    a.req = d_req; a.counts = R.counts; a.blk = m->lrank_u; a.err = m->d_err;      // lrank_u: [cap] ints of sort scratch
    HIPCHK(hipMemsetAsync(d_req, 0xff, (size_t)world * slot_cap * 4, s));             // every slot unused (-1)
    launch_route_compact(a, s);
    HIPCHK(hipGetLastError());
    {   // the local samples sorted by global item id: distinct ids become adjacent and grouped by owner
        const int32_t* keys[2] = {R.it, nullptr};           // (into the set's own arrays: the step buffers may be in use by a step
        const int bits[2] = {bits_for(I_global + 1), 0};    //  that runs beside this routing)
        int32_t* ks[2] = {R.ks_i, nullptr};
        int32_t* ps[2] = {R.ps_i, nullptr};
        if ((rc = radix_sort_columns(m, 1, keys, bits, ks, ps, sample_cap))) return rc;
    }
    a.ks = R.ks_i; a.ps = R.ps_i;
    launch_route_slots(a, s);
    HIPCHK(hipGetLastError());
    return TFR_OK;
}

int tfr_shard_route(tfr_model* m, const int32_t* d_user, const int32_t* d_item, const float* d_rate, int64_t Bg,
                    int32_t rank, int32_t world, int64_t U_global, int64_t I_global, int32_t sample_cap, int32_t slot_cap,
                    int32_t* d_req) {
    MODEL_ENTER(m);
    return shard_route_core(m, d_user, d_item, d_rate, nullptr, Bg, rank, world, U_global, I_global, sample_cap, slot_cap, d_req);
}

int tfr_shard_route_ids(tfr_model* m, const int64_t* d_ids, int64_t Bg, int32_t rank, int32_t world, int64_t U_global,
                        int64_t I_global, int32_t sample_cap, int32_t slot_cap, int32_t* d_req) {
    MODEL_ENTER(m);
    if (!m->N) return fail(TFR_ERR_STATE, "no resident triples: call tfr_upload_triples / tfr_set_triples_dev first (global row ids)");
    if (Bg > 0 && !d_ids) return fail(TFR_ERR_ARG, "shard_route_ids: null ids");
    return shard_route_core(m, nullptr, nullptr, nullptr, d_ids, Bg, rank, world, U_global, I_global, sample_cap, slot_cap, d_req);
}

int tfr_shard_bucket_ids(tfr_model* m, const int64_t* d_ids, int64_t B, int32_t world, int64_t U_global, int32_t pair_cap,
                         void* d_send) {
    MODEL_ENTER(m);
    if (!m->N) return fail(TFR_ERR_STATE, "no resident triples: call tfr_upload_triples / tfr_set_triples_dev first (global row ids)");
    if (B < 0 || world < 1 || world > 4096 || U_global < 1 || pair_cap < 1 || !d_send || (B > 0 && !d_ids))
        return fail(TFR_ERR_ARG, "shard_bucket_ids: bad arguments");
    int rc;
    const int64_t nblocks = (B + 1023) / 1024 > 0 ? (B + 1023) / 1024 : 1;
    const int64_t need = nblocks * world > B ? nblocks * world : (B > 0 ? B : 1);
    if ((rc = ensure_capacity(m, need))) return rc;
    hipStream_t s = m->stream;
    HIPCHK(hipMemsetAsync(d_send, 0xff, (size_t)world * pair_cap * sizeof(int4), s));      // every slot unused (u = -1)
    BucketArgs a;
    memset(&a, 0, sizeof(a));
    a.ids = d_ids; a.store = m->store; a.N = m->N; a.B = B; a.U = U_global; a.per_u = (U_global + world - 1) / world;
    a.world = world; a.cap = pair_cap; a.send = reinterpret_cast<int4*>(d_send); a.blk = m->lrank_u; a.err = m->d_err;
    launch_bucket(a, s);
    HIPCHK(hipGetLastError());
    return TFR_OK;
}

int tfr_shard_route_recs(tfr_model* m, const void* d_recs, int64_t n, int32_t rank, int32_t world, int64_t U_global,
                         int64_t I_global, int32_t sample_cap, int32_t slot_cap, int32_t* d_req) {
    MODEL_ENTER(m);
    if (n > 0 && !d_recs) return fail(TFR_ERR_ARG, "shard_route_recs: null records");
    return shard_route_core(m, nullptr, nullptr, nullptr, nullptr, n, rank, world, U_global, I_global, sample_cap, slot_cap, d_req,
                            reinterpret_cast<const int4*>(d_recs));
}

int tfr_shard_routed_devptrs(tfr_model* m, void** mine, void** u_local, void** slot, void** counts) {
    MODEL_ENTER(m);
    const tfr_model::RouteSet& R = m->rt[m->rt_sel];
    if (!R.counts) return fail(TFR_ERR_STATE, "no routed batch: call tfr_shard_route first");
    if (mine) *mine = R.mine;
    if (u_local) *u_local = R.u;
    if (slot) *slot = R.slot;
    if (counts) *counts = R.counts;
    return TFR_OK;
}

int tfr_shard_gather(tfr_model* m, const int32_t* d_req_recv, int64_t n, float* d_rows_out) {
    MODEL_ENTER(m);
    if (n < 0 || (n > 0 && (!d_req_recv || !d_rows_out))) return fail(TFR_ERR_ARG, "shard_gather: bad arguments");
    if (n == 0) return TFR_OK;
    { const int rcq = settle_q(m); if (rcq) return rcq; }
    GatherPackedArgs g;
    g.ids = d_req_recv; g.table = m->w[TFR_Q]; g.bias = m->w[TFR_BI]; g.out = d_rows_out; g.err = m->d_err;
    g.n = n; g.rows = m->I; g.D = m->D; g.stride = shard_stride(m);
    launch_gather_packed(g, m->G, m->VEC, m->stream);
    HIPCHK(hipGetLastError());
    return TFR_OK;
}

// part: 1 = the item half (sort, forward + item-side reduce into the exchange buffer, local scalars), 2 = the user half (user-side
// reduce + apply; needs nothing the gradient exchange touches, so a caller may run it beside that exchange), 3 = both
static int shard_forward_reduce_part(tfr_model* m, const float* d_item_rows, float* d_logits, float* d_item_grad, float* d_scalars4, int part) {
    tfr_model::RouteSet& R = m->rt[m->rt_sel];
    if (!R.counts) return fail(TFR_ERR_STATE, "no routed batch: call tfr_shard_route first");
    if (!d_item_rows || ((part & 1) && (!d_item_grad || !d_scalars4))) return fail(TFR_ERR_ARG, "shard_forward_reduce: null pointer");
    const int64_t B = R.B, nI = R.slots;
    const int DS = shard_stride(m);
    const int32_t* du = R.u; const int32_t* dslot = R.slot; const float* dr = R.r;
    const int32_t* dB = R.counts;                         // the local batch size, on the device
    int rc;
    if ((rc = ensure_capacity(m, B > nI ? B : nI))) return rc;
    const tfr_opts& o = m->o;
    const bool adam = o.optimizer == TFR_OPT_ADAM;
    const bool tf1 = adam && o.adam_mode == TFR_ADAM_TF1;
    const float alpha = adam ? o.lr * sqrtf(1.f - m->b2p) / (1.f - m->b1p) : 0.f;
    hipStream_t s = m->stream;
    int nblk = (int)((B + 1024 / m->G - 1) / (1024 / m->G));
    RedArgs r;
    memset(&r, 0, sizeof(r));
    r.g = m->d_g; r.err = m->d_err; r.B = B; r.D = m->D; r.dB = dB;
    r.item_abs = o.item_abs; r.reg_bias = o.reg_bias;
    r.lam = o.reg; r.alpha = alpha; r.b1 = o.beta1; r.b2 = o.beta2; r.eps = o.eps; r.lr = o.lr;
    RedPair pr;
    ApplyArgs ap;
    memset(&ap, 0, sizeof(ap));
    ap.err = m->d_err; ap.B = B; ap.D = m->D; ap.dB = dB;
    ap.alpha = alpha; ap.b1 = o.beta1; ap.b2 = o.beta2; ap.eps = o.eps; ap.lr = o.lr;
    ApplyPair app;
    if (part & 1) {
        // a peer that had to void this step (capacity overflow, id out of range) said so beside its rows: void it here too, before
        // anything is updated - every rank then skips the same step and reports it at its next sync
        launch_adopt_peer_err(d_item_rows, (int64_t)(nI / R.world) * DS, R.world, m->D, m->d_err, s);
        HIPCHK(hipGetLastError());
        // K1 runs inside the item-side reduce, on the rows it has in registers anyway (as in the single-GPU big-table step);
        // a separate k_forward launch cost 68 us of the 464 (world-1 rehearsal)
        if (R.sorted_fwd) {             // tfr_shard_presort made the sorted orders ahead of time
            R.fks_u = R.ks_u; R.fps_u = R.ps_u; R.fks_i = R.ks_i; R.fps_i = R.ps_i;
        } else {
            Prof p(m, TFR_K_SORT);                        // unused sample slots carry keys one past the last row: they sort last
            const int32_t* keys[2] = {du, dslot};
            const int bits[2] = {bits_for(m->U + 1), bits_for(nI + 1)};
            int32_t* ks[2] = {m->ks_u, m->ks_i};
            int32_t* ps[2] = {m->ps_u, m->ps_i};
            if ((rc = radix_sort_columns(m, 2, keys, bits, ks, ps, B))) return rc;
            R.fks_u = m->ks_u; R.fps_u = m->ps_u; R.fks_i = m->ks_i; R.fps_i = m->ps_i;
        }
        RedArgs ri = r;                 // item side: own = the fetched rows, indexed by slot
        ri.side = 1;
        ri.ks = R.fks_i; ri.ps = R.fps_i; ri.other = du;
        ri.own = d_item_rows; ri.ostride = DS; ri.own_bias = d_item_rows + m->D; ri.obstride = DS; ri.partner = m->w[TFR_P];
        ri.grad_rows = m->gq; ri.grad_bias = m->gbq;
        // a slot whose samples lie in one block of the sorted order (nearly all) goes straight into the exchange buffer;
        // k_apply_rows then only finishes the slots cut by a block boundary (it emitted every slot before: 102 us)
        ri.dense_rows = d_item_grad; ri.dstride = DS; ri.dense_bias = d_item_grad + m->D; ri.dbstride = DS;
        ri.partner_bias = m->w[TFR_BU]; ri.mu = m->w[TFR_MU]; ri.r = dr; ri.loss = o.loss;
        ri.g_out = m->d_g; ri.logits_out = d_logits; ri.partials = m->partials; ri.stage_sum = 1;
        pr.a[0] = ri;
        {
            Prof p(m, TFR_K_REDUCE_ITEM);
            launch_seg_reduce(pr, 1, RMODE_SCRATCH, m->G, m->VEC, s, true);
        }
        HIPCHK(hipGetLastError());
        app.a[0] = ap;                  // reduced gradient row (+ bias gradient) of the split slots, in the exchange layout
        app.a[0].only_split = 1;
        app.a[0].ks = R.fks_i; app.a[0].grad_rows = m->gq; app.a[0].grad_bias = m->gbq;
        app.a[0].w = d_item_grad; app.a[0].wstride = DS; app.a[0].bias_w = d_item_grad + m->D; app.a[0].wbstride = DS;
        {
            Prof p(m, TFR_K_APPLY);
            launch_apply_rows(app, 1, 2, m->G, m->VEC, s);
        }
        HIPCHK(hipGetLastError());
        FinArgs f;                      // local {loss, reg, sum g}; bias_global waits for the all-reduce
        memset(&f, 0, sizeof(f));
        f.partials = m->partials; f.nblk = nblk; f.scalars = m->scalars; f.out = d_scalars4; f.err = m->d_err;
        f.mu = m->w[TFR_MU];
        {
            Prof p(m, TFR_K_FINALIZE);
            launch_finalize(f, s);
        }
        HIPCHK(hipGetLastError());
    }
    if (part & 2) {
        RedArgs ru = r;                 // user side: rows are local; partner = fetched item rows
        ru.side = 0;
        if (!R.fks_u) return fail(TFR_ERR_STATE, "shard_reduce_users: call tfr_shard_forward_items on this routed batch first");
        ru.ks = R.fks_u; ru.ps = R.fps_u; ru.other = dslot;
        ru.own = m->w[TFR_P]; ru.partner = d_item_rows; ru.pstride = DS; ru.own_bias = m->w[TFR_BU];
        ru.own_w = m->w[TFR_P]; ru.m = m->m[TFR_P]; ru.v = m->v[TFR_P];
        ru.bias_w = m->w[TFR_BU]; ru.bias_m = m->m[TFR_BU]; ru.bias_v = m->v[TFR_BU];
        ru.grad_bias = m->gbp; ru.map = tf1 ? m->map_u : nullptr;
        ru.grad_rows = tf1 ? m->gp : m->gq + (size_t)m->cap * m->D;
        ru.frozen_rows = (m->frozen >> TFR_P) & 1; ru.frozen_bias = (m->frozen >> TFR_BU) & 1;
        pr.a[0] = ru;
        {
            Prof p(m, TFR_K_REDUCE_USER);
            launch_seg_reduce(pr, 1, tf1 ? RMODE_SCRATCH : (adam ? RMODE_ADAM : RMODE_SGD), m->G, m->VEC, s);
        }
        HIPCHK(hipGetLastError());
        if (!tf1) {
            app.a[0] = ap;
            app.a[0].only_split = 1;
            app.a[0].ks = R.fks_u; app.a[0].grad_rows = ru.grad_rows; app.a[0].grad_bias = m->gbp;
            app.a[0].w = m->w[TFR_P]; app.a[0].m = m->m[TFR_P]; app.a[0].v = m->v[TFR_P];
            app.a[0].bias_w = m->w[TFR_BU]; app.a[0].bias_m = m->m[TFR_BU]; app.a[0].bias_v = m->v[TFR_BU];
            app.a[0].frozen_rows = ru.frozen_rows; app.a[0].frozen_bias = ru.frozen_bias;
            Prof p(m, TFR_K_APPLY);
            launch_apply_rows(app, 1, adam ? 0 : 1, m->G, m->VEC, s);
        }
        HIPCHK(hipGetLastError());
    }
    if (tf1 && (part & 2)) {            // dense sweep of the local user rows (every row moves)
        DensePair dp;
        memset(&dp, 0, sizeof(dp));
        DenseArgs& d = dp.a[0];
        d.err = m->d_err; d.D = m->D; d.B = B;
        d.alpha = alpha; d.b1 = o.beta1; d.b2 = o.beta2; d.eps = o.eps;
        d.map = m->map_u; d.ks = R.fks_u; d.grad_rows = m->gp; d.grad_bias = m->gbp; d.rows = m->U;
        d.w = m->w[TFR_P]; d.m = m->m[TFR_P]; d.v = m->v[TFR_P];
        d.bias_w = m->w[TFR_BU]; d.bias_m = m->m[TFR_BU]; d.bias_v = m->v[TFR_BU];
        d.frozen_rows = (m->frozen >> TFR_P) & 1; d.frozen_bias = (m->frozen >> TFR_BU) & 1;
        Prof p(m, TFR_K_APPLY);
        launch_adam_dense(dp, 1, m->G, m->VEC, s);
        HIPCHK(hipGetLastError());
    }
    return TFR_OK;
}

int tfr_shard_forward_reduce(tfr_model* m, const float* d_item_rows, float* d_logits, float* d_item_grad, float* d_scalars4) {
    MODEL_ENTER(m);
    return shard_forward_reduce_part(m, d_item_rows, d_logits, d_item_grad, d_scalars4, 3);
}

int tfr_shard_forward_items(tfr_model* m, const float* d_item_rows, float* d_logits, float* d_item_grad, float* d_scalars4) {
    MODEL_ENTER(m);
    return shard_forward_reduce_part(m, d_item_rows, d_logits, d_item_grad, d_scalars4, 1);
}

int tfr_shard_reduce_users(tfr_model* m, const float* d_item_rows) {
    MODEL_ENTER(m);
    return shard_forward_reduce_part(m, d_item_rows, nullptr, nullptr, nullptr, 2);
}

int tfr_shard_apply_items(tfr_model* m, const int32_t* d_req_recv, const float* d_grad_recv, int64_t n) {
    MODEL_ENTER(m);
    if (n < 0 || (n > 0 && (!d_req_recv || !d_grad_recv))) return fail(TFR_ERR_ARG, "shard_apply_items: bad arguments");
    tfr_model::RouteSet& R = m->rt[m->rt_sel];
    if (!R.counts) return fail(TFR_ERR_STATE, "no routed batch: call tfr_shard_route first");
    int rc;
    if ((rc = settle_q(m))) return rc;
    if ((rc = ensure_capacity(m, n > 0 ? n : 1))) return rc;
    const tfr_opts& o = m->o;
    const bool adam = o.optimizer == TFR_OPT_ADAM;
    const bool tf1 = adam && o.adam_mode == TFR_ADAM_TF1;
    const float alpha = adam ? o.lr * sqrtf(1.f - m->b2p) / (1.f - m->b1p) : 0.f;
    const int DS = shard_stride(m);
    hipStream_t s = m->stream;
    int32_t* d_nvalid = R.counts + R.world + 2;           // requests actually received (unused slots excluded)
    const int32_t* aks = m->ks_i; const int32_t* aps = m->ps_i;
    if (n > 0) {
        if (R.sorted_req == d_req_recv && R.sorted_req_n == n) {       // tfr_shard_presort has sorted these requests already
            aks = R.aks; aps = R.aps;
        } else {
            // unused slots (-1) get the key one past the last row: they sort behind every real request and fall outside the count
            launch_pad_keys(d_req_recv, m->d_i, n, (int32_t)m->I, d_nvalid, s);
            HIPCHK(hipGetLastError());
            Prof p(m, TFR_K_SORT);
            const int32_t* keys[2] = {m->d_i, nullptr};
            const int bits[2] = {bits_for(m->I + 1), 0};
            int32_t* ks[2] = {m->ks_i, nullptr};
            int32_t* ps[2] = {m->ps_i, nullptr};
            if ((rc = radix_sort_columns(m, 1, keys, bits, ks, ps, n))) return rc;
        }
        RedPair pr;
        RedArgs& r = pr.a[0];
        memset(&r, 0, sizeof(r));
        r.err = m->d_err; r.B = n; r.D = m->D; r.side = 1; r.dB = d_nvalid;
        r.alpha = alpha; r.b1 = o.beta1; r.b2 = o.beta2; r.eps = o.eps; r.lr = o.lr;
        r.ks = aks; r.ps = aps; r.rows_in = d_grad_recv; r.rstride = DS; r.bias_in = d_grad_recv + m->D; r.rbstride = DS;
        r.own = m->w[TFR_Q]; r.own_bias = m->w[TFR_BI];
        r.own_w = m->w[TFR_Q]; r.m = m->m[TFR_Q]; r.v = m->v[TFR_Q];
        r.bias_w = m->w[TFR_BI]; r.bias_m = m->m[TFR_BI]; r.bias_v = m->v[TFR_BI];
        r.grad_rows = m->gq; r.grad_bias = m->gbq; r.map = tf1 ? m->map_i : nullptr;
        r.frozen_rows = (m->frozen >> TFR_Q) & 1; r.frozen_bias = (m->frozen >> TFR_BI) & 1;
        {
            Prof p(m, TFR_K_REDUCE_ITEM);
            launch_seg_reduce(pr, 1, tf1 ? RMODE_SCRATCH : (adam ? RMODE_ADAM : RMODE_SGD), m->G, m->VEC, s);
        }
        HIPCHK(hipGetLastError());
        if (!tf1) {
            ApplyPair app;
            ApplyArgs& ap = app.a[0];
            memset(&ap, 0, sizeof(ap));
            ap.err = m->d_err; ap.B = n; ap.D = m->D; ap.only_split = 1; ap.dB = d_nvalid;
            ap.alpha = alpha; ap.b1 = o.beta1; ap.b2 = o.beta2; ap.eps = o.eps; ap.lr = o.lr;
            ap.ks = aks; ap.grad_rows = m->gq; ap.grad_bias = m->gbq;
            ap.w = m->w[TFR_Q]; ap.m = m->m[TFR_Q]; ap.v = m->v[TFR_Q];
            ap.bias_w = m->w[TFR_BI]; ap.bias_m = m->m[TFR_BI]; ap.bias_v = m->v[TFR_BI];
            ap.frozen_rows = r.frozen_rows; ap.frozen_bias = r.frozen_bias;
            Prof p(m, TFR_K_APPLY);
            launch_apply_rows(app, 1, adam ? 0 : 1, m->G, m->VEC, s);
        }
        HIPCHK(hipGetLastError());
    }
    if (tf1) {
        DensePair dp;
        memset(&dp, 0, sizeof(dp));
        DenseArgs& d = dp.a[0];
        d.err = m->d_err; d.D = m->D; d.B = n;
        d.alpha = alpha; d.b1 = o.beta1; d.b2 = o.beta2; d.eps = o.eps;
        d.map = m->map_i; d.ks = aks; d.grad_rows = m->gq; d.grad_bias = m->gbq; d.rows = m->I;
        d.w = m->w[TFR_Q]; d.m = m->m[TFR_Q]; d.v = m->v[TFR_Q];
        d.bias_w = m->w[TFR_BI]; d.bias_m = m->m[TFR_BI]; d.bias_v = m->v[TFR_BI];
        d.frozen_rows = (m->frozen >> TFR_Q) & 1; d.frozen_bias = (m->frozen >> TFR_BI) & 1;
        Prof p(m, TFR_K_APPLY);
        launch_adam_dense(dp, 1, m->G, m->VEC, s);
        HIPCHK(hipGetLastError());
    }
    return TFR_OK;
}

// which of the two routed-batch sets the shard calls fill (route_*) and consume (forward / reduce / apply): a caller that
// prepares batch s+1 on a side stream while step s runs alternates between them
int tfr_shard_select(tfr_model* m, int32_t which) {
    MODEL_ENTER(m);
    if (which != 0 && which != 1) return fail(TFR_ERR_ARG, "shard_select: 0 or 1");
    m->rt_sel = which;
    return TFR_OK;
}

// the index work of a step, ahead of time (all of it depends on the routed batch and the requests only, not on any table):
// the routed samples sorted by local user row and by request slot, and - with d_req_recv - the requests received as an owner
// padded and sorted by item row.  tfr_shard_forward_items / _reduce_users / _apply_items then skip their own sorts.
int tfr_shard_presort(tfr_model* m, const int32_t* d_req_recv, int64_t n) {
    MODEL_ENTER(m);
    tfr_model::RouteSet& R = m->rt[m->rt_sel];
    if (!R.counts) return fail(TFR_ERR_STATE, "no routed batch: call tfr_shard_route first");
    if (n < 0 || (n > 0 && !d_req_recv)) return fail(TFR_ERR_ARG, "shard_presort: bad arguments");
    int rc;
    const int64_t B = R.B, nI = R.slots;
    if ((rc = ensure_capacity(m, (B > nI ? B : nI) > n ? (B > nI ? B : nI) : n))) return rc;
    {
        const int32_t* keys[2] = {R.u, R.slot};
        const int bits[2] = {bits_for(m->U + 1), bits_for(nI + 1)};
        int32_t* ks[2] = {R.ks_u, R.ks_i};
        int32_t* ps[2] = {R.ps_u, R.ps_i};
        if ((rc = radix_sort_columns(m, 2, keys, bits, ks, ps, B))) return rc;
        R.sorted_fwd = true;
    }
    if (n > 0) {
        if (n > R.acap) {
            HIPCHK(hipStreamSynchronize(m->stream));
            dfree(R.akeys); dfree(R.aks); dfree(R.aps);
            R.akeys = R.aks = R.aps = nullptr; R.acap = 0;
            if ((rc = dmalloc(&R.akeys, (size_t)n)) || (rc = dmalloc(&R.aks, (size_t)n)) || (rc = dmalloc(&R.aps, (size_t)n))) return rc;
            R.acap = n;
        }
        launch_pad_keys(d_req_recv, R.akeys, n, (int32_t)m->I, R.counts + R.world + 2, m->stream);
        HIPCHK(hipGetLastError());
        const int32_t* keys[2] = {R.akeys, nullptr};
        const int bits[2] = {bits_for(m->I + 1), 0};
        int32_t* ks[2] = {R.aks, nullptr};
        int32_t* ps[2] = {R.aps, nullptr};
        if ((rc = radix_sort_columns(m, 1, keys, bits, ks, ps, n))) return rc;
        R.sorted_req = d_req_recv; R.sorted_req_n = n;
    }
    return TFR_OK;
}

int tfr_shard_finish_step(tfr_model* m, const float* d_scalars4) {
    MODEL_ENTER(m);
    if (!d_scalars4) return fail(TFR_ERR_ARG, "shard_finish_step: null scalars");
    const tfr_opts& o = m->o;
    const bool adam = o.optimizer == TFR_OPT_ADAM;
    const float alpha = adam ? o.lr * sqrtf(1.f - m->b2p) / (1.f - m->b1p) : 0.f;
    FinArgs f;
    memset(&f, 0, sizeof(f));
    f.partials = d_scalars4; f.nblk = 1; f.scalars = m->scalars; f.out = nullptr;
    f.mu = m->w[TFR_MU]; f.mu_m = m->m[TFR_MU]; f.mu_v = m->v[TFR_MU]; f.err = m->d_err;
    f.update_mu = !((m->frozen >> TFR_MU) & 1); f.opt = adam ? 0 : 1;
    f.alpha = alpha; f.b1 = o.beta1; f.b2 = o.beta2; f.eps = o.eps; f.lr = o.lr;
    {
        Prof p(m, TFR_K_FINALIZE);
        launch_finalize(f, m->stream);
    }
    HIPCHK(hipGetLastError());
    if (adam) {
        m->b1p *= o.beta1;
        m->b2p *= o.beta2;
    }
    m->step += 1;
    return TFR_OK;
}

// ---- data-parallel building blocks: replicated tables, one all-reduce per step -----------
// flat gradient buffer layout (floats): [P grads U*D | Q grads I*D | user_bias U | item_bias I |
// loss, reg, sum_g, 0].  It must be all zeros before the first tfr_dp_local_grads (tfr_dp_apply
// leaves it zeroed again, so one cudaMemset at allocation is enough).
int64_t tfr_dp_flat_size(tfr_model* m) {
    if (!m) return 0;
    return m->U * m->D + m->I * m->D + m->U + m->I + 4;
}

int tfr_dp_hint_next(tfr_model* m, const int64_t* d_next_store_ids) {
    MODEL_ENTER(m);
    m->dp_next_ids = d_next_store_ids;
    return TFR_OK;
}

int tfr_dp_local_grads(tfr_model* m, const int32_t* du, const int32_t* di, const float* dr, int64_t B,
                       const int64_t* d_store_ids, float* d_flat) {
    MODEL_ENTER(m);
    if (!d_flat || B < 0) return fail(TFR_ERR_ARG, "dp_local_grads: bad arguments");
    if (!d_store_ids && B > 0 && (!du || !di || !dr)) return fail(TFR_ERR_ARG, "dp_local_grads: null batch pointers");
    if (d_store_ids && !m->N) return fail(TFR_ERR_STATE, "no resident triples: call tfr_upload_triples first");
    const tfr_opts& o = m->o;
    if (o.optimizer == TFR_OPT_ADAM && o.adam_mode != TFR_ADAM_TF1)
        return fail(TFR_ERR_STATE, "data-parallel steps need dense semantics: Adam tf1 or SGD");
    int rc;
    if ((rc = settle_q(m))) return rc;
    const int64_t* next_ids = d_store_ids ? m->dp_next_ids : nullptr;     // one-shot hint (tfr_dp_hint_next)
    m->dp_next_ids = nullptr;
    if ((rc = ensure_capacity(m, B > 0 ? B : 1))) return rc;
    float* gP = d_flat;
    float* gQ = gP + m->U * m->D;
    float* gbu = gQ + m->I * m->D;
    float* gbi = gbu + m->U;
    float* tail = gbi + m->I;
    hipStream_t s = m->stream;
    int nblk = 0;
    bool fin_done = false;
    FinArgs f;                         // local {loss, reg, sum g} -> tail of the flat buffer; no mu update yet
    memset(&f, 0, sizeof(f));
    f.partials = m->partials; f.scalars = m->scalars; f.out = tail; f.err = m->d_err;
    f.mu = m->w[TFR_MU];
    if (B > 0 && tiles_eligible(m, B)) {
        // small tables: k_tile_step (look-ahead sort of the hinted next batch included), then one sweep that
        // writes every touched row's gradient into the flat buffer and reduces the local scalars (K4 without
        // the mu update)
        float* gp_rows = m->gp ? m->gp : m->gq + (size_t)m->cap * m->D;
        int par = 0;
        if ((rc = tile_step_launch(m, du, di, dr, B, nullptr, d_store_ids, next_ids, gp_rows, &par, &nblk))) return rc;
        f.nblk = nblk;
        TileDenseLaunch L;
        memset(&L, 0, sizeof(L));
        TileDenseArgs d;
        memset(&d, 0, sizeof(d));
        d.err = m->d_err; d.D = m->D; d.ntiles = (int32_t)((B + CSORT_TILE - 1) / CSORT_TILE);
        L.a[0] = d;
        L.a[0].tab = par ? m->offs_i : m->hist_i; L.a[0].nbins = 1 << m->bits_i; L.a[0].rows = m->I;
        L.a[0].grad_rows = m->gq; L.a[0].grad_bias = m->gbq; L.a[0].out_rows = gQ; L.a[0].out_bias = gbi;
        L.a[1] = d;
        L.a[1].tab = par ? m->offs_u : m->hist_u; L.a[1].nbins = 1 << m->bits_u; L.a[1].rows = m->U;
        L.a[1].grad_rows = gp_rows; L.a[1].grad_bias = m->gbp; L.a[1].out_rows = gP; L.a[1].out_bias = gbu;
        L.f = f;
        {
            Prof p(m, TFR_K_APPLY);
            launch_dense_tiles(L, true, true, m->G, m->VEC, s);
        }
        HIPCHK(hipGetLastError());
        fin_done = true;
    } else if (B > 0) {
        if ((rc = front_and_sort(m, du, di, dr, B, nullptr, d_store_ids, f, nblk, fin_done))) return rc;
        RedArgs r;
        memset(&r, 0, sizeof(r));
        r.g = m->d_g; r.err = m->d_err; r.B = B; r.D = m->D;
        r.item_abs = o.item_abs; r.reg_bias = o.reg_bias; r.lam = o.reg;
        RedPair pr;
        pr.a[0] = r;                   // whole runs go straight to their row of the flat buffer
        pr.a[0].side = 1; pr.a[0].ks = m->ks_i; pr.a[0].ps = m->ps_i; pr.a[0].other = du;
        pr.a[0].own = m->w[TFR_Q]; pr.a[0].partner = m->w[TFR_P]; pr.a[0].own_bias = m->w[TFR_BI];
        pr.a[0].grad_rows = m->gq; pr.a[0].grad_bias = m->gbq;
        pr.a[0].dense_rows = gQ; pr.a[0].dense_bias = gbi;
        pr.a[1] = r;
        pr.a[1].side = 0; pr.a[1].ks = m->ks_u; pr.a[1].ps = m->ps_u; pr.a[1].other = di;
        pr.a[1].own = m->w[TFR_P]; pr.a[1].partner = m->w[TFR_Q]; pr.a[1].own_bias = m->w[TFR_BU];
        pr.a[1].grad_rows = m->gp ? m->gp : m->gq + (size_t)m->cap * m->D; pr.a[1].grad_bias = m->gbp;
        pr.a[1].dense_rows = gP; pr.a[1].dense_bias = gbu;
        {
            Prof p(m, TFR_K_REDUCE_ITEM);
            launch_seg_reduce(pr, 2, RMODE_SCRATCH, m->G, m->VEC, s);
        }
        HIPCHK(hipGetLastError());
        ApplyPair app;                 // runs split over several reduce blocks: add their pieces, emit the row
        memset(&app, 0, sizeof(app));
        app.a[0].err = m->d_err; app.a[0].B = B; app.a[0].D = m->D; app.a[0].only_split = 1;
        app.a[0].ks = m->ks_i; app.a[0].grad_rows = m->gq; app.a[0].grad_bias = m->gbq;
        app.a[0].w = gQ; app.a[0].bias_w = gbi;
        app.a[1] = app.a[0];
        app.a[1].ks = m->ks_u; app.a[1].grad_rows = pr.a[1].grad_rows; app.a[1].grad_bias = m->gbp;
        app.a[1].w = gP; app.a[1].bias_w = gbu;
        {
            Prof p(m, TFR_K_APPLY);
            launch_apply_rows(app, 2, 2, m->G, m->VEC, s);
        }
        HIPCHK(hipGetLastError());
    }
    if (!fin_done) {
        f.nblk = nblk;
        Prof p(m, TFR_K_FINALIZE);
        launch_finalize(f, s);
    }
    HIPCHK(hipGetLastError());
    return TFR_OK;
}

int tfr_dp_apply(tfr_model* m, float* d_flat) {
    MODEL_ENTER(m);
    if (!d_flat) return fail(TFR_ERR_ARG, "dp_apply: null buffer");
    const tfr_opts& o = m->o;
    const bool adam = o.optimizer == TFR_OPT_ADAM;
    if (adam && o.adam_mode != TFR_ADAM_TF1)
        return fail(TFR_ERR_STATE, "data-parallel steps need dense semantics: Adam tf1 or SGD");
    const float alpha = adam ? o.lr * sqrtf(1.f - m->b2p) / (1.f - m->b1p) : 0.f;
    { const int rcq = settle_q(m); if (rcq) return rcq; }
    float* gP = d_flat;
    float* gQ = gP + m->U * m->D;
    float* gbu = gQ + m->I * m->D;
    float* gbi = gbu + m->U;
    float* tail = gbi + m->I;
    DensePair dp;
    memset(&dp, 0, sizeof(dp));
    DenseArgs d;
    memset(&d, 0, sizeof(d));
    d.err = m->d_err; d.D = m->D; d.opt = adam ? 0 : 1;
    d.alpha = alpha; d.b1 = o.beta1; d.b2 = o.beta2; d.eps = o.eps; d.lr = o.lr;
    dp.a[0] = d;
    dp.a[0].dense_grad = gP; dp.a[0].dense_gbias = gbu; dp.a[0].rows = m->U;
    dp.a[0].w = m->w[TFR_P]; dp.a[0].m = m->m[TFR_P]; dp.a[0].v = m->v[TFR_P];
    dp.a[0].bias_w = m->w[TFR_BU]; dp.a[0].bias_m = m->m[TFR_BU]; dp.a[0].bias_v = m->v[TFR_BU];
    dp.a[0].frozen_rows = (m->frozen >> TFR_P) & 1; dp.a[0].frozen_bias = (m->frozen >> TFR_BU) & 1;
    dp.a[1] = d;
    dp.a[1].dense_grad = gQ; dp.a[1].dense_gbias = gbi; dp.a[1].rows = m->I;
    dp.a[1].w = m->w[TFR_Q]; dp.a[1].m = m->m[TFR_Q]; dp.a[1].v = m->v[TFR_Q];
    dp.a[1].bias_w = m->w[TFR_BI]; dp.a[1].bias_m = m->m[TFR_BI]; dp.a[1].bias_v = m->v[TFR_BI];
    dp.a[1].frozen_rows = (m->frozen >> TFR_Q) & 1; dp.a[1].frozen_bias = (m->frozen >> TFR_BI) & 1;
    FinArgs f;                         // bias_global from the all-reduced {loss, reg, sum g}; rides in the sweep
    memset(&f, 0, sizeof(f));
    f.partials = tail; f.nblk = 1; f.scalars = m->scalars; f.out = nullptr;
    f.mu = m->w[TFR_MU]; f.mu_m = m->m[TFR_MU]; f.mu_v = m->v[TFR_MU]; f.err = m->d_err;
    f.update_mu = !((m->frozen >> TFR_MU) & 1); f.opt = adam ? 0 : 1;
    f.alpha = alpha; f.b1 = o.beta1; f.b2 = o.beta2; f.eps = o.eps; f.lr = o.lr;
    f.clear_partials = 1;                                        // scalars consumed: clean for the next step
    {
        Prof p(m, TFR_K_APPLY);
        launch_adam_dense(dp, 2, m->G, m->VEC, m->stream, &f);
    }
    HIPCHK(hipGetLastError());
    if (adam) {
        m->b1p *= o.beta1;
        m->b2p *= o.beta2;
    }
    m->step += 1;
    return TFR_OK;
}

int tfr_staged_ids_devptr(tfr_model* m, void** ptr, int64_t* n) {
    MODEL_ENTER(m);
    if (ptr) *ptr = m->d_ids;
    if (n) *n = m->n_ids;
    return TFR_OK;
}

}  // extern "C"

// ---- second-order FM (BASELINE config 5) -------------------------------------
// The handle wraps a regular model: V = its user_features [F,D], W = its user_bias [F],
// mu = its bias_global; so the FM backward reuses the radix sort, the segmented reduce (K3) and
// the fused SGD / lazy-Adam apply unchanged.
struct tfr_fm {
    tfr_model* m = nullptr;
    int64_t cap_rows = 0, cap_nnz = 0, s_cap = 0;
    int64_t* d_indptr = nullptr;
    int32_t* d_indices = nullptr;
    float *d_data = nullptr, *d_y = nullptr, *d_out = nullptr, *s_rows = nullptr;
    int4* ent = nullptr; int64_t ent_cap = 0;            // per non-zero {row, g x, lam - g x^2, -} of the training step
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

static int fm_stage_csr(tfr_fm* f, const int64_t* indptr, const int32_t* indices, const float* data,
                        const float* y, int64_t n_rows, int64_t* nnz_out) {
    tfr_model* m = f->m;
    const int64_t nnz = indptr[n_rows];
    if (nnz < 0 || indptr[0] != 0) return fail(TFR_ERR_ARG, "indptr must start at 0 and be non-decreasing");
    for (int64_t r = 0; r < n_rows; ++r)
        if (indptr[r + 1] < indptr[r]) return fail(TFR_ERR_ARG, "indptr must be non-decreasing");
    if (nnz > 0 && (!indices || !data)) return fail(TFR_ERR_ARG, "null indices/data");
    if (n_rows > f->cap_rows) {
        HIPCHK(hipStreamSynchronize(m->stream));
        dfree(f->d_indptr); dfree(f->d_out); dfree(f->d_y);
        f->d_indptr = nullptr; f->d_out = nullptr; f->d_y = nullptr; f->cap_rows = 0;
        int rc;
        if ((rc = dmalloc(&f->d_indptr, (size_t)n_rows + 1))) return rc;
        if ((rc = dmalloc(&f->d_out, (size_t)n_rows))) return rc;
        if ((rc = dmalloc(&f->d_y, (size_t)n_rows))) return rc;
        f->cap_rows = n_rows;
    }
    if (nnz > f->cap_nnz) {
        HIPCHK(hipStreamSynchronize(m->stream));
        dfree(f->d_indices); dfree(f->d_data);
        f->d_indices = nullptr; f->d_data = nullptr; f->cap_nnz = 0;
        int rc;
        if ((rc = dmalloc(&f->d_indices, (size_t)nnz))) return rc;
        if ((rc = dmalloc(&f->d_data, (size_t)nnz))) return rc;
        f->cap_nnz = nnz;
    }
    HIPCHK(hipMemcpyAsync(f->d_indptr, indptr, (size_t)(n_rows + 1) * 8, hipMemcpyHostToDevice, m->stream));
    if (nnz > 0) {
        HIPCHK(hipMemcpyAsync(f->d_indices, indices, (size_t)nnz * 4, hipMemcpyHostToDevice, m->stream));
        HIPCHK(hipMemcpyAsync(f->d_data, data, (size_t)nnz * 4, hipMemcpyHostToDevice, m->stream));
    }
    if (y) HIPCHK(hipMemcpyAsync(f->d_y, y, (size_t)n_rows * 4, hipMemcpyHostToDevice, m->stream));
    *nnz_out = nnz;
    return TFR_OK;
}

// k_fm_forward variant (launch_fm): TFR_FM_VARIANT=<bits> overrides for A/B
static int fm_variant(const tfr_model* m) {
    static int ov = -2;
    if (ov == -2) { const char* e = getenv("TFR_FM_VARIANT"); ov = e ? atoi(e) : -1; }
    if (ov >= 0) return ov;
    return ((size_t)m->U * m->D * 4 >= ((size_t)128 << 20)) ? 2 : 0;      // V beyond half the Infinity Cache: stream it
}

static int fm_forward_core(tfr_fm* f, const int64_t* d_indptr, const int32_t* d_indices, const float* d_data,
                           int64_t n_rows, float* d_out) {
    tfr_model* m = f->m;
    FmArgs a;
    memset(&a, 0, sizeof(a));
    a.V = m->w[TFR_P]; a.W = m->w[TFR_BU]; a.mu = m->w[TFR_MU];
    a.indptr = d_indptr; a.indices = d_indices; a.data = d_data; a.out = d_out; a.err = m->d_err;
    a.n_rows = n_rows; a.F = m->U; a.D = m->D;
    a.variant = fm_variant(m);
    (void)hipEventRecord(f->ev0, m->stream);
    launch_fm(a, false, m->G, m->VEC, fm_grid(n_rows, m->G, false), m->stream);
    (void)hipEventRecord(f->ev1, m->stream);
    HIPCHK(hipGetLastError());
    return TFR_OK;
}

// one minibatch: forward (+ s, g, per-entry coefficients) -> radix sort of the non-zeros by
// feature id -> segmented reduce with fused SGD / lazy Adam on V and W -> mu
static int fm_train_core(tfr_fm* f, const int64_t* d_indptr, const int32_t* d_indices, const float* d_data,
                         const float* d_y, int64_t n_rows, int64_t nnz, float* d_pred, float* out3) {
    tfr_model* m = f->m;
    const tfr_opts& o = m->o;
    const bool adam = o.optimizer == TFR_OPT_ADAM;
    if (adam && o.adam_mode != TFR_ADAM_LAZY) return fail(TFR_ERR_STATE, "FM training supports SGD and lazy Adam");
    int rc;
    if ((rc = ensure_capacity(m, nnz > 0 ? nnz : 1))) return rc;
    if (n_rows * m->D > f->s_cap) {
        HIPCHK(hipStreamSynchronize(m->stream));
        dfree(f->s_rows);
        f->s_rows = nullptr; f->s_cap = 0;
        if ((rc = dmalloc(&f->s_rows, (size_t)n_rows * m->D))) return rc;
        f->s_cap = n_rows * m->D;
    }
    if (nnz > f->ent_cap) {
        HIPCHK(hipStreamSynchronize(m->stream));
        dfree(f->ent);
        f->ent = nullptr; f->ent_cap = 0;
        if ((rc = dmalloc(&f->ent, (size_t)nnz))) return rc;
        f->ent_cap = nnz;
    }
    const float alpha = adam ? o.lr * sqrtf(1.f - m->b2p) / (1.f - m->b1p) : 0.f;
    hipStream_t s = m->stream;
    FmArgs a;
    memset(&a, 0, sizeof(a));
    a.V = m->w[TFR_P]; a.W = m->w[TFR_BU]; a.mu = m->w[TFR_MU];
    a.indptr = d_indptr; a.indices = d_indices; a.data = d_data; a.out = d_pred; a.err = m->d_err;
    a.y = d_y; a.s_rows = f->s_rows; a.ent = f->ent; a.partials = m->partials;
    a.n_rows = n_rows; a.F = m->U; a.D = m->D; a.loss = o.loss; a.lam = o.reg;
    a.variant = fm_variant(m);
    const int grid = fm_grid(n_rows, m->G, true);
    (void)hipEventRecord(f->ev0, s);
    {
        Prof p(m, TFR_K_FORWARD);
        launch_fm(a, true, m->G, m->VEC, grid, s);
    }
    HIPCHK(hipGetLastError());
    if (nnz > 0) {
        const int32_t* keys[2] = {d_indices, nullptr};
        const int bits[2] = {m->bits_u, 0};
        int32_t* ks[2] = {m->ks_u, nullptr};
        int32_t* ps[2] = {m->ps_u, nullptr};
        {
            Prof p(m, TFR_K_SORT);
            if ((rc = radix_sort_columns(m, 1, keys, bits, ks, ps, nnz))) return rc;
        }
        RedPair pr;
        RedArgs& r = pr.a[0];
        memset(&r, 0, sizeof(r));
        r.err = m->d_err; r.B = nnz; r.D = m->D; r.side = 0; r.reg_bias = 1;
        r.lam = o.reg; r.alpha = alpha; r.b1 = o.beta1; r.b2 = o.beta2; r.eps = o.eps; r.lr = o.lr;
        r.ks = m->ks_u; r.ps = m->ps_u; r.ent = f->ent;
        r.own = m->w[TFR_P]; r.partner = f->s_rows; r.own_bias = m->w[TFR_BU];
        r.own_w = m->w[TFR_P]; r.m = m->m[TFR_P]; r.v = m->v[TFR_P];
        r.bias_w = m->w[TFR_BU]; r.bias_m = m->m[TFR_BU]; r.bias_v = m->v[TFR_BU];
        r.grad_rows = m->gq + (size_t)m->cap * m->D; r.grad_bias = m->gbp;
        {
            Prof p(m, TFR_K_REDUCE_USER);
            launch_seg_reduce(pr, 1, adam ? RMODE_ADAM : RMODE_SGD, m->G, m->VEC, s);
        }
        HIPCHK(hipGetLastError());
        ApplyPair app;
        ApplyArgs& ap = app.a[0];
        memset(&ap, 0, sizeof(ap));
        ap.err = m->d_err; ap.B = nnz; ap.D = m->D; ap.only_split = 1;
        ap.alpha = alpha; ap.b1 = o.beta1; ap.b2 = o.beta2; ap.eps = o.eps; ap.lr = o.lr;
        ap.ks = m->ks_u; ap.grad_rows = r.grad_rows; ap.grad_bias = m->gbp;
        ap.w = m->w[TFR_P]; ap.m = m->m[TFR_P]; ap.v = m->v[TFR_P];
        ap.bias_w = m->w[TFR_BU]; ap.bias_m = m->m[TFR_BU]; ap.bias_v = m->v[TFR_BU];
        {
            Prof p(m, TFR_K_APPLY);
            launch_apply_rows(app, 1, adam ? 0 : 1, m->G, m->VEC, s);
        }
        HIPCHK(hipGetLastError());
    }
    FinArgs fin;
    memset(&fin, 0, sizeof(fin));
    fin.partials = m->partials; fin.nblk = grid; fin.scalars = m->scalars; fin.out = out3;
    fin.mu = m->w[TFR_MU]; fin.mu_m = m->m[TFR_MU]; fin.mu_v = m->v[TFR_MU]; fin.err = m->d_err;
    fin.update_mu = 1; fin.opt = adam ? 0 : 1;
    fin.alpha = alpha; fin.b1 = o.beta1; fin.b2 = o.beta2; fin.eps = o.eps; fin.lr = o.lr;
    {
        Prof p(m, TFR_K_FINALIZE);
        launch_finalize(fin, s);
    }
    (void)hipEventRecord(f->ev1, s);
    HIPCHK(hipGetLastError());
    if (adam) {
        m->b1p *= o.beta1;
        m->b2p *= o.beta2;
    }
    m->step += 1;
    return TFR_OK;
}

extern "C" {

const char* tfr_fm_last_error(void) { return g_err; }

int tfr_fm_destroy(tfr_fm* f) {
    if (!f) return TFR_OK;
    if (f->m) {
        (void)hipSetDevice(f->m->device);
        (void)hipStreamSynchronize(f->m->stream);
    }
    dfree(f->d_indptr); dfree(f->d_indices); dfree(f->d_data); dfree(f->d_y); dfree(f->d_out); dfree(f->s_rows); dfree(f->ent);
    if (f->ev0) (void)hipEventDestroy(f->ev0);
    if (f->ev1) (void)hipEventDestroy(f->ev1);
    tfr_destroy(f->m);
    delete f;
    return TFR_OK;
}

int tfr_fm_create(tfr_fm** out, int64_t n_features, int32_t dim, const tfr_opts* opts) {
    if (!out) return fail(TFR_ERR_ARG, "out is null");
    *out = nullptr;
    tfr_opts o;
    if (opts) o = *opts;
    else {
        tfr_default_opts(&o);
        o.loss = TFR_LOSS_NLL;          // fm.py:104 task = 'classification'
        o.optimizer = TFR_OPT_SGD;
        o.lr = 0.01f; o.reg = 0.0f;
    }
    if (o.optimizer == TFR_OPT_ADAM) o.adam_mode = TFR_ADAM_LAZY;
    tfr_fm* f = new (std::nothrow) tfr_fm();
    if (!f) return fail(TFR_ERR_NOMEM, "host allocation failed");
    int rc = tfr_create(&f->m, n_features, 1, dim, &o);
    if (rc == TFR_OK && (hipEventCreate(&f->ev0) != hipSuccess || hipEventCreate(&f->ev1) != hipSuccess))
        rc = fail(TFR_ERR_HIP, "event creation failed");
    if (rc) {
        char keep[512];
        strncpy(keep, g_err, sizeof(keep));
        tfr_fm_destroy(f);
        strncpy(g_err, keep, sizeof(g_err));
        return rc;
    }
    *out = f;
    return TFR_OK;
}

int tfr_fm_set(tfr_fm* f, float mu, const float* W, const float* V) {
    if (!f || !W || !V) return fail(TFR_ERR_ARG, "null argument");
    int rc = tfr_set_table(f->m, TFR_MU, &mu, 1);
    if (!rc) rc = tfr_set_table(f->m, TFR_BU, W, f->m->U);
    if (!rc) rc = tfr_set_table(f->m, TFR_P, V, f->m->U * f->m->D);
    return rc;
}

int tfr_fm_get(tfr_fm* f, float* mu, float* W, float* V) {
    if (!f) return fail(TFR_ERR_ARG, "null model");
    int rc = TFR_OK;
    if (mu) rc = tfr_get_table(f->m, TFR_MU, mu, 1);
    if (!rc && W) rc = tfr_get_table(f->m, TFR_BU, W, f->m->U);
    if (!rc && V) rc = tfr_get_table(f->m, TFR_P, V, f->m->U * f->m->D);
    return rc;
}

int tfr_fm_init(tfr_fm* f, uint64_t seed, float stddev) {
    if (!f) return fail(TFR_ERR_ARG, "null model");
    tfr_model* m = f->m;
    HIPCHK(hipSetDevice(m->device));
    launch_init_trunc_normal(m->w[TFR_P], m->n[TFR_P], stddev, seed * 2 + 0, m->stream);
    launch_init_trunc_normal(m->w[TFR_BU], m->n[TFR_BU], stddev, seed * 2 + 1, m->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(m->w[TFR_MU], 0, 4, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return TFR_OK;
}

int tfr_fm_forward_dev(tfr_fm* f, const int64_t* d_indptr, const int32_t* d_indices, const float* d_data,
                       int64_t n_rows, float* d_out) {
    if (!f || n_rows < 0 || (n_rows > 0 && (!d_indptr || !d_out))) return fail(TFR_ERR_ARG, "bad arguments");
    HIPCHK(hipSetDevice(f->m->device));
    if (n_rows == 0) return TFR_OK;
    int rc = ensure_capacity(f->m, 1);
    if (rc) return rc;
    return fm_forward_core(f, d_indptr, d_indices, d_data, n_rows, d_out);
}

int tfr_fm_forward(tfr_fm* f, const int64_t* indptr, const int32_t* indices, const float* data,
                   int64_t n_rows, float* out) {
    if (!f || n_rows < 0 || (n_rows > 0 && (!indptr || !out))) return fail(TFR_ERR_ARG, "bad arguments");
    tfr_model* m = f->m;
    HIPCHK(hipSetDevice(m->device));
    if (n_rows == 0) return TFR_OK;
    int64_t nnz = 0;
    int rc = fm_stage_csr(f, indptr, indices, data, nullptr, n_rows, &nnz);
    if (rc) return rc;
    if ((rc = fm_forward_core(f, f->d_indptr, f->d_indices, f->d_data, n_rows, f->d_out))) return rc;
    HIPCHK(hipMemcpyAsync(out, f->d_out, (size_t)n_rows * 4, hipMemcpyDeviceToHost, m->stream));
    return check_device_error(m);
}

int tfr_fm_train_step_dev(tfr_fm* f, const int64_t* d_indptr, const int32_t* d_indices, const float* d_data,
                          const float* d_y, int64_t n_rows, int64_t nnz, float* d_pred) {
    if (!f || n_rows < 1 || nnz < 0 || !d_indptr || !d_y) return fail(TFR_ERR_ARG, "bad arguments");
    HIPCHK(hipSetDevice(f->m->device));
    return fm_train_core(f, d_indptr, d_indices, d_data, d_y, n_rows, nnz, d_pred, nullptr);
}

int tfr_fm_train_step(tfr_fm* f, const int64_t* indptr, const int32_t* indices, const float* data, const float* y,
                      int64_t n_rows, float* pred_out, float* loss_out) {
    if (!f || n_rows < 1 || !indptr || !y) return fail(TFR_ERR_ARG, "bad arguments");
    tfr_model* m = f->m;
    HIPCHK(hipSetDevice(m->device));
    int64_t nnz = 0;
    int rc = fm_stage_csr(f, indptr, indices, data, y, n_rows, &nnz);
    if (rc) return rc;
    const int64_t step0 = m->step;
    const float b1p0 = m->b1p, b2p0 = m->b2p;
    if ((rc = fm_train_core(f, f->d_indptr, f->d_indices, f->d_data, f->d_y, n_rows, nnz, f->d_out, nullptr))) return rc;
    float sc[4] = {0.f, 0.f, 0.f, 0.f};
    if (pred_out) HIPCHK(hipMemcpyAsync(pred_out, f->d_out, (size_t)n_rows * 4, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipMemcpyAsync(sc, m->scalars, 16, hipMemcpyDeviceToHost, m->stream));
    if ((rc = check_device_error(m))) {
        rollback_step(m, step0, b1p0, b2p0);
        return rc;
    }
    if (loss_out) *loss_out = sc[0];
    return TFR_OK;
}

int tfr_fm_sync(tfr_fm* f, float* last_kernel_ms) {
    if (!f) return fail(TFR_ERR_ARG, "null model");
    HIPCHK(hipSetDevice(f->m->device));
    int rc = check_device_error(f->m);
    if (rc) return rc;
    if (last_kernel_ms) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, f->ev0, f->ev1) != hipSuccess) ms = 0.f;
        *last_kernel_ms = ms;
    }
    return TFR_OK;
}

}  // extern "C"
