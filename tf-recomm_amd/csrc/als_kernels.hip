// als_kernels.hip - the ALS baseline of als3.py on the GPU, float64 like the reference.
//   fit_user / fit_work (als3.py:67-108): for one user u with rated works J, ratings R:
//       A = V[J]^T V[J] + lambda * N * I        b = (R - W_work[J] - (W_user[u] + bias)) . V[J]
//       U[u] = solve(A, b)                      W_user[u] = mean(R - U[u].V[J]^T - W_work[J]) / (1 + lambda) - bias
//   Inside one half-sweep every entity only reads the OTHER side's tables, so all users (then all
//   works) are independent: one 256-thread block per entity builds the d x d normal equations from
//   32-row tiles staged in LDS, factors them (Cholesky; A is SPD because of the lambda*N*I term) and
//   back-substitutes.  Lists longer than 512 ratings are cut into chunks: k_als_partial builds each chunk's
//   partial sums in its own block, the entity's block adds them in list order.  Fixed summation order ->
//   deterministic.  d <= 32.
//   predict (als3.py:110-113) at explicit (user, work) pairs - the dense U V^T is never formed.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <new>
#include <vector>
#include <algorithm>
#include "tfrecomm.h"

namespace {

constexpr int ALS_MAXD = 32;
constexpr int ALS_TILE = 32;         // ratings staged per LDS tile (128 measured no faster: not latency-bound)

struct AlsFitArgs {
    const int32_t* list; int64_t n_list;               // entities to fit
    const int64_t* ptr; const int32_t* ids; const double* vals;   // their rating lists (other-side ids, ratings)
    double* own; double* w_own; const double* other; const double* w_other;
    double bias, lambda;
    int32_t d;
    // long rating lists are cut into chunks whose partial normal equations are built by other blocks first
    // (k_als_partial): cfirst[entity] = its first chunk or -1, ccount[entity] chunks, in list order
    const int32_t* cfirst; const int32_t* ccount;
    const int32_t* chunk_ent; const int64_t* chunk_lo; const int64_t* chunk_hi; int64_t n_chunks;
    double* partial;                                   // [n_chunks][d*d + d]
};

// normal-equation sums over the ratings [lo, hi) of one entity: acc[q] += sum_k rows[k][r] rows[k][c] for the
// A entries t = tid + 256 q, accb += sum_k coef[k] rows[k][tid]; 32-row tiles staged in LDS; 256 threads
__device__ __forceinline__ void als_accumulate(const AlsFitArgs& a, int64_t lo, int64_t hi, double b0, double (&acc)[4], double& accb,
                                               double (*rows)[ALS_MAXD + 1], double* coef) {
    const int tid = threadIdx.x, d = a.d;
    for (int64_t s = lo; s < hi; s += ALS_TILE) {
        const int nk = (int)((hi - s < ALS_TILE) ? hi - s : ALS_TILE);
        for (int t = tid; t < nk * d; t += 256) {
            const int k = t / d, c = t % d;
            rows[k][c] = a.other[(size_t)a.ids[s + k] * d + c];
        }
        if (tid < nk) coef[tid] = a.vals[s + tid] - a.w_other[a.ids[s + tid]] - b0;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = tid + 256 * q;
            if (t < d * d) {
                const int r = t / d, c = t % d;
                double sacc = acc[q];
                for (int k = 0; k < nk; ++k) sacc += rows[k][r] * rows[k][c];
                acc[q] = sacc;
            }
        }
        if (tid < d) {
            double sb = accb;
            for (int k = 0; k < nk; ++k) sb += coef[k] * rows[k][tid];
            accb = sb;
        }
        __syncthreads();
    }
}

// partial normal equations of one chunk of a long rating list
__global__ __launch_bounds__(256) void k_als_partial(AlsFitArgs a) {
    __shared__ double rows[ALS_TILE][ALS_MAXD + 1];
    __shared__ double coef[ALS_TILE];
    const int tid = threadIdx.x, d = a.d;
    for (int64_t c = blockIdx.x; c < a.n_chunks; c += gridDim.x) {
        const int32_t idx = a.chunk_ent[c];
        const double b0 = a.w_own[idx] + a.bias;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        double accb = 0.0;
        __syncthreads();
        als_accumulate(a, a.chunk_lo[c], a.chunk_hi[c], b0, acc, accb, rows, coef);
        double* pp = a.partial + (size_t)c * (d * d + d);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int t = tid + 256 * q; if (t < d * d) pp[t] = acc[q]; }
        if (tid < d) pp[d * d + tid] = accb;
    }
}

__global__ __launch_bounds__(256) void k_als_fit(AlsFitArgs a) {
    __shared__ double A[ALS_MAXD][ALS_MAXD + 1];
    __shared__ double rows[ALS_TILE][ALS_MAXD + 1];
    __shared__ double coef[ALS_TILE];
    __shared__ double bvec[ALS_MAXD], xvec[ALS_MAXD];
    __shared__ double red[256];
    const int tid = threadIdx.x, d = a.d;
    for (int64_t e = blockIdx.x; e < a.n_list; e += gridDim.x) {
        const int32_t idx = a.list[e];
        const int64_t lo = a.ptr[idx], hi = a.ptr[idx + 1];
        const int64_t N = hi - lo;
        const double b0 = a.w_own[idx] + a.bias;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};           // A entries t = tid + 256*q  (d*d <= 1024)
        double accb = 0.0;                               // b entry tid (< d)
        __syncthreads();
        const int32_t nch = a.ccount ? a.ccount[idx] : 0;
        if (nch > 0) {                                   // a long list: add the chunks' partial sums, in list order
            const int32_t c0 = a.cfirst[idx];
            for (int32_t ch = 0; ch < nch; ++ch) {
                const double* pp = a.partial + (size_t)(c0 + ch) * (d * d + d);
#pragma unroll
                for (int q = 0; q < 4; ++q) { const int t = tid + 256 * q; if (t < d * d) acc[q] += pp[t]; }
                if (tid < d) accb += pp[d * d + tid];
            }
        } else {
            als_accumulate(a, lo, hi, b0, acc, accb, rows, coef);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = tid + 256 * q;
            if (t < d * d) {
                const int r = t / d, c = t % d;
                A[r][c] = acc[q] + ((r == c) ? a.lambda * (double)N : 0.0);
            }
        }
        if (tid < d) bvec[tid] = accb;
        __syncthreads();
        // Cholesky A = L L^T (in place, lower triangle) and the two triangular solves, by ONE wave: lane r owns row r, a column
        // step is j LDS reads of row j (broadcast) and of the lane's own row - no block barrier (a wave's LDS accesses are made
        // in program order; volatile keeps the compiler from moving a read of another lane's element over the write it follows).
        // The block-wide version (thread 0 alone on the diagonal and in the substitutions, a barrier pair per column) cost
        // ~50 us per entity, most of the sweep.  Same operations in the same order per element.
        if (tid < 64) {
            const int r = tid;
            volatile double (*L)[ALS_MAXD + 1] = A;
            for (int j = 0; j < d; ++j) {
                double s = 0.0;
                if (r >= j && r < d) {
                    s = L[r][j];
                    for (int k = 0; k < j; ++k) s -= L[r][k] * L[j][k];
                }
                const double piv = sqrt(__shfl(s, j, 64));
                if (r == j) L[j][j] = piv;
                else if (r > j && r < d) L[r][j] = s / piv;
            }
            // L y = b, column by column: lane i finishes y_i, the lanes below take it off their right-hand sides
            double y = (r < d) ? bvec[r] : 0.0;
            for (int i = 0; i < d; ++i) {
                const double yi = __shfl(y / ((r == i) ? L[i][i] : 1.0), i, 64);
                if (r == i) y = yi;
                else if (r > i && r < d) y -= L[r][i] * yi;
            }
            // L^T x = y, from the last column up: column i of L^T is row i of L
            for (int i = d - 1; i >= 0; --i) {
                const double xi = __shfl(y / ((r == i) ? L[i][i] : 1.0), i, 64);
                if (r == i) y = xi;
                else if (r < i) y -= L[i][r] * xi;
            }
            if (r < d) xvec[r] = y;
        }
        __syncthreads();
        if (tid < d) a.own[(size_t)idx * d + tid] = xvec[tid];
        // W_own = mean(R - x.V[J] - W_other[J]) / (1 + lambda) - bias
        double part = 0.0;
        for (int64_t k = lo + tid; k < hi; k += 256) {
            const int32_t j = a.ids[k];
            const double* v = a.other + (size_t)j * d;
            double dot = 0.0;
            for (int c = 0; c < d; ++c) dot += xvec[c] * v[c];
            part += a.vals[k] - dot - a.w_other[j];
        }
        red[tid] = part;
        __syncthreads();
        for (int o = 128; o >= 1; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        if (tid == 0) a.w_own[idx] = red[0] / (double)N / (1.0 + a.lambda) - a.bias;
    }
}

__global__ __launch_bounds__(256) void k_als_predict(const double* U, const double* V, const double* Wu, const double* Ww,
                                                     double bias, const int32_t* u, const int32_t* w, int64_t n, int d,
                                                     double* out) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const double* pu = U + (size_t)u[k] * d;
        const double* pv = V + (size_t)w[k] * d;
        double s = 0.0;
        for (int c = 0; c < d; ++c) s += pu[c] * pv[c];
        out[k] = s + Wu[u[k]] + Ww[w[k]] + bias;
    }
}

thread_local char g_als_err[512] = "";
int als_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_als_err, sizeof(g_als_err), fmt, ap);
    va_end(ap);
    return code;
}
#define ALSCHK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return als_fail(e_ == hipErrorOutOfMemory ? TFR_ERR_NOMEM : TFR_ERR_HIP, \
                                              "%s: %s", #expr, hipGetErrorString(e_));                \
    } while (0)

void als_free(void* p) { if (p) (void)hipFree(p); }

}  // namespace

struct tfr_als {
    int64_t nu = 0, nw = 0, n = 0, n_users = 0, n_works = 0;
    int32_t d = 0;
    double lambda = 0.1, bias = 0.0;
    int device = 0;
    hipStream_t stream = nullptr;
    double *U = nullptr, *V = nullptr, *Wu = nullptr, *Ww = nullptr;
    int64_t *ptr_u = nullptr, *ptr_w = nullptr;
    int32_t *ids_u = nullptr, *ids_w = nullptr, *users = nullptr, *works = nullptr;
    double *val_u = nullptr, *val_w = nullptr;
    // chunk tables for long rating lists, per side (0 = users, 1 = works)
    int32_t *cfirst[2] = {nullptr, nullptr}, *ccount[2] = {nullptr, nullptr}, *chunk_ent[2] = {nullptr, nullptr};
    int64_t *chunk_lo[2] = {nullptr, nullptr}, *chunk_hi[2] = {nullptr, nullptr};
    int64_t n_chunks[2] = {0, 0};
    double* partial = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

extern "C" {

const char* tfr_als_last_error(void) { return g_als_err; }

int tfr_als_destroy(tfr_als* m) {
    if (!m) return TFR_OK;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    void* ps[] = {m->U, m->V, m->Wu, m->Ww, m->ptr_u, m->ptr_w, m->ids_u, m->ids_w, m->users, m->works, m->val_u, m->val_w};
    for (void* p : ps) als_free(p);
    for (int z = 0; z < 2; ++z) { als_free(m->cfirst[z]); als_free(m->ccount[z]); als_free(m->chunk_ent[z]); als_free(m->chunk_lo[z]); als_free(m->chunk_hi[z]); }
    als_free(m->partial);
    if (m->ev0) (void)hipEventDestroy(m->ev0);
    if (m->ev1) (void)hipEventDestroy(m->ev1);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
    return TFR_OK;
}

int tfr_als_create(tfr_als** out, int64_t nb_users, int64_t nb_works, int32_t nb_components, double lambda_, int32_t device) {
    if (!out) return als_fail(TFR_ERR_ARG, "out is null");
    *out = nullptr;
    if (nb_users < 1 || nb_works < 1 || nb_users > 0x7fffffffLL || nb_works > 0x7fffffffLL || nb_components < 1 || nb_components > ALS_MAXD)
        return als_fail(TFR_ERR_ARG, "need 1 <= nb_components <= %d and positive int32 table sizes", ALS_MAXD);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return als_fail(TFR_ERR_HIP, "no HIP device available - this library has no CPU path");
    if (device < 0 || device >= ndev) return als_fail(TFR_ERR_ARG, "device %d not in [0,%d)", device, ndev);
    ALSCHK(hipSetDevice(device));
    tfr_als* m = new (std::nothrow) tfr_als();
    if (!m) return als_fail(TFR_ERR_NOMEM, "host allocation failed");
    m->nu = nb_users; m->nw = nb_works; m->d = nb_components; m->lambda = lambda_; m->device = device;
    hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&m->U, (size_t)nb_users * m->d * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&m->V, (size_t)nb_works * m->d * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&m->Wu, (size_t)nb_users * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&m->Ww, (size_t)nb_works * 8);
    if (e == hipSuccess) e = hipMemsetAsync(m->U, 0, (size_t)nb_users * m->d * 8, m->stream);
    if (e == hipSuccess) e = hipMemsetAsync(m->V, 0, (size_t)nb_works * m->d * 8, m->stream);
    if (e == hipSuccess) e = hipMemsetAsync(m->Wu, 0, (size_t)nb_users * 8, m->stream);
    if (e == hipSuccess) e = hipMemsetAsync(m->Ww, 0, (size_t)nb_works * 8, m->stream);
    if (e == hipSuccess) e = hipEventCreate(&m->ev0);
    if (e == hipSuccess) e = hipEventCreate(&m->ev1);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    if (e != hipSuccess) {
        als_fail(e == hipErrorOutOfMemory ? TFR_ERR_NOMEM : TFR_ERR_HIP, "als_create: %s", hipGetErrorString(e));
        char keep[512];
        strncpy(keep, g_als_err, sizeof(keep));
        tfr_als_destroy(m);
        strncpy(g_als_err, keep, sizeof(g_als_err));
        return e == hipErrorOutOfMemory ? TFR_ERR_NOMEM : TFR_ERR_HIP;
    }
    *out = m;
    return TFR_OK;
}

int tfr_als_set(tfr_als* m, const double* U, const double* V, const double* W_user, const double* W_work) {
    if (!m || !U || !V || !W_user || !W_work) return als_fail(TFR_ERR_ARG, "null argument");
    ALSCHK(hipSetDevice(m->device));
    ALSCHK(hipMemcpyAsync(m->U, U, (size_t)m->nu * m->d * 8, hipMemcpyHostToDevice, m->stream));
    ALSCHK(hipMemcpyAsync(m->V, V, (size_t)m->nw * m->d * 8, hipMemcpyHostToDevice, m->stream));
    ALSCHK(hipMemcpyAsync(m->Wu, W_user, (size_t)m->nu * 8, hipMemcpyHostToDevice, m->stream));
    ALSCHK(hipMemcpyAsync(m->Ww, W_work, (size_t)m->nw * 8, hipMemcpyHostToDevice, m->stream));
    ALSCHK(hipStreamSynchronize(m->stream));
    return TFR_OK;
}

int tfr_als_get(tfr_als* m, double* U, double* V, double* W_user, double* W_work, double* bias) {
    if (!m) return als_fail(TFR_ERR_ARG, "null model");
    ALSCHK(hipSetDevice(m->device));
    if (U) ALSCHK(hipMemcpyAsync(U, m->U, (size_t)m->nu * m->d * 8, hipMemcpyDeviceToHost, m->stream));
    if (V) ALSCHK(hipMemcpyAsync(V, m->V, (size_t)m->nw * m->d * 8, hipMemcpyDeviceToHost, m->stream));
    if (W_user) ALSCHK(hipMemcpyAsync(W_user, m->Wu, (size_t)m->nu * 8, hipMemcpyDeviceToHost, m->stream));
    if (W_work) ALSCHK(hipMemcpyAsync(W_work, m->Ww, (size_t)m->nw * 8, hipMemcpyDeviceToHost, m->stream));
    ALSCHK(hipStreamSynchronize(m->stream));
    if (bias) *bias = m->bias;
    return TFR_OK;
}

// X = (user_ids, work_ids), y: the training ratings (als3.py:20-29).  Builds the per-user and per-work
// rating lists in insertion order (als3.py:36-55), bias = mean(y), and the sweep sets.
int tfr_als_load(tfr_als* m, const int64_t* user_ids, const int64_t* work_ids, const double* y, int64_t n) {
    if (!m || n < 1 || !user_ids || !work_ids || !y) return als_fail(TFR_ERR_ARG, "load: need n >= 1 and non-null columns");
    ALSCHK(hipSetDevice(m->device));
    for (int64_t k = 0; k < n; ++k)
        if (user_ids[k] < 0 || user_ids[k] >= m->nu || work_ids[k] < 0 || work_ids[k] >= m->nw)
            return als_fail(TFR_ERR_OOB, "rating %lld: id out of range", (long long)k);
    double sum = 0.0;
    for (int64_t k = 0; k < n; ++k) sum += y[k];
    // numpy's mean uses pairwise summation; reproduce its value exactly enough by summing in long double
    long double ls = 0.0L;
    for (int64_t k = 0; k < n; ++k) ls += (long double)y[k];
    (void)sum;
    m->bias = (double)(ls / (long double)n);
    auto build = [&](const int64_t* key, const int64_t* oth, int64_t rows, std::vector<int64_t>& ptr,
                     std::vector<int32_t>& ids, std::vector<double>& vals, std::vector<int32_t>& list) {
        ptr.assign((size_t)rows + 1, 0);
        for (int64_t k = 0; k < n; ++k) ptr[(size_t)key[k] + 1]++;
        for (int64_t r = 0; r < rows; ++r) ptr[(size_t)r + 1] += ptr[(size_t)r];
        std::vector<int64_t> cur(ptr.begin(), ptr.end() - 1);
        ids.resize((size_t)n); vals.resize((size_t)n);
        std::vector<char> nz((size_t)rows, 0);
        for (int64_t k = 0; k < n; ++k) {                 // stable: insertion order inside each list
            const int64_t p = cur[(size_t)key[k]]++;
            ids[(size_t)p] = (int32_t)oth[k];
            vals[(size_t)p] = y[k];
            if (y[k] != 0.0) nz[(size_t)key[k]] = 1;      // als3.py:29: entities with a non-zero rating
        }
        list.clear();
        for (int64_t r = 0; r < rows; ++r) if (nz[(size_t)r]) list.push_back((int32_t)r);
    };
    std::vector<int64_t> pu, pw;
    std::vector<int32_t> iu, iw, lu, lw;
    std::vector<double> vu, vw;
    build(user_ids, work_ids, m->nu, pu, iu, vu, lu);
    build(work_ids, user_ids, m->nw, pw, iw, vw, lw);
    ALSCHK(hipStreamSynchronize(m->stream));
    void* old[] = {m->ptr_u, m->ptr_w, m->ids_u, m->ids_w, m->users, m->works, m->val_u, m->val_w};
    for (void* p : old) als_free(p);
    m->ptr_u = m->ptr_w = nullptr; m->ids_u = m->ids_w = m->users = m->works = nullptr; m->val_u = m->val_w = nullptr;
    ALSCHK(hipMalloc((void**)&m->ptr_u, pu.size() * 8));
    ALSCHK(hipMalloc((void**)&m->ptr_w, pw.size() * 8));
    ALSCHK(hipMalloc((void**)&m->ids_u, (size_t)n * 4));
    ALSCHK(hipMalloc((void**)&m->ids_w, (size_t)n * 4));
    ALSCHK(hipMalloc((void**)&m->val_u, (size_t)n * 8));
    ALSCHK(hipMalloc((void**)&m->val_w, (size_t)n * 8));
    ALSCHK(hipMalloc((void**)&m->users, std::max<size_t>(1, lu.size()) * 4));
    ALSCHK(hipMalloc((void**)&m->works, std::max<size_t>(1, lw.size()) * 4));
    ALSCHK(hipMemcpy(m->ptr_u, pu.data(), pu.size() * 8, hipMemcpyHostToDevice));
    ALSCHK(hipMemcpy(m->ptr_w, pw.data(), pw.size() * 8, hipMemcpyHostToDevice));
    ALSCHK(hipMemcpy(m->ids_u, iu.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    ALSCHK(hipMemcpy(m->ids_w, iw.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    ALSCHK(hipMemcpy(m->val_u, vu.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    ALSCHK(hipMemcpy(m->val_w, vw.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    if (!lu.empty()) ALSCHK(hipMemcpy(m->users, lu.data(), lu.size() * 4, hipMemcpyHostToDevice));
    if (!lw.empty()) ALSCHK(hipMemcpy(m->works, lw.data(), lw.size() * 4, hipMemcpyHostToDevice));
    m->n = n; m->n_users = (int64_t)lu.size(); m->n_works = (int64_t)lw.size();
    // long lists (a blockbuster item can hold a few per cent of all ratings) would leave one block working
    // long after the rest of the sweep has finished: cut them into chunks of CH ratings
    int64_t CH = 512;                                     // A/B in one call: 2048 1.33 ms per iteration, 1024 1.18, 512 1.12, 256 1.16, 128 1.38
    if (const char* e = getenv("TFR_ALS_CHUNK")) { const long v = atol(e); if (v >= ALS_TILE) CH = (v / ALS_TILE) * ALS_TILE; }
    size_t max_chunks = 0;
    for (int z = 0; z < 2; ++z) {
        const std::vector<int64_t>& ptr = z == 0 ? pu : pw;
        const int64_t rows = z == 0 ? m->nu : m->nw;
        std::vector<int32_t> cf((size_t)rows, -1), cc((size_t)rows, 0), ce;
        std::vector<int64_t> cl, chh;
        for (int64_t r = 0; r < rows; ++r) {
            const int64_t lo = ptr[(size_t)r], hi = ptr[(size_t)r + 1];
            if (hi - lo <= CH) continue;
            cf[(size_t)r] = (int32_t)ce.size();
            for (int64_t s0 = lo; s0 < hi; s0 += CH) {
                ce.push_back((int32_t)r); cl.push_back(s0); chh.push_back(std::min(hi, s0 + CH));
                cc[(size_t)r]++;
            }
        }
        als_free(m->cfirst[z]); als_free(m->ccount[z]); als_free(m->chunk_ent[z]); als_free(m->chunk_lo[z]); als_free(m->chunk_hi[z]);
        m->cfirst[z] = m->ccount[z] = m->chunk_ent[z] = nullptr; m->chunk_lo[z] = m->chunk_hi[z] = nullptr;
        m->n_chunks[z] = (int64_t)ce.size();
        ALSCHK(hipMalloc((void**)&m->cfirst[z], (size_t)rows * 4));
        ALSCHK(hipMalloc((void**)&m->ccount[z], (size_t)rows * 4));
        ALSCHK(hipMemcpy(m->cfirst[z], cf.data(), (size_t)rows * 4, hipMemcpyHostToDevice));
        ALSCHK(hipMemcpy(m->ccount[z], cc.data(), (size_t)rows * 4, hipMemcpyHostToDevice));
        if (!ce.empty()) {
            ALSCHK(hipMalloc((void**)&m->chunk_ent[z], ce.size() * 4));
            ALSCHK(hipMalloc((void**)&m->chunk_lo[z], ce.size() * 8));
            ALSCHK(hipMalloc((void**)&m->chunk_hi[z], ce.size() * 8));
            ALSCHK(hipMemcpy(m->chunk_ent[z], ce.data(), ce.size() * 4, hipMemcpyHostToDevice));
            ALSCHK(hipMemcpy(m->chunk_lo[z], cl.data(), ce.size() * 8, hipMemcpyHostToDevice));
            ALSCHK(hipMemcpy(m->chunk_hi[z], chh.data(), ce.size() * 8, hipMemcpyHostToDevice));
        }
        max_chunks = std::max(max_chunks, ce.size());
    }
    als_free(m->partial);
    m->partial = nullptr;
    if (max_chunks) ALSCHK(hipMalloc((void**)&m->partial, max_chunks * (size_t)(m->d * m->d + m->d) * 8));
    return TFR_OK;
}

int tfr_als_set_bias(tfr_als* m, double bias) {
    if (!m) return als_fail(TFR_ERR_ARG, "null model");
    m->bias = bias;
    return TFR_OK;
}

// n_iterations x (every user, then every work) - the loop body of als3.py:30-35
int tfr_als_sweep(tfr_als* m, int32_t n_iterations, float* elapsed_ms) {
    if (!m || n_iterations < 0) return als_fail(TFR_ERR_ARG, "bad arguments");
    if (!m->n) return als_fail(TFR_ERR_STATE, "no ratings: call tfr_als_load first");
    ALSCHK(hipSetDevice(m->device));
    (void)hipEventRecord(m->ev0, m->stream);
    for (int it = 0; it < n_iterations; ++it) {
        AlsFitArgs a;
        a.bias = m->bias; a.lambda = m->lambda; a.d = m->d;
        a.list = m->users; a.n_list = m->n_users; a.ptr = m->ptr_u; a.ids = m->ids_u; a.vals = m->val_u;
        a.own = m->U; a.w_own = m->Wu; a.other = m->V; a.w_other = m->Ww;
        a.cfirst = m->cfirst[0]; a.ccount = m->ccount[0]; a.chunk_ent = m->chunk_ent[0]; a.chunk_lo = m->chunk_lo[0];
        a.chunk_hi = m->chunk_hi[0]; a.n_chunks = m->n_chunks[0]; a.partial = m->partial;
        if (a.n_list && a.n_chunks) hipLaunchKernelGGL(k_als_partial, dim3((unsigned)std::min<int64_t>(a.n_chunks, 65535)), dim3(256), 0, m->stream, a);
        if (a.n_list) hipLaunchKernelGGL(k_als_fit, dim3((unsigned)std::min<int64_t>(a.n_list, 65535)), dim3(256), 0, m->stream, a);
        a.list = m->works; a.n_list = m->n_works; a.ptr = m->ptr_w; a.ids = m->ids_w; a.vals = m->val_w;
        a.own = m->V; a.w_own = m->Ww; a.other = m->U; a.w_other = m->Wu;
        a.cfirst = m->cfirst[1]; a.ccount = m->ccount[1]; a.chunk_ent = m->chunk_ent[1]; a.chunk_lo = m->chunk_lo[1];
        a.chunk_hi = m->chunk_hi[1]; a.n_chunks = m->n_chunks[1]; a.partial = m->partial;
        if (a.n_list && a.n_chunks) hipLaunchKernelGGL(k_als_partial, dim3((unsigned)std::min<int64_t>(a.n_chunks, 65535)), dim3(256), 0, m->stream, a);
        if (a.n_list) hipLaunchKernelGGL(k_als_fit, dim3((unsigned)std::min<int64_t>(a.n_list, 65535)), dim3(256), 0, m->stream, a);
    }
    (void)hipEventRecord(m->ev1, m->stream);
    ALSCHK(hipGetLastError());
    ALSCHK(hipStreamSynchronize(m->stream));
    if (elapsed_ms) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, m->ev0, m->ev1) != hipSuccess) ms = 0.f;
        *elapsed_ms = ms;
    }
    return TFR_OK;
}

int tfr_als_predict(tfr_als* m, const int64_t* user_ids, const int64_t* work_ids, int64_t n, double* out) {
    if (!m || n < 0 || (n > 0 && (!user_ids || !work_ids || !out))) return als_fail(TFR_ERR_ARG, "bad arguments");
    if (n == 0) return TFR_OK;
    ALSCHK(hipSetDevice(m->device));
    std::vector<int32_t> u((size_t)n), w((size_t)n);
    for (int64_t k = 0; k < n; ++k) {
        if (user_ids[k] < 0 || user_ids[k] >= m->nu || work_ids[k] < 0 || work_ids[k] >= m->nw)
            return als_fail(TFR_ERR_OOB, "pair %lld: id out of range", (long long)k);
        u[(size_t)k] = (int32_t)user_ids[k];
        w[(size_t)k] = (int32_t)work_ids[k];
    }
    int32_t *du = nullptr, *dw = nullptr;
    double* dout = nullptr;
    hipError_t e = hipMalloc((void**)&du, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&dw, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&dout, (size_t)n * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(du, u.data(), (size_t)n * 4, hipMemcpyHostToDevice, m->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dw, w.data(), (size_t)n * 4, hipMemcpyHostToDevice, m->stream);
    if (e == hipSuccess) {
        int64_t nb = (n + 255) / 256;
        if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL(k_als_predict, dim3((unsigned)nb), dim3(256), 0, m->stream, m->U, m->V, m->Wu, m->Ww, m->bias, du, dw, n,
                           m->d, dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout, (size_t)n * 8, hipMemcpyDeviceToHost, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    als_free(du); als_free(dw); als_free(dout);
    if (e != hipSuccess) return als_fail(TFR_ERR_HIP, "als_predict: %s", hipGetErrorString(e));
    return TFR_OK;
}

}  // extern "C"
