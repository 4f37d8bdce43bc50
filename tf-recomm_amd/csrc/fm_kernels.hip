// fm_kernels.hip - second-order factorisation-machine forward on CSR rows (BASELINE config 5).
// Restates forward.py:21-22:   y(x) = mu + x.W + 0.5 * ( ||x V||^2 - sum_j x_j^2 ||V_j||^2 )
// on the design matrix fm.py:61-93 builds (one-hot user/item blocks + count-valued blocks).
// forward.py:22 writes the last term as x.dot(V**2) - without squaring x - which is the same
// thing on 0/1 features only; this kernel computes the general (libFM) form.
// One lane group (G lanes x VEC floats = one V row) per CSR row; the row's non-zeros are walked
// four at a time so four 4*D-byte row gathers are in flight per group.  HBM-bound gather.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <new>
#include "tfrecomm.h"
#include "svd_kernels.h"

namespace tfr {

struct FmArgs {
    const float* V; const float* W; const float* mu;
    const int64_t* indptr; const int32_t* indices; const float* data;
    float* out; int32_t* err;
    int64_t n_rows, F;
    int32_t D;
};

template <int G, int VEC>
__global__ __launch_bounds__(256) void k_fm_forward(FmArgs a) {
    constexpr int GPB = 256 / G;
    const int gl = threadIdx.x % G;
    const int d0 = gl * VEC;
    const int D = a.D;
    const float mu = *a.mu;
    bool oob = false;
    for (int64_t row = (int64_t)blockIdx.x * GPB + threadIdx.x / G; row < a.n_rows;
         row += (int64_t)gridDim.x * GPB) {
        const int64_t lo = a.indptr[row], hi = a.indptr[row + 1];
        float s[VEC], q[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) { s[e] = 0.f; q[e] = 0.f; }
        float lin = 0.f;
        for (int64_t p = lo; p < hi; p += 4) {
            int32_t f[4];
            float x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = p + j < hi;
                f[j] = ok ? a.indices[p + j] : 0;
                x[j] = ok ? a.data[p + j] : 0.f;
                if ((uint64_t)(int64_t)f[j] >= (uint64_t)a.F) { oob = true; f[j] = 0; x[j] = 0.f; }
            }
            float v[4][VEC];
            float w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* vr = a.V + (size_t)f[j] * D;
                if constexpr (VEC == 4) {
                    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (d0 < D) t = *reinterpret_cast<const float4*>(vr + d0);
                    v[j][0] = t.x; v[j][1] = t.y; v[j][2] = t.z; v[j][3] = t.w;
                } else {
                    v[j][0] = (d0 < D) ? vr[d0] : 0.f;
                }
                w[j] = a.W[f[j]];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float xv = x[j] * v[j][e];
                    s[e] += xv;
                    q[e] = fmaf(xv, xv, q[e]);
                }
                lin = fmaf(x[j], w[j], lin);
            }
        }
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc += s[e] * s[e] - q[e];
#pragma unroll
        for (int o = G / 2; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (gl == 0) a.out[row] = (mu + lin) + 0.5f * acc;
    }
    if (oob) atomicOr(a.err, 1);
}

static void launch_fm(const FmArgs& a, int G, int VEC, hipStream_t s) {
    const int gpb = 256 / G;
    int64_t nb = (a.n_rows + gpb - 1) / gpb;
    if (nb > 8192) nb = 8192;
    if (nb < 1) nb = 1;
#define TFR_FM_CASE(g, v) \
    if (G == g && VEC == v) { hipLaunchKernelGGL((k_fm_forward<g, v>), dim3((int)nb), dim3(256), 0, s, a); return; }
    TFR_FM_CASE(4, 4) TFR_FM_CASE(8, 4) TFR_FM_CASE(16, 4) TFR_FM_CASE(32, 4) TFR_FM_CASE(64, 4)
    TFR_FM_CASE(4, 1) TFR_FM_CASE(8, 1) TFR_FM_CASE(16, 1) TFR_FM_CASE(32, 1) TFR_FM_CASE(64, 1)
#undef TFR_FM_CASE
}

}  // namespace tfr

using namespace tfr;

struct tfr_fm {
    int64_t F = 0;
    int32_t D = 0, G = 0, VEC = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    float *V = nullptr, *W = nullptr, *mu = nullptr;
    int32_t* d_err = nullptr;
    // staging for host CSR
    int64_t cap_rows = 0, cap_nnz = 0;
    int64_t* d_indptr = nullptr;
    int32_t* d_indices = nullptr;
    float *d_data = nullptr, *d_out = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

static thread_local char g_fm_err[512] = "";
static int fm_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_fm_err, sizeof(g_fm_err), fmt, ap);
    va_end(ap);
    return code;
}
#define FMCHK(expr)                                                                              \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fm_fail(e_ == hipErrorOutOfMemory ? TFR_ERR_NOMEM : TFR_ERR_HIP, \
                                             "%s: %s", #expr, hipGetErrorString(e_));           \
    } while (0)

extern "C" {

const char* tfr_fm_last_error(void) { return g_fm_err; }

int tfr_fm_destroy(tfr_fm* m) {
    if (!m) return TFR_OK;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    void* ps[] = {m->V, m->W, m->mu, m->d_err, m->d_indptr, m->d_indices, m->d_data, m->d_out};
    for (void* p : ps) if (p) (void)hipFree(p);
    if (m->ev0) (void)hipEventDestroy(m->ev0);
    if (m->ev1) (void)hipEventDestroy(m->ev1);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
    return TFR_OK;
}

int tfr_fm_create(tfr_fm** out, int64_t n_features, int32_t dim, int32_t device) {
    if (!out) return fm_fail(TFR_ERR_ARG, "out is null");
    *out = nullptr;
    int G, VEC;
    if (n_features < 1 || n_features > 0x7fffffffLL || !geometry(dim, &G, &VEC))
        return fm_fail(TFR_ERR_ARG, "bad n_features / unsupported dim %d", dim);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fm_fail(TFR_ERR_HIP, "no HIP device available - this library has no CPU path");
    if (device < 0 || device >= ndev) return fm_fail(TFR_ERR_ARG, "device %d not in [0,%d)", device, ndev);
    FMCHK(hipSetDevice(device));
    tfr_fm* m = new (std::nothrow) tfr_fm();
    if (!m) return fm_fail(TFR_ERR_NOMEM, "host allocation failed");
    m->F = n_features; m->D = dim; m->G = G; m->VEC = VEC; m->device = device;
    hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&m->V, (size_t)n_features * dim * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&m->W, (size_t)n_features * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&m->mu, 4);
    if (e == hipSuccess) e = hipMalloc((void**)&m->d_err, 4);
    if (e == hipSuccess) e = hipMemsetAsync(m->V, 0, (size_t)n_features * dim * 4, m->stream);
    if (e == hipSuccess) e = hipMemsetAsync(m->W, 0, (size_t)n_features * 4, m->stream);
    if (e == hipSuccess) e = hipMemsetAsync(m->mu, 0, 4, m->stream);
    if (e == hipSuccess) e = hipMemsetAsync(m->d_err, 0, 4, m->stream);
    if (e == hipSuccess) e = hipEventCreate(&m->ev0);
    if (e == hipSuccess) e = hipEventCreate(&m->ev1);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    if (e != hipSuccess) {
        fm_fail(e == hipErrorOutOfMemory ? TFR_ERR_NOMEM : TFR_ERR_HIP, "fm_create: %s", hipGetErrorString(e));
        char keep[512];
        strncpy(keep, g_fm_err, sizeof(keep));
        tfr_fm_destroy(m);
        strncpy(g_fm_err, keep, sizeof(g_fm_err));
        return e == hipErrorOutOfMemory ? TFR_ERR_NOMEM : TFR_ERR_HIP;
    }
    *out = m;
    return TFR_OK;
}

int tfr_fm_set(tfr_fm* m, float mu, const float* W, const float* V) {
    if (!m || !W || !V) return fm_fail(TFR_ERR_ARG, "null argument");
    FMCHK(hipSetDevice(m->device));
    FMCHK(hipMemcpyAsync(m->mu, &mu, 4, hipMemcpyHostToDevice, m->stream));
    FMCHK(hipMemcpyAsync(m->W, W, (size_t)m->F * 4, hipMemcpyHostToDevice, m->stream));
    FMCHK(hipMemcpyAsync(m->V, V, (size_t)m->F * m->D * 4, hipMemcpyHostToDevice, m->stream));
    FMCHK(hipStreamSynchronize(m->stream));
    return TFR_OK;
}

int tfr_fm_init(tfr_fm* m, uint64_t seed, float stddev) {
    if (!m) return fm_fail(TFR_ERR_ARG, "null model");
    FMCHK(hipSetDevice(m->device));
    launch_init_trunc_normal(m->V, m->F * m->D, stddev, seed * 2 + 0, m->stream);
    launch_init_trunc_normal(m->W, m->F, stddev, seed * 2 + 1, m->stream);
    FMCHK(hipGetLastError());
    FMCHK(hipStreamSynchronize(m->stream));
    return TFR_OK;
}

static int fm_run(tfr_fm* m, const int64_t* d_indptr, const int32_t* d_indices, const float* d_data,
                  int64_t n_rows, float* d_out) {
    FmArgs a;
    a.V = m->V; a.W = m->W; a.mu = m->mu;
    a.indptr = d_indptr; a.indices = d_indices; a.data = d_data; a.out = d_out; a.err = m->d_err;
    a.n_rows = n_rows; a.F = m->F; a.D = m->D;
    (void)hipEventRecord(m->ev0, m->stream);
    launch_fm(a, m->G, m->VEC, m->stream);
    (void)hipEventRecord(m->ev1, m->stream);
    FMCHK(hipGetLastError());
    return TFR_OK;
}

static int fm_check(tfr_fm* m) {
    int32_t e = 0;
    FMCHK(hipMemcpyAsync(&e, m->d_err, 4, hipMemcpyDeviceToHost, m->stream));
    FMCHK(hipStreamSynchronize(m->stream));
    if (e) {
        FMCHK(hipMemsetAsync(m->d_err, 0, 4, m->stream));
        FMCHK(hipStreamSynchronize(m->stream));
        return fm_fail(TFR_ERR_OOB, "feature index out of range [0,%lld)", (long long)m->F);
    }
    return TFR_OK;
}

/* device CSR; asynchronous */
int tfr_fm_forward_dev(tfr_fm* m, const int64_t* d_indptr, const int32_t* d_indices, const float* d_data,
                       int64_t n_rows, float* d_out) {
    if (!m || n_rows < 0 || (n_rows > 0 && (!d_indptr || !d_out))) return fm_fail(TFR_ERR_ARG, "bad arguments");
    FMCHK(hipSetDevice(m->device));
    if (n_rows == 0) return TFR_OK;
    return fm_run(m, d_indptr, d_indices, d_data, n_rows, d_out);
}

/* host CSR (scipy.sparse layout: indptr int64 [n_rows+1], indices int32 [nnz], data f32 [nnz]) */
int tfr_fm_forward(tfr_fm* m, const int64_t* indptr, const int32_t* indices, const float* data,
                   int64_t n_rows, float* out) {
    if (!m || n_rows < 0 || (n_rows > 0 && (!indptr || !out))) return fm_fail(TFR_ERR_ARG, "bad arguments");
    FMCHK(hipSetDevice(m->device));
    if (n_rows == 0) return TFR_OK;
    const int64_t nnz = indptr[n_rows];
    if (nnz < 0 || indptr[0] != 0) return fm_fail(TFR_ERR_ARG, "indptr must start at 0 and be non-decreasing");
    for (int64_t r = 0; r < n_rows; ++r)
        if (indptr[r + 1] < indptr[r]) return fm_fail(TFR_ERR_ARG, "indptr must be non-decreasing");
    if (nnz > 0 && (!indices || !data)) return fm_fail(TFR_ERR_ARG, "null indices/data");
    if (n_rows > m->cap_rows) {
        FMCHK(hipStreamSynchronize(m->stream));
        if (m->d_indptr) (void)hipFree(m->d_indptr);
        if (m->d_out) (void)hipFree(m->d_out);
        m->d_indptr = nullptr; m->d_out = nullptr; m->cap_rows = 0;
        FMCHK(hipMalloc((void**)&m->d_indptr, (size_t)(n_rows + 1) * 8));
        FMCHK(hipMalloc((void**)&m->d_out, (size_t)n_rows * 4));
        m->cap_rows = n_rows;
    }
    if (nnz > m->cap_nnz) {
        FMCHK(hipStreamSynchronize(m->stream));
        if (m->d_indices) (void)hipFree(m->d_indices);
        if (m->d_data) (void)hipFree(m->d_data);
        m->d_indices = nullptr; m->d_data = nullptr; m->cap_nnz = 0;
        FMCHK(hipMalloc((void**)&m->d_indices, (size_t)nnz * 4));
        FMCHK(hipMalloc((void**)&m->d_data, (size_t)nnz * 4));
        m->cap_nnz = nnz;
    }
    FMCHK(hipMemcpyAsync(m->d_indptr, indptr, (size_t)(n_rows + 1) * 8, hipMemcpyHostToDevice, m->stream));
    if (nnz > 0) {
        FMCHK(hipMemcpyAsync(m->d_indices, indices, (size_t)nnz * 4, hipMemcpyHostToDevice, m->stream));
        FMCHK(hipMemcpyAsync(m->d_data, data, (size_t)nnz * 4, hipMemcpyHostToDevice, m->stream));
    }
    int rc = fm_run(m, m->d_indptr, m->d_indices, m->d_data, n_rows, m->d_out);
    if (rc) return rc;
    FMCHK(hipMemcpyAsync(out, m->d_out, (size_t)n_rows * 4, hipMemcpyDeviceToHost, m->stream));
    return fm_check(m);
}

/* synchronise; report a deferred out-of-range feature index; elapsed ms of the last launch */
int tfr_fm_sync(tfr_fm* m, float* last_kernel_ms) {
    if (!m) return fm_fail(TFR_ERR_ARG, "null model");
    FMCHK(hipSetDevice(m->device));
    int rc = fm_check(m);
    if (rc) return rc;
    if (last_kernel_ms) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, m->ev0, m->ev1) != hipSuccess) ms = 0.f;
        *last_kernel_ms = ms;
    }
    return TFR_OK;
}

}  // extern "C"
