/* svd_oracle.c - scalar float32 C restatement of the SVD minibatch step.  TEST INFRASTRUCTURE
 * ONLY: used by tests/ (checked against oracle/svd_oracle.py) and by bench.py's cpu_baseline leg
 * (kind "port", all host cores through OpenMP).  The product never links or loads this file.
 *
 * Threading keeps every result independent of the thread count except the three batch scalars:
 * rows are summed per unique row in batch order (exactly the serial order), dense sweeps are
 * element-wise (TF's five dense passes over a table - decay m, scatter-add, decay v, scatter-add,
 * update - are walked row by row in one loop: the same operations on every element, in the same order);
 * loss / regulariser / sum g are added per fixed chunk of 256 ratings, chunks in order.
 *
 * PARITY UNPINNED by reference fixtures (the reference has none for this path; TensorFlow is not
 * installable here) - see the header of oracle/svd_oracle.py.
 *
 * It follows the reference graph op by op, materialising what TensorFlow materialises:
 *   gathers            ops.py:13-14,37-38      -> pu[B,D], qi[B,D], bu[B], bi[B]
 *   logits             ops.py:44-47
 *   regulariser        ops.py:81-89
 *   loss               ops.py:124 (canonical) / ops.py:125-126 (fork)
 *   minimize           ops.py:143-149: per-occurrence IndexedSlices values, then
 *       Adam  [TF1-lib]: unique (first-occurrence order) + unsorted_segment_sum (batch order),
 *                        dense decay of m and v, scatter-add, dense variable update
 *       lazy Adam      : the same arithmetic on touched rows only
 *       SGD   [TF1-lib]: scatter_sub of lr*value, duplicates accumulate in batch order
 *   step               svd_train_val.py:66-72 (returns the pre-update logits of the batch)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int svdo_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void svdo_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

enum { T_MU = 0, T_BU = 1, T_BI = 2, T_P = 3, T_Q = 4 };
#define CHUNK 256   /* ratings per partial sum of the batch scalars */

typedef struct svdo {
    int64_t U, I;
    int D, loss, item_abs, reg_bias, optimizer, adam_mode;
    float lr, reg, b1, b2, eps, b1p, b2p;
    uint32_t frozen;
    int64_t step;
    float* w[5];
    float* m[5];
    float* v[5];
    int64_t n[5];
    int32_t* slot_u; /* row -> unique slot, -1 when absent */
    int32_t* slot_i;
    /* batch scratch */
    int64_t cap;
    float *pu, *qi, *dP, *dQ, *dbu, *dbi, *g, *gsum;
    int32_t* uniq;
    int32_t *slot_of, *start, *order;   /* occurrences grouped by unique slot, batch order inside a slot */
    float* part;                        /* per-chunk partial scalars */
} svdo;

static void free_batch(svdo* o) {
    free(o->pu); free(o->qi); free(o->dP); free(o->dQ); free(o->dbu); free(o->dbi); free(o->g);
    free(o->gsum); free(o->uniq); free(o->slot_of); free(o->start); free(o->order); free(o->part);
    o->pu = o->qi = o->dP = o->dQ = o->dbu = o->dbi = o->g = o->gsum = NULL;
    o->uniq = NULL; o->slot_of = o->start = o->order = NULL; o->part = NULL;
    o->cap = 0;
}

void svdo_destroy(svdo* o) {
    if (!o) return;
    for (int t = 0; t < 5; ++t) { free(o->w[t]); free(o->m[t]); free(o->v[t]); }
    free(o->slot_u); free(o->slot_i);
    free_batch(o);
    free(o);
}

svdo* svdo_create(int64_t U, int64_t I, int D, int loss, int item_abs, int reg_bias, int optimizer,
                  int adam_mode, float lr, float reg, float b1, float b2, float eps) {
    svdo* o = (svdo*)calloc(1, sizeof(svdo));
    if (!o) return NULL;
    o->U = U; o->I = I; o->D = D; o->loss = loss; o->item_abs = item_abs; o->reg_bias = reg_bias;
    o->optimizer = optimizer; o->adam_mode = adam_mode;
    o->lr = lr; o->reg = reg; o->b1 = b1; o->b2 = b2; o->eps = eps; o->b1p = b1; o->b2p = b2;
    o->n[T_MU] = 1; o->n[T_BU] = U; o->n[T_BI] = I; o->n[T_P] = U * D; o->n[T_Q] = I * D;
    for (int t = 0; t < 5; ++t) {
        o->w[t] = (float*)calloc((size_t)o->n[t], 4);
        o->m[t] = (float*)calloc((size_t)o->n[t], 4);
        o->v[t] = (float*)calloc((size_t)o->n[t], 4);
        if (!o->w[t] || !o->m[t] || !o->v[t]) { svdo_destroy(o); return NULL; }
    }
    o->slot_u = (int32_t*)malloc((size_t)U * 4);
    o->slot_i = (int32_t*)malloc((size_t)I * 4);
    if (!o->slot_u || !o->slot_i) { svdo_destroy(o); return NULL; }
    memset(o->slot_u, 0xff, (size_t)U * 4);
    memset(o->slot_i, 0xff, (size_t)I * 4);
    return o;
}

float* svdo_table(svdo* o, int which) {
    int t = which & 7;
    if (t > T_Q) return NULL;
    if (which & 8) return o->m[t];
    if (which & 16) return o->v[t];
    return o->w[t];
}

void svdo_set_frozen(svdo* o, uint32_t mask) { o->frozen = mask; }
int64_t svdo_step(const svdo* o) { return o->step; }

static int ensure(svdo* o, int64_t B) {
    if (B <= o->cap) return 0;
    free_batch(o);
    size_t bd = (size_t)B * o->D;
    o->pu = (float*)malloc(bd * 4); o->qi = (float*)malloc(bd * 4);
    o->dP = (float*)malloc(bd * 4); o->dQ = (float*)malloc(bd * 4);
    o->gsum = (float*)malloc(bd * 4);
    o->dbu = (float*)malloc((size_t)B * 4); o->dbi = (float*)malloc((size_t)B * 4);
    o->g = (float*)malloc((size_t)B * 4); o->uniq = (int32_t*)malloc((size_t)B * 4);
    o->slot_of = (int32_t*)malloc((size_t)B * 4); o->start = (int32_t*)malloc(((size_t)B + 1) * 4);
    o->order = (int32_t*)malloc((size_t)B * 4); o->part = (float*)malloc(((size_t)B / CHUNK + 1) * 8 * 4);
    if (!o->pu || !o->qi || !o->dP || !o->dQ || !o->gsum || !o->dbu || !o->dbi || !o->g || !o->uniq ||
        !o->slot_of || !o->start || !o->order || !o->part) return -5;
    o->cap = B;
    return 0;
}

static int check_ids(const svdo* o, const int32_t* u, const int32_t* it, int64_t B) {
    for (int64_t k = 0; k < B; ++k)
        if (u[k] < 0 || u[k] >= o->U || it[k] < 0 || it[k] >= o->I) return -2;
    return 0;
}

/* ops.py:13-14,37-38,44-47 */
int svdo_forward(svdo* o, const int32_t* u, const int32_t* it, int64_t B, float* logits) {
    if (check_ids(o, u, it, B)) return -2;
    const int D = o->D;
    const float mu = o->w[T_MU][0];
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < B; ++k) {
        const float* p = o->w[T_P] + (size_t)u[k] * D;
        const float* q = o->w[T_Q] + (size_t)it[k] * D;
        float s = 0.f;
        if (o->item_abs) for (int d = 0; d < D; ++d) s += p[d] * fabsf(q[d]);
        else for (int d = 0; d < D; ++d) s += p[d] * q[d];
        s = s + mu;
        s = s + o->w[T_BU][u[k]];
        s = s + o->w[T_BI][it[k]];
        logits[k] = s;
    }
    return 0;
}

static float lr_t(const svdo* o) { return o->lr * sqrtf(1.f - o->b2p) / (1.f - o->b1p); }

/* unique (first-occurrence order) + unsorted_segment_sum (batch order) [TF1-lib]: the slot assignment
 * is a serial integer pass; each unique row then adds ITS occurrences in batch order - the serial order -
 * so the sums do not depend on the number of threads.  Returns the number of unique rows. */
static int64_t group_by_slot(svdo* o, const int32_t* ids, int64_t B, int32_t* slot, int keep) {
    int64_t nu = 0;
    for (int64_t k = 0; k < B; ++k) {
        int32_t s = slot[ids[k]];
        if (s < 0) { s = (int32_t)nu++; slot[ids[k]] = s; o->uniq[s] = ids[k]; o->start[s] = 0; }
        o->slot_of[k] = s;
        o->start[s] += 1;
    }
    int32_t run = 0;
    for (int64_t s = 0; s < nu; ++s) { const int32_t c = o->start[s]; o->start[s] = run; run += c; }
    o->start[nu] = run;
    for (int64_t k = 0; k < B; ++k) o->order[o->start[o->slot_of[k]]++] = (int32_t)k;   /* stable */
    for (int64_t s = nu; s > 0; --s) o->start[s] = o->start[s - 1];
    o->start[0] = 0;
    if (!keep) for (int64_t s = 0; s < nu; ++s) slot[o->uniq[s]] = -1;
    return nu;
}

/* AdamOptimizer sparse apply on one table [TF1-lib] */
static void adam_table(svdo* o, int t, int width, const int32_t* ids, const float* occ, int64_t B,
                       int32_t* slot, int64_t rows) {
    if ((o->frozen >> t) & 1) return;
    float* w = o->w[t]; float* m = o->m[t]; float* v = o->v[t];
    const float a = lr_t(o), b1 = o->b1, b2 = o->b2, omb1 = 1.f - o->b1, omb2 = 1.f - o->b2, eps = o->eps;
    const int64_t nu = group_by_slot(o, ids, B, slot, 1);
#pragma omp parallel for schedule(static)
    for (int64_t s = 0; s < nu; ++s) {
        float* gs = o->gsum + (size_t)s * width;
        memset(gs, 0, (size_t)width * 4);
        for (int32_t j = o->start[s]; j < o->start[s + 1]; ++j) {
            const float* oc = occ + (size_t)o->order[j] * width;
            for (int d = 0; d < width; ++d) gs[d] += oc[d];
        }
    }
    if (o->adam_mode == 0) {                /* TF1: dense decay, scatter-add, dense update - every row moves */
#pragma omp parallel for schedule(static)
        for (int64_t row = 0; row < rows; ++row) {
            const int32_t sl = slot[row];
            const float* gs = sl >= 0 ? o->gsum + (size_t)sl * width : NULL;
            const size_t off = (size_t)row * width;
            for (int d = 0; d < width; ++d) {
                float mm = m[off + d] * b1;
                if (gs) mm += gs[d] * omb1;
                float vv = v[off + d] * b2;
                if (gs) vv += (gs[d] * gs[d]) * omb2;
                m[off + d] = mm; v[off + d] = vv;
                w[off + d] -= a * mm / (sqrtf(vv) + eps);
            }
        }
    } else {                                /* lazy: touched rows only */
#pragma omp parallel for schedule(static)
        for (int64_t s = 0; s < nu; ++s) {
            const size_t off = (size_t)o->uniq[s] * width;
            const float* gs = o->gsum + (size_t)s * width;
            for (int d = 0; d < width; ++d) {
                const float mm = m[off + d] * b1 + gs[d] * omb1;
                const float vv = v[off + d] * b2 + (gs[d] * gs[d]) * omb2;
                m[off + d] = mm; v[off + d] = vv;
                w[off + d] -= a * mm / (sqrtf(vv) + eps);
            }
        }
    }
    for (int64_t s = 0; s < nu; ++s) slot[o->uniq[s]] = -1;
}

/* GradientDescentOptimizer sparse apply = scatter_sub [TF1-lib]: duplicates accumulate in batch order;
 * rows are independent, so each row replays its own occurrences in that order */
static void sgd_table(svdo* o, int t, int width, const int32_t* ids, const float* occ, int64_t B, int32_t* slot) {
    if ((o->frozen >> t) & 1) return;
    float* w = o->w[t];
    const int64_t nu = group_by_slot(o, ids, B, slot, 0);
#pragma omp parallel for schedule(static)
    for (int64_t s = 0; s < nu; ++s) {
        float* wr = w + (size_t)o->uniq[s] * width;
        for (int32_t j = o->start[s]; j < o->start[s + 1]; ++j) {
            const float* oc = occ + (size_t)o->order[j] * width;
            for (int d = 0; d < width; ++d) wr[d] -= o->lr * oc[d];
        }
    }
}

/* one sess.run([train_op, logits, infer]) - svd_train_val.py:70-72 */
int svdo_train_step(svdo* o, const int32_t* u, const int32_t* it, const float* r, int64_t B,
                    float* logits_out, float* loss_out, float* reg_out) {
    if (check_ids(o, u, it, B)) return -2;
    if (ensure(o, B > 0 ? B : 1)) return -5;
    const int D = o->D;
    const float lam = o->reg;
    const float mu = o->w[T_MU][0];
    float loss = 0.f, reg = 0.f, l2u = 0.f, l2i = 0.f, l2bu = 0.f, l2bi = 0.f, dmu = 0.f;
    const int64_t nchunk = (B + CHUNK - 1) / CHUNK;
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < nchunk; ++c) {
        float closs = 0.f, cl2u = 0.f, cl2i = 0.f, cl2bu = 0.f, cl2bi = 0.f, cdmu = 0.f;
        const int64_t k1 = (c + 1) * CHUNK < B ? (c + 1) * CHUNK : B;
        /* gathers (materialised, like feat_users / feat_items) */
        for (int64_t k = c * CHUNK; k < k1; ++k) {
            memcpy(o->pu + (size_t)k * D, o->w[T_P] + (size_t)u[k] * D, (size_t)D * 4);
            memcpy(o->qi + (size_t)k * D, o->w[T_Q] + (size_t)it[k] * D, (size_t)D * 4);
        }
        for (int64_t k = c * CHUNK; k < k1; ++k) {
            const float* p = o->pu + (size_t)k * D;
            const float* q = o->qi + (size_t)k * D;
            const float bu = o->w[T_BU][u[k]], bi = o->w[T_BI][it[k]];
            float s = 0.f;
            for (int d = 0; d < D; ++d) {
                s += p[d] * (o->item_abs ? fabsf(q[d]) : q[d]);
                cl2u += p[d] * p[d];
                cl2i += q[d] * q[d];
            }
            const float x = ((s + mu) + bu) + bi;
            if (logits_out) logits_out[k] = x;
            float g;
            if (o->loss == 0) { g = x - r[k]; closs += g * g; }
            else {
                g = 1.f / (1.f + expf(-x)) - r[k];
                closs += fmaxf(x, 0.f) - x * r[k] + log1pf(expf(-fabsf(x)));
            }
            o->g[k] = g;
            cdmu += g;
            cl2bu += bu * bu; cl2bi += bi * bi;
            /* per-occurrence gradient values */
            float* dp = o->dP + (size_t)k * D;
            float* dq = o->dQ + (size_t)k * D;
            for (int d = 0; d < D; ++d) {
                const float qd = q[d];
                if (o->item_abs) {
                    const float sg = (qd > 0.f) ? 1.f : ((qd < 0.f) ? -1.f : 0.f);
                    dp[d] = g * fabsf(qd) + lam * p[d];
                    dq[d] = g * p[d] * sg + lam * qd;
                } else {
                    dp[d] = g * qd + lam * p[d];
                    dq[d] = g * p[d] + lam * qd;
                }
            }
            o->dbu[k] = o->reg_bias ? g + lam * bu : g;
            o->dbi[k] = o->reg_bias ? g + lam * bi : g;
        }
        float* pc = o->part + (size_t)c * 8;
        pc[0] = closs; pc[1] = cl2u; pc[2] = cl2i; pc[3] = cl2bu; pc[4] = cl2bi; pc[5] = cdmu;
    }
    for (int64_t c = 0; c < nchunk; ++c) {                /* chunks in order: independent of the thread count */
        const float* pc = o->part + (size_t)c * 8;
        loss += pc[0]; l2u += pc[1]; l2i += pc[2]; l2bu += pc[3]; l2bi += pc[4]; dmu += pc[5];
    }
    if (o->loss == 0) loss *= 0.5f;                       /* tf.nn.l2_loss = sum(x^2)/2 */
    reg = 0.5f * l2u + 0.5f * l2i;
    if (o->reg_bias) reg = (reg + 0.5f * l2bu) + 0.5f * l2bi;
    if (loss_out) *loss_out = loss;
    if (reg_out) *reg_out = reg;
    if (o->optimizer == 0) {
        adam_table(o, T_P, D, u, o->dP, B, o->slot_u, o->U);
        adam_table(o, T_Q, D, it, o->dQ, B, o->slot_i, o->I);
        adam_table(o, T_BU, 1, u, o->dbu, B, o->slot_u, o->U);
        adam_table(o, T_BI, 1, it, o->dbi, B, o->slot_i, o->I);
        if (!((o->frozen >> T_MU) & 1)) {                 /* dense ApplyAdam [TF1-lib] */
            const float a = lr_t(o);
            float* m = o->m[T_MU]; float* v = o->v[T_MU];
            m[0] += (dmu - m[0]) * (1.f - o->b1);
            v[0] += (dmu * dmu - v[0]) * (1.f - o->b2);
            o->w[T_MU][0] -= (a * m[0]) / (sqrtf(v[0]) + o->eps);
        }
        o->b1p *= o->b1;
        o->b2p *= o->b2;
    } else {
        sgd_table(o, T_P, D, u, o->dP, B, o->slot_u);
        sgd_table(o, T_Q, D, it, o->dQ, B, o->slot_i);
        sgd_table(o, T_BU, 1, u, o->dbu, B, o->slot_u);
        sgd_table(o, T_BI, 1, it, o->dbi, B, o->slot_i);
        if (!((o->frozen >> T_MU) & 1)) o->w[T_MU][0] -= o->lr * dmu;
    }
    o->step += 1;
    return 0;
}

/* ---- FM second-order forward (forward.py:21-22), float64 like the reference's NumPy:
 *      y(x) = mu + x.W + 0.5 * (||x V||^2 - sum_j x_j^2 ||V_j||^2)  on CSR rows (the general x^2 form; equals forward.py:22's
 *      x.dot(V**2) on the 0/1 design matrices fm.py:61-93 builds).  W [F], V [F, D] float32 as the tables are stored; rows are
 *      independent -> OpenMP over rows (the all-core CPU baseline of BASELINE configs[4]). */
int fmo_forward(double mu, const float* W, const float* V, int64_t F, int D, const int64_t* indptr, const int32_t* indices,
                const float* data, int64_t n, double* out) {
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int64_t r = 0; r < n; ++r) {
        double s[256];
        if (D > 256) { bad |= 1; continue; }
        for (int d = 0; d < D; ++d) s[d] = 0.0;
        double lin = 0.0, sq = 0.0;
        for (int64_t k = indptr[r]; k < indptr[r + 1]; ++k) {
            const int64_t f = indices[k];
            if (f < 0 || f >= F) { bad |= 2; continue; }
            const double x = (double)data[k];
            lin += x * (double)W[f];
            const float* v = V + (size_t)f * D;
            for (int d = 0; d < D; ++d) {
                const double xv = x * (double)v[d];
                s[d] += xv;
                sq += xv * xv;
            }
        }
        double ss = 0.0;
        for (int d = 0; d < D; ++d) ss += s[d] * s[d];
        out[r] = mu + lin + 0.5 * (ss - sq);
    }
    return bad;
}
