// snpm_api_strict.hpp -- reference-order launches: dense, the chain of chunk totals, the sparse re-evaluation tier (anonymous namespace of snpm_api.hip).
// Part of the one translation unit of libsnpmatch_hip.so: included by snpm_api.hip at this place, not on its own.
// ---- reference-order (strict) launches ---------------------------------------------------------------------
// Dense: every accession, segments [seg0, seg0 + n_seg) -> ctx->ws_seg_* [n_seg, ld].  Segments are explicit
// (d_seg_off: windows) or `chunk`-row pieces of the query.  gate (may be NULL): device count; the launch is a
// no-op unless *gate > REEVAL_CAP.
// A panel of ONE accession: the reference's per-call sums are numpy's vector sums (snpm_kernels_single.hpp)
static inline bool single_accession(const snpm_panel *p) { return p->n_acc_total == 1; }

// k_strict_single in place of the strict kernel of a site; tier / pairs / count as in the kernel's header
static int launch_strict_single(snpm_ctx *ctx, const snpm_panel *p, const int64_t *d_row_idx, int64_t row0, const double *d_w,
                                int skip, const int64_t *seg_off, int64_t chunk, int64_t n, int64_t seg0, int64_t n_seg,
                                const int32_t *pairs, const int *count, int cap, int tier, int64_t kmax, dim3 grid,
                                double *out_score, uint32_t *out_miss, int64_t ld)
{
    const bool gather = d_row_idx != nullptr;
#define LAUNCH_SINGLE(S, G)                                                                                          \
    hipLaunchKernelGGL((k_strict_single<S, G>), grid, dim3(SINGLE_THREADS), 0, ctx->stream, p->d, p->kpitch, p->desc,  \
                       d_row_idx, row0, d_w, seg_off, chunk, n, seg0, n_seg, pairs, count, cap, tier, kmax, out_score,  \
                       out_miss, ld)
    if (skip) {
        if (gather) LAUNCH_SINGLE(true, true); else LAUNCH_SINGLE(true, false);
    } else {
        if (gather) LAUNCH_SINGLE(false, true); else LAUNCH_SINGLE(false, false);
    }
#undef LAUNCH_SINGLE
    HIPCHK(ctx, hipGetLastError());
    return SNPM_OK;
}

int launch_strict_dense(snpm_query *q, int skip, const int64_t *d_seg_off, int64_t chunk, int64_t seg0, int64_t n_seg,
                        const int *gate)
{
    snpm_ctx *ctx = q->panel->ctx;
    snpm_panel *p = q->panel;
    const int64_t ncols = p->n_acc, ld = p->ld;
    const bool gather = q->d_row_idx != nullptr;
    const int64_t *seg_off = d_seg_off ? d_seg_off + seg0 : nullptr;
    if (n_seg == 0) return SNPM_OK;
    if (single_accession(p)) {
        ProfScope ps(ctx, PK_STRICT);
        return launch_strict_single(ctx, p, q->d_row_idx, q->row0, q->d_w, skip, seg_off, chunk, q->n, seg0, n_seg, nullptr, gate,
                                    REEVAL_CAP, gate ? 2 : 0, 0, dim3((unsigned)std::min<int64_t>(n_seg, 65535)),
                                    (double *)ctx->ws_seg_score.p, (uint32_t *)ctx->ws_seg_miss.p, ld);
    }
    if (ctx->strict4) {
        // 4 columns per lane (one dword of an int8 panel, one byte of a packed panel)
        const int64_t lanes = (ncols + 3) / 4;
        const int t4 = lanes >= 256 ? 256 : (lanes > 64 ? 128 : 64);
        // a gated launch (the certificate's dense tier) usually has nothing to do: a bounded grid that walks the segments
        dim3 grid4((unsigned)(gate ? std::min<int64_t>(n_seg, 2048) : n_seg), (unsigned)((lanes + t4 - 1) / t4));
        ProfScope ps(ctx, PK_STRICT);
#define LAUNCH_STRICT4(S, G)                                                                                     \
    do {                                                                                                         \
        if (p->packed)                                                                                           \
            hipLaunchKernelGGL((k_strict4<S, G, true>), grid4, dim3(t4), 0, ctx->stream, p->d, p->kpitch, q->d_row_idx, q->row0, \
                               q->d_w, seg_off, chunk, q->n, seg0, n_seg, ncols, (double *)ctx->ws_seg_score.p,  \
                               (uint32_t *)ctx->ws_seg_miss.p, ld, (const int *)nullptr, gate, REEVAL_CAP, p->desc); \
        else                                                                                                     \
            hipLaunchKernelGGL((k_strict4<S, G, false>), grid4, dim3(t4), 0, ctx->stream, p->d, p->pitch, q->d_row_idx, q->row0, \
                               q->d_w, seg_off, chunk, q->n, seg0, n_seg, ncols, (double *)ctx->ws_seg_score.p,  \
                               (uint32_t *)ctx->ws_seg_miss.p, ld, (const int *)p->d_other, gate, REEVAL_CAP);   \
    } while (0)
        if (skip) {
            if (gather) LAUNCH_STRICT4(true, true); else LAUNCH_STRICT4(true, false);
        } else {
            if (gather) LAUNCH_STRICT4(false, true); else LAUNCH_STRICT4(false, false);
        }
#undef LAUNCH_STRICT4
        HIPCHK(ctx, hipGetLastError());
        return SNPM_OK;
    }
    const int thr = ncols > 128 ? 256 : (ncols > 64 ? 128 : 64);
    dim3 grid((unsigned)(gate ? std::min<int64_t>(n_seg, 2048) : n_seg), (unsigned)((ncols + thr - 1) / thr));
    ProfScope ps(ctx, PK_STRICT);
#define LAUNCH_STRICT(S, G)                                                                                       \
    hipLaunchKernelGGL((k_strict<S, G>), grid, dim3(thr), 0, ctx->stream, p->d, p->kpitch, p->desc, q->d_row_idx, q->row0,  \
                       q->d_w, seg_off, chunk, q->n, seg0, n_seg, (const int32_t *)nullptr, ncols,                \
                       (double *)ctx->ws_seg_score.p, (uint32_t *)ctx->ws_seg_miss.p, ld, gate, REEVAL_CAP)
    if (skip) {
        if (gather) LAUNCH_STRICT(true, true); else LAUNCH_STRICT(true, false);
    } else {
        if (gather) LAUNCH_STRICT(false, true); else LAUNCH_STRICT(false, false);
    }
#undef LAUNCH_STRICT
    HIPCHK(ctx, hipGetLastError());
    return SNPM_OK;
}

// The reference's whole chunk loop for every accession: strict chunk sums in groups of segments (bounded
// workspace), each group's sums added onto the running totals in order (ScoreList += chunk, core/snpmatch.py:224).
// carry_*: totals of earlier SNP slabs (or NULL).  gate as above.
constexpr size_t kStrictGroupBytes = size_t(512) << 20;

int run_strict_chain(snpm_query *q, int skip, int64_t chunk, const int *gate, const double *carry_score,
                     const int64_t *carry_ninfo, double *dst_score, int64_t *dst_ninfo)
{
    snpm_ctx *ctx = q->panel->ctx;
    snpm_panel *p = q->panel;
    const int64_t n_seg = (q->n + chunk - 1) / chunk;
    const int64_t per_seg = p->ld * (int64_t)(sizeof(double) + sizeof(uint32_t));
    const int64_t group = std::max<int64_t>(1, std::min<int64_t>(std::max<int64_t>(n_seg, 1), (int64_t)kStrictGroupBytes / per_seg));
    int rc = ensure(ctx, ctx->ws_seg_score, (size_t)group * p->ld * sizeof(double));
    if (rc) return rc;
    rc = ensure(ctx, ctx->ws_seg_miss, (size_t)group * p->ld * sizeof(uint32_t));
    if (rc) return rc;
    const int thr = 256;
    const unsigned cb = (unsigned)((p->n_acc + thr - 1) / thr);
    bool first = true;
    for (int64_t g0 = 0; g0 < n_seg || first; g0 += group) {
        const int64_t ns = std::max<int64_t>(0, std::min<int64_t>(group, n_seg - g0));
        rc = launch_strict_dense(q, skip, nullptr, chunk, g0, ns, gate);
        if (rc) return rc;
        const int64_t rows = std::min<int64_t>(q->n, (g0 + ns) * chunk) - std::min<int64_t>(q->n, g0 * chunk);
        ProfScope ps(ctx, PK_SCAN);
        hipLaunchKernelGGL(k_scan, dim3(cb), dim3(thr), 0, ctx->stream, (const double *)ctx->ws_seg_score.p,
                           (const uint32_t *)ctx->ws_seg_miss.p, rows, ns, p->ld, p->n_acc, dst_score, dst_ninfo,
                           first ? carry_score : (const double *)dst_score, first ? carry_ninfo : (const int64_t *)dst_ninfo,
                           gate, REEVAL_CAP);
        HIPCHK(ctx, hipGetLastError());
        first = false;
    }
    return SNPM_OK;
}

// Sparse tier: reference-order chunk sums of the accessions listed on the device (d_cols, *d_ncols <= REEVAL_CAP;
// the kernels do nothing for other counts) -> ws_seg_score [n_seg, REEVAL_CAP] -> chain of additions ->
// ws_tmp_score [REEVAL_CAP].  carry (may be NULL): compact totals of earlier slabs, continued by the chain.
// patch_score (may be NULL): the totals also replace patch_score[d_cols[i]] (k_patch inside the chain's kernel: one launch less).
// the one-call path's last step rides on the chain kernel of the sparse tier (k_once_tail instead of k_scan_few + k_once_finish)
struct OnceTail {
    const int64_t *d_ninfo = nullptr;
    int64_t n_acc = 0;
    int want_lik = 0;
    unsigned *state = nullptr;
    double *lik_tmp = nullptr;
    int64_t *out = nullptr;
};

int run_strict_sparse(snpm_query *q, int skip, int64_t chunk, const int32_t *d_cols, const int *d_ncols, const double *carry,
                      const int64_t *d_seg_off = nullptr, int64_t n_seg_explicit = 0, double *patch_score = nullptr,
                      const OnceTail *tail = nullptr)
{
    snpm_ctx *ctx = q->panel->ctx;
    snpm_panel *p = q->panel;
    const int64_t n_seg = d_seg_off ? n_seg_explicit : (q->n + chunk - 1) / chunk;
    const int64_t ld = REEVAL_CAP;
    int rc = ensure(ctx, ctx->ws_seg_score, (size_t)std::max<int64_t>(n_seg, 1) * ld * sizeof(double));
    if (rc) return rc;
    rc = ensure(ctx, ctx->ws_seg_miss, (size_t)std::max<int64_t>(n_seg, 1) * ld * sizeof(uint32_t));
    if (rc) return rc;
    rc = ensure(ctx, ctx->ws_tmp_score, (size_t)ld * sizeof(double));
    if (rc) return rc;
    const bool gather = q->d_row_idx != nullptr;
    const int64_t total = n_seg * ld;
    if (total > 0) {
        dim3 sgrid((unsigned)std::min<int64_t>((total + 255) / 256, (int64_t)ctx->n_cu * 8));     // grid-stride over (segment, column)
        const bool use_T = !q->transient_panel && q->n >= ctx->acc_major_min_rows && p->dT_state == 1;
        ProfScope ps(ctx, PK_STRICT);
        if (single_accession(p)) {
            // the only column that can be flagged is column 0; its segment sums land in slot 0 of the compact [n_seg, ld] rows
            q->reeval_path = 2;
            rc = launch_strict_single(ctx, p, q->d_row_idx, q->row0, q->d_w, skip, d_seg_off, chunk, q->n, 0, n_seg, nullptr, d_ncols,
                                      REEVAL_CAP, 1, 0, dim3((unsigned)std::min<int64_t>(n_seg, 65535)),
                                      (double *)ctx->ws_seg_score.p, (uint32_t *)ctx->ws_seg_miss.p, ld);
            if (rc) return rc;
        } else if (use_T) {
            q->reeval_path = 1;
#define LAUNCH_SPARSE_T(S, G)                                                                                      \
    hipLaunchKernelGGL((k_strict_sparse_T<S, G>), sgrid, dim3(256), 0, ctx->stream, p->dT, p->pitchT, q->d_row_idx, \
                       q->row0, q->d_w, d_seg_off, chunk, q->n, n_seg, d_cols, d_ncols, REEVAL_CAP, \
                       (double *)ctx->ws_seg_score.p, (uint32_t *)ctx->ws_seg_miss.p, ld)
            if (skip) {
                if (gather) LAUNCH_SPARSE_T(true, true); else LAUNCH_SPARSE_T(true, false);
            } else {
                if (gather) LAUNCH_SPARSE_T(false, true); else LAUNCH_SPARSE_T(false, false);
            }
#undef LAUNCH_SPARSE_T
        } else {
            q->reeval_path = 2;
#define LAUNCH_SPARSE(S, G)                                                                                        \
    hipLaunchKernelGGL((k_strict_sparse<S, G>), sgrid, dim3(256), 0, ctx->stream, p->d, p->kpitch, p->desc, q->d_row_idx, \
                       q->row0, q->d_w, d_seg_off, chunk, q->n, n_seg, d_cols, d_ncols, REEVAL_CAP, \
                       (double *)ctx->ws_seg_score.p, (uint32_t *)ctx->ws_seg_miss.p, ld)
            if (skip) {
                if (gather) LAUNCH_SPARSE(true, true); else LAUNCH_SPARSE(true, false);
            } else {
                if (gather) LAUNCH_SPARSE(false, true); else LAUNCH_SPARSE(false, false);
            }
#undef LAUNCH_SPARSE
        }
        HIPCHK(ctx, hipGetLastError());
    }
    ProfScope ps(ctx, PK_SCAN);
    if (tail)
        hipLaunchKernelGGL(k_once_tail, dim3(1), dim3(256), 0, ctx->stream, (const double *)ctx->ws_seg_score.p, n_seg, ld, d_ncols,
                           REEVAL_CAP, (double *)ctx->ws_tmp_score.p, d_cols, patch_score, tail->d_ninfo, tail->n_acc, tail->want_lik,
                           tail->state, tail->lik_tmp, tail->out);
    else
        hipLaunchKernelGGL(k_scan_few, dim3(1), dim3(256), 0, ctx->stream, (const double *)ctx->ws_seg_score.p, n_seg, ld,
                           d_ncols, REEVAL_CAP, (double *)ctx->ws_tmp_score.p, carry, d_cols, patch_score);
    HIPCHK(ctx, hipGetLastError());
    return SNPM_OK;
}

