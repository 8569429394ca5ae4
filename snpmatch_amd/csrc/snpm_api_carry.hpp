// snpm_api_carry.hpp -- C ABI: slab-streamed jobs -- carries (inside the extern "C" block of snpm_api.hip).
// Part of the one translation unit of libsnpmatch_hip.so: included by snpm_api.hip at this place, not on its own.
// ---------------------------------------------------------------------------------------------- slab-streamed jobs
// A panel larger than HBM is scored SNP slab after SNP slab; a snpm_carry holds the running per-accession totals
// so that the result equals one pass of the reference's chunk loop over the whole SNP axis (core/snpmatch.py:218-225).
int snpm_carry_create(snpm_ctx *ctx, int64_t n_acc, snpm_carry **out)
try {
    if (!ctx) return SNPM_ERR_BADARG;
    CHECK_ARG(ctx, out != nullptr && n_acc >= 1, "carry needs n_acc >= 1");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    snpm_carry *c = new snpm_carry();
    c->ctx = ctx;
    c->n_acc = n_acc;
    c->ld = ((n_acc + 255) / 256) * 256;
    hipError_t e = hipMalloc((void **)&c->own_score, (size_t)c->ld * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&c->own_ninfo, (size_t)c->ld * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_E, 256 + REEVAL_CAP * sizeof(int32_t));
    c->d_score = c->own_score;
    c->d_ninfo = c->own_ninfo;
    c->len = c->ld;
    if (e != hipSuccess) {
        if (c->own_score) (void)hipFree(c->own_score);
        if (c->own_ninfo) (void)hipFree(c->own_ninfo);
        delete c;
        return set_err(ctx, SNPM_ERR_OOM, "carry allocation failed: %s", hipGetErrorString(e));
    }
    c->d_ncols = (int *)((char *)c->d_E + 8);
    c->d_cols = (int32_t *)((char *)c->d_E + 256);
    ctx->carries.push_back(c);
    *out = c;
    return snpm_carry_reset(c);
} SNPM_GUARD(ctx)

int snpm_carry_reset(snpm_carry *c)
{
    CHECK_CARRY(c);
    snpm_ctx *ctx = c->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemsetAsync(c->d_score, 0, (size_t)c->len * sizeof(double), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(c->d_ninfo, 0, (size_t)c->len * sizeof(int64_t), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(c->d_E, 0, 256 + REEVAL_CAP * sizeof(int32_t), ctx->stream));
    c->n_rows = 0;
    c->n_slabs = 0;
    c->wsum = 0;
    c->mode = -1;
    c->n_cols = -1;
    c->all_integer = true;
    c->finished = false;
    return SNPM_OK;
}

int snpm_carry_free(snpm_carry *c)
{
    if (!c) return SNPM_OK;
    snpm_ctx *ctx = c->ctx;
    if (ctx) {
        if (hip_alive()) {
            (void)hipSetDevice(ctx->device);
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipFree(c->own_score);
            (void)hipFree(c->own_ninfo);
            (void)hipFree(c->d_E);
        }
        ctx->carries.erase(std::remove(ctx->carries.begin(), ctx->carries.end(), c), ctx->carries.end());
    }
    delete c;
    return SNPM_OK;
}

// totals live in caller-owned DEVICE buffers (float64 [n_acc], int64 [n_acc]; e.g. torch tensors feeding an
// all-gather) from the next reset on; NULL, NULL restores the carry's own buffers
int snpm_carry_bind_outputs(snpm_carry *c, void *d_score, void *d_ninfo)
{
    CHECK_CARRY(c);
    snpm_ctx *ctx = c->ctx;
    CHECK_ARG(ctx, (d_score == nullptr) == (d_ninfo == nullptr), "bind both outputs or neither");
    CHECK_ARG(ctx, c->n_slabs == 0, "bind the outputs before the first slab");
    c->d_score = d_score ? (double *)d_score : c->own_score;
    c->d_ninfo = d_ninfo ? (int64_t *)d_ninfo : c->own_ninfo;
    c->len = d_score ? c->n_acc : c->ld;
    return snpm_carry_reset(c);
}

int snpm_carry_set_columns(snpm_carry *c, const int32_t *cols, int64_t ncols)
{
    CHECK_CARRY(c);
    snpm_ctx *ctx = c->ctx;
    CHECK_ARG(ctx, c->n_slabs == 0, "set the column list before the first slab");
    CHECK_ARG(ctx, ncols >= 1 && ncols <= REEVAL_CAP && cols, "a column list holds 1..64 accessions (more: a strict pass over all of them)");
    for (int64_t i = 0; i < ncols; ++i) CHECK_ARG(ctx, cols[i] >= 0 && cols[i] < c->n_acc, "accession index outside the panel");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int nc = (int)ncols;
    HIPCHK(ctx, hipMemcpyAsync(c->d_cols, cols, (size_t)ncols * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(c->d_ncols, &nc, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    c->n_cols = ncols;
    return SNPM_OK;
}

int snpm_query_run_carry(snpm_query *q, int64_t chunk, int skip_hets, int mode, int64_t chunks_after, snpm_carry *c)
try {
    CHECK_QUERY(q);
    CHECK_CARRY(c);
    snpm_ctx *ctx = q->panel->ctx;
    snpm_panel *p = q->panel;
    CHECK_ARG(ctx, c->ctx == ctx && c->n_acc == p->n_acc, "the carry belongs to another context or panel width");
    CHECK_ARG(ctx, chunk >= 1 && chunks_after >= 0, "chunk must be >= 1 and chunks_after >= 0");
    CHECK_ARG(ctx, mode == SNPM_MODE_EXACT || mode == SNPM_MODE_STRICT || mode == SNPM_MODE_FAST, "unknown mode");
    CHECK_ARG(ctx, !c->finished, "the carry was finished: reset it first");
    CHECK_ARG(ctx, c->mode < 0 || c->mode == mode, "every slab of a job is scored in the same mode");
    // the reference's chunks are cut over the whole SNP axis: a slab boundary must be a chunk boundary
    CHECK_ARG(ctx, chunks_after == 0 || q->n % chunk == 0, "every slab but the last must hold a multiple of `chunk` rows");
    CHECK_ARG(ctx, c->n_cols < 0 || mode == SNPM_MODE_STRICT, "a column-list carry takes strict slabs");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = wait_upload(p);
    if (rc) return rc;
    const int skip = skip_hets ? 1 : 0;
    q->transient_panel = true;
    q->count_valid = false;
    if (c->n_cols >= 0) {
        // second pass: the listed accessions only, chain continued from the compact totals
        rc = run_strict_sparse(q, skip, chunk, c->d_cols, c->d_ncols, c->d_score);
        if (rc) return rc;
        HIPCHK(ctx, hipMemcpyAsync(c->d_score, ctx->ws_tmp_score.p, (size_t)c->n_cols * sizeof(double),
                                   hipMemcpyDeviceToDevice, ctx->stream));
    } else if (mode == SNPM_MODE_STRICT) {
        q->last_kernel = "k_strict4";
        rc = run_strict_chain(q, skip, chunk, nullptr, c->d_score, c->d_ninfo, c->d_score, c->d_ninfo);
        if (rc) return rc;
    } else {
        Certify cert;
        cert.on = (mode == SNPM_MODE_EXACT);
        cert.flag = false;                  // certified once, over the totals (snpm_carry_finish)
        cert.chunk = chunk;
        cert.chunks_after = chunks_after;
        FastGeom g;
        rc = run_fast(q, skip, &g, cert);
        if (rc) return rc;
        // The job's bound takes the reference-order term of EVERY slab: a slab of integer weights is exact on its own
        // (run_fast skips its bound), but in a job that also holds non-integer slabs the reference adds this slab's chunk
        // sums onto a non-integer running total, so its terms pick up gamma(chunks left) like any others.  Only a job
        // whose slabs are all integer is exact in any order (snpm_carry_finish then flags nothing).
        const bool bounded = cert.on && q->n > 0;
        if (bounded && q->all_integer) {
            rc = ensure_eref(q, chunk, chunks_after);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(k_carry_add, dim3((unsigned)((p->n_acc + 255) / 256)), dim3(256), 0, ctx->stream, c->d_score,
                           c->d_ninfo, (const double *)q->d_score, (const int64_t *)q->d_ninfo, p->n_acc, c->d_E,
                           bounded ? (const double *)q->cert_eref() : (const double *)nullptr,
                           (bounded && !q->all_integer) ? efast_bound(q, g) : 0.0);
        HIPCHK(ctx, hipGetLastError());
        c->all_integer = c->all_integer && (q->all_integer || q->n == 0);
    }
    c->mode = mode;
    c->n_rows += q->n;
    c->n_slabs += 1;
    c->wsum += q->wsum;
    return SNPM_OK;
} SNPM_GUARD(((q && q->panel) ? q->panel->ctx : nullptr))

// integer weights in every slab and totals below 2^53: every partial sum of either order is exact
static bool carry_is_exact(const snpm_carry *c) { return c->all_integer && c->wsum < 9.0e15L; }

// the slab totals were added in slab order: n_slabs more additions per term
static double carry_e_extra(const snpm_carry *c)
{
    const double u = 1.1102230246251565e-16;
    const double m = (double)(c->n_slabs + 1);
    return (double)(c->wsum * (long double)(m * u / (1.0 - m * u))) * 1.0000001;
}

int snpm_carry_error_bound(snpm_carry *c, double *bound)
{
    CHECK_CARRY(c);
    snpm_ctx *ctx = c->ctx;
    CHECK_ARG(ctx, bound != nullptr, "bound is NULL");
    *bound = 0.0;
    if (c->mode != SNPM_MODE_EXACT || carry_is_exact(c)) return SNPM_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = ensure_pinned(ctx, 64);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, c->d_E, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *bound = *(const double *)ctx->h_pinned + carry_e_extra(c);
    return SNPM_OK;
}

int snpm_carry_finish(snpm_carry *c, double *score, int64_t *ninfo, int32_t *flagged, int64_t cap, int64_t *n_flagged)
{
    CHECK_CARRY(c);
    snpm_ctx *ctx = c->ctx;
    CHECK_ARG(ctx, c->n_cols < 0, "a column-list carry is read with snpm_carry_patch");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int64_t nf = 0;
    if (c->mode == SNPM_MODE_EXACT && !carry_is_exact(c)) {
        const double e_extra = carry_e_extra(c);
        HIPCHK(ctx, hipMemsetAsync(c->d_ncols, 0, sizeof(int), ctx->stream));
        hipLaunchKernelGGL(k_carry_flag, dim3((unsigned)((c->n_acc + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const double *)c->d_score, c->n_acc, (const double *)c->d_E, e_extra, ctx->debug_reeval,
                           c->d_cols, c->d_ncols, REEVAL_CAP);
        HIPCHK(ctx, hipGetLastError());
        int rc = ensure_pinned(ctx, 1024);
        if (rc) return rc;
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, c->d_ncols, 256 - 8 + REEVAL_CAP * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        nf = *(const int *)ctx->h_pinned;
        const int32_t *hc = (const int32_t *)((const char *)ctx->h_pinned + 256 - 8);
        if (flagged)
            for (int64_t i = 0; i < std::min<int64_t>(std::min<int64_t>(nf, REEVAL_CAP), cap); ++i) flagged[i] = hc[i];
    }
    c->finished = true;
    if (n_flagged) *n_flagged = nf;
    if (score) HIPCHK(ctx, hipMemcpyAsync(score, c->d_score, (size_t)c->n_acc * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (ninfo) HIPCHK(ctx, hipMemcpyAsync(ninfo, c->d_ninfo, (size_t)c->n_acc * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SNPM_OK;
}

// totals[cols[i]] = reference-order totals of the second pass (a column-list carry)
int snpm_carry_patch(snpm_carry *totals, const snpm_carry *cols_pass)
{
    CHECK_CARRY(totals);
    CHECK_CARRY(cols_pass);
    snpm_ctx *ctx = totals->ctx;
    CHECK_ARG(ctx, cols_pass->ctx == ctx && cols_pass->n_cols >= 1 && cols_pass->n_acc == totals->n_acc, "not a column-list carry of this job");
    CHECK_ARG(ctx, cols_pass->n_rows == totals->n_rows, "the second pass covered other rows than the first");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_patch, dim3(1), dim3(REEVAL_CAP), 0, ctx->stream, (const double *)cols_pass->d_score,
                       (const int32_t *)cols_pass->d_cols, (const int *)cols_pass->d_ncols, REEVAL_CAP, totals->d_score);
    HIPCHK(ctx, hipGetLastError());
    return SNPM_OK;
}

int snpm_carry_device_ptrs(snpm_carry *c, void **d_score, void **d_ninfo)
{
    CHECK_CARRY(c);
    if (d_score) *d_score = c->d_score;
    if (d_ninfo) *d_ninfo = c->d_ninfo;
    return SNPM_OK;
}

