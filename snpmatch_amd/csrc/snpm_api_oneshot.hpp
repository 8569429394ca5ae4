// snpm_api_oneshot.hpp -- C ABI: one-shot forms -- dense matchGTsAccs, likelihood, identity, --refine scan, F1 pairs, host memory (inside the extern "C" block of snpm_api.hip).
// Part of the one translation unit of libsnpmatch_hip.so: included by snpm_api.hip at this place, not on its own.
// ---------------------------------------------------------------------------------------------- one-shot
int snpm_score_dense_host(snpm_ctx *ctx, const int8_t *db, int64_t db_pitch, int64_t n, int64_t n_acc,
                          const double *wei, int skip_hets, double *score, int64_t *ninfo)
{
    if (!ctx) return SNPM_ERR_BADARG;
    CHECK_ARG(ctx, n >= 0 && n_acc >= 1, "please provide same number of positions for both sample and db");
    CHECK_ARG(ctx, n == 0 || (db != nullptr && wei != nullptr), "NULL input");
    CHECK_ARG(ctx, db_pitch >= n_acc, "db_pitch smaller than n_acc");
    snpm_panel *p = nullptr;
    snpm_query *q = nullptr;
    int rc = snpm_panel_create(ctx, n, n_acc, &p);
    if (rc) return rc;
    rc = snpm_panel_upload_rows(p, 0, n, db, db_pitch);
    if (!rc) rc = snpm_query_create(p, nullptr, 0, n, wei, &q);
    if (!rc) {
        // one matchGTsAccs call == one segment over all n rows, reference order
        rc = snpm_query_run(q, std::max<int64_t>(n, 1), skip_hets, SNPM_MODE_STRICT, score, ninfo, nullptr);
    }
    std::string keep = ctx->err;
    if (q) snpm_query_free(q);
    (void)snpm_panel_upload_wait(p);
    snpm_panel_free(p);
    if (rc) ctx->err = keep;
    return rc;
}

int snpm_likelihood_device(snpm_ctx *ctx, const void *d_y, const void *d_n, int64_t m, int64_t len, int truncate,
                           double amin_or_nan, void *d_lik, void *d_lrt, int *domain_error)
{
    if (!ctx) return SNPM_ERR_BADARG;
    CHECK_ARG(ctx, m >= 0 && len >= 0, "negative size");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (domain_error) *domain_error = 0;
    if (m == 0 || len == 0) return SNPM_OK;
    int rc = ensure(ctx, ctx->ws_flags, sizeof(int));
    if (rc) return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->ws_flags.p, 0, sizeof(int), ctx->stream));
    {
        ProfScope ps(ctx, PK_LIK);
        const int thr = len >= 1024 ? 1024 : (len > 256 ? 512 : 256);
        for (int64_t r0 = 0; r0 < m; r0 += 1 << 30) {
            const int64_t mr = std::min<int64_t>(m - r0, 1 << 30);
            hipLaunchKernelGGL(k_likelihood, dim3((unsigned)mr), dim3(thr), 0, ctx->stream, (const double *)d_y + r0 * len,
                               (const int64_t *)d_n + r0 * len, len, truncate, amin_or_nan, (double *)d_lik + r0 * len,
                               (double *)d_lrt + r0 * len, (int *)ctx->ws_flags.p);
            HIPCHK(ctx, hipGetLastError());
        }
    }
    if (domain_error) {
        int flag = 0;
        HIPCHK(ctx, hipMemcpyAsync(&flag, ctx->ws_flags.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        *domain_error = flag & 1;
    }
    return SNPM_OK;
}

int snpm_likelihood(snpm_ctx *ctx, const double *y, const int64_t *n, int64_t m, int64_t len, int truncate,
                    double amin_or_nan, double *lik, double *lrt)
{
    if (!ctx) return SNPM_ERR_BADARG;
    CHECK_ARG(ctx, m >= 0 && len >= 0, "negative size");
    const size_t tot = (size_t)m * (size_t)len;
    if (tot == 0) return SNPM_OK;
    CHECK_ARG(ctx, y && n && lik && lrt, "NULL pointer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->ws_lik_y, tot * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, ctx->ws_lik_n, tot * sizeof(int64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->ws_lik_l, tot * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, ctx->ws_lik_r, tot * sizeof(double)))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->ws_lik_y.p, y, tot * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->ws_lik_n.p, n, tot * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    int dom = 0;
    rc = snpm_likelihood_device(ctx, ctx->ws_lik_y.p, ctx->ws_lik_n.p, m, len, truncate, amin_or_nan, ctx->ws_lik_l.p,
                                ctx->ws_lik_r.p, &dom);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(lik, ctx->ws_lik_l.p, tot * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(lrt, ctx->ws_lik_r.p, tot * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (dom) return set_err(ctx, SNPM_ERR_DOMAIN, "provided y is greater than n");
    return SNPM_OK;
}

// np_test_identity on the device (k_binom_identity); host pointers in and out.
int snpm_binom_identity(snpm_ctx *ctx, const double *x, const int64_t *n, int64_t len, double error_rate,
                        double pthres, int64_t *out, double *sf)
{
    if (!ctx) return SNPM_ERR_BADARG;
    CHECK_ARG(ctx, len >= 0, "negative size");
    if (len == 0) return SNPM_OK;
    CHECK_ARG(ctx, x && n && out, "NULL pointer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    const size_t L = (size_t)len;
    if ((rc = ensure(ctx, ctx->ws_lik_y, L * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, ctx->ws_lik_n, L * sizeof(int64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->ws_lik_l, L * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, ctx->ws_lik_r, L * sizeof(int64_t)))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->ws_lik_y.p, x, L * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->ws_lik_n.p, n, L * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_binom_identity, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const double *)ctx->ws_lik_y.p, (const int64_t *)ctx->ws_lik_n.p, len, error_rate, pthres,
                       (int64_t *)ctx->ws_lik_r.p, (double *)ctx->ws_lik_l.p);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->ws_lik_r.p, L * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    if (sf) HIPCHK(ctx, hipMemcpyAsync(sf, ctx->ws_lik_l.p, L * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SNPM_OK;
}

// host twin of k_binom_identity's arithmetic (no device needed): lets the CPU test-suite pin the algorithm
int snpm_binom_sf_host(const double *k, const double *n, int64_t len, double p, double *sf)
{
    if (len < 0 || (len > 0 && (!k || !n || !sf))) return SNPM_ERR_BADARG;
    for (int64_t i = 0; i < len; ++i) sf[i] = binom_sf_eval(k[i], n[i], p);
    return SNPM_OK;
}

// snpm_intersect_sorted / snpm_intersect_sorted_search: pure host code, in snpm_host.cpp (also built with
// -fsanitize=address,undefined by the CPU test-suite)

// identify_segregating_snps on the resident panel: mask [n_snp] (host, uint8); first (may be NULL): the first
// informative call of the listed accessions per row (0xFF = none), for accession-sharded DBs
static int panel_segregating(snpm_panel *p, const int32_t *cols, int64_t ncols, uint8_t *mask, uint8_t *first)
{
    CHECK_PANEL(p);
    snpm_ctx *ctx = p->ctx;
    CHECK_ARG(ctx, ncols >= 0 && (ncols == 0 || cols) && mask, "provide an np array for list of indices to be considered");
    for (int64_t i = 0; i < ncols; ++i) CHECK_ARG(ctx, cols[i] >= 0 && cols[i] < p->n_acc, "accession index outside the panel");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = wait_upload(p);
    if (rc) return rc;
    if (p->n_snp == 0) return SNPM_OK;
    if ((rc = ensure(ctx, ctx->ws_cols, (size_t)std::max<int64_t>(ncols, 1) * sizeof(int32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->ws_tmp_ninfo, (size_t)p->n_snp * 2))) return rc;
    if (ncols > 0)
        HIPCHK(ctx, hipMemcpyAsync(ctx->ws_cols.p, cols, (size_t)ncols * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    uint8_t *d_mask = (uint8_t *)ctx->ws_tmp_ninfo.p, *d_first = d_mask + p->n_snp;
    hipLaunchKernelGGL(k_segregating, dim3((unsigned)((p->n_snp + 255) / 256)), dim3(256), 0, ctx->stream, p->d, p->kpitch,
                       p->desc, p->n_snp, (const int32_t *)ctx->ws_cols.p, (int)ncols, d_mask, first ? d_first : (uint8_t *)nullptr);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(mask, d_mask, (size_t)p->n_snp, hipMemcpyDeviceToHost, ctx->stream));
    if (first) HIPCHK(ctx, hipMemcpyAsync(first, d_first, (size_t)p->n_snp, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SNPM_OK;
}

int snpm_panel_segregating(snpm_panel *p, const int32_t *cols, int64_t ncols, uint8_t *mask)
{
    if (p && p->ctx) CHECK_ARG(p->ctx, ncols >= 1, "provide an np array for list of indices to be considered");
    return panel_segregating(p, cols, ncols, mask, nullptr);
}

int snpm_panel_segregating_first(snpm_panel *p, const int32_t *cols, int64_t ncols, uint8_t *mask, uint8_t *first)
{
    if (!first) return SNPM_ERR_BADARG;
    return panel_segregating(p, cols, ncols, mask, first);
}

// calls of the listed accessions at the query's matched rows: codes [ncols, n] (uint8: 0 ref, 1 alt, 2 het, 3 other,
// 0xFF missing), host.  The g_acc.snps[:, i] reads of the reference (core/csmatch.py:116-117) for accession-sharded
// DBs: a rank hands the columns it holds to the rank that crosses them in silico.
int snpm_query_gather_columns(snpm_query *q, const int32_t *acc_idx, int ncols, uint8_t *codes)
try {
    CHECK_QUERY(q);
    snpm_panel *p = q->panel;
    snpm_ctx *ctx = p->ctx;
    CHECK_ARG(ctx, ncols >= 0 && ncols <= 4096, "between 0 and 4096 columns");
    if (ncols == 0 || q->n == 0) return SNPM_OK;
    CHECK_ARG(ctx, acc_idx && codes, "NULL argument");
    for (int i = 0; i < ncols; ++i) CHECK_ARG(ctx, acc_idx[i] >= 0 && acc_idx[i] < p->n_acc, "accession index outside the panel");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = wait_upload(p);
    if (rc) return rc;
    const int64_t n = q->n;
    const int64_t stride = (n + 255) / 256 * 256;
    if ((rc = ensure(ctx, ctx->ws_cols, (size_t)ncols * sizeof(int32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->ws_tmp_ninfo, (size_t)ncols * (size_t)stride))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->ws_cols.p, acc_idx, (size_t)ncols * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_f1_gather, dim3((unsigned)(stride / 256)), dim3(256), 0, ctx->stream, p->d, p->kpitch, p->desc,
                       (const int64_t *)q->d_row_idx, q->row0, n, (const int32_t *)ctx->ws_cols.p, ncols,
                       (uint8_t *)ctx->ws_tmp_ninfo.p, stride);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpy2DAsync(codes, (size_t)n, ctx->ws_tmp_ninfo.p, (size_t)stride, (size_t)n, (size_t)ncols,
                                 hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SNPM_OK;
} SNPM_GUARD(((q && q->panel) ? q->panel->ctx : nullptr))

// match_insilico_f1s (core/csmatch.py:115-125) on the resident panel: scores of all pairs of the selected
// accessions over the query's rows, in numpy's summation order (k_f1_* in snpm_kernels.hpp)
int snpm_query_f1_pairs(snpm_query *q, const int32_t *acc_idx, int n_sel, double *score, int64_t *ninfo)
try {
    CHECK_QUERY(q);
    snpm_panel *p = q->panel;
    snpm_ctx *ctx = p->ctx;
    CHECK_ARG(ctx, n_sel >= 0 && n_sel <= F1_MAX_SEL, "between 0 and 32 accessions can be crossed in silico");
    const int n_pairs = n_sel * (n_sel - 1) / 2;
    if (n_pairs == 0) return SNPM_OK;
    CHECK_ARG(ctx, acc_idx && score && ninfo, "NULL argument");
    for (int i = 0; i < n_sel; ++i) CHECK_ARG(ctx, acc_idx[i] >= 0 && acc_idx[i] < p->n_acc, "accession index outside the panel");
    CHECK_ARG(ctx, q->n < (int64_t(1) << 31), "too many matched SNPs for the in-silico crosses");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = wait_upload(p);
    if (rc) return rc;
    const int64_t n = q->n;
    if (n == 0) {
        for (int k = 0; k < n_pairs; ++k) { score[k] = 0.0; ninfo[k] = 0; }
        return SNPM_OK;
    }
    const int64_t stride = (n + F1_ROWS_PER_BLOCK - 1) / F1_ROWS_PER_BLOCK * F1_ROWS_PER_BLOCK;
    const int64_t nblk = stride / F1_ROWS_PER_BLOCK;
    const int64_t max_chunks = (n + NP_SUM_CHUNK - 1) / NP_SUM_CHUNK;
    // pairs per batch: the compacted weights of a pair take n doubles; keep the slab around 2 GiB
    const int batch = (int)std::max<int64_t>(1, std::min<int64_t>(n_pairs, ctx->f1_slab_bytes / (n * 8)));

    std::vector<int32_t> tab((size_t)n_sel + 2 * (size_t)n_pairs);      // [acc | (i, j) per pair, combination order]
    for (int i = 0; i < n_sel; ++i) tab[(size_t)i] = acc_idx[i];
    for (int i = 0, k = 0; i < n_sel; ++i)
        for (int j = i + 1; j < n_sel; ++j, ++k) {
            tab[(size_t)n_sel + 2 * k] = i;
            tab[(size_t)n_sel + 2 * k + 1] = j;
        }

    struct Scratch {                    // freed on every return path
        std::vector<void *> ptrs;
        ~Scratch() { for (void *x : ptrs) (void)hipFree(x); }
        hipError_t get(void **out, size_t bytes)
        {
            hipError_t e = hipMalloc(out, std::max<size_t>(bytes, 256));
            if (e == hipSuccess) ptrs.push_back(*out);
            return e;
        }
    } scratch;
    uint8_t *d_codes = nullptr;
    int32_t *d_tab = nullptr;
    uint32_t *d_cnt = nullptr, *d_m = nullptr;
    double *d_cw = nullptr, *d_chunk = nullptr, *d_score = nullptr;
    int64_t *d_ninfo = nullptr;
    HIPCHK(ctx, scratch.get((void **)&d_codes, (size_t)n_sel * stride));
    HIPCHK(ctx, scratch.get((void **)&d_tab, tab.size() * sizeof(int32_t)));
    HIPCHK(ctx, scratch.get((void **)&d_cnt, (size_t)batch * 3 * nblk * sizeof(uint32_t)));
    HIPCHK(ctx, scratch.get((void **)&d_m, (size_t)batch * 3 * sizeof(uint32_t)));
    HIPCHK(ctx, scratch.get((void **)&d_cw, (size_t)batch * n * sizeof(double)));
    HIPCHK(ctx, scratch.get((void **)&d_chunk, (size_t)batch * 3 * max_chunks * sizeof(double)));
    HIPCHK(ctx, scratch.get((void **)&d_score, (size_t)n_pairs * sizeof(double)));
    HIPCHK(ctx, scratch.get((void **)&d_ninfo, (size_t)n_pairs * sizeof(int64_t)));

    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(d_tab, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_f1_gather, dim3((unsigned)(stride / 256)), dim3(256), 0, st, p->d, p->kpitch, p->desc,
                       (const int64_t *)q->d_row_idx, q->row0, n, (const int32_t *)d_tab, n_sel, d_codes, stride);
    HIPCHK(ctx, hipGetLastError());
    const int32_t *d_pairs = d_tab + n_sel;
    for (int pair0 = 0; pair0 < n_pairs; pair0 += batch) {
        const int nb = std::min(batch, n_pairs - pair0);
        hipLaunchKernelGGL(k_f1_count, dim3((unsigned)nblk, (unsigned)nb), dim3(F1_BLOCK), 0, st, (const uint8_t *)d_codes,
                           stride, d_pairs, pair0, d_cnt, nblk);
        hipLaunchKernelGGL(k_f1_scan, dim3((unsigned)(nb * 3)), dim3(256), 0, st, d_cnt, nblk, d_m);
        hipLaunchKernelGGL(k_f1_compact, dim3((unsigned)nblk, (unsigned)nb), dim3(F1_BLOCK), 0, st, (const uint8_t *)d_codes,
                           stride, d_pairs, pair0, (const double *)q->d_w, n, (const uint32_t *)d_cnt, nblk,
                           (const uint32_t *)d_m, d_cw);
        hipLaunchKernelGGL(k_f1_chunks, dim3((unsigned)max_chunks, (unsigned)(nb * 3)), dim3(256), 0, st,
                           (const double *)d_cw, (const uint32_t *)d_m, n, max_chunks, d_chunk);
        hipLaunchKernelGGL(k_f1_finish, dim3((unsigned)nb), dim3(192), 0, st, (const double *)d_chunk,
                           (const uint32_t *)d_m, max_chunks, pair0, d_score, d_ninfo);
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipMemcpyAsync(score, d_score, (size_t)n_pairs * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(ninfo, d_ninfo, (size_t)n_pairs * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return SNPM_OK;
} SNPM_GUARD(((q && q->panel) ? q->panel->ctx : nullptr))

// PMC calibration helper: stream the whole panel once (known byte count = n_snp * pitch)
int snpm_debug_stream_read(snpm_panel *p, int64_t *bytes_read)
{
    CHECK_PANEL(p);
    snpm_ctx *ctx = p->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = ensure(ctx, ctx->ws_flags, sizeof(int));
    if (rc) return rc;
    rc = wait_upload(p);
    if (rc) return rc;
    const int64_t n_dwords = p->n_snp * p->pitch / 4;
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_dwords + 1023) / 1024, (int64_t)ctx->n_cu * 8));
    hipLaunchKernelGGL(k_calib_read, dim3(blocks), dim3(256), 0, ctx->stream, (const uint32_t *)p->d, n_dwords,
                       (uint32_t *)ctx->ws_flags.p);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (bytes_read) *bytes_read = n_dwords * 4;
    return SNPM_OK;
}

// pinned host memory for callers that want their batch inputs to travel at full PCIe speed without the staging copy
int snpm_host_alloc(snpm_ctx *ctx, int64_t bytes, void **out)
{
    if (!ctx) return SNPM_ERR_BADARG;
    CHECK_ARG(ctx, out != nullptr && bytes >= 0, "bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, (size_t)std::max<int64_t>(bytes, 1), hipHostMallocDefault);
    if (e != hipSuccess) return set_err(ctx, SNPM_ERR_OOM, "hipHostMalloc of %lld bytes failed: %s", (long long)bytes, hipGetErrorString(e));
    return SNPM_OK;
}

int snpm_host_free(snpm_ctx *ctx, void *ptr)
{
    if (!ctx) return SNPM_ERR_BADARG;
    if (ptr && hip_alive()) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipHostFree(ptr);
    }
    return SNPM_OK;
}

