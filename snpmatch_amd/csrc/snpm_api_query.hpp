// snpm_api_query.hpp -- C ABI: queries, windows, batches of samples (inside the extern "C" block of snpm_api.hip).
// Part of the one translation unit of libsnpmatch_hip.so: included by snpm_api.hip at this place, not on its own.
// ---------------------------------------------------------------------------------------------- query
// buffers of a new query; weights / row list are filled by the caller
static int query_alloc_all(snpm_panel *p, int64_t n, bool gather, snpm_query **out)
{
    snpm_ctx *ctx = p->ctx;
    snpm_query *q = new snpm_query();
    q->panel = p;
    p->queries.push_back(q);
    q->n = n;
    const size_t nn = (size_t)std::max<int64_t>(n, 1);
    hipError_t e = hipSuccess;
    if (gather) e = query_alloc(q, (void **)&q->d_row_idx, (nn + PREFETCH_PAD_ROWS) * sizeof(int64_t));
    if (e == hipSuccess) e = query_alloc(q, (void **)&q->d_w, nn * 3 * sizeof(double));
    if (e == hipSuccess) e = query_alloc(q, (void **)&q->d_lut, nn * 4 * sizeof(double));
    if (e == hipSuccess) e = query_alloc(q, (void **)&q->own_score, (size_t)p->ld * sizeof(double));
    if (e == hipSuccess) e = query_alloc(q, (void **)&q->own_ninfo, (size_t)p->ld * sizeof(int64_t));
    if (e == hipSuccess) e = query_alloc(q, &q->d_cert, 16 + REEVAL_CAP * sizeof(int32_t));
    q->d_score = q->own_score;
    q->d_ninfo = q->own_ninfo;
    if (e != hipSuccess) {
        snpm_query_free(q);
        return set_err(ctx, SNPM_ERR_OOM, "query allocation failed: %s", hipGetErrorString(e));
    }
    *out = q;
    return SNPM_OK;
}

// weight properties from the device copy (k_wprops): wsum, all-integer, hard 0/1 calls (+ the weight bits for the
// bit-parallel pass on packed panels).  Synchronises: the caller's host buffers are free afterwards.
static int query_finish_setup(snpm_query *q)
{
    snpm_panel *p = q->panel;
    snpm_ctx *ctx = p->ctx;
    const int64_t n = q->n;
    if (q->d_row_idx)      // pad entries: a valid row (0), only ever prefetched
        HIPCHK(ctx, hipMemsetAsync(q->d_row_idx + n, 0, PREFETCH_PAD_ROWS * sizeof(int64_t), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(q->d_cert, 0, 16, ctx->stream));
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 1024));
    int rc = ensure(ctx, ctx->ws_wprops, (size_t)grid * sizeof(double) + 64);
    if (rc) return rc;
    rc = ensure_pinned(ctx, (size_t)grid * sizeof(double) + 64);
    if (rc) return rc;
    int *d_flags = (int *)((char *)ctx->ws_wprops.p + (size_t)grid * sizeof(double));
    HIPCHK(ctx, hipMemsetAsync(d_flags, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_wprops, dim3((unsigned)grid), dim3(256), 0, ctx->stream, (const double *)q->d_w, n,
                       (double *)ctx->ws_wprops.p, d_flags);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, ctx->ws_wprops.p, (size_t)grid * sizeof(double) + sizeof(int),
                               hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    long double tot = 0;
    const double *part = (const double *)ctx->h_pinned;
    for (int i = 0; i < grid; ++i) tot += part[i];
    const int flags = *(const int *)((const char *)ctx->h_pinned + (size_t)grid * sizeof(double));
    if (flags & 4) return set_err(ctx, SNPM_ERR_BADARG, "SNP weights must be finite (a NaN or infinite weight was given)");
    q->wsum = (double)tot * 1.0000001;              // block sums carry ~1e-13 relative rounding: round up
    q->all_integer = !(flags & 1) && tot < 9.0e15L; // every partial sum exactly representable
    q->hard01 = q->all_integer && !(flags & 2);
    if (q->hard01 && p->packed && n > 0) {
        const int64_t padded = n + 16;
        hipError_t e2 = query_alloc(q, (void **)&q->d_wbits, (size_t)padded);
        if (e2 != hipSuccess) return set_err(ctx, SNPM_ERR_OOM, "query allocation failed: %s", hipGetErrorString(e2));
        hipLaunchKernelGGL(k_wbits, dim3((unsigned)((padded + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const double *)q->d_w, n, padded, q->d_wbits);
        HIPCHK(ctx, hipGetLastError());
    } else {
        q->hard01 = q->hard01 && p->packed;
    }
    return SNPM_OK;
}

int snpm_query_create(snpm_panel *p, const int64_t *row_idx, int64_t row0, int64_t n, const double *wei,
                      snpm_query **out)
try {
    CHECK_PANEL(p);
    snpm_ctx *ctx = p->ctx;
    CHECK_ARG(ctx, out != nullptr, "out is NULL");
    CHECK_ARG(ctx, n >= 0, "n must be >= 0");
    CHECK_ARG(ctx, n == 0 || wei != nullptr, "SNP weights should be a np.array with  shape == n,3");
    if (row_idx) {
        for (int64_t i = 0; i < n; ++i)
            if (row_idx[i] < 0 || row_idx[i] >= p->n_snp)
                return set_err(ctx, SNPM_ERR_BADARG, "row index %lld at %lld outside the panel (n_snp %lld)",
                               (long long)row_idx[i], (long long)i, (long long)p->n_snp);
    } else {
        CHECK_ARG(ctx, row0 >= 0 && row0 + n <= p->n_snp, "dense row range outside the panel");
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    snpm_query *q = nullptr;
    int rc = query_alloc_all(p, n, row_idx != nullptr, &q);
    if (rc) return rc;
    q->row0 = row_idx ? 0 : row0;
    hipError_t e = hipSuccess;
    if (n > 0 && row_idx)
        e = hipMemcpyAsync(q->d_row_idx, row_idx, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
    if (n > 0 && e == hipSuccess)
        e = hipMemcpyAsync(q->d_w, wei, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) rc = set_err(ctx, SNPM_ERR_HIP, "query upload failed: %s", hipGetErrorString(e));
    if (!rc) rc = query_finish_setup(q);
    if (rc) {
        const std::string keep = ctx->err;
        (void)hipStreamSynchronize(ctx->stream);
        snpm_query_free(q);
        ctx->err = keep;
        return rc;
    }
    *out = q;
    return SNPM_OK;
} SNPM_GUARD((p ? p->ctx : nullptr))

// Same query from DEVICE arrays (row list and weights already in HBM, e.g. produced by snpm_sample_synthetic or
// by a caller's own kernels); both are copied, the caller keeps ownership of its buffers.  Row indices must lie
// inside the panel: the caller guarantees it (they are not read back to the host).
int snpm_query_create_device(snpm_panel *p, const void *d_row_idx, int64_t row0, int64_t n, const void *d_wei,
                             snpm_query **out)
try {
    CHECK_PANEL(p);
    snpm_ctx *ctx = p->ctx;
    CHECK_ARG(ctx, out != nullptr, "out is NULL");
    CHECK_ARG(ctx, n >= 0, "n must be >= 0");
    CHECK_ARG(ctx, n == 0 || d_wei != nullptr, "SNP weights should be a np.array with  shape == n,3");
    if (!d_row_idx) CHECK_ARG(ctx, row0 >= 0 && row0 + n <= p->n_snp, "dense row range outside the panel");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    snpm_query *q = nullptr;
    int rc = query_alloc_all(p, n, d_row_idx != nullptr, &q);
    if (rc) return rc;
    q->row0 = d_row_idx ? 0 : row0;
    hipError_t e = hipSuccess;
    if (n > 0 && d_row_idx)
        e = hipMemcpyAsync(q->d_row_idx, d_row_idx, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToDevice, ctx->stream);
    if (n > 0 && e == hipSuccess)
        e = hipMemcpyAsync(q->d_w, d_wei, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) rc = set_err(ctx, SNPM_ERR_HIP, "query copy failed: %s", hipGetErrorString(e));
    if (!rc) rc = query_finish_setup(q);
    if (rc) {
        const std::string keep = ctx->err;
        (void)hipStreamSynchronize(ctx->stream);
        snpm_query_free(q);
        ctx->err = keep;
        return rc;
    }
    *out = q;
    return SNPM_OK;
} SNPM_GUARD((p ? p->ctx : nullptr))

int snpm_query_free(snpm_query *q)
{
    if (!q) return SNPM_OK;
    snpm_panel *p = q->panel;
    if (p) {                                // NULL: the panel or the context went first
        const bool use_hip = hip_alive() && p->ctx;
        if (use_hip) (void)hipSetDevice(p->ctx->device);
        p->queries.erase(std::remove(p->queries.begin(), p->queries.end(), q), p->queries.end());
        orphan_query(q, use_hip);
    }
    delete q;
    return SNPM_OK;
}

int snpm_query_bind_outputs(snpm_query *q, void *d_score, void *d_ninfo)
{
    CHECK_QUERY(q);
    snpm_ctx *ctx = q->panel->ctx;
    CHECK_ARG(ctx, (d_score == nullptr) == (d_ninfo == nullptr), "bind both outputs or neither");
    q->d_score = d_score ? (double *)d_score : q->own_score;
    q->d_ninfo = d_ninfo ? (int64_t *)d_ninfo : q->own_ninfo;
    return SNPM_OK;
}

int snpm_query_error_bound(snpm_query *q, int64_t chunk, double *bound)
try {
    CHECK_QUERY(q);
    if (!bound) return SNPM_ERR_BADARG;
    snpm_ctx *ctx = q->panel->ctx;
    CHECK_ARG(ctx, chunk >= 1, "chunk must be >= 1");
    if (q->all_integer || q->n == 0) { *bound = 0.0; return SNPM_OK; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = ensure_eref(q, chunk, 0);
    if (rc) return rc;
    rc = ensure_pinned(ctx, 64);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, q->cert_eref(), sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const double eref = *(const double *)ctx->h_pinned;
    // The fast pass's share, bounded over EVERY geometry a run may pick (two probe geometries did not dominate the
    // one run_fast chooses from the measured occupancy, its part multipliers and block shapes): a term passes through at most
    // min(n, rows of an epoch) additions inside a part, REDUCE_GROUP in its group and one per group afterwards, and there are
    // at most (parts + epochs) / REDUCE_GROUP + 1 groups with parts <= 16 resident blocks x 16 parts each per CU and an epoch
    // never shorter than EPOCH_TILES tiles of 16 rows.
    const double u = 1.1102230246251565e-16;
    const int64_t epoch_adds = (int64_t)EPOCH_TILES * 255;                                  // the longest tile any kernel walks
    const int64_t max_parts = (int64_t)ctx->n_cu * 16 * 16;
    const int64_t max_slots = max_parts + q->n / ((int64_t)EPOCH_TILES * 16) + 2;
    const double m = (double)(std::min<int64_t>(q->n, epoch_adds) + REDUCE_GROUP + (max_slots + REDUCE_GROUP - 1) / REDUCE_GROUP + 3);
    *bound = eref + (q->wsum * (m * u / (1.0 - m * u))) * 1.0000001;
    return SNPM_OK;
} SNPM_GUARD(((q && q->panel) ? q->panel->ctx : nullptr))

// After a certified fast pass: both re-evaluation tiers are enqueued behind it and decide on the device whether
// they have anything to do (see dense_tier_off / the sparse kernels), so the host never waits for the flag count.
// The accession-major copy is built the first time something is flagged on a long query -- the one case that
// reads the count back (once per panel).
// dense_tier = false (snpm_genotype_once): the > REEVAL_CAP tier is left to the caller, who sees the count with its results and
// runs run_strict_chain only then -- two launches that almost never have work stay out of a short sample's critical path.
static int enqueue_reevaluation(snpm_query *q, int skip, int64_t chunk, bool dense_tier = true, const OnceTail *tail = nullptr)
{
    snpm_ctx *ctx = q->panel->ctx;
    snpm_panel *p = q->panel;
    const bool want_T = ctx->use_acc_major && !q->transient_panel && q->n >= ctx->acc_major_min_rows && !single_accession(p);
    if (want_T && p->dT_state == 0) {
        int64_t cnt = 0;
        int rc = read_count(q, &cnt);
        if (rc) return rc;
        if (cnt >= 1 && cnt <= REEVAL_CAP) (void)ensure_acc_major(p);
    }
    int rc = run_strict_sparse(q, skip, chunk, q->cert_cols(), q->cert_count(), nullptr, nullptr, 0, q->d_score, tail);
    if (rc || !dense_tier) return rc;
    // more than REEVAL_CAP flagged (many exact-integer scores, e.g. clonal accessions): everything in reference order
    return run_strict_chain(q, skip, chunk, q->cert_count(), nullptr, nullptr, q->d_score, q->d_ninfo);
}

int snpm_query_run_device(snpm_query *q, int64_t chunk, int skip_hets, int mode, void **d_score, void **d_ninfo,
                          int64_t *info)
try {
    CHECK_QUERY(q);
    snpm_ctx *ctx = q->panel->ctx;
    snpm_panel *p = q->panel;
    CHECK_ARG(ctx, chunk >= 1, "chunk must be >= 1");
    CHECK_ARG(ctx, mode == SNPM_MODE_EXACT || mode == SNPM_MODE_STRICT || mode == SNPM_MODE_FAST, "unknown mode");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = wait_upload(p);
    if (rc) return rc;
    const int skip = skip_hets ? 1 : 0;
    bool certified = false;

    if (mode == SNPM_MODE_STRICT) {
        q->count_valid = false;
        q->last_kernel = "k_strict4";
        rc = run_strict_chain(q, skip, chunk, nullptr, nullptr, nullptr, q->d_score, q->d_ninfo);
        if (rc) return rc;
    } else {
        Certify cert;
        cert.on = (mode == SNPM_MODE_EXACT);
        cert.chunk = chunk;
        rc = run_fast(q, skip, nullptr, cert);
        if (rc) return rc;
        certified = cert.on && !q->all_integer && q->n > 0;
        if (certified) {
            rc = enqueue_reevaluation(q, skip, chunk);
            if (rc) return rc;
        }
    }
    if (d_score) *d_score = q->d_score;
    if (d_ninfo) *d_ninfo = q->d_ninfo;
    if (info) {                 // asking for the counters costs a synchronisation
        int64_t n_flag = 0;
        if (certified) {
            rc = read_count(q, &n_flag);
            if (rc) return rc;
        }
        info[0] = n_flag;
        info[1] = q->all_integer ? 1 : 0;
        info[2] = n_flag > REEVAL_CAP ? 3 : (n_flag > 0 ? q->reeval_path : 0);
        info[3] = 0;
    }
    return SNPM_OK;
} SNPM_GUARD(((q && q->panel) ? q->panel->ctx : nullptr))

int snpm_query_last_reeval(snpm_query *q, int64_t *n_flagged)
{
    CHECK_QUERY(q);
    if (!n_flagged) return SNPM_ERR_BADARG;
    HIPCHK(q->panel->ctx, hipSetDevice(q->panel->ctx->device));
    if (q->all_integer || q->n == 0) { *n_flagged = 0; return SNPM_OK; }
    return read_count(q, n_flagged);
}

const char *snpm_query_last_kernel(const snpm_query *q) { return q ? q->last_kernel : ""; }

int snpm_query_run(snpm_query *q, int64_t chunk, int skip_hets, int mode, double *score, int64_t *ninfo, int64_t *info)
{
    int rc = snpm_query_run_device(q, chunk, skip_hets, mode, nullptr, nullptr, info);
    if (rc) return rc;
    snpm_ctx *ctx = q->panel->ctx;
    const size_t na = (size_t)q->panel->n_acc;
    if (score) HIPCHK(ctx, hipMemcpyAsync(score, q->d_score, na * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (ninfo) HIPCHK(ctx, hipMemcpyAsync(ninfo, q->d_ninfo, na * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SNPM_OK;
}

static int run_windows_impl(snpm_query *q, const int64_t *win_off, int64_t n_win, int skip_hets, double *score,
                            int64_t *ninfo, double *tot_score, int64_t *tot_ninfo, snpm_carry *carry)
{
    snpm_ctx *ctx = q->panel->ctx;
    snpm_panel *p = q->panel;
    CHECK_ARG(ctx, n_win >= 0 && win_off != nullptr, "window offsets missing");
    for (int64_t w = 0; w < n_win; ++w)
        CHECK_ARG(ctx, win_off[w] <= win_off[w + 1], "window offsets must be non-decreasing");
    CHECK_ARG(ctx, win_off[0] >= 0 && win_off[n_win] <= q->n, "window offsets outside the matched list");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = wait_upload(p);
    if (rc) return rc;
    const int skip = skip_hets ? 1 : 0;
    q->count_valid = false;
    std::vector<int64_t> off(win_off, win_off + n_win + 1);
    rc = upload_seg_off(ctx, off);
    if (rc) return rc;
    rc = ensure(ctx, ctx->ws_seg_score, (size_t)std::max<int64_t>(n_win, 1) * p->ld * sizeof(double));
    if (rc) return rc;
    rc = ensure(ctx, ctx->ws_seg_miss, (size_t)std::max<int64_t>(n_win, 1) * p->ld * sizeof(uint32_t));
    if (rc) return rc;
    rc = launch_strict_dense(q, skip, (const int64_t *)ctx->ws_seg_off.p, 0, 0, n_win, nullptr);
    if (rc) return rc;
    const int thr = 256;
    const size_t na = (size_t)p->n_acc;
    // results through the pinned slab (HostFetch) -- except for slab-streamed jobs, which do not wait here and hand in pinned
    // buffers of their own
    HostFetch fetch(ctx);
    {
        const size_t per_win = (size_t)n_win * na * 8, tot = na * 8;
        if ((rc = fetch.reserve(carry ? 0 : ((score ? per_win + 64 : 0) + (ninfo ? per_win + 64 : 0) + (tot_score ? tot + 64 : 0) + (tot_ninfo ? tot + 64 : 0))))) return rc;
    }
    if (carry) {
        // the windows of a DB scored slab after slab: the reference's totals run window after window over the whole genome
        // (core/csmatch.py:88-90), so the chain of additions continues from the carry, in place
        ProfScope ps(ctx, PK_SCAN);
        hipLaunchKernelGGL(k_scan, dim3((unsigned)((p->n_acc + thr - 1) / thr)), dim3(thr), 0, ctx->stream,
                           (const double *)ctx->ws_seg_score.p, (const uint32_t *)ctx->ws_seg_miss.p,
                           win_off[n_win] - win_off[0], n_win, p->ld, p->n_acc, carry->d_score, carry->d_ninfo,
                           (const double *)carry->d_score, (const int64_t *)carry->d_ninfo, (const int *)nullptr, 0);
        HIPCHK(ctx, hipGetLastError());
    } else if (tot_score || tot_ninfo) {
        {
            ProfScope ps(ctx, PK_SCAN);
            hipLaunchKernelGGL(k_scan, dim3((unsigned)((p->n_acc + thr - 1) / thr)), dim3(thr), 0, ctx->stream,
                               (const double *)ctx->ws_seg_score.p, (const uint32_t *)ctx->ws_seg_miss.p,
                               win_off[n_win] - win_off[0], n_win, p->ld, p->n_acc, q->d_score, q->d_ninfo,
                               (const double *)nullptr, (const int64_t *)nullptr, (const int *)nullptr, 0);
            HIPCHK(ctx, hipGetLastError());
        }
        if ((rc = fetch.add(tot_score, q->d_score, na * sizeof(double)))) return rc;
        if ((rc = fetch.add(tot_ninfo, q->d_ninfo, na * sizeof(int64_t)))) return rc;
    }
    if ((score || ninfo) && n_win > 0) {
        rc = ensure(ctx, ctx->ws_tmp_score, (size_t)n_win * na * sizeof(double));
        if (rc) return rc;
        rc = ensure(ctx, ctx->ws_tmp_ninfo, (size_t)n_win * na * sizeof(int64_t));
        if (rc) return rc;
        // grid.y is limited to 65535: loop in slabs of windows
        for (int64_t w0 = 0; w0 < n_win; w0 += 32768) {
            const int64_t nw = std::min<int64_t>(32768, n_win - w0);
            hipLaunchKernelGGL(k_seg_pack, dim3((unsigned)((p->n_acc + thr - 1) / thr), (unsigned)nw), dim3(thr), 0,
                               ctx->stream, (const double *)ctx->ws_seg_score.p + w0 * p->ld,
                               (const uint32_t *)ctx->ws_seg_miss.p + w0 * p->ld,
                               (const int64_t *)ctx->ws_seg_off.p + w0, nw, p->ld, p->n_acc,
                               (double *)ctx->ws_tmp_score.p + w0 * p->n_acc, (int64_t *)ctx->ws_tmp_ninfo.p + w0 * p->n_acc);
            HIPCHK(ctx, hipGetLastError());
        }
        if ((rc = fetch.add(score, ctx->ws_tmp_score.p, (size_t)n_win * na * sizeof(double)))) return rc;
        if ((rc = fetch.add(ninfo, ctx->ws_tmp_ninfo.p, (size_t)n_win * na * sizeof(int64_t)))) return rc;
    }
    // slab-streamed jobs do not wait here: the next slab is loaded while this one is scored; the per-window rows arrive in the
    // caller's (pinned) buffers by the time snpm_carry_finish / snpm_synchronize returns
    if (!carry) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        fetch.finish();
    }
    return SNPM_OK;
}

int snpm_query_run_windows(snpm_query *q, const int64_t *win_off, int64_t n_win, int skip_hets, double *score,
                           int64_t *ninfo, double *tot_score, int64_t *tot_ninfo)
try {
    CHECK_QUERY(q);
    return run_windows_impl(q, win_off, n_win, skip_hets, score, ninfo, tot_score, tot_ninfo, nullptr);
} SNPM_GUARD(((q && q->panel) ? q->panel->ctx : nullptr))

// the same for one SNP slab of a DB that is scored slab after slab: slabs hold whole windows, the totals continue in `carry`
// (reference order, fp64 bits of one pass over all windows); read them with snpm_carry_finish
int snpm_query_run_windows_carry(snpm_query *q, const int64_t *win_off, int64_t n_win, int skip_hets, double *score,
                                 int64_t *ninfo, snpm_carry *c)
try {
    CHECK_QUERY(q);
    CHECK_CARRY(c);
    snpm_ctx *ctx = q->panel->ctx;
    CHECK_ARG(ctx, c->ctx == ctx && c->n_acc == q->panel->n_acc, "the carry belongs to another context or panel width");
    CHECK_ARG(ctx, !c->finished && c->n_cols < 0, "the carry was finished (reset it first) or holds a column list");
    CHECK_ARG(ctx, c->mode < 0 || c->mode == SNPM_MODE_STRICT, "every slab of a job is scored in the same mode");
    q->transient_panel = true;
    int rc = run_windows_impl(q, win_off, n_win, skip_hets, score, ninfo, nullptr, nullptr, c);
    if (rc) return rc;
    c->mode = SNPM_MODE_STRICT;
    c->n_rows += win_off[n_win] - win_off[0];
    c->n_slabs += 1;
    return SNPM_OK;
} SNPM_GUARD(((q && q->panel) ? q->panel->ctx : nullptr))

// CrossIdentifier.window_genotyper at streaming speed: one segmented fast pass over all windows, then the
// certificate per (window, accession) -- pairs whose int(score) is not proven are re-scored in reference order and
// patched in -- and once more for the totals.  snps_match = int(score), snps_info and the totals' counts are
// bit-exact; fp64 window scores are within the per-window bound (~1e-12) of the reference's, likelihoods follow at
// that relative accuracy.  snpm_query_run_windows stays the mode whose fp64 scores carry the reference's bits.
int snpm_query_run_windows_fast(snpm_query *q, const int64_t *win_off, int64_t n_win, int skip_hets, double *score,
                                int64_t *ninfo, double *tot_score, int64_t *tot_ninfo, int64_t *info)
try {
    CHECK_QUERY(q);
    snpm_ctx *ctx = q->panel->ctx;
    snpm_panel *p = q->panel;
    CHECK_ARG(ctx, n_win >= 0 && win_off != nullptr, "window offsets missing");
    CHECK_ARG(ctx, n_win <= 65535, "at most 65535 windows per call");
    int64_t maxlen = 1;
    for (int64_t w = 0; w < n_win; ++w) {
        CHECK_ARG(ctx, win_off[w] <= win_off[w + 1], "window offsets must be non-decreasing");
        maxlen = std::max(maxlen, win_off[w + 1] - win_off[w]);
    }
    CHECK_ARG(ctx, win_off[0] >= 0 && win_off[n_win] <= q->n, "window offsets outside the matched list");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (info) info[0] = info[1] = info[2] = info[3] = 0;
    if (n_win == 0) return snpm_query_run_windows(q, win_off, n_win, skip_hets, score, ninfo, tot_score, tot_ninfo);
    int rc = wait_upload(p);
    if (rc) return rc;
    const int skip = skip_hets ? 1 : 0;
    rc = ensure_lut(q, skip);
    if (rc) return rc;
    const size_t na = (size_t)p->n_acc;
    if ((rc = ensure(ctx, ctx->ws_bscore, (size_t)n_win * na * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, ctx->ws_bninfo, (size_t)n_win * na * sizeof(int64_t)))) return rc;
    q->count_valid = false;
    SegJob j;
    j.p = p;
    j.d_row_idx = q->d_row_idx;
    j.row0 = q->row0;
    j.n_total = q->n;
    j.d_w = q->d_w;
    j.d_lut = q->d_lut;
    j.seg_off = win_off;
    j.n_seg = n_win;
    j.chunk = maxlen;                          // a window is ONE matchGTsAccs call
    j.skip = skip;
    j.certify = true;
    j.d_score = (double *)ctx->ws_bscore.p;
    j.d_ninfo = (int64_t *)ctx->ws_bninfo.p;
    j.ldo = p->n_acc;
    rc = run_segmented(ctx, j);
    if (rc) return rc;
    // totals in window order, certified against the reference's chain over its own (bit-different) window scores
    rc = ensure(ctx, ctx->ws_flags, 64);
    if (rc) return rc;
    double *d_etot = (double *)((char *)ctx->ws_flags.p + 8);
    hipLaunchKernelGGL(k_tot_seg, dim3((unsigned)((p->n_acc + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const double *)ctx->ws_bscore.p, (const int64_t *)ctx->ws_bninfo.p, n_win, p->n_acc, p->n_acc,
                       q->d_score, q->d_ninfo, (const double *)ctx->ws_eseg.p, d_etot);
    HIPCHK(ctx, hipGetLastError());
    const double u = 1.1102230246251565e-16;
    const double m = (double)(n_win + 2);
    const double e_extra = q->all_integer ? 0.0 : 2.0 * q->wsum * (m * u / (1.0 - m * u)) * 1.0000001;
    HIPCHK(ctx, hipMemsetAsync(q->cert_count(), 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_carry_flag, dim3((unsigned)((p->n_acc + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const double *)q->d_score, p->n_acc, (const double *)d_etot, e_extra, ctx->debug_reeval,
                       q->cert_cols(), q->cert_count(), REEVAL_CAP);
    HIPCHK(ctx, hipGetLastError());
    rc = run_strict_sparse(q, skip, maxlen, q->cert_cols(), q->cert_count(), nullptr, j.d_seg_off, n_win, q->d_score);
    if (rc) return rc;
    HostFetch fetch(ctx);
    {
        const size_t per_win = (size_t)n_win * na * 8, tot = na * 8;
        if ((rc = fetch.reserve((score ? per_win + 64 : 0) + (ninfo ? per_win + 64 : 0) + (tot_score ? tot + 64 : 0) + (tot_ninfo ? tot + 64 : 0)))) return rc;
    }
    int *h_cnt = (int *)ctx->h_pinned;
    HIPCHK(ctx, hipMemcpyAsync(h_cnt, seg_pair_count(ctx), sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h_cnt + 1, q->cert_count(), sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = fetch.add(score, ctx->ws_bscore.p, (size_t)n_win * na * sizeof(double)))) return rc;
    if ((rc = fetch.add(ninfo, ctx->ws_bninfo.p, (size_t)n_win * na * sizeof(int64_t)))) return rc;
    if ((rc = fetch.add(tot_score, q->d_score, na * sizeof(double)))) return rc;
    if ((rc = fetch.add(tot_ninfo, q->d_ninfo, na * sizeof(int64_t)))) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const int n_pairs = h_cnt[0], n_tot = h_cnt[1];
    if (!(n_pairs > j.cap || n_tot > REEVAL_CAP)) fetch.finish();      // (else: every window again in reference order, below)
    if (info) { info[0] = n_pairs; info[1] = n_tot; }
    if (n_pairs > j.cap || n_tot > REEVAL_CAP) {
        // more uncertain results than the sparse tiers take: every window in reference order
        if (info) info[2] = 1;
        return snpm_query_run_windows(q, win_off, n_win, skip_hets, score, ninfo, tot_score, tot_ninfo);
    }
    return SNPM_OK;
} SNPM_GUARD(((q && q->panel) ? q->panel->ctx : nullptr))

// Many samples against one resident panel in ONE call (the reference scores one sample per process,
// core/snpmatch.py:256-268): sample b owns entries [sample_off[b], sample_off[b+1]) of the concatenated matched-row list
// and weights.  One segmented fast pass (sample = segment), certificate per (sample, accession), reference-order
// re-evaluation of the flagged pairs, one likelihood launch with a row per sample, one copy back.
static int score_batch_impl(snpm_panel *p, int64_t n_samples, const int64_t *sample_off, const void *row_idx, const void *wei,
                            const uint16_t *codes, const double *table, int64_t table_len, int device_inputs, int64_t chunk, int skip_hets,
                            int mode, double *score, int64_t *ninfo, double *lik, double *lrt, int64_t *info, bool allow_shared = true)
{
    CHECK_PANEL(p);
    snpm_ctx *ctx = p->ctx;
    CHECK_ARG(ctx, n_samples >= 0 && n_samples <= 65535 && sample_off, "between 0 and 65535 samples per call");
    CHECK_ARG(ctx, chunk >= 1, "chunk must be >= 1");
    CHECK_ARG(ctx, mode == SNPM_MODE_EXACT || mode == SNPM_MODE_STRICT || mode == SNPM_MODE_FAST, "unknown mode");
    if (info) info[0] = info[1] = info[2] = info[3] = 0;
    if (n_samples == 0) return SNPM_OK;
    CHECK_ARG(ctx, sample_off[0] == 0, "sample offsets start at 0");
    for (int64_t b = 0; b < n_samples; ++b) CHECK_ARG(ctx, sample_off[b] <= sample_off[b + 1], "sample offsets must be non-decreasing");
    const int64_t N = sample_off[n_samples];
    CHECK_ARG(ctx, N == 0 || (row_idx && (wei || (codes && table))), "please provide same number of positions for both sample and db");
    CHECK_ARG(ctx, (lik == nullptr) == (lrt == nullptr), "ask for both likelihood outputs or neither");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = wait_upload(p);
    if (rc) return rc;
    const int skip = skip_hets ? 1 : 0;
    const size_t NN = (size_t)std::max<int64_t>(N, 1);
    const size_t na = (size_t)p->n_acc, B = (size_t)n_samples;
    if ((rc = ensure(ctx, ctx->ws_brows, (NN + PREFETCH_PAD_ROWS) * sizeof(int64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->ws_blut, NN * 4 * sizeof(double)))) return rc;
    // the results of the batch in ONE device block, [status line | score | ninfo | likelihood | ratio]: they return to the host
    // in one copy (five copies of 0.6 MB and four of a flag each were ~0.16 ms of GPU timeline behind a 0.6-ms batch)
    const size_t out_slot = (B * na * sizeof(double) + 255) & ~size_t(255);
    if ((rc = ensure(ctx, ctx->ws_bout, 256 + 4 * out_slot))) return rc;
    int *const d_status = (int *)ctx->ws_bout.p;
    double *const d_bscore = (double *)((char *)ctx->ws_bout.p + 256);
    int64_t *const d_bninfo = (int64_t *)((char *)ctx->ws_bout.p + 256 + out_slot);
    double *const d_blik = (double *)((char *)ctx->ws_bout.p + 256 + 2 * out_slot);
    double *const d_blrt = (double *)((char *)ctx->ws_bout.p + 256 + 3 * out_slot);
    if ((rc = ensure(ctx, ctx->ws_flags2, sizeof(int)))) return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->ws_flags2.p, 0, sizeof(int), ctx->stream));
    int64_t *d_rows = (int64_t *)ctx->ws_brows.p;
    const double *d_w = nullptr;
    if (device_inputs) {
        d_w = (const double *)wei;
    } else {
        if ((rc = ensure(ctx, ctx->ws_bw, NN * 3 * sizeof(double)))) return rc;
        if ((rc = ensure(ctx, ctx->ws_brows32, NN * sizeof(int32_t)))) return rc;
        if (codes) {
            if ((rc = ensure(ctx, ctx->ws_bcodes, NN * 3 * sizeof(uint16_t) + 65536 * sizeof(double) + 64))) return rc;
            // the table travels first (ordered before every expansion kernel on the compute stream)
            HIPCHK(ctx, hipMemsetAsync(ctx->ws_bcodes.p, 0, 65536 * sizeof(double), ctx->stream));
            HIPCHK(ctx, hipMemcpyAsync(ctx->ws_bcodes.p, table, (size_t)table_len * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));      // `table` is the caller's
        }
        d_w = (const double *)ctx->ws_bw.p;
    }
    // the scoring kernel prefetches (never scores) a few row-list entries past a part: every entry it can reach must be
    // a row of THIS panel before the first launch -- also the ones whose upload is still on its way
    SegJob j;
    j.p = p;
    j.d_row_idx = d_rows;
    j.n_total = N;
    j.d_w = d_w;
    j.d_lut = (const double *)ctx->ws_blut.p;
    j.seg_off = sample_off;
    j.n_seg = n_samples;
    j.chunk = chunk;
    j.skip = skip;
    j.certify = (mode == SNPM_MODE_EXACT);
    j.d_score = d_bscore;
    j.d_ninfo = d_bninfo;
    j.ldo = p->n_acc;
    // rows [r0, r1) of the concatenated inputs are on the device (or on their way, ordered before what follows):
    // sanitise the row list, build the LUT rows
    auto check_rows = [&](int64_t r0, int64_t r1, const int32_t *rows32) -> int {
        if (r1 <= r0) return SNPM_OK;
        const int64_t n = r1 - r0;
        hipLaunchKernelGGL(k_check_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_rows + r0, rows32, n,
                           p->n_snp, (int *)ctx->ws_flags2.p);
        HIPCHK(ctx, hipGetLastError());
        return SNPM_OK;
    };
    auto build_lut = [&](int64_t r0, int64_t r1) -> int {
        if (r1 <= r0) return SNPM_OK;
        const int64_t n = r1 - r0;
        ProfScope ps(ctx, PK_LUT);
        hipLaunchKernelGGL(k_build_lut, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_w + 3 * r0,
                           (double *)ctx->ws_blut.p + 4 * r0, n, skip, (int *)ctx->ws_flags2.p);
        HIPCHK(ctx, hipGetLastError());
        return SNPM_OK;
    };
    auto prepare_rows = [&](int64_t r0, int64_t r1, const int32_t *rows32) -> int {
        int r = check_rows(r0, r1, rows32);
        return r ? r : build_lut(r0, r1);
    };
    // every sample through the reference-order chain (requested, or more uncertain pairs than the sparse tier takes)
    auto strict_every_sample = [&]() -> int {
        for (int64_t b = 0; b < n_samples; ++b) {
            snpm_query *q = nullptr;
            const int64_t o = sample_off[b], nb = sample_off[b + 1] - o;
            int r = snpm_query_create_device(p, j.d_row_idx + o, 0, nb, d_w + 3 * o, &q);
            if (r) return r;
            r = run_strict_chain(q, skip, chunk, nullptr, nullptr, nullptr, d_bscore + b * na, d_bninfo + b * na);
            const std::string keep = ctx->err;
            (void)hipStreamSynchronize(ctx->stream);
            snpm_query_free(q);
            if (r) { ctx->err = keep; return r; }
        }
        return SNPM_OK;
    };
    bool strict_all = (mode == SNPM_MODE_STRICT);
    const bool trace = getenv("SNPM_BATCH_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    double t_stage = 0, t_launch = 0;
    // Shared-row scan (snpm_api_shared.hpp): when the samples were genotyped on largely the same markers, every DB row is read
    // once and scored against all of them (int8 MFMA contraction of fixed-point weight digits), same certificate, same
    // reference-order re-evaluation of the unproven pairs.  It needs the whole batch on the device first: batches from host
    // memory keep the per-sample pass and its upload overlap unless the policy says "whenever possible".
    const int policy = ctx->batch_shared;
    const bool try_shared = allow_shared && !strict_all && policy != 0 && (policy > 0 || device_inputs);
    SharedStats shst;
    shst.reason = 1;
    bool shared_done = false;
    if (device_inputs && try_shared) {
        // inputs already in HBM: the pass reads the caller's row list and weights in place (its first kernel vets every index
        // and weight); only a batch it declines pays for the sanitised copy and the LUT rows of the per-sample pass
        SegJob js = j;
        js.d_row_idx = (const int64_t *)row_idx;
        rc = shared_rows_try(ctx, js, policy > 0, shst);
        if (rc) return rc;
        if (shst.taken) {
            j = js;
            shared_done = true;
        }
    }
    if (device_inputs && !shared_done) {
        HIPCHK(ctx, hipMemcpyAsync(d_rows, row_idx, (size_t)N * sizeof(int64_t), hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(d_rows + N, 0, PREFETCH_PAD_ROWS * sizeof(int64_t), ctx->stream));
    } else if (!device_inputs) {
        HIPCHK(ctx, hipMemsetAsync(d_rows, 0, ((size_t)N + PREFETCH_PAD_ROWS) * sizeof(int64_t), ctx->stream));
    }
    SegPlan pl;
    if (!strict_all && !shared_done) {
        rc = seg_plan(ctx, j, pl);
        if (rc) return rc;
    }
    const double t_planned = now();
    auto shared_or_segments = [&]() -> int {        // host batch, whole on the device now (ordered before what is enqueued here)
        const SegJob keep = j;
        int r = shared_rows_try(ctx, j, policy > 0, shst);
        if (r) return r;
        if (shst.taken) {
            shared_done = true;
        } else {
            j = keep;
            r = build_lut(0, N);
            if (!r) r = seg_launch(ctx, j, pl, 0, n_samples);
        }
        return r;
    };
    if (shared_done) {
        // scored above
    } else if (device_inputs) {
        rc = prepare_rows(0, N, nullptr);
        if (!rc && !strict_all) rc = seg_launch(ctx, j, pl, 0, n_samples);
        if (rc) return rc;
    } else {
        // Host inputs: the batch is cut into runs of samples of about one staging slab; while run k is scored, run
        // k + 1 travels over PCIe on the copy stream and the host fills the slab of run k + 2.
        rc = ensure_stage(ctx);
        if (rc) return rc;
        if (!ctx->batch_ev) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->batch_ev, hipEventDisableTiming));
        if (!ctx->compute_mark) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->compute_mark, hipEventDisableTiming));
        // the device arenas may still be read by the previous call's kernels: the copy stream waits for them
        HIPCHK(ctx, hipEventRecord(ctx->compute_mark, ctx->stream));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->compute_mark, 0));
        const bool pinned_codes = codes && N > 0 && host_pointer_is_pinned(codes);
        const bool pinned = N > 0 && host_pointer_is_pinned(row_idx) && (codes ? pinned_codes : host_pointer_is_pinned(wei));
        // a run = what one staging slab takes of the widest per-row item (24 B of fp64 weights, 6 B of codes), but about a
        // quarter of the batch at most, so that uploads and launches overlap without the launches becoming small
        const int64_t slab_rows = (int64_t)(snpm_ctx::kStageBytes / (codes ? 8 : 32));
        const int64_t rows_per_run = std::max<int64_t>(1, std::min<int64_t>(slab_rows, std::max<int64_t>(N / 4 + 1, 262144)));
        int64_t s0 = 0;
        while (s0 < n_samples) {
            int64_t s1 = s0 + 1;
            while (s1 < n_samples && sample_off[s1 + 1] - sample_off[s0] <= rows_per_run) ++s1;
            const int64_t r0 = sample_off[s0], r1 = sample_off[s1];
            // the row list crosses PCIe as int32 (the link is what bounds a batch: 28 instead of 32 bytes per matched SNP);
            // it is narrowed while the staging slab is filled and widened again by k_check_rows
            const int32_t *rows32 = nullptr;
            const double ts0 = now();
            if (!pinned && r1 - r0 <= rows_per_run) {
                rc = stage_rows32(ctx, (int32_t *)ctx->ws_brows32.p + r0, (const int64_t *)row_idx + r0, r1 - r0);
                rows32 = (const int32_t *)ctx->ws_brows32.p + r0;
            } else {
                rc = stage_bytes(ctx, d_rows + r0, (const int64_t *)row_idx + r0, (size_t)(r1 - r0) * sizeof(int64_t), pinned);
            }
            uint16_t *d_codes = codes ? (uint16_t *)((char *)ctx->ws_bcodes.p + 65536 * sizeof(double)) : nullptr;
            if (!rc && codes) rc = stage_bytes(ctx, d_codes + 3 * r0, codes + 3 * r0, (size_t)(r1 - r0) * 3 * sizeof(uint16_t), pinned_codes);
            else if (!rc) rc = stage_bytes(ctx, (double *)ctx->ws_bw.p + 3 * r0, (const double *)wei + 3 * r0,
                                           (size_t)(r1 - r0) * 3 * sizeof(double), pinned);
            if (rc) return rc;
            HIPCHK(ctx, hipEventRecord(ctx->batch_ev, ctx->copy_stream));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->batch_ev, 0));
            const double ts1 = now();
            t_stage += ts1 - ts0;
            if (codes && r1 > r0) {      // weights of these rows from their codes (3 B per matched SNP crossed PCIe)
                const int64_t n3 = (r1 - r0) * 3;
                hipLaunchKernelGGL(k_expand_codes, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, ctx->stream,
                                   (const uint16_t *)d_codes + 3 * r0, (const double *)ctx->ws_bcodes.p, n3,
                                   (double *)ctx->ws_bw.p + 3 * r0);
                HIPCHK(ctx, hipGetLastError());
            }
            rc = try_shared ? check_rows(r0, r1, rows32) : prepare_rows(r0, r1, rows32);
            if (!rc && !strict_all && !try_shared) rc = seg_launch(ctx, j, pl, s0, s1);
            if (rc) return rc;
            t_launch += now() - ts1;
            s0 = s1;
        }
        if (try_shared && (rc = shared_or_segments())) return rc;
    }
    ctx->shared_last[0] = shst.taken; ctx->shared_last[1] = shst.reason; ctx->shared_last[2] = shst.union_rows;
    ctx->shared_last[3] = (int64_t)(shst.density * 1e6); ctx->shared_last[4] = shst.tiles; ctx->shared_last[5] = shst.groups;
    ctx->shared_last[6] = shst.passes; ctx->shared_last[7] = shst.digits;
    const double t_enqueued = now();
    int n_pairs = 0;
    // Results return through the context's pinned slab: a copy into pageable memory is staged by the runtime piece by piece with the
    // host waiting in between (2.3 MB of results of 64 samples: ~0.35 ms of a 1.4-ms call), a copy into pinned memory is one DMA;
    // the host threads then move the slab into the caller's arrays.  Nothing waits before the end: the count of unproven pairs,
    // the likelihoods' domain flag and the input flags come back with the results (a batch with more unproven pairs than the
    // sparse tier takes -- rare -- is scored again in reference order and delivered a second time).
    const size_t out_elems = B * na;
    HostFetch fetch(ctx);
    if ((rc = fetch.reserve(256 + 4 * out_slot))) return rc;
    const int *h_status = (const int *)ctx->h_pinned;     // [0] unproven pairs, [1] likelihood domain flag, [2] input flags, [4..5] the contraction's flags
    auto deliver = [&]() -> int {
        int r;
        fetch.finish();                             // (a second delivery starts from an empty list)
        if (lik) {
            r = snpm_likelihood_device(ctx, d_bscore, d_bninfo, n_samples, p->n_acc, 1, __builtin_nan(""), d_blik, d_blrt, nullptr);
            if (r) return r;
        }
        hipLaunchKernelGGL(k_batch_status, dim3(1), dim3(64), 0, ctx->stream, (j.certify && !strict_all) ? (const int *)seg_pair_count(ctx) : nullptr,
                           lik ? (const int *)ctx->ws_flags.p : nullptr, (const int *)ctx->ws_flags2.p,
                           shared_done ? (const unsigned long long *)(shst.d_meta + 1) : nullptr, d_status);
        HIPCHK(ctx, hipGetLastError());
        void *dsts[4] = {score, ninfo, lik, lrt};
        const void *srcs[4] = {d_bscore, d_bninfo, d_blik, d_blrt};
        int last = -1;
        bool pinned_dst = false;
        for (int i = 0; i < 4; ++i)
            if (dsts[i]) { last = i; pinned_dst = pinned_dst || host_pointer_is_pinned(dsts[i]); }
        const size_t span = 256 + (last < 0 ? 0 : (size_t)last * out_slot + out_elems * 8);
        if (out_elems * 8 <= (size_t(1) << 20) && !pinned_dst && span <= fetch.cap) {
            // one copy: the status line and every array up to the last one asked for (an array not asked for in between rides along)
            char *slab = (char *)ctx->h_pinned + 256;
            HIPCHK(ctx, hipMemcpyAsync(slab, ctx->ws_bout.p, span, hipMemcpyDeviceToHost, ctx->stream));
            h_status = (const int *)slab;
            for (int i = 0; i < 4; ++i)
                if (dsts[i]) fetch.items[fetch.n_items++] = HostFetch::Item{dsts[i], slab + 256 + (size_t)i * out_slot, out_elems * 8};
        } else {
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, d_status, 64, hipMemcpyDeviceToHost, ctx->stream));
            h_status = (const int *)ctx->h_pinned;
            for (int i = 0; i < 4; ++i)
                if ((r = fetch.add(dsts[i], srcs[i], out_elems * 8))) return r;
        }
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if ((shared_done && (*(const long long *)(h_status + 4) & 2))               // a weight outside [0, 1]: not a batch for the contraction
            || (j.certify && !strict_all && h_status[0] > j.cap)) {                  // the caller scores again in reference order
            fetch.n_items = 0;
            fetch.used = 0;
            return SNPM_OK;
        }
        fetch.finish();
        return SNPM_OK;
    };
    if (strict_all) {
        rc = strict_every_sample();
        if (rc) return rc;
    } else {
        rc = seg_finish(ctx, j);
        if (rc) return rc;
    }
    if ((rc = deliver())) return rc;
    if (shared_done && (*(const long long *)(h_status + 4) & 2)) {
        // the expansion met a weight outside [0, 1] (it vets the weights while it converts them: the decision to take the shared-row
        // pass is made before): the whole batch once more through the per-sample pass, which takes any finite weight
        int r = score_batch_impl(p, n_samples, sample_off, row_idx, wei, codes, table, table_len, device_inputs, chunk, skip_hets, mode, score,
                                 ninfo, lik, lrt, info, false);
        ctx->shared_last[0] = 0;
        ctx->shared_last[1] = 4;
        if (info && !r) { info[2] = 0; }
        return r;
    }
    if (j.certify && !strict_all) {
        n_pairs = h_status[0];
        if (n_pairs > j.cap) {
            strict_all = true;
            if ((rc = strict_every_sample())) return rc;
            if ((rc = deliver())) return rc;
        }
    }
    const int *h_bad = h_status + 2;
    if (lik && !(*h_bad) && (h_status[1] & 1)) return set_err(ctx, SNPM_ERR_DOMAIN, "provided y is greater than n");
    if (trace)
        fprintf(stderr, "[snpm batch] plan %.3f ms, enqueue %.3f ms (staging %.3f, launches %.3f), finish+likelihood+copy back %.3f ms\n",
                t_planned - t_begin, t_enqueued - t_planned, t_stage, t_launch, now() - t_enqueued);
    if (*h_bad & 4) return set_err(ctx, SNPM_ERR_BADARG, "SNP weights must be finite (a NaN or infinite weight was given)");
    if (*h_bad) return set_err(ctx, SNPM_ERR_BADARG, "a row index lies outside the panel (n_snp %lld)", (long long)p->n_snp);
    if (info) { info[0] = n_pairs; info[1] = (strict_all && mode != SNPM_MODE_STRICT) ? 1 : 0; info[2] = shst.taken; info[3] = shst.union_rows; }
    return SNPM_OK;
}

int snpm_batch_configure(snpm_ctx *ctx, int shared_rows, int digits, double min_density)
{
    if (!ctx) return set_err(nullptr, SNPM_ERR_BADARG, "ctx is NULL");
    CHECK_ARG(ctx, shared_rows >= -1 && shared_rows <= 1, "shared_rows: -1 auto, 0 never, 1 whenever the batch allows it");
    CHECK_ARG(ctx, digits == -1 || digits == 0 || (digits >= 3 && digits <= 7), "digits: 3..7, -1 chosen by the longest sample, 0 keeps the current value");
    ctx->batch_shared = shared_rows;
    if (digits) ctx->shared_digits = digits < 0 ? 0 : digits;
    if (min_density >= 0.0) ctx->shared_min_density = min_density;
    return SNPM_OK;
}

int snpm_batch_last_stats(snpm_ctx *ctx, int64_t *stats)
{
    if (!ctx) return set_err(nullptr, SNPM_ERR_BADARG, "ctx is NULL");
    CHECK_ARG(ctx, stats != nullptr, "stats is NULL");
    for (int i = 0; i < 8; ++i) stats[i] = ctx->shared_last[i];
    return SNPM_OK;
}

int snpm_score_batch(snpm_panel *p, int64_t n_samples, const int64_t *sample_off, const void *row_idx, const void *wei,
                     int device_inputs, int64_t chunk, int skip_hets, int mode, double *score, int64_t *ninfo,
                     double *lik, double *lrt, int64_t *info)
try {
    return score_batch_impl(p, n_samples, sample_off, row_idx, wei, nullptr, nullptr, 0, device_inputs, chunk, skip_hets, mode,
                            score, ninfo, lik, lrt, info);
} SNPM_GUARD((p ? p->ctx : nullptr))

// The same batch with DICTIONARY-CODED weights: wei[r, c] = table[codes[r, c]] (codes uint16 [N, 3], table float64
// [table_len <= 65536], both host; codes >= table_len read 0.0).  A VCF sample's weights are exp(-PL/10) of integer PLs
// (core/parsers.py:141-151): the caller computes the table entries with its own libm (numpy), so the device weights carry
// exactly the bits the fp64 path would have received, while 6 + 4 instead of 24 + 8 bytes per matched SNP cross PCIe --
// the link is what bounds a batch from host memory.
int snpm_score_batch_coded(snpm_panel *p, int64_t n_samples, const int64_t *sample_off, const int64_t *row_idx,
                           const uint16_t *codes, const double *table, int64_t table_len, int64_t chunk, int skip_hets,
                           int mode, double *score, int64_t *ninfo, double *lik, double *lrt, int64_t *info)
try {
    if (p && p->ctx) CHECK_ARG(p->ctx, codes && table && table_len >= 1 && table_len <= 65536, "codes and a table of 1..65536 entries are required");
    return score_batch_impl(p, n_samples, sample_off, row_idx, nullptr, codes, table, table_len, 0, chunk, skip_hets, mode, score,
                            ninfo, lik, lrt, info);
} SNPM_GUARD((p ? p->ctx : nullptr))

