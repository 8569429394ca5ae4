// snpm_once.hpp -- ONE sample against a resident panel in ONE call: snpm_genotype_once.
//
// Genotyper.genotyper (core/snpmatch.py:207-241) for a single sample used to be three synchronising ABI calls from Python --
// snpm_query_create (two pageable uploads, a device pass over the weights and a read-back of its flags), snpm_query_run (two
// copies back), snpm_likelihood (two copies up, two back) -- plus a numpy gather of the sample's matched weight rows in front:
// 0.55-0.7 ms per 200k-SNP sample around 0.06 ms of scoring (profiles/r04_real_panel_*).  Here:
//   1. the host thread pool gathers wei[sample_idx[i]] and the row list straight into ONE pinned slab and, on the way, checks
//      the row indices and collects the weight properties the kernels are chosen by (sum of max |w|, all-integer, hard 0/1,
//      finite) -- no device pass over the weights, no read-back before the launch decisions;
//   2. the slab goes up piece by piece behind the fill (copy stream), the compute stream waits for the last piece;
//   3. LUT, fast pass, ordered reduce + certificate, the gated reference-order tiers, likelihood / nanmin / ratio (on the
//      truncated counts, as GenotyperOutput does, :96, :106-117) are enqueued back to back;
//   4. (score, ninfo, likelihood, lrt, #re-evaluated) are packed into one buffer and come back in ONE copy, one synchronisation.
// Included by snpm_api.hip inside its extern "C" block.

namespace {

struct OnceProps {          // per task of the fill
    long double wsum = 0;
    int flags = 0;          // bit 0: a weight is not an integer (or huge); bit 1: a weight is neither 0 nor 1; bit 2: NaN / infinite
    int64_t bad_row = -1;   // position of a row index outside the panel
    char pad[64];
};

}  // namespace

int snpm_genotype_once(snpm_panel *p, const int64_t *row_idx, const double *wei, const int64_t *sample_idx, int64_t n_wei,
                       int64_t n, int64_t chunk, int skip_hets, int mode, double *score, int64_t *ninfo, double *lik,
                       double *lrt, int64_t *info)
try {
    CHECK_PANEL(p);
    snpm_ctx *ctx = p->ctx;
    CHECK_ARG(ctx, n >= 0 && n_wei >= 0, "n must be >= 0");
    CHECK_ARG(ctx, chunk >= 1, "chunk must be >= 1");
    CHECK_ARG(ctx, mode == SNPM_MODE_EXACT || mode == SNPM_MODE_STRICT || mode == SNPM_MODE_FAST, "unknown mode");
    CHECK_ARG(ctx, n == 0 || (row_idx && wei), "please provide same number of positions for both sample and db");
    CHECK_ARG(ctx, sample_idx || n <= n_wei, "please provide same number of positions for both sample and db");
    CHECK_ARG(ctx, score && ninfo && ((lik == nullptr) == (lrt == nullptr)), "score / ninfo outputs missing, or only one likelihood output");
    if (info) info[0] = info[1] = info[2] = info[3] = 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = wait_upload(p);
    if (rc) return rc;
    const size_t na = (size_t)p->n_acc;
    snpm_query *q = nullptr;
    rc = query_alloc_all(p, n, true, &q);
    if (rc) return rc;
    struct Guard {          // every exit frees the query (its buffers return to the context's cache)
        snpm_query *q;
        snpm_ctx *ctx;
        ~Guard()
        {
            const std::string keep = ctx->err;
            snpm_query_free(q);
            ctx->err = keep;
        }
    } guard{q, ctx};
    q->row0 = 0;

    // ---- 1 + 2: fill the pinned slab [rows int64 | weights fp64 x 3] with the pool, pieces go up behind the fill
    const size_t row_bytes = (size_t)n * sizeof(int64_t), wei_bytes = (size_t)n * 3 * sizeof(double);
    const size_t out_words = 4 * na + 2;
    rc = ensure_pinned(ctx, std::max<size_t>(row_bytes + wei_bytes + 64, out_words * sizeof(int64_t)));
    if (rc) return rc;
    int64_t *h_rows = (int64_t *)ctx->h_pinned;
    double *h_wei = (double *)((char *)ctx->h_pinned + row_bytes);
    HostPool *pool = host_pool(ctx);
    const int64_t piece = 4096;                                    // rows per task
    const int n_tasks = (int)((n + piece - 1) / piece);
    std::vector<OnceProps> props((size_t)std::max(n_tasks, 1));
    const int64_t n_snp = p->n_snp;
    // weight properties with integer tests on the bit patterns (no libm call per weight: this loop runs once per matched SNP)
    auto bits_of = [](double v) -> uint64_t { uint64_t b; memcpy(&b, &v, 8); return b; };
    auto fill = [&](int t) {
        const int64_t i0 = (int64_t)t * piece, i1 = std::min<int64_t>(n, i0 + piece);
        OnceProps pr;
        double wsum = 0.0;                                         // <= 4096 non-negative terms: good to 1e-12, rounded up below
        uint64_t any_frac = 0, any_not01 = 0, any_nonfinite = 0, any_huge = 0;
        for (int64_t i = i0; i < i1; ++i) {
            const int64_t r = row_idx[i];
            if ((uint64_t)r >= (uint64_t)n_snp && pr.bad_row < 0) pr.bad_row = i;
            h_rows[i] = r;
            int64_t s = sample_idx ? sample_idx[i] : i;
            if ((uint64_t)s >= (uint64_t)n_wei) {                  // a sample index outside the weight array: reported like a bad row
                if (pr.bad_row < 0) pr.bad_row = i;
                s = 0;
            }
            double m = 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double v = wei[3 * s + c];
                h_wei[3 * i + c] = v;
                const uint64_t b = bits_of(v), mag = b & 0x7FFFFFFFFFFFFFFFull;
                any_nonfinite |= (uint64_t)(mag >= 0x7FF0000000000000ull);
                any_not01 |= (uint64_t)(!(mag == 0 || b == 0x3FF0000000000000ull));
                const double a = fabs(v);
                any_huge |= (uint64_t)(!(a < 9.0e15));             // beyond 2^53 every double is an integer, but sums of them are not exact
                any_frac |= (uint64_t)(a < 9.0e15 && (double)(int64_t)a != a);
                m = a > m ? a : m;
            }
            wsum += m;
        }
        pr.wsum = wsum;
        pr.flags = ((any_frac | any_huge) ? 1 : 0) | (any_not01 ? 2 : 0) | (any_nonfinite ? 4 : 0);
        props[(size_t)t] = pr;
    };
    if (n > 0 && n_wei == 0) return set_err(ctx, SNPM_ERR_BADARG, "please provide same number of positions for both sample and db");
    static const bool trace = getenv("SNPM_ONCE_TRACE") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = trace ? now() : 0.0;
    // The slab goes up BEHIND the fill: task 0 of the pool run is the uploader -- it waits (in order) for the fill tasks of each
    // piece and enqueues that piece's two copies, while the other threads keep filling.  Without pool threads the
    // calling thread fills everything first.
    static const int kTasksPerPiece = std::max(1, getenv("SNPM_ONCE_PIECE_TASKS") ? atoi(getenv("SNPM_ONCE_PIECE_TASKS")) : 25);   // x 4096 rows: a 200k-SNP sample
                                                                                   // goes up in two pieces (every copy costs ~25 us on its own:
                                                                                   // 13 pieces 0.38 ms, 2 pieces 0.19 ms, 1 piece 0.21 ms of GPU-side wait)
    const int n_pieces = (n_tasks + kTasksPerPiece - 1) / kTasksPerPiece;
    std::vector<std::atomic<int>> piece_done((size_t)std::max(n_pieces, 1));
    for (auto &c : piece_done) c.store(0, std::memory_order_relaxed);
    std::atomic<int> upload_error{0};
    auto upload_piece = [&](int k) -> bool {
        const int64_t i0 = (int64_t)k * kTasksPerPiece * piece, i1 = std::min<int64_t>(n, i0 + (int64_t)kTasksPerPiece * piece);
        if (hipMemcpyAsync(q->d_row_idx + i0, h_rows + i0, (size_t)(i1 - i0) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(q->d_w + 3 * i0, h_wei + 3 * i0, (size_t)(i1 - i0) * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
            return false;
        return true;
    };
    const bool overlapped = pool->size() > 0 && n_pieces > 1;
    if (overlapped) {
        pool->run(n_tasks + 1, [&](int t) {
            if (t == 0) {                                          // the uploader
                if (hipSetDevice(ctx->device) != hipSuccess) { upload_error.store(1); return; }
                for (int k = 0; k < n_pieces; ++k) {
                    const int need = std::min(kTasksPerPiece, n_tasks - k * kTasksPerPiece);
                    while (piece_done[(size_t)k].load(std::memory_order_acquire) < need) {
#if defined(__x86_64__)
                        _mm_pause();
#endif
                    }
                    if (!upload_piece(k)) { upload_error.store(1); return; }
                }
                return;
            }
            fill(t - 1);
            piece_done[(size_t)((t - 1) / kTasksPerPiece)].fetch_add(1, std::memory_order_release);
        });
    } else {
        pool->run(n_tasks, fill);
    }
    const double t_filled = trace ? now() : 0.0;
    long double tot = 0;
    int flags = 0;
    int64_t bad_at = -1;
    for (int t = 0; t < n_tasks; ++t) {
        if (props[(size_t)t].bad_row >= 0 && bad_at < 0) bad_at = props[(size_t)t].bad_row;
        tot += props[(size_t)t].wsum;
        flags |= props[(size_t)t].flags;
    }
    if (bad_at >= 0 || (flags & 4) || upload_error.load()) {
        (void)hipStreamSynchronize(ctx->stream);                   // pieces already on their way read the slab
        if (bad_at >= 0)
            return set_err(ctx, SNPM_ERR_BADARG, "row index %lld at %lld outside the panel (n_snp %lld), or a sample index outside the weights",
                           (long long)row_idx[bad_at], (long long)bad_at, (long long)n_snp);
        if (flags & 4) return set_err(ctx, SNPM_ERR_BADARG, "SNP weights must be finite (a NaN or infinite weight was given)");
        return set_err(ctx, SNPM_ERR_HIP, "upload of the sample failed");
    }
    if (!overlapped)
        for (int k = 0; k < n_pieces; ++k)
            if (!upload_piece(k)) return set_err(ctx, SNPM_ERR_HIP, "upload of the sample failed");
    HIPCHK(ctx, hipMemsetAsync(q->d_row_idx + n, 0, PREFETCH_PAD_ROWS * sizeof(int64_t), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(q->d_cert, 0, 16, ctx->stream));
    q->wsum = (double)tot * 1.0000001;                  // the same margin as the device sum of query_finish_setup
    q->all_integer = !(flags & 1) && tot < 9.0e15L;
    q->hard01 = q->all_integer && !(flags & 2);
    if (q->hard01 && p->packed && n > 0) {
        const int64_t padded = n + 16;
        hipError_t e2 = query_alloc(q, (void **)&q->d_wbits, (size_t)padded);
        if (e2 != hipSuccess) return set_err(ctx, SNPM_ERR_OOM, "query allocation failed: %s", hipGetErrorString(e2));
        hipLaunchKernelGGL(k_wbits, dim3((unsigned)((padded + 255) / 256)), dim3(256), 0, ctx->stream, (const double *)q->d_w, n,
                           padded, q->d_wbits);
        HIPCHK(ctx, hipGetLastError());
    } else {
        q->hard01 = q->hard01 && p->packed;
    }

    // ---- 3: the scoring pipeline of snpm_query_run_device, then the likelihoods of the truncated counts
    void *d_s = nullptr, *d_n = nullptr;
    rc = snpm_query_run_device(q, chunk, skip_hets, mode, &d_s, &d_n, nullptr);
    if (rc) return rc;
    const bool certified = (mode == SNPM_MODE_EXACT) && !q->all_integer && n > 0;
    if ((rc = ensure(ctx, ctx->ws_lik_l, na * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, ctx->ws_lik_r, na * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, ctx->ws_once, out_words * sizeof(int64_t)))) return rc;
    if (lik) {
        rc = snpm_likelihood_device(ctx, d_s, d_n, 1, (int64_t)na, 1, __builtin_nan(""), ctx->ws_lik_l.p, ctx->ws_lik_r.p, nullptr);
        if (rc) return rc;
    }
    // ---- 4: one packed copy back, one synchronisation
    hipLaunchKernelGGL(k_once_pack, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, ctx->stream, (const double *)d_s,
                       (const int64_t *)d_n, lik ? (const double *)ctx->ws_lik_l.p : nullptr,
                       lik ? (const double *)ctx->ws_lik_r.p : nullptr, certified ? (const int *)q->cert_count() : nullptr,
                       lik ? (const int *)ctx->ws_flags.p : nullptr, (int64_t)na, (int64_t *)ctx->ws_once.p);
    HIPCHK(ctx, hipGetLastError());
    int64_t *h_out = (int64_t *)ctx->h_pinned;          // the inputs have left the slab by the time this copy runs (same stream)
    HIPCHK(ctx, hipMemcpyAsync(h_out, ctx->ws_once.p, (4 * na + 2) * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    const double t_enqueued = trace ? now() : 0.0;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (trace)
        fprintf(stderr, "[snpm once] n %lld: fill %.3f ms (%d tasks), enqueue %.3f ms, wait %.3f ms\n", (long long)n, t_filled - t_begin,
                n_tasks, t_enqueued - t_filled, now() - t_enqueued);
    if (h_out[4 * na + 1] & 1) return set_err(ctx, SNPM_ERR_DOMAIN, "provided y is greater than n");       // core/snpmatch.py:43
    memcpy(score, h_out, na * sizeof(double));
    memcpy(ninfo, h_out + na, na * sizeof(int64_t));
    if (lik) {
        memcpy(lik, h_out + 2 * na, na * sizeof(double));
        memcpy(lrt, h_out + 3 * na, na * sizeof(double));
    }
    if (info) {
        const int64_t n_flag = h_out[4 * na];
        info[0] = n_flag;
        info[1] = q->all_integer ? 1 : 0;
        info[2] = n_flag > REEVAL_CAP ? 3 : (n_flag > 0 ? q->reeval_path : 0);
    }
    return SNPM_OK;
} SNPM_GUARD((p ? p->ctx : nullptr))
