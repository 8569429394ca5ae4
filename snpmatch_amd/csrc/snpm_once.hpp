// snpm_once.hpp -- ONE sample against a resident panel in ONE call: snpm_genotype_once / snpm_genotype_once_coded.
//
// Genotyper.genotyper (core/snpmatch.py:207-241) for a single sample used to be three synchronising ABI calls from Python --
// snpm_query_create (two pageable uploads, a device pass over the weights and a read-back of its flags), snpm_query_run (two
// copies back), snpm_likelihood (two copies up, two back) -- plus a numpy gather of the sample's matched weight rows in front:
// 0.55-0.7 ms per 200k-SNP sample around 0.06 ms of scoring (profiles/r04_real_panel_*).  Two forms live here:
//   genotype_once_fused (default)  the host thread pool gathers wei[sample_idx[i]] / the weight codes and the row list into ONE
//      pinned slab, collecting the weight properties the kernels are chosen by on the way; k_once_prep reads the slab where it
//      lies (or what the copy engine brought behind the fill) and prepares everything the scoring needs in one launch; fast pass,
//      reduce + certificate, the sparse tier, k_once_finish (likelihood / nanmin / ratio of the truncated counts, :96, :106-117,
//      + status) written straight into the slab: seven launches, one synchronisation.
//   genotype_once_impl's own body (SNPM_ONCE_FUSED=0, chunk > ONCE_MAX_CHUNK, n == 0)  the first version: the slab goes up in
//      pieces behind the fill, row check / code expansion / LUT / error bound / tiers / likelihood as separate launches of the
//      query path, one packed copy back.
// Both return the same bits (tests/test_gpu_once.py).
// Included by snpm_api.hip inside its extern "C" block.

namespace {

struct OnceProps {          // per task of the fill
    long double wsum = 0;
    int flags = 0;          // bit 0: a weight is not an integer (or huge); bit 1: a weight is neither 0 nor 1; bit 2: NaN / infinite
    int64_t bad_row = -1;   // position of a row index outside the panel
    char pad[64];
};

}  // namespace

// The weight table of the coded form on the device (uploaded when it is not the one of the previous call) and, with it, the
// properties of every code: a flag byte and |entry| (65536 of each; codes past the table carry bit 3).  BOTH forms of the call
// go through here: the per-code tables always describe ctx->once_table.
static int once_set_table(snpm_ctx *ctx, const double *table, int64_t table_len)
{
    int rc = ensure(ctx, ctx->ws_once_table, 65536 * sizeof(double));
    if (rc) return rc;
    if (ctx->once_table.size() == (size_t)table_len && ctx->once_code_flags.size() == 65536 &&
        memcmp(ctx->once_table.data(), table, (size_t)table_len * sizeof(double)) == 0)
        return SNPM_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));         // a previous call's expansion may still read the old table / staging copy
    ctx->once_table.assign(table, table + table_len);
    HIPCHK(ctx, hipMemcpyAsync(ctx->ws_once_table.p, ctx->once_table.data(), (size_t)table_len * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ctx->once_code_flags.assign(65536, (uint8_t)8);
    ctx->once_code_abs.assign(65536, 0.0);
    for (int64_t k = 0; k < table_len; ++k) {
        const double v = table[k], a = fabs(v);
        uint64_t b;
        memcpy(&b, &v, 8);
        const uint64_t mag = b & 0x7FFFFFFFFFFFFFFFull;
        ctx->once_code_flags[(size_t)k] = (uint8_t)((uint32_t)((a < 9.0e15 && (double)(int64_t)a != a) || !(a < 9.0e15)) |
                                                    ((uint32_t)(!(mag == 0 || b == 0x3FF0000000000000ull)) << 1) |
                                                    ((uint32_t)(mag >= 0x7FF0000000000000ull) << 2));
        ctx->once_code_abs[(size_t)k] = (mag < 0x7FF0000000000000ull) ? a : 0.0;
    }
    return SNPM_OK;
}

// ---- the short form of the call (default; SNPM_ONCE_FUSED=0 or a chunk above ONCE_MAX_CHUNK take genotype_once_impl's body) ----
// GPU timeline of a coded 200k-SNP sample before: 2 copies up (67 us with their gaps), 15 kernels and fills of ~4.8 us each around
// the 53 us of k_fast, 1 copy back = 224 us (profiles/r04_once_timeline_before.txt).  Here: the slab stays in pinned host memory
// and k_once_prep reads it over the bus while it expands it (SNPM_ONCE_ZEROCOPY=0: staged through the copy engine as before),
// k_fast, two reduce kernels, the sparse tier (its patch inside k_scan_few), k_once_finish writing into the pinned slab:
// 7 launches, no copy.  The > REEVAL_CAP tier runs only when the count that comes back says so (second round trip, rare).
// Coded samples: the weight properties come from two small per-code tables (a flag byte, |entry|) made once per weight table.
static int genotype_once_fused(snpm_panel *p, const int64_t *row_idx, const double *wei, const uint16_t *codes, const double *table,
                               int64_t table_len, const int64_t *sample_idx, int64_t n_wei, int64_t n, int64_t chunk, int skip_hets,
                               int mode, double *score, int64_t *ninfo, double *lik, double *lrt, int64_t *info)
{
    snpm_ctx *ctx = p->ctx;
    const bool coded = codes != nullptr;
    const size_t na = (size_t)p->n_acc;
    const int skip = skip_hets ? 1 : 0;
    static const bool trace = getenv("SNPM_ONCE_TRACE") != nullptr;
    const bool zero_copy = ctx->once_zero_copy != 0;        // (fp64 samples too: 6.4 MB per 200k SNPs read in place 0.33-0.35 ms, through the copy engine 0.37-0.38)
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = trace ? now() : 0.0;
    snpm_query *q = nullptr;
    int rc = query_alloc_all(p, n, true, &q);
    if (rc) return rc;
    struct Guard {
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
    const size_t row_bytes = ((size_t)n * (coded ? sizeof(int32_t) : sizeof(int64_t)) + 7) / 8 * 8;
    const size_t wei_bytes = (size_t)n * 3 * (coded ? sizeof(uint16_t) : sizeof(double));
    const size_t out_words = 4 * na + 2;
    if ((rc = ensure_pinned(ctx, std::max<size_t>(row_bytes + wei_bytes + 64, out_words * sizeof(int64_t))))) return rc;
    char *h_slab = (char *)ctx->h_pinned;
    int64_t *h_rows = (int64_t *)h_slab;
    int32_t *h_rows32 = (int32_t *)h_slab;
    double *h_wei = (double *)(h_slab + row_bytes);
    uint16_t *h_codes = (uint16_t *)(h_slab + row_bytes);
    if ((rc = ensure(ctx, ctx->ws_once_state, 64))) return rc;
    if (!ctx->once_state_clean) {                                   // first use, or a call that failed between prep and finish
        HIPCHK(ctx, hipMemsetAsync(ctx->ws_once_state.p, 0, 64, ctx->stream));
        ctx->once_state_clean = true;
    }
    if (coded && (rc = once_set_table(ctx, table, table_len))) return rc;

    // ---- fill: the pool gathers the matched rows into the slab
    HostPool *pool = host_pool(ctx);
    // rows per task: small enough that threads which wake late still find work (28 tasks of 7168 rows filled a 200k-SNP fp64
    // sample in 0.20 ms, 49 of 4096 in 0.13), at most ~256 tasks
    const int64_t piece = std::max<int64_t>(4096, ((n + 255) / 256 + 1023) / 1024 * 1024);
    const int n_tasks = (int)((n + piece - 1) / piece);
    std::vector<OnceProps> props((size_t)std::max(n_tasks, 1));
    const int64_t n_snp = p->n_snp;
    const uint8_t *code_flags = coded ? ctx->once_code_flags.data() : nullptr;
    const double *code_abs = coded ? ctx->once_code_abs.data() : nullptr;
    auto fill = [&](int t) {
        const int64_t i0 = (int64_t)t * piece, i1 = std::min<int64_t>(n, i0 + piece);
        OnceProps pr;
        double wsum = 0.0;
        uint32_t fl = 0;
        for (int64_t i = i0; i < i1; ++i) {
            const int64_t r = row_idx[i];
            if ((uint64_t)r >= (uint64_t)n_snp && pr.bad_row < 0) pr.bad_row = i;
            if (coded) h_rows32[i] = (uint64_t)r < (uint64_t)n_snp ? (int32_t)r : -1;
            else h_rows[i] = r;
            int64_t s = sample_idx ? sample_idx[i] : i;
            if ((uint64_t)s >= (uint64_t)n_wei) {
                if (pr.bad_row < 0) pr.bad_row = i;
                s = 0;
            }
            if (coded) {
                const uint16_t c0 = codes[3 * s], c1 = codes[3 * s + 1], c2 = codes[3 * s + 2];
                h_codes[3 * i] = c0; h_codes[3 * i + 1] = c1; h_codes[3 * i + 2] = c2;
                fl |= (uint32_t)(code_flags[c0] | code_flags[c1] | code_flags[c2]);
                const double a0 = code_abs[c0], a1 = code_abs[c1], a2 = code_abs[c2];
                const double m01 = a0 > a1 ? a0 : a1;
                wsum += m01 > a2 ? m01 : a2;
            } else {
                double m = 0.0;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double v = wei[3 * s + c];
                    h_wei[3 * i + c] = v;
                    uint64_t b;
                    memcpy(&b, &v, 8);
                    const uint64_t mag = b & 0x7FFFFFFFFFFFFFFFull;
                    const double a = fabs(v);
                    // branch-free (bit 0: fractional or huge): the loop runs once per matched SNP
                    fl |= ((uint32_t)(a < 9.0e15) & (uint32_t)((double)(int64_t)a != a)) | (uint32_t)(!(a < 9.0e15)) |
                          ((uint32_t)(!((mag == 0) | (b == 0x3FF0000000000000ull))) << 1) | ((uint32_t)(mag >= 0x7FF0000000000000ull) << 2);
                    m = a > m ? a : m;
                }
                wsum += m;
            }
        }
        pr.wsum = wsum;
        pr.flags = (int)fl;
        props[(size_t)t] = pr;
    };
    if (n_wei == 0) return set_err(ctx, SNPM_ERR_BADARG, "please provide same number of positions for both sample and db");
    // through the copy engine (SNPM_ONCE_ZEROCOPY=0): the slab goes up in pieces BEHIND the fill -- task 0 of the pool run waits,
    // in order, for the fill tasks of each piece and enqueues its two copies -- into the query's own arrays (fp64: k_once_prep
    // then works in place) or a staging pair (coded)
    int32_t *d_rows32 = nullptr;
    uint16_t *d_codes = nullptr;
    if (!zero_copy && coded)
        if (query_alloc(q, (void **)&d_rows32, row_bytes) != hipSuccess || query_alloc(q, (void **)&d_codes, wei_bytes + 8) != hipSuccess)
            return set_err(ctx, SNPM_ERR_OOM, "query allocation failed");
    const int tasks_per_piece = std::max(1, (n_tasks + 1) / 2);             // two pieces: every copy costs ~25 us of its own
    const int n_pieces = (n_tasks + tasks_per_piece - 1) / tasks_per_piece;
    std::vector<std::atomic<int>> piece_done((size_t)std::max(n_pieces, 1));
    for (auto &c : piece_done) c.store(0, std::memory_order_relaxed);
    std::atomic<int> upload_error{0};
    auto upload_piece = [&](int k) -> bool {
        const int64_t i0 = (int64_t)k * tasks_per_piece * piece, i1 = std::min<int64_t>(n, i0 + (int64_t)tasks_per_piece * piece);
        if (coded)
            return hipMemcpyAsync(d_rows32 + i0, h_rows32 + i0, (size_t)(i1 - i0) * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
                   hipMemcpyAsync(d_codes + 3 * i0, h_codes + 3 * i0, (size_t)(i1 - i0) * 3 * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
        return hipMemcpyAsync(q->d_row_idx + i0, h_rows + i0, (size_t)(i1 - i0) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
               hipMemcpyAsync(q->d_w + 3 * i0, h_wei + 3 * i0, (size_t)(i1 - i0) * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
    };
    const bool overlapped = !zero_copy && pool->size() > 0 && n_pieces > 1;
    if (overlapped) {
        pool->run(n_tasks + 1, [&](int t) {
            if (t == 0) {                                          // the uploader
                if (hipSetDevice(ctx->device) != hipSuccess) { upload_error.store(1); return; }
                for (int k = 0; k < n_pieces; ++k) {
                    const int need = std::min(tasks_per_piece, n_tasks - k * tasks_per_piece);
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
            piece_done[(size_t)((t - 1) / tasks_per_piece)].fetch_add(1, std::memory_order_release);
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
    if (coded) {
        if ((flags & 8) && bad_at < 0) {                                     // a code outside the table: find it for the message
            for (int64_t i = 0; i < n && bad_at < 0; ++i) {
                const int64_t s = sample_idx ? sample_idx[i] : i;
                for (int c = 0; c < 3; ++c)
                    if ((int64_t)codes[3 * s + c] >= table_len) bad_at = i;
            }
        }
    }
    if (bad_at >= 0 || (flags & 4) || upload_error.load()) {
        if (!zero_copy) (void)hipStreamSynchronize(ctx->stream);   // pieces already on their way read the slab
        if (bad_at >= 0)
            return set_err(ctx, SNPM_ERR_BADARG, "row index %lld at %lld outside the panel (n_snp %lld), or a sample index / weight code outside the weights",
                           (long long)row_idx[bad_at], (long long)bad_at, (long long)n_snp);
        if (flags & 4) return set_err(ctx, SNPM_ERR_BADARG, "SNP weights must be finite (a NaN or infinite weight was given)");
        return set_err(ctx, SNPM_ERR_HIP, "upload of the sample failed");
    }

    // ---- prep: one launch (reads the pinned slab in place, or what the copy engine has brought)
    const void *src_rows = h_slab, *src_wei = h_slab + row_bytes;
    if (zero_copy) {
        void *dp = nullptr;
        HIPCHK(ctx, hipHostGetDevicePointer(&dp, h_slab, 0));
        src_rows = dp;
        src_wei = (const char *)dp + row_bytes;
    } else {
        if (!overlapped)
            for (int k = 0; k < n_pieces; ++k)
                if (!upload_piece(k)) return set_err(ctx, SNPM_ERR_HIP, "upload of the sample failed");
        src_rows = coded ? (const void *)d_rows32 : (const void *)q->d_row_idx;
        src_wei = coded ? (const void *)d_codes : (const void *)q->d_w;
    }
    const int64_t K = (n + chunk - 1) / chunk;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(K, 2048));
    if ((rc = ensure(ctx, ctx->ws_epart, (size_t)grid * sizeof(double)))) return rc;
    ctx->once_state_clean = false;                                  // until k_once_finish has cleared the words again
    if (coded)
        hipLaunchKernelGGL((k_once_prep<true>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, src_rows, src_wei, (const double *)ctx->ws_once_table.p,
                           (int)table_len, n, n_snp, chunk, skip, q->d_row_idx, q->d_w, q->d_lut, (double *)ctx->ws_epart.p, q->cert_eref(),
                           q->cert_count(), (unsigned *)ctx->ws_once_state.p, (int)PREFETCH_PAD_ROWS);
    else
        hipLaunchKernelGGL((k_once_prep<false>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, src_rows, src_wei, (const double *)nullptr, 0, n,
                           n_snp, chunk, skip, q->d_row_idx, q->d_w, q->d_lut, (double *)ctx->ws_epart.p, q->cert_eref(), q->cert_count(),
                           (unsigned *)ctx->ws_once_state.p, (int)PREFETCH_PAD_ROWS);
    HIPCHK(ctx, hipGetLastError());
    q->lut_skip = skip;
    q->eref_chunk = chunk;
    q->eref_after = 0;
    q->cert_count_clean = true;
    q->wsum = (double)tot * 1.0000001;
    q->all_integer = !(flags & 1) && tot < 9.0e15L;
    q->hard01 = q->all_integer && !(flags & 2) && p->packed;
    if (q->hard01) {
        const int64_t padded = n + 16;
        hipError_t e2 = query_alloc(q, (void **)&q->d_wbits, (size_t)padded);
        if (e2 != hipSuccess) return set_err(ctx, SNPM_ERR_OOM, "query allocation failed: %s", hipGetErrorString(e2));
        hipLaunchKernelGGL(k_wbits, dim3((unsigned)((padded + 255) / 256)), dim3(256), 0, ctx->stream, (const double *)q->d_w, n, padded, q->d_wbits);
        HIPCHK(ctx, hipGetLastError());
    }

    // ---- score: fast pass + reduce (+ certificate), the sparse tier behind it; strict mode: the chain
    bool certified = false;
    if (mode == SNPM_MODE_STRICT) {
        q->count_valid = false;
        q->last_kernel = "k_strict4";
        if ((rc = run_strict_chain(q, skip, chunk, nullptr, nullptr, nullptr, q->d_score, q->d_ninfo))) return rc;
    } else {
        Certify cert;
        cert.on = (mode == SNPM_MODE_EXACT);
        cert.chunk = chunk;
        if ((rc = run_fast(q, skip, nullptr, cert))) return rc;
        certified = cert.on && !q->all_integer;
    }
    if ((rc = ensure(ctx, ctx->ws_lik_l, na * sizeof(double)))) return rc;
    int64_t *h_out = (int64_t *)h_slab;
    int64_t *d_out = h_out;
    if (zero_copy) {
        void *dp = nullptr;
        HIPCHK(ctx, hipHostGetDevicePointer(&dp, h_slab, 0));
        d_out = (int64_t *)dp;
    } else {
        if ((rc = ensure(ctx, ctx->ws_once, out_words * sizeof(int64_t)))) return rc;
        d_out = (int64_t *)ctx->ws_once.p;
    }
    // the sparse tier behind the fast pass; its chain kernel also does the likelihood / ratio / status step (k_once_tail)
    bool tail_done = false;
    if (certified) {
        OnceTail tail;
        tail.d_ninfo = q->d_ninfo; tail.n_acc = (int64_t)na; tail.want_lik = lik ? 1 : 0;
        tail.state = (unsigned *)ctx->ws_once_state.p; tail.lik_tmp = (double *)ctx->ws_lik_l.p; tail.out = d_out;
        const bool fuse = ctx->once_tail && !single_accession(p);
        if ((rc = enqueue_reevaluation(q, skip, chunk, false, fuse ? &tail : nullptr))) return rc;
        if (fuse) {
            if (!zero_copy) HIPCHK(ctx, hipMemcpyAsync(h_out, d_out, out_words * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
            tail_done = true;
        }
    }
    auto finish = [&]() -> int {
        hipLaunchKernelGGL(k_once_finish, dim3(1), dim3(1024), 0, ctx->stream, (const double *)q->d_score, (const int64_t *)q->d_ninfo, (int64_t)na,
                           lik ? 1 : 0, certified ? (const int *)q->cert_count() : (const int *)nullptr, (unsigned *)ctx->ws_once_state.p,
                           (double *)ctx->ws_lik_l.p, d_out);
        HIPCHK(ctx, hipGetLastError());
        if (!zero_copy) HIPCHK(ctx, hipMemcpyAsync(h_out, d_out, out_words * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        return SNPM_OK;
    };
    if (!tail_done && (rc = finish())) return rc;
    const double t_enqueued = trace ? now() : 0.0;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->once_state_clean = true;
    const int64_t status = h_out[4 * na + 1];
    int64_t n_flag = h_out[4 * na];
    if (status & 6)
        return set_err(ctx, SNPM_ERR_BADARG, "a row index outside the panel (n_snp %lld), or a sample index / weight code outside the weights", (long long)n_snp);
    if (certified && n_flag > REEVAL_CAP) {             // the dense tier, now that the count is known: everything in reference order
        if ((rc = run_strict_chain(q, skip, chunk, q->cert_count(), nullptr, nullptr, q->d_score, q->d_ninfo))) return rc;
        ctx->once_state_clean = false;
        if ((rc = finish())) return rc;
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        ctx->once_state_clean = true;
        n_flag = h_out[4 * na];
    }
    if (trace)
        fprintf(stderr, "[snpm once] n %lld: fill %.3f ms (%d tasks), enqueue %.3f ms, wait %.3f ms (fused%s)\n", (long long)n, t_filled - t_begin,
                n_tasks, t_enqueued - t_filled, now() - t_enqueued, zero_copy ? ", zero-copy" : "");
    if (h_out[4 * na + 1] & 1) return set_err(ctx, SNPM_ERR_DOMAIN, "provided y is greater than n");       // core/snpmatch.py:43
    memcpy(score, h_out, na * sizeof(double));
    memcpy(ninfo, h_out + na, na * sizeof(int64_t));
    if (lik) {
        memcpy(lik, h_out + 2 * na, na * sizeof(double));
        memcpy(lrt, h_out + 3 * na, na * sizeof(double));
    }
    if (info) {
        info[0] = n_flag;
        info[1] = q->all_integer ? 1 : 0;
        info[2] = n_flag > REEVAL_CAP ? 3 : (n_flag > 0 ? q->reeval_path : 0);
    }
    return SNPM_OK;
}

// codes != NULL: the sample's weights as dictionary codes, wei[r, c] = table[codes[3 r + c]] (snpm_genotype_once_coded)
static int genotype_once_impl(snpm_panel *p, const int64_t *row_idx, const double *wei, const uint16_t *codes, const double *table,
                              int64_t table_len, const int64_t *sample_idx, int64_t n_wei, int64_t n, int64_t chunk, int skip_hets,
                              int mode, double *score, int64_t *ninfo, double *lik, double *lrt, int64_t *info)
{
    CHECK_PANEL(p);
    snpm_ctx *ctx = p->ctx;
    const bool coded = codes != nullptr;
    if (coded) {
        CHECK_ARG(ctx, table != nullptr && table_len >= 1 && table_len <= 65536, "weight table of 1 .. 65536 entries expected");
        CHECK_ARG(ctx, p->n_snp < ((int64_t)1 << 31), "coded samples travel with 32-bit row indices");
    }
    CHECK_ARG(ctx, n >= 0 && n_wei >= 0, "n must be >= 0");
    CHECK_ARG(ctx, chunk >= 1, "chunk must be >= 1");
    CHECK_ARG(ctx, mode == SNPM_MODE_EXACT || mode == SNPM_MODE_STRICT || mode == SNPM_MODE_FAST, "unknown mode");
    CHECK_ARG(ctx, n == 0 || (row_idx && (wei || coded)), "please provide same number of positions for both sample and db");
    CHECK_ARG(ctx, sample_idx || n <= n_wei, "please provide same number of positions for both sample and db");
    CHECK_ARG(ctx, score && ninfo && ((lik == nullptr) == (lrt == nullptr)), "score / ninfo outputs missing, or only one likelihood output");
    if (info) info[0] = info[1] = info[2] = info[3] = 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = wait_upload(p);
    if (rc) return rc;
    if (ctx->once_fused && n > 0 && chunk <= ONCE_MAX_CHUNK)
        return genotype_once_fused(p, row_idx, wei, codes, table, table_len, sample_idx, n_wei, n, chunk, skip_hets, mode, score, ninfo, lik, lrt, info);
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
    // plain: [rows int64 | weights fp64 x 3] = 32 B per matched SNP; coded: [rows int32 | codes uint16 x 3] = 10 B
    const size_t row_bytes = ((size_t)n * (coded ? sizeof(int32_t) : sizeof(int64_t)) + 7) / 8 * 8;
    const size_t wei_bytes = (size_t)n * 3 * (coded ? sizeof(uint16_t) : sizeof(double));
    const size_t out_words = 4 * na + 2;
    rc = ensure_pinned(ctx, std::max<size_t>(row_bytes + wei_bytes + 64, out_words * sizeof(int64_t)));
    if (rc) return rc;
    int64_t *h_rows = (int64_t *)ctx->h_pinned;
    int32_t *h_rows32 = (int32_t *)ctx->h_pinned;
    double *h_wei = (double *)((char *)ctx->h_pinned + row_bytes);
    uint16_t *h_codes = (uint16_t *)((char *)ctx->h_pinned + row_bytes);
    int32_t *d_rows32 = nullptr;
    uint16_t *d_codes = nullptr;
    if (coded && n > 0) {
        if ((rc = once_set_table(ctx, table, table_len))) return rc;
        if (query_alloc(q, (void **)&d_rows32, row_bytes) != hipSuccess || query_alloc(q, (void **)&d_codes, wei_bytes) != hipSuccess)
            return set_err(ctx, SNPM_ERR_OOM, "query allocation failed");
    }
    HostPool *pool = host_pool(ctx);
    const int64_t piece = coded ? 8192 : 4096;                     // rows per task
    const int n_tasks = (int)((n + piece - 1) / piece);
    std::vector<OnceProps> props((size_t)std::max(n_tasks, 1));
    const int64_t n_snp = p->n_snp;
    // weight properties with integer tests on the bit patterns (no libm call per weight: this loop runs once per matched SNP)
    auto bits_of = [](double v) -> uint64_t { uint64_t b; memcpy(&b, &v, 8); return b; };
    auto flags_of = [&](double v) -> uint32_t {          // bit 0 fractional, 1 not 0 / 1, 2 non-finite, 3 huge
        const uint64_t b = bits_of(v), mag = b & 0x7FFFFFFFFFFFFFFFull;
        const double a = fabs(v);
        return (uint32_t)(a < 9.0e15 && (double)(int64_t)a != a) | ((uint32_t)(!(mag == 0 || b == 0x3FF0000000000000ull)) << 1) |
               ((uint32_t)(mag >= 0x7FF0000000000000ull) << 2) | ((uint32_t)(!(a < 9.0e15)) << 3);
    };
    // coded samples: the properties of every table entry once, then one small lookup per weight
    std::vector<uint32_t> tab_flags;
    std::vector<double> tab_abs;
    if (coded) {
        tab_flags.resize((size_t)table_len);
        tab_abs.resize((size_t)table_len);
        for (int64_t k = 0; k < table_len; ++k) {
            tab_flags[(size_t)k] = flags_of(table[k]);
            tab_abs[(size_t)k] = fabs(table[k]);
        }
    }
    auto fill = [&](int t) {
        const int64_t i0 = (int64_t)t * piece, i1 = std::min<int64_t>(n, i0 + piece);
        OnceProps pr;
        double wsum = 0.0;                                         // <= 4096 non-negative terms: good to 1e-12, rounded up below
        uint64_t any_frac = 0, any_not01 = 0, any_nonfinite = 0, any_huge = 0;
        for (int64_t i = i0; i < i1; ++i) {
            const int64_t r = row_idx[i];
            if ((uint64_t)r >= (uint64_t)n_snp && pr.bad_row < 0) pr.bad_row = i;
            if (coded) h_rows32[i] = (int32_t)r;
            else h_rows[i] = r;
            int64_t s = sample_idx ? sample_idx[i] : i;
            if ((uint64_t)s >= (uint64_t)n_wei) {                  // a sample index outside the weight array: reported like a bad row
                if (pr.bad_row < 0) pr.bad_row = i;
                s = 0;
            }
            double m = 0.0;
            if (coded) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const uint16_t k = codes[3 * s + c];
                    h_codes[3 * i + c] = k;
                    if ((int64_t)k >= table_len) {                 // a code outside the table: reported like a bad index
                        if (pr.bad_row < 0) pr.bad_row = i;
                        continue;
                    }
                    const uint32_t f = tab_flags[k];
                    any_frac |= f & 1u;
                    any_not01 |= (f >> 1) & 1u;
                    any_nonfinite |= (f >> 2) & 1u;
                    any_huge |= (f >> 3) & 1u;
                    m = tab_abs[k] > m ? tab_abs[k] : m;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double v = wei[3 * s + c];
                    h_wei[3 * i + c] = v;
                    const uint32_t f = flags_of(v);
                    any_frac |= f & 1u;
                    any_not01 |= (f >> 1) & 1u;
                    any_nonfinite |= (f >> 2) & 1u;
                    any_huge |= (f >> 3) & 1u;
                    const double a = fabs(v);
                    m = a > m ? a : m;
                }
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
        if (coded)
            return hipMemcpyAsync(d_rows32 + i0, h_rows32 + i0, (size_t)(i1 - i0) * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
                   hipMemcpyAsync(d_codes + 3 * i0, h_codes + 3 * i0, (size_t)(i1 - i0) * 3 * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
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
            return set_err(ctx, SNPM_ERR_BADARG, "row index %lld at %lld outside the panel (n_snp %lld), or a sample index / weight code outside the weights",
                           (long long)row_idx[bad_at], (long long)bad_at, (long long)n_snp);
        if (flags & 4) return set_err(ctx, SNPM_ERR_BADARG, "SNP weights must be finite (a NaN or infinite weight was given)");
        return set_err(ctx, SNPM_ERR_HIP, "upload of the sample failed");
    }
    if (!overlapped)
        for (int k = 0; k < n_pieces; ++k)
            if (!upload_piece(k)) return set_err(ctx, SNPM_ERR_HIP, "upload of the sample failed");
    if (coded && n > 0) {          // widen the row list, expand the codes: the query then looks like any other
        if ((rc = ensure(ctx, ctx->ws_flags2, 64))) return rc;
        hipLaunchKernelGGL(k_check_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, q->d_row_idx, (const int32_t *)d_rows32, n,
                           p->n_snp, (int *)ctx->ws_flags2.p);
        hipLaunchKernelGGL(k_expand_codes, dim3((unsigned)((3 * n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint16_t *)d_codes,
                           (const double *)ctx->ws_once_table.p, 3 * n, q->d_w);
        HIPCHK(ctx, hipGetLastError());
    }
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
}

int snpm_genotype_once(snpm_panel *p, const int64_t *row_idx, const double *wei, const int64_t *sample_idx, int64_t n_wei,
                       int64_t n, int64_t chunk, int skip_hets, int mode, double *score, int64_t *ninfo, double *lik,
                       double *lrt, int64_t *info)
try {
    return genotype_once_impl(p, row_idx, wei, nullptr, nullptr, 0, sample_idx, n_wei, n, chunk, skip_hets, mode, score, ninfo, lik, lrt, info);
} SNPM_GUARD((p ? p->ctx : nullptr))

// The same call with DICTIONARY-CODED weights: wei[r, c] = table[codes[3 r + c]] (codes uint16 [n_wei, 3], table fp64 [table_len]).
// A VCF sample's weights are exp(-PL / 10) of small integer PLs (core/parsers.py:141-151): the caller computes the table with its
// own libm, so the device weights carry the fp64 path's bits, and 10 instead of 32 bytes per matched SNP cross PCIe.
int snpm_genotype_once_coded(snpm_panel *p, const int64_t *row_idx, const uint16_t *codes, const double *table, int64_t table_len,
                             const int64_t *sample_idx, int64_t n_wei, int64_t n, int64_t chunk, int skip_hets, int mode,
                             double *score, int64_t *ninfo, double *lik, double *lrt, int64_t *info)
try {
    if (p && p->ctx && !codes) return set_err(p->ctx, SNPM_ERR_BADARG, "weight codes missing");
    return genotype_once_impl(p, row_idx, nullptr, codes, table, table_len, sample_idx, n_wei, n, chunk, skip_hets, mode, score, ninfo, lik,
                              lrt, info);
} SNPM_GUARD((p ? p->ctx : nullptr))
