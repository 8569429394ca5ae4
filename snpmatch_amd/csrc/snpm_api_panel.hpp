// snpm_api_panel.hpp -- C ABI: panels (inside the extern "C" block of snpm_api.hip).
// Part of the one translation unit of libsnpmatch_hip.so: included by snpm_api.hip at this place, not on its own.
// ---------------------------------------------------------------------------------------------- panel
// bytes per row of a panel of n_acc accessions (see panel_create_fmt for the measurements behind the rule)
// SPLIT layout of a packed panel (snpm_k_common.hpp): the whole 256-B column blocks of a row form the main matrix, its ragged
// tail (<= 128 B, padded to a power of two) a matrix of its own -- chosen when that saves >= 5 % of the row against padding the
// tail to 256 B.  1135 accessions: 256 + 32 B per row instead of 512; 512 / 256 / 128 accessions: a 128 / 64 / 32-B pitch.
// main / tail = 0 / 0: whole rows (panel_row_pitch below).
static void packed_split_of(const snpm_ctx *ctx, int64_t n_acc, int64_t *main_pitch, int64_t *tail_pitch)
{
    *main_pitch = *tail_pitch = 0;
    if (!ctx->packed_split || ctx->pitch_align_forced) return;
    if (const char *e = getenv("SNPM_PACKED_SPLIT"))               // also read per panel: one process may hold both layouts (tests)
        if (atoi(e) == 0) return;
    const int64_t row_bytes = (n_acc + 3) / 4;
    const int64_t main = row_bytes / 256 * 256, rem = row_bytes - main;
    if (rem == 0 || rem > 128) return;
    int64_t tp = 4;
    while (tp < rem) tp <<= 1;
    if ((256 - tp) * 20 < main + 256) return;
    *main_pitch = main;
    *tail_pitch = tp;
}

static int64_t panel_row_pitch(const snpm_ctx *ctx, int64_t n_acc, int packed)
{
    if (packed) {
        int64_t mp, tp;
        packed_split_of(ctx, n_acc, &mp, &tp);
        if (tp) return mp + tp;
    }
    int64_t align = ctx->pitch_align;
    if (!packed && !ctx->pitch_align_forced) {
        const int64_t p256 = (n_acc + 255) / 256 * 256, p128 = (n_acc + 127) / 128 * 128;
        if ((p256 - p128) * 20 >= p256) align = 128;
    }
    int64_t pitch = packed ? (((n_acc + 3) / 4 + align - 1) / align) * align : ((n_acc + align - 1) / align) * align;
    // A pitch that is a multiple of 8 KiB puts the same columns of consecutive rows on the same memory channels: 256 B more
    // per row (round 3, profiles/r03k_ab_pow2_pitch.txt: 8192 accessions int8 0.801 -> 0.827 of HBM peak, 16 384: 0.780 -> 0.797,
    // 32 768 accessions packed with hard calls 0.697 -> 0.741, with PL weights +1 %; at 4 KiB the gain is 1 %, at 2 KiB the
    // padding costs more than it brings)
    if (!ctx->pitch_align_forced && pitch % 8192 == 0) pitch += 256;
    return pitch;
}

static int panel_create_fmt(snpm_ctx *ctx, int64_t n_snp, int64_t n_acc, int packed, snpm_panel **out)
try {
    if (!ctx) return SNPM_ERR_BADARG;
    CHECK_ARG(ctx, out != nullptr, "out is NULL");
    CHECK_ARG(ctx, n_snp >= 0 && n_acc >= 1, "panel needs n_snp >= 0 and n_acc >= 1");
    // 2^27 accessions: a group of 8 rows stays below 2^31 bytes (the kernels address row groups through 32-bit buffer offsets)
    CHECK_ARG(ctx, n_acc <= (int64_t)1 << 27, "n_acc too large (at most 2^27 accessions per panel)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    snpm_panel *p = new snpm_panel();
    p->ctx = ctx;
    p->n_snp = n_snp;
    p->n_acc = n_acc;
    p->n_acc_total = n_acc;
    p->packed = packed ? 1 : 0;
    p->ld = ((n_acc + 255) / 256) * 256;
    // Row pitch: padded to 256 B, the width of a wave's read (10 000 accessions -> 10 240 B: 2.4 % of every pass is padding).
    // Round 3 tried whole 64-B sectors instead (SNPM_PITCH_ALIGN=64: lanes past the pitch are masked, 10 048 B per row move; all
    // tests pass): 1135 x 11M +3 %, 2500 x 50M +1 %, but 10 000 x 20M 0.816 -> 0.784 and 5000 x 40M 0.787 -> 0.740 of HBM peak
    // (profiles/r03e_ab_pitch.txt) -- a wave's 256-B read that straddles two 256-B units costs more than the padding saves.
    // int8 panels whose 256-B padding would be 5 % of the row or more take whole 128-B cache lines instead (round 3,
    // profiles/r03h_ab_pitch128.txt: the 1135 accessions of the 1001 Genomes panel 1280 -> 1152 B per row, 0.720 -> 0.740 of HBM
    // peak and a tenth less HBM; 10 000 accessions would LOSE 0.3 % and keep their 10 240 B; packed panels measured no gain)
    p->pitch = panel_row_pitch(ctx, n_acc, packed);
    p->kpitch = p->pitch;
    // PREFETCH_PAD_ROWS extra rows: the fast pass prefetches (and never scores) a few rows past a part
    size_t row_bytes = (size_t)(n_snp + PREFETCH_PAD_ROWS) * (size_t)p->pitch;
    if (packed) {
        int64_t mp, tp;
        packed_split_of(ctx, n_acc, &mp, &tp);
        if (tp) {               // main matrix, then (256-B aligned) the tail matrix; phased waves read up to 64 rows past a part
            const size_t rows_alloc = (size_t)(n_snp + PREFETCH_PAD_ROWS + 64);
            p->kpitch = mp;
            p->tail_pitch = tp;
            p->tail_off = (int64_t)(((rows_alloc * (size_t)mp) + 255) / 256 * 256);
            row_bytes = (size_t)p->tail_off + (rows_alloc * (size_t)tp + 255) / 256 * 256;
            int lg = 0;
            while (((int64_t)1 << lg) < tp) ++lg;
            p->desc = 1 | ((int64_t)(lg + 1) << 1) | ((p->tail_off / 256) << 8);
        } else {
            p->desc = 1;
        }
    }
    const size_t bytes = row_bytes + 256;                            // + the flag word d_other
    hipError_t e = hipMalloc((void **)&p->d, bytes);
    if (e != hipSuccess) {
        delete p;
        return set_err(ctx, SNPM_ERR_OOM, "hipMalloc of %zu panel bytes failed: %s", bytes, hipGetErrorString(e));
    }
    p->d_other = (int *)(p->d + row_bytes);
    if (hipMemsetAsync(p->d_other, 0, sizeof(int), ctx->copy_stream) != hipSuccess) {
        (void)hipFree(p->d);
        delete p;
        return set_err(ctx, SNPM_ERR_HIP, "hipMemsetAsync failed");
    }
    if (hipEventCreateWithFlags(&p->uploaded, hipEventDisableTiming) != hipSuccess) {
        (void)hipFree(p->d);
        delete p;
        return set_err(ctx, SNPM_ERR_HIP, "hipEventCreate failed");
    }
    (void)hipEventRecord(p->uploaded, ctx->copy_stream);             // the flag word is cleared before anything reads it
    p->upload_pending = true;
    ctx->panels.push_back(p);
    *out = p;
    return SNPM_OK;
} SNPM_GUARD(ctx)

int snpm_panel_create(snpm_ctx *ctx, int64_t n_snp, int64_t n_acc, snpm_panel **out)
{
    return panel_create_fmt(ctx, n_snp, n_acc, 0, out);
}

int snpm_panel_create_packed(snpm_ctx *ctx, int64_t n_snp, int64_t n_acc, snpm_panel **out)
{
    return panel_create_fmt(ctx, n_snp, n_acc, 1, out);
}

int snpm_panel_row_pitch(snpm_ctx *ctx, int64_t n_acc, int packed, int64_t *pitch)
{
    if (!ctx || !pitch || n_acc < 1) return SNPM_ERR_BADARG;
    *pitch = panel_row_pitch(ctx, n_acc, packed ? 1 : 0);
    return SNPM_OK;
}

int snpm_panel_is_packed(const snpm_panel *p, int *packed)
{
    if (!p || !packed) return SNPM_ERR_BADARG;
    *packed = p->packed;
    return SNPM_OK;
}

int snpm_panel_set_total_accessions(snpm_panel *p, int64_t n_acc_total)
{
    if (!p || !p->ctx) return SNPM_ERR_BADARG;
    CHECK_ARG(p->ctx, n_acc_total >= p->n_acc, "the whole panel cannot be narrower than this shard of it");
    p->n_acc_total = n_acc_total;
    return SNPM_OK;
}

int snpm_panel_free(snpm_panel *p)
{
    if (!p) return SNPM_OK;
    snpm_ctx *ctx = p->ctx;
    if (ctx) {                              // NULL: the context was destroyed first, the device memory went with it
        const bool use_hip = hip_alive();
        if (use_hip) {
            (void)hipSetDevice(ctx->device);
            (void)hipStreamSynchronize(ctx->copy_stream);
            (void)hipStreamSynchronize(ctx->stream);
        }
        ctx->panels.erase(std::remove(ctx->panels.begin(), ctx->panels.end(), p), ctx->panels.end());
        orphan_panel(p, use_hip);
    }
    delete p;
    return SNPM_OK;
}

int snpm_panel_info(const snpm_panel *p, int64_t *n_snp, int64_t *n_acc, int64_t *pitch, void **device_ptr)
{
    CHECK_PANEL(p);
    if (n_snp) *n_snp = p->n_snp;
    if (n_acc) *n_acc = p->n_acc;
    if (pitch) *pitch = p->pitch;
    if (device_ptr) *device_ptr = p->d;
    return SNPM_OK;
}

int snpm_panel_upload_wait(snpm_panel *p)
{
    CHECK_PANEL(p);
    HIPCHK(p->ctx, hipStreamSynchronize(p->ctx->copy_stream));
    p->upload_pending = false;
    p->ctx->stage_busy[0] = p->ctx->stage_busy[1] = false;
    for (int i = 0; i < snpm_ctx::kLdStages; ++i) p->ctx->ld_busy[i] = false;
    return SNPM_OK;
}

int snpm_panel_download_rows(snpm_panel *p, int64_t row0, int64_t nrows, int8_t *host, int64_t host_pitch)
{
    CHECK_PANEL(p);
    snpm_ctx *ctx = p->ctx;
    CHECK_ARG(ctx, row0 >= 0 && nrows >= 0 && row0 + nrows <= p->n_snp, "download rows outside the panel");
    CHECK_ARG(ctx, host_pitch >= p->n_acc, "host_pitch smaller than n_acc");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (nrows == 0) return SNPM_OK;
    if (!p->packed) {
        HIPCHK(ctx, hipMemcpy2D(host, (size_t)host_pitch, p->d + row0 * p->pitch, (size_t)p->pitch, (size_t)p->n_acc,
                                (size_t)nrows, hipMemcpyDeviceToHost));
        return SNPM_OK;
    }
    // packed: unpack slab by slab into a device scratch buffer, then copy out
    const int64_t slab = std::max<int64_t>(1, (int64_t)((64u << 20) / (size_t)p->n_acc));
    int rc = ensure(ctx, ctx->ws_stage_dev, std::max<size_t>(2 * snpm_ctx::kStageBytes, (size_t)slab * p->n_acc));
    if (rc) return rc;
    for (int64_t r = 0; r < nrows; r += slab) {
        const int64_t nr = std::min(slab, nrows - r);
        const int64_t total = nr * p->n_acc;
        hipLaunchKernelGGL(k_unpack_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const uint8_t *)p->d, p->kpitch, p->desc, row0 + r, nr, p->n_acc,
                           (int8_t *)ctx->ws_stage_dev.p, p->n_acc);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipMemcpy2D(host + r * host_pitch, (size_t)host_pitch, ctx->ws_stage_dev.p, (size_t)p->n_acc,
                                (size_t)p->n_acc, (size_t)nr, hipMemcpyDeviceToHost));
    }
    return SNPM_OK;
}

int snpm_panel_fill_synthetic_rows(snpm_panel *p, uint64_t seed, int64_t snp0, int64_t acc0, int64_t row0, int64_t nrows)
{
    CHECK_PANEL(p);
    snpm_ctx *ctx = p->ctx;
    CHECK_ARG(ctx, (acc0 & 3) == 0 && acc0 >= 0 && snp0 >= 0, "acc0 must be a non-negative multiple of 4");
    CHECK_ARG(ctx, row0 >= 0 && nrows >= 0 && row0 + nrows <= p->n_snp, "fill rows outside the panel");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (nrows == 0) return SNPM_OK;
    int rc = wait_upload(p);
    if (rc) return rc;
    ProfScope ps(ctx, PK_SYNTH);
    const int thr = 256;
    const unsigned gy = (unsigned)std::min<int64_t>(nrows, 2048);
    if (p->packed) {
        hipLaunchKernelGGL(k_synth_packed, dim3((unsigned)((p->pitch + thr - 1) / thr), gy), dim3(thr), 0, ctx->stream,
                           (uint8_t *)p->d, p->kpitch, p->desc, row0, nrows, p->n_acc, seed, snp0, acc0);
    } else {
        hipLaunchKernelGGL(k_synth, dim3((unsigned)((p->pitch / 4 + thr - 1) / thr), gy), dim3(thr), 0, ctx->stream,
                           (uint32_t *)(p->d + row0 * p->pitch), p->pitch, nrows, p->n_acc, seed, snp0, acc0);
    }
    HIPCHK(ctx, hipGetLastError());
    p->dT_state = 0;
    return SNPM_OK;
}

int snpm_panel_fill_synthetic(snpm_panel *p, uint64_t seed, int64_t snp0, int64_t acc0)
{
    CHECK_PANEL(p);
    return snpm_panel_fill_synthetic_rows(p, seed, snp0, acc0, 0, p->n_snp);
}

int snpm_sample_synthetic(snpm_ctx *ctx, uint64_t seed, int64_t snp0, int64_t n, int64_t planted, int err_permille,
                          int pl_permille, const double *exp_table, void *d_wei)
{
    if (!ctx) return SNPM_ERR_BADARG;
    CHECK_ARG(ctx, n >= 0 && snp0 >= 0 && planted >= 0, "negative size");
    CHECK_ARG(ctx, err_permille >= 0 && err_permille <= 1000 && pl_permille >= 0 && pl_permille <= 1000, "permille out of range");
    if (n == 0) return SNPM_OK;
    CHECK_ARG(ctx, exp_table && d_wei, "NULL pointer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = ensure(ctx, ctx->ws_lik_y, 256 * sizeof(double));
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->ws_lik_y.p, exp_table, 256 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));      // exp_table is the caller's
    hipLaunchKernelGGL(k_synth_sample, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, seed, snp0, n, planted,
                       (uint32_t)err_permille, (uint32_t)pl_permille, (const double *)ctx->ws_lik_y.p, (double *)d_wei);
    HIPCHK(ctx, hipGetLastError());
    return SNPM_OK;
}

