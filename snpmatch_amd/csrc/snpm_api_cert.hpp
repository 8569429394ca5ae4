// snpm_api_cert.hpp -- the certificate on the device: error bounds, the fast pass with its ordered reduce, the accession-major copy (anonymous namespace of snpm_api.hip).
// Part of the one translation unit of libsnpmatch_hip.so: included by snpm_api.hip at this place, not on its own.
// ---- certificate: error bounds on the device (see DESIGN.md "Exactness") -------------------------------
// For sums of terms x_i with |x_i| <= wmax_i, a computed sum differs from the exact one by at most
// sum_i wmax_i * gamma(m_i), gamma(m) = m*u/(1-m*u), u = 2^-53, m_i = number of fp64 additions the
// term passes through.  Reference order: m_i <= (rows of its chunk) + 3 + (chunks left, later slabs included):
// k_eref / k_efinish evaluate that sum where the weights live and leave it in q->cert_eref()[0].
int ensure_pinned(snpm_ctx *ctx, size_t bytes)
{
    if (ctx->h_pinned_cap >= bytes) return SNPM_OK;
    if (ctx->h_pinned) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipHostFree(ctx->h_pinned);
    }
    ctx->h_pinned = nullptr;
    ctx->h_pinned_cap = 0;
    const size_t want = std::max<size_t>(bytes, 64 << 10);
    HIPCHK(ctx, hipHostMalloc(&ctx->h_pinned, want, hipHostMallocDefault));
    ctx->h_pinned_cap = want;
    return SNPM_OK;
}

int ensure_eref(snpm_query *q, int64_t chunk, int64_t chunks_after)
{
    snpm_ctx *ctx = q->panel->ctx;
    if (q->eref_chunk == chunk && q->eref_after == chunks_after) return SNPM_OK;
    const int64_t K = (q->n + chunk - 1) / chunk;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(K, 2048));
    int rc = ensure(ctx, ctx->ws_epart, (size_t)grid * sizeof(double));
    if (rc) return rc;
    hipLaunchKernelGGL(k_eref, dim3((unsigned)grid), dim3(256), 0, ctx->stream, (const double *)q->d_w, q->n, chunk,
                       chunks_after, (double *)ctx->ws_epart.p);
    hipLaunchKernelGGL(k_efinish, dim3(1), dim3(256), 0, ctx->stream, (const double *)ctx->ws_epart.p, grid, q->n, chunk,
                       chunks_after, q->cert_eref());
    HIPCHK(ctx, hipGetLastError());
    q->eref_chunk = chunk;
    q->eref_after = chunks_after;
    return SNPM_OK;
}

double efast_bound(const snpm_query *q, const FastGeom &g)
{
    const double u = 1.1102230246251565e-16;
    // a term passes through <= EPOCH_TILES*TILE_ROWS adds inside k_fast, REDUCE_GROUP in its group, n_groups after
    // (int8 kernel: an accumulator takes one addition per row of its epoch = EPOCH_TILES tiles of g.tile_rows rows; the packed
    // kernels add pre-summed quads of rows or run on integer weights only: the static_asserts beside Q4_RUN keep them below
    // EPOCH_TILES * TILE_ROWS additions per epoch)
    const int64_t epoch_adds = (int64_t)EPOCH_TILES * (g.bpl == 4 ? std::max(g.tile_rows, TILE_ROWS) : TILE_ROWS);
    const double m = (double)(std::min<int64_t>(g.part_rows, epoch_adds) + REDUCE_GROUP + g.n_groups + 2);
    return (q->wsum * (m * u / (1.0 - m * u))) * 1.0000001;
}

struct Certify {            // what the last reduce step of a fast pass should certify against (on == false: nothing)
    bool on = false;
    bool flag = true;       // false: only the bound is prepared (slab-streamed jobs certify their totals at the end)
    int64_t chunk = 1000, chunks_after = 0;
};

// rows per LUT tile of the fast pass for this query: packed panels have their own tile sizes; the int8 kernel walks longer tiles
// on long scans (LONG_TILE_ROWS, snpm_kernels.hpp), where the part count is bounded by the resident blocks, not by the tiles
int fast_tile_rows(const snpm_query *q, bool bits)
{
    const snpm_panel *p = q->panel;
    if (p->packed) return bits ? BITS_TILE_ROWS : Q4_TILE_ROWS * Q4_RUN;
    // (panels of one or two waves keep the 128-row tiles: their blocks are small, the 16 KB of a long tile would bound the
    // resident blocks -- 256 accessions x 100M rows 0.528 -> 0.597 of HBM peak, 512 accessions 0.685 -> 0.773,
    // profiles/r03j_ab_int8_narrow.txt)
    const bool long_tiles = q->n >= p->ctx->long_scan_rows && pick_bpl(p->ctx, p->n_acc) == 4 && p->n_acc > 2 * WAVE * 4;
    return long_tiles ? LONG_TILE_ROWS : TILE_ROWS;
}

// fast pass + ordered reduce -> q->d_score / q->d_ninfo (+ the list of accessions the certificate cannot vouch
// for, left on the device); returns the geometry used
int run_fast(snpm_query *q, int skip, FastGeom *geom_out, const Certify &cert)
{
    snpm_ctx *ctx = q->panel->ctx;
    snpm_panel *p = q->panel;
    int rc = ensure_lut(q, skip);
    if (rc) return rc;
    const bool gather = q->d_row_idx != nullptr;
    // packed panels: 16 accessions (one dword) per lane and row, four rows per table lookup (k_fast_packed_q4)
    const bool p16 = p->packed != 0;
    const int bpl = p16 ? 16 : pick_bpl(ctx, p->n_acc);
    const bool bits = p16 && q->hard01 && ctx->bits_path;      // counts instead of weighted sums
    const int tile_rows = fast_tile_rows(q, bits);
    // k_fast_bits has no LDS tile and no barrier: one wave per block fills every wave slot of a CU evenly (measured on the
    // packed 10k x 50M panel: 22.4 ms with 1- or 2-wave blocks, 26.9 ms with the 5-wave blocks of the LUT kernels, 30.7 with 3)
    // k_fast_packed_q4: 4-wave blocks (one wave per SIMD; 33.5 ms against 34.5 with 5-wave blocks on 10 000 accessions,
    // although 2 of its 12 waves there only help to build the tables; 2- and 3-wave blocks 40-41 ms) -- except for panels
    // of exactly five waves (4097-5120 accessions): one 5-wave block instead of two 4-wave blocks with three idle waves
    // (17.5 against 23.4 ms on 5000 x 50M)
    const int wpb_fixed = bits ? 1 : (p16 ? q4_waves_per_block(p->n_acc) : 0);
    FastGeom g0 = fast_geom(ctx, p->n_acc, q->n, 2, bpl, tile_rows, wpb_fixed);   // wpb does not depend on occupancy
    int occ = 0;
    const bool nt = ctx->nt_loads != 0;
    const int thr = WAVE * g0.wpb;
    if (bits) (void)launch_bits(q, g0, skip, gather, nt, &occ, thr);
    else if (p16) (void)launch_p16(q, g0, skip, gather, nt, &occ, thr);
    else if (bpl == 16) occ = nt ? occ_b<16, true>(skip, gather, thr) : occ_b<16, false>(skip, gather, thr);
    else if (bpl == 8) occ = nt ? occ_b<8, true>(skip, gather, thr) : occ_b<8, false>(skip, gather, thr);
    else occ = nt ? occ_b<4, true>(skip, gather, thr) : occ_b<4, false>(skip, gather, thr);
    // Parts per resident block (round 3, profiles/r03j_ab_q4_parts_mult.txt, r03j_ab_parts_mult_all.txt): with as many parts as
    // resident blocks every block of k_fast_packed_q4 walks its tiles in step with all the others -- the whole chip builds tables,
    // then the whole chip looks up; eight times as many, shorter parts take the blocks out of step: 10 000 accessions 11.77 ->
    // 10.50 ms per 20M SNPs (0.536 -> 0.601 of HBM peak on packed bytes), 8192: 0.644 -> 0.673, 4096: 0.628 -> 0.661, 2400: 0.448 ->
    // 0.506, 1135: 0.393 -> 0.430, 512: 0.279 -> 0.332; on the whole 10 000 x 50M job 2 / 4 / 8 / 16 / 24 parts per block take
    // 27.9 / 27.0 / 25.9 / 25.3 / 25.1 ms (r03j_ab_parts_mult_full.txt; 28.9 with one): sixteen.  k_fast_bits and the int8 k_fast keep
    // one part per resident block (more: +1.5 % on 20M rows but -3 % on 50M for the bits kernel, -6 ... -1 % on every int8 shape).
    const int kmult = (p16 && !bits) ? 16 : 1;
    // GATHERED rows on an int8 panel, blocks of 5, 6 or 7 waves (1025-1792 accessions: the 1001 Genomes width): 4 / 3 / 3 resident
    // blocks per CU instead of the 5 / 4 / 4 the occupancy query allows -- a 1M-row sample takes 0.205 instead of 0.267 ms on 1135
    // accessions (5.66 vs 4.35 TB/s), 0.277 instead of 0.327 ms on 1500, 0.299 instead of 0.345 ms on 1700; 200k rows and blocks of
    // 1-4 or 8 waves do not care (profiles/r04_gather_occupancy.txt)
    if (gather && !p16 && bpl == 4 && g0.wpb >= 5 && g0.wpb <= 7 && !ctx->full_occupancy) occ = std::min(occ, g0.wpb == 5 ? 4 : 3);
    FastGeom g = fast_geom(ctx, p->n_acc, q->n, occ, bpl, tile_rows, wpb_fixed, kmult);
    // Short int8 queries (a sample's 200k gathered rows: one or two 128-row tiles per resident block): tiles of such a length that
    // every part walks the same number of them -- 1563 tiles over 1024 parts make the launch as long as its two-tile parts (76 %
    // of the blocks' time is work); with k = ceil(tiles / parts) tiles of ceil(n / (k parts)) rows (a multiple of 8) it is 94 %.
    if (!p16 && bpl == 4 && tile_rows == TILE_ROWS && ctx->even_tiles && g.n_epochs == 1 && g.n_parts > 1) {
        const int64_t n_tiles = (q->n + TILE_ROWS - 1) / TILE_ROWS;
        const int64_t k = (n_tiles + g.n_parts - 1) / g.n_parts;
        if (k <= 8 && n_tiles > g.n_parts) {
            int64_t tr = ((q->n + k * g.n_parts - 1) / (k * g.n_parts) + 7) / 8 * 8;
            tr = std::max<int64_t>(32, std::min<int64_t>(TILE_ROWS, tr));
            if (tr < TILE_ROWS) g = fast_geom(ctx, p->n_acc, q->n, occ, bpl, (int)tr, wpb_fixed, kmult);
        }
    }
    if (geom_out) *geom_out = g;
    q->last_kernel = bits ? "k_fast_bits" : (p16 ? "k_fast_packed_q4" : "k_fast");
    rc = ensure(ctx, ctx->ws_part_score, (size_t)g.n_slots * p->ld * sizeof(double));
    if (rc) return rc;
    rc = ensure(ctx, ctx->ws_part_miss, (size_t)g.n_slots * p->ld * sizeof(uint32_t));
    if (rc) return rc;
    rc = ensure(ctx, ctx->ws_grp_score, (size_t)g.n_groups * p->ld * sizeof(double));
    if (rc) return rc;
    rc = ensure(ctx, ctx->ws_grp_miss, (size_t)g.n_groups * p->ld * sizeof(uint32_t));
    if (rc) return rc;
    const bool certify = cert.on && !q->all_integer && q->n > 0;
    if (certify) {
        rc = ensure_eref(q, cert.chunk, cert.chunks_after);
        if (rc) return rc;
    }
    if (!q->cert_count_clean) HIPCHK(ctx, hipMemsetAsync(q->cert_count(), 0, sizeof(int), ctx->stream));
    q->cert_count_clean = false;
    q->count_valid = cert.on;
    if (q->n > 0) {
        if (g.n_epochs > 1) {
            // parts with fewer tiles never reach the last epoch slot: those slots must read as zero
            const size_t off = (size_t)(g.n_epochs - 1) * g.n_parts * p->ld;
            HIPCHK(ctx, hipMemsetAsync((double *)ctx->ws_part_score.p + off, 0, (size_t)g.n_parts * p->ld * sizeof(double), ctx->stream));
            HIPCHK(ctx, hipMemsetAsync((uint32_t *)ctx->ws_part_miss.p + off, 0, (size_t)g.n_parts * p->ld * sizeof(uint32_t), ctx->stream));
        }
        if (bits) rc = launch_bits(q, g, skip, gather, nt, nullptr, thr);
        else if (p16) rc = launch_p16(q, g, skip, gather, nt, nullptr, thr);
        else if (bpl == 16) rc = nt ? launch_fast_b<16, true>(q, g, skip, gather) : launch_fast_b<16, false>(q, g, skip, gather);
        else if (bpl == 8) rc = nt ? launch_fast_b<8, true>(q, g, skip, gather) : launch_fast_b<8, false>(q, g, skip, gather);
        else rc = nt ? launch_fast_b<4, true>(q, g, skip, gather) : launch_fast_b<4, false>(q, g, skip, gather);
        if (rc) return rc;
    }
    {
        ProfScope ps(ctx, PK_REDUCE);
        const int thr = 64;       // one wave per block: narrow panels still spread over many CUs
        const unsigned cb = (unsigned)((p->n_acc + thr - 1) / thr);
        const int64_t n_groups = q->n > 0 ? g.n_groups : 0;
        const double *eref = (certify && cert.flag) ? (const double *)q->cert_eref() : (const double *)nullptr;
        const double efast = certify ? efast_bound(q, g) : 0.0;
        if (n_groups > 0 && n_groups <= 65535 && ctx->fused_reduce) {
            // both steps in one launch (tickets per column block, zero between launches)
            if (ctx->ws_tickets.cap < (size_t)cb * sizeof(unsigned)) {
                rc = ensure(ctx, ctx->ws_tickets, std::max<size_t>((size_t)cb * sizeof(unsigned), 4096));
                if (rc) return rc;
                HIPCHK(ctx, hipMemsetAsync(ctx->ws_tickets.p, 0, ctx->ws_tickets.cap, ctx->stream));
            }
            hipLaunchKernelGGL(k_reduce_all, dim3(cb, (unsigned)n_groups), dim3(thr), 0, ctx->stream,
                               (const double *)ctx->ws_part_score.p, (const uint32_t *)ctx->ws_part_miss.p, g.n_slots, p->ld, p->n_acc,
                               q->n, (double *)ctx->ws_grp_score.p, (uint32_t *)ctx->ws_grp_miss.p, q->d_score, q->d_ninfo, eref, efast,
                               ctx->debug_reeval, q->cert_cols(), q->cert_count(), REEVAL_CAP, (unsigned *)ctx->ws_tickets.p);
            HIPCHK(ctx, hipGetLastError());
        } else {
            if (n_groups > 0) {
                hipLaunchKernelGGL(k_reduce_groups, dim3(cb, (unsigned)n_groups), dim3(thr), 0, ctx->stream,
                                   (const double *)ctx->ws_part_score.p, (const uint32_t *)ctx->ws_part_miss.p, g.n_slots,
                                   p->ld, p->n_acc, (double *)ctx->ws_grp_score.p, (uint32_t *)ctx->ws_grp_miss.p);
                HIPCHK(ctx, hipGetLastError());
            }
            hipLaunchKernelGGL(k_reduce, dim3(cb), dim3(thr), 0, ctx->stream, (const double *)ctx->ws_grp_score.p,
                               (const uint32_t *)ctx->ws_grp_miss.p, n_groups, p->ld, p->n_acc, q->n, q->d_score,
                               q->d_ninfo, eref, efast, ctx->debug_reeval, q->cert_cols(), q->cert_count(), REEVAL_CAP);
            HIPCHK(ctx, hipGetLastError());
        }
    }
    return SNPM_OK;
}

// Build (or reuse) the accession-major packed copy; returns true when it can be used.
bool ensure_acc_major(snpm_panel *p)
{
    snpm_ctx *ctx = p->ctx;
    if (!ctx->use_acc_major) return false;
    if (p->dT_state == 1) return true;
    if (p->dT_state == -1 || p->n_snp == 0) return false;
    const int64_t pitchT = (((p->n_snp + 3) / 4 + 255) / 256) * 256 + 256;    // + one tile of slack for the last block
    if (!p->dT) {
        if (hipMalloc((void **)&p->dT, (size_t)p->n_acc * (size_t)pitchT) != hipSuccess) {
            (void)hipGetLastError();
            p->dT = nullptr;
            p->dT_state = -1;           // not enough memory: keep the strided path
            return false;
        }
        p->pitchT = pitchT;
    }
    if (ensure(ctx, ctx->ws_flags, sizeof(int)) != SNPM_OK) return false;
    if (hipMemsetAsync(ctx->ws_flags.p, 0, sizeof(int), ctx->stream) != hipSuccess) return false;
    if (p->packed) {
        dim3 grid((unsigned)((p->n_snp + PTP_ROWS - 1) / PTP_ROWS), (unsigned)((p->n_acc + PTP_COLS - 1) / PTP_COLS));
        hipLaunchKernelGGL(k_pack_transpose_packed, grid, dim3(256), 0, ctx->stream, (const uint8_t *)p->d, p->kpitch, p->n_snp,
                           p->n_acc, p->dT, p->pitchT, p->desc);
    } else {
        dim3 grid((unsigned)((p->n_snp + PT_ROWS - 1) / PT_ROWS), (unsigned)((p->n_acc + PT_COLS - 1) / PT_COLS));
        hipLaunchKernelGGL(k_pack_transpose, grid, dim3(256), 0, ctx->stream, p->d, p->pitch, p->n_snp, p->n_acc, p->dT,
                           p->pitchT, (int *)ctx->ws_flags.p);
    }
    int bad = 0;
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(&bad, ctx->ws_flags.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        p->dT_state = -1;
        return false;
    }
    p->dT_state = bad ? -1 : 1;
    return p->dT_state == 1;
}

