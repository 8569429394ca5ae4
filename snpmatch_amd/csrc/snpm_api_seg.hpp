// snpm_api_seg.hpp -- segmented scoring (batches of samples, windows) and the host staging helpers (anonymous namespace of snpm_api.hip).
// Part of the one translation unit of libsnpmatch_hip.so: included by snpm_api.hip at this place, not on its own.
// ---- segmented scoring: many row ranges of one matched list in one launch (batches of samples, windows) -----------
struct SegJob {
    snpm_panel *p = nullptr;
    const int64_t *d_row_idx = nullptr;     // device; NULL = dense rows row0 + r
    int64_t row0 = 0, n_total = 0;
    const double *d_w = nullptr, *d_lut = nullptr;      // device [n_total, 3] / [n_total, 4]
    const int64_t *seg_off = nullptr;       // HOST [n_seg + 1], non-decreasing, inside [0, n_total]
    int64_t n_seg = 0;
    int64_t chunk = 1000;                   // rows per matchGTsAccs call of the reference inside a segment
    int skip = 0;
    bool certify = true;
    double *d_score = nullptr;              // device outputs [n_seg, ldo]
    int64_t *d_ninfo = nullptr;
    int64_t ldo = 0;
    // filled by run_segmented
    int64_t kmax = 1;
    int cap = 0;
    const int64_t *d_seg_off = nullptr;
};

constexpr int SEG_PAIR_CAP = 32768;

int *seg_pair_count(snpm_ctx *ctx) { return (int *)ctx->ws_pairs.p; }
int32_t *seg_pairs(snpm_ctx *ctx) { return (int32_t *)((char *)ctx->ws_pairs.p + 16); }

template <bool NT>
static int launch_q4_seg(snpm_ctx *ctx, const SegJob &j, dim3 grid, dim3 block, int64_t n_parts, const int64_t *d_desc)
{
    snpm_panel *p = j.p;
    const bool gather = j.d_row_idx != nullptr;
    ProfScope ps(ctx, PK_FAST);
    const int tr = q4_tile_rows(ctx, (int)(block.x / WAVE));
#define LAUNCH_SEG_TR(S, G, TR)                                                                                   \
    hipLaunchKernelGGL((k_fast_packed_q4<S, G, NT, true, TR>), grid, block, 0, ctx->stream, p->d, p->kpitch, j.d_row_idx, j.row0, \
                       n_parts, j.d_lut, (double *)ctx->ws_part_score.p, (uint32_t *)ctx->ws_part_miss.p, p->ld, p->n_acc, p->desc, d_desc)
#define LAUNCH_SEG(S, G)                                                                                          \
    do {                                                                                                          \
        if (tr == 16) LAUNCH_SEG_TR(S, G, 16);                                                                    \
        else if (tr == 32) LAUNCH_SEG_TR(S, G, 32);                                                               \
        else LAUNCH_SEG_TR(S, G, 64);                                                                             \
    } while (0)
    if (j.skip) {
        if (gather) LAUNCH_SEG(true, true); else LAUNCH_SEG(true, false);
    } else {
        if (gather) LAUNCH_SEG(false, true); else LAUNCH_SEG(false, false);
    }
#undef LAUNCH_SEG_TR
#undef LAUNCH_SEG
    HIPCHK(ctx, hipGetLastError());
    return SNPM_OK;
}

template <int BPL, bool NT>
static int launch_fast_seg(snpm_ctx *ctx, const SegJob &j, dim3 grid, dim3 block, int64_t n_parts, const int64_t *d_desc)
{
    snpm_panel *p = j.p;
    const bool gather = j.d_row_idx != nullptr;
    ProfScope ps(ctx, PK_FAST);
#define LAUNCH_SEG(S, G)                                                                                          \
    hipLaunchKernelGGL((k_fast<BPL, S, G, NT, true>), grid, block, 0, ctx->stream, p->d, p->pitch, j.d_row_idx, j.row0,    \
                       n_parts, j.d_lut, (double *)ctx->ws_part_score.p, (uint32_t *)ctx->ws_part_miss.p, p->ld, d_desc)
    if (j.skip) {
        if (gather) LAUNCH_SEG(true, true); else LAUNCH_SEG(true, false);
    } else {
        if (gather) LAUNCH_SEG(false, true); else LAUNCH_SEG(false, false);
    }
#undef LAUNCH_SEG
    HIPCHK(ctx, hipGetLastError());
    return SNPM_OK;
}

// Plan of a segmented pass: parts (contiguous runs of <= EPOCH_TILES tiles inside one segment), their partial slots,
// the descriptor tables on the device.  seg_launch scores a range of segments (fast pass + ordered reduce + the
// certificate's list of uncertain (segment, accession) pairs), seg_finish re-scores those pairs in reference order
// and patches them in.  Everything is enqueued; nothing waits.
struct SegPlan {
    FastGeom g0;
    int64_t n_parts = 0, tiles_per_part = 0;
    std::vector<int64_t> slot0;             // host: first part / slot of every segment, [n_seg + 1]
    const int64_t *d_slot0 = nullptr, *d_desc = nullptr;
};

static int seg_plan(snpm_ctx *ctx, SegJob &j, SegPlan &pl)
{
    snpm_panel *p = j.p;
    const int64_t n_seg = j.n_seg;
    // int8: a dword (4 accessions) per lane, k_fast<4, SEG>; packed: a dword (16 accessions) per lane, k_fast_packed_q4<SEG>
    const bool q4 = p->packed != 0;
    pl.g0 = q4 ? fast_geom(ctx, p->n_acc, TILE_ROWS, 2, 16, TILE_ROWS, q4_waves_per_block(p->n_acc))
               : fast_geom(ctx, p->n_acc, TILE_ROWS, 2, 4, TILE_ROWS);
    int64_t total_tiles = 0, kmax = 1;
    for (int64_t s = 0; s < n_seg; ++s) {
        const int64_t len = j.seg_off[s + 1] - j.seg_off[s];
        total_tiles += (len + TILE_ROWS - 1) / TILE_ROWS;
        kmax = std::max<int64_t>(kmax, (len + j.chunk - 1) / j.chunk);
    }
    // enough parts to fill the chip a few times over
    // (int8 panels: 32 parts per CU and column block, gathered batches of 64 samples 3.07 -> 2.72 ms with them; packed panels
    // measure the same from 8 to 64 and keep 8 -- profiles/r03j_ab_seg_blocks.txt)
    const int per_cu = ctx->seg_blocks_per_cu > 0 ? ctx->seg_blocks_per_cu : (q4 ? 8 : 32);
    // ... but k_reduce_seg adds a segment's slots one after the other: a batch of 8 samples cut into 8192 parts spent 0.21 of its
    // 0.76 ms there (1024 dependent additions per accession); at most 256 slots per segment
    const int64_t want_blocks = std::max<int64_t>(1, std::min<int64_t>((int64_t)ctx->n_cu * per_cu / std::max<int64_t>(1, pl.g0.n_colblocks),
                                                                       std::max<int64_t>((int64_t)ctx->n_cu, 256 * n_seg)));
    pl.tiles_per_part = std::max<int64_t>(2, std::min<int64_t>(EPOCH_TILES, (total_tiles + want_blocks - 1) / want_blocks));
    // [seg_off | slot0 | part_desc], built in pinned memory (the copy below is asynchronous)
    const int64_t max_parts = total_tiles / pl.tiles_per_part + n_seg + 1;
    const size_t words = 2 * ((size_t)n_seg + 1) + 3 * (size_t)max_parts;
    if (ctx->h_desc_cap < words * sizeof(int64_t)) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->h_desc) (void)hipHostFree(ctx->h_desc);
        ctx->h_desc = nullptr;
        ctx->h_desc_cap = 0;
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_desc, std::max<size_t>(words * sizeof(int64_t), 1 << 16), hipHostMallocDefault));
        ctx->h_desc_cap = std::max<size_t>(words * sizeof(int64_t), 1 << 16);
    } else {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));     // the previous plan's copy may still read the buffer
    }
    int64_t *h = ctx->h_desc;
    memcpy(h, j.seg_off, ((size_t)n_seg + 1) * sizeof(int64_t));
    const size_t o_slot0 = (size_t)n_seg + 1, o_desc = 2 * ((size_t)n_seg + 1);
    pl.slot0.assign((size_t)n_seg + 1, 0);
    int64_t n_parts = 0;
    size_t w = o_desc;
    for (int64_t s = 0; s < n_seg; ++s) {
        pl.slot0[(size_t)s] = n_parts;
        const int64_t r0 = j.seg_off[s], r1 = j.seg_off[s + 1];
        for (int64_t r = r0; r < r1; r += pl.tiles_per_part * TILE_ROWS) {
            h[w++] = r;
            h[w++] = std::min<int64_t>(r1, r + pl.tiles_per_part * TILE_ROWS);
            h[w++] = n_parts++;
        }
    }
    pl.slot0[(size_t)n_seg] = n_parts;
    memcpy(h + o_slot0, pl.slot0.data(), ((size_t)n_seg + 1) * sizeof(int64_t));
    pl.n_parts = n_parts;
    int rc = ensure(ctx, ctx->ws_seg_desc, w * sizeof(int64_t));
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->ws_seg_desc.p, h, w * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    j.d_seg_off = (const int64_t *)ctx->ws_seg_desc.p;
    pl.d_slot0 = j.d_seg_off + o_slot0;
    pl.d_desc = j.d_seg_off + o_desc;
    j.kmax = kmax;
    j.cap = (int)std::max<int64_t>(64, std::min<int64_t>(SEG_PAIR_CAP, (int64_t(8) << 20) / kmax));
    rc = ensure(ctx, ctx->ws_part_score, (size_t)std::max<int64_t>(n_parts, 1) * p->ld * sizeof(double));
    if (rc) return rc;
    rc = ensure(ctx, ctx->ws_part_miss, (size_t)std::max<int64_t>(n_parts, 1) * p->ld * sizeof(uint32_t));
    if (rc) return rc;
    rc = ensure(ctx, ctx->ws_pairs, 16 + (size_t)SEG_PAIR_CAP * 2 * sizeof(int32_t));
    if (rc) return rc;
    HIPCHK(ctx, hipMemsetAsync(seg_pair_count(ctx), 0, sizeof(int), ctx->stream));
    if (j.certify) {
        rc = ensure(ctx, ctx->ws_eseg, (size_t)std::max<int64_t>(n_seg, 1) * sizeof(double));
        if (rc) return rc;
        rc = ensure(ctx, ctx->ws_pair_sums, (size_t)j.cap * (size_t)kmax * sizeof(double));
        if (rc) return rc;
    }
    return SNPM_OK;
}

static int seg_launch(snpm_ctx *ctx, const SegJob &j, const SegPlan &pl, int64_t s0, int64_t s1)
{
    snpm_panel *p = j.p;
    if (s1 <= s0) return SNPM_OK;
    int rc;
    if (j.certify) {
        // fast-pass additions a term passes through: its part (<= tiles_per_part tiles) + the parts of its segment
        const int64_t seg_parts = (j.kmax * j.chunk) / (pl.tiles_per_part * TILE_ROWS) + 2;
        const int npart = (int)((j.kmax + 3) / 4);
        rc = ensure(ctx, ctx->ws_epart, (size_t)(s1 - s0) * (size_t)npart * 3 * sizeof(double));
        if (rc) return rc;
        hipLaunchKernelGGL(k_eseg_part, dim3((unsigned)npart, (unsigned)(s1 - s0)), dim3(256), 0, ctx->stream, j.d_w,
                           j.d_seg_off, j.chunk, s0, npart, (double *)ctx->ws_epart.p);
        hipLaunchKernelGGL(k_eseg_finish, dim3((unsigned)(s1 - s0)), dim3(256), 0, ctx->stream,
                           (const double *)ctx->ws_epart.p, j.d_seg_off, j.chunk,
                           pl.tiles_per_part * TILE_ROWS + seg_parts + 2, s0, npart, (double *)ctx->ws_eseg.p);
        HIPCHK(ctx, hipGetLastError());
    }
    const int64_t p0 = pl.slot0[(size_t)s0], p1 = pl.slot0[(size_t)s1];
    if (p1 > p0) {
        const int64_t np = p1 - p0;
        const unsigned gy = (unsigned)std::min<int64_t>(np, 65535);
        const unsigned gz = (unsigned)((np + gy - 1) / gy);
        dim3 grid((unsigned)pl.g0.n_colblocks, gy, gz), block(WAVE * pl.g0.wpb);
        const bool nt = ctx->nt_loads != 0;
        const int64_t *desc = pl.d_desc + 3 * p0;
        if (p->packed) rc = nt ? launch_q4_seg<true>(ctx, j, grid, block, np, desc) : launch_q4_seg<false>(ctx, j, grid, block, np, desc);
        else rc = nt ? launch_fast_seg<4, true>(ctx, j, grid, block, np, desc) : launch_fast_seg<4, false>(ctx, j, grid, block, np, desc);
        if (rc) return rc;
    }
    ProfScope ps(ctx, PK_REDUCE);
    const int thr = 64;
    hipLaunchKernelGGL(k_reduce_seg, dim3((unsigned)((p->n_acc + thr - 1) / thr), (unsigned)(s1 - s0)), dim3(thr), 0, ctx->stream,
                       (const double *)ctx->ws_part_score.p, (const uint32_t *)ctx->ws_part_miss.p, pl.d_slot0, j.d_seg_off,
                       p->ld, p->n_acc, j.d_score, j.d_ninfo, j.ldo,
                       j.certify ? (const double *)ctx->ws_eseg.p : (const double *)nullptr, ctx->debug_reeval,
                       seg_pairs(ctx), seg_pair_count(ctx), j.cap, s0);
    HIPCHK(ctx, hipGetLastError());
    return SNPM_OK;
}

static int seg_finish(snpm_ctx *ctx, const SegJob &j)
{
    snpm_panel *p = j.p;
    if (!j.certify || j.n_seg == 0) return SNPM_OK;
    const bool gather = j.d_row_idx != nullptr;
    if (single_accession(p)) {
        ProfScope ps(ctx, PK_STRICT);
        int rc = launch_strict_single(ctx, p, j.d_row_idx, j.row0, j.d_w, j.skip, j.d_seg_off, j.chunk, j.n_total, 0, j.n_seg,
                                      (const int32_t *)seg_pairs(ctx), (const int *)seg_pair_count(ctx), j.cap, 0, j.kmax,
                                      dim3((unsigned)std::min<int64_t>(std::max<int64_t>(j.kmax, 1), 1024), (unsigned)std::min(j.cap, 256)),
                                      (double *)ctx->ws_pair_sums.p, nullptr, 0);
        if (rc) return rc;
    } else {
        ProfScope ps(ctx, PK_STRICT);
        dim3 grid((unsigned)std::min<int64_t>(std::max<int64_t>(j.kmax, 1), 256), (unsigned)std::min(j.cap, 128));   // a wave per (pair, chunk); both axes walk (the usual launch finds no pair: 128 rows of blocks leave in 8 us, 512 took 23)
#define LAUNCH_PAIRS(S, G)                                                                                        \
    hipLaunchKernelGGL((k_strict_pairs<S, G>), grid, dim3(WAVE), 0, ctx->stream, p->d, p->kpitch, p->desc, j.d_row_idx,     \
                       j.row0, j.d_w, j.d_seg_off, j.chunk, (const int32_t *)seg_pairs(ctx), (const int *)seg_pair_count(ctx), \
                       j.cap, j.kmax, (double *)ctx->ws_pair_sums.p)
        if (j.skip) {
            if (gather) LAUNCH_PAIRS(true, true); else LAUNCH_PAIRS(true, false);
        } else {
            if (gather) LAUNCH_PAIRS(false, true); else LAUNCH_PAIRS(false, false);
        }
#undef LAUNCH_PAIRS
        HIPCHK(ctx, hipGetLastError());
    }
    ProfScope ps(ctx, PK_SCAN);
    hipLaunchKernelGGL(k_scan_pairs, dim3((unsigned)std::min(j.cap, 1024)), dim3(64), 0, ctx->stream, (const double *)ctx->ws_pair_sums.p,
                       j.d_seg_off, j.chunk, (const int32_t *)seg_pairs(ctx), (const int *)seg_pair_count(ctx), j.cap, j.kmax,
                       j.d_score, j.ldo);
    HIPCHK(ctx, hipGetLastError());
    return SNPM_OK;
}

static int run_segmented(snpm_ctx *ctx, SegJob &j)
{
    if (j.n_seg == 0) return SNPM_OK;
    SegPlan pl;
    int rc = seg_plan(ctx, j, pl);
    if (!rc) rc = seg_launch(ctx, j, pl, 0, j.n_seg);
    if (!rc) rc = seg_finish(ctx, j);
    return rc;
}

// Host bytes -> device through the double-buffered pinned staging slabs on the copy stream (the slab is filled by a few
// threads while the previous one is in flight); memory the caller pinned itself goes straight to hipMemcpyAsync.
static int ensure_stage(snpm_ctx *ctx)
{
    for (int i = 0; i < 2; ++i) {
        if (!ctx->stage[i]) {
            hipError_t e = hipHostMalloc(&ctx->stage[i], snpm_ctx::kStageBytes, hipHostMallocDefault);
            if (e != hipSuccess) return set_err(ctx, SNPM_ERR_OOM, "hipHostMalloc staging failed: %s", hipGetErrorString(e));
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->stage_done[i], hipEventDisableTiming));
        }
    }
    return SNPM_OK;
}

void parallel_copy(snpm_ctx *ctx, int8_t *dst, const int8_t *src, size_t n);

static int stage_bytes(snpm_ctx *ctx, void *d_dst, const void *h_src, size_t nbytes, bool src_pinned)
{
    if (nbytes == 0) return SNPM_OK;
    if (src_pinned) {
        HIPCHK(ctx, hipMemcpyAsync(d_dst, h_src, nbytes, hipMemcpyHostToDevice, ctx->copy_stream));
        return SNPM_OK;
    }
    for (size_t o = 0; o < nbytes; o += snpm_ctx::kStageBytes) {
        const size_t piece = std::min(snpm_ctx::kStageBytes, nbytes - o);
        const int which = ctx->stage_which;
        ctx->stage_which ^= 1;
        if (ctx->stage_busy[which]) {
            HIPCHK(ctx, hipEventSynchronize(ctx->stage_done[which]));
            ctx->stage_busy[which] = false;
        }
        parallel_copy(ctx, (int8_t *)ctx->stage[which], (const int8_t *)h_src + o, piece);
        HIPCHK(ctx, hipMemcpyAsync((char *)d_dst + o, ctx->stage[which], piece, hipMemcpyHostToDevice, ctx->copy_stream));
        HIPCHK(ctx, hipEventRecord(ctx->stage_done[which], ctx->copy_stream));
        ctx->stage_busy[which] = true;
    }
    return SNPM_OK;
}

// n int64 row indices (n * 4 bytes <= one slab) -> int32 on the device; values that do not fit become -1
static int stage_rows32(snpm_ctx *ctx, int32_t *d_dst, const int64_t *h_src, int64_t n)
{
    if (n == 0) return SNPM_OK;
    const int which = ctx->stage_which;
    ctx->stage_which ^= 1;
    if (ctx->stage_busy[which]) {
        HIPCHK(ctx, hipEventSynchronize(ctx->stage_done[which]));
        ctx->stage_busy[which] = false;
    }
    int32_t *slab = (int32_t *)ctx->stage[which];
    const int nthreads = (int)std::min<int64_t>(ctx->stage_threads, std::max<int64_t>(1, n >> 18));
    auto narrow = [=](int64_t a, int64_t b) {
        for (int64_t i = a; i < b; ++i) {
            const int64_t v = h_src[i];
            slab[i] = (v >= 0 && v <= 0x7fffffff) ? (int32_t)v : -1;
        }
    };
    if (nthreads <= 1) {
        narrow(0, n);
    } else {
        std::vector<std::thread> pool;
        const int64_t per = (n + nthreads - 1) / nthreads;
        for (int t = 0; t < nthreads; ++t)
            if (t * per < n) pool.emplace_back(narrow, t * per, std::min<int64_t>(n, (t + 1) * per));
        for (auto &th : pool) th.join();
    }
    HIPCHK(ctx, hipMemcpyAsync(d_dst, slab, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->copy_stream));
    HIPCHK(ctx, hipEventRecord(ctx->stage_done[which], ctx->copy_stream));
    ctx->stage_busy[which] = true;
    return SNPM_OK;
}

static bool host_pointer_is_pinned(const void *ptr)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

// Results -> the caller's host arrays through the context's pinned slab.  A device-to-host copy into pageable memory is staged by
// the runtime piece by piece with the host waiting in between (2.3 MB of results of 64 samples: ~0.35 ms of a 1.4-ms call; the 7 MB
// of 399 windows: 0.2 ms of a 0.38-ms call); a copy into pinned memory is one DMA, and the host threads then move the slab into
// the caller's arrays.  Usage: add() every output (enqueues the copy on ctx->stream), synchronise the stream, finish().
// Outputs the caller pinned itself, and sets too large for the slab, are copied directly.  The slab starts 256 bytes into
// ctx->h_pinned (the first words hold the small read-backs of the calls).
struct HostFetch {
    snpm_ctx *ctx;
    struct Item { void *dst; const void *src; size_t bytes; };
    Item items[8];
    int n_items = 0;
    size_t used = 0, cap = 0;
    explicit HostFetch(snpm_ctx *c) : ctx(c) {}
    // total: bytes of every output that may come; call before the first add()
    int reserve(size_t total)
    {
        cap = 0;
        if (total == 0 || total > (size_t(256) << 20)) return ensure_pinned(ctx, 256);
        int rc = ensure_pinned(ctx, 256 + total);
        if (!rc) cap = total;
        return rc;
    }
    int add(void *host_dst, const void *dev_src, size_t bytes)
    {
        if (!host_dst || bytes == 0) return SNPM_OK;
        // (arrays above 1 MiB go directly: the runtime pipelines a large pageable copy well -- the 2 x 3.6 MB of 399 windows took
        // 0.12 ms LONGER through the slab and a host copy than directly -- what the slab saves is the fixed cost per copy)
        if (n_items < 8 && bytes <= (size_t(1) << 20) && used + bytes <= cap && !host_pointer_is_pinned(host_dst)) {
            char *slab = (char *)ctx->h_pinned + 256 + used;
            HIPCHK(ctx, hipMemcpyAsync(slab, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
            items[n_items++] = Item{host_dst, slab, bytes};
            used += (bytes + 63) & ~size_t(63);
        } else {
            HIPCHK(ctx, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        }
        return SNPM_OK;
    }
    void finish()                   // after the stream has been synchronised
    {
        for (int i = 0; i < n_items; ++i) parallel_copy(ctx, (int8_t *)items[i].dst, (const int8_t *)items[i].src, items[i].bytes);
        n_items = 0;
        used = 0;
    }
};

// number of accessions the last certified run flagged (synchronises the stream)
int read_count(snpm_query *q, int64_t *out)
{
    snpm_ctx *ctx = q->panel->ctx;
    *out = 0;
    if (!q->count_valid) return SNPM_OK;
    int rc = ensure_pinned(ctx, 64);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, q->cert_count(), sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *out = *(const int *)ctx->h_pinned;
    return SNPM_OK;
}

int upload_seg_off(snpm_ctx *ctx, const std::vector<int64_t> &off)
{
    int rc = ensure(ctx, ctx->ws_seg_off, off.size() * sizeof(int64_t));
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->ws_seg_off.p, off.data(), off.size() * sizeof(int64_t), hipMemcpyHostToDevice,
                               ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // `off` is a host temporary
    return SNPM_OK;
}

