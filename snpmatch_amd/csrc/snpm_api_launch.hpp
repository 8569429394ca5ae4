// snpm_api_launch.hpp -- launch geometry of the fast passes and their launches (inside the anonymous namespace of snpm_api.hip).
// Part of the one translation unit of libsnpmatch_hip.so: included by snpm_api.hip at this place, not on its own.
// ---- launch geometry of the fast pass ---------------------------------------------------------
struct FastGeom {
    int bpl, wpb;
    int tile_rows = TILE_ROWS;
    int64_t n_wc, n_colblocks, n_parts, part_rows;
    int64_t n_epochs, n_slots, n_groups;      // partial slots = n_epochs * n_parts, reduced in groups
};

template <int BPL, bool SKIP, bool GATHER, bool NT>
int occupancy_of(int threads)
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_fast<BPL, SKIP, GATHER, NT>, threads, 0) != hipSuccess) nb = 0;
    return nb;
}

int pick_bpl(snpm_ctx *ctx, int64_t n_acc)
{
    // Bytes per lane of the fast pass.  Measured on MI355X (10k x 6.25M panel, round 1): 4 B per lane
    // streams at 6.5 TB/s, 8 B at 4.4, 16 B at 5.3 -- the kernel is latency-bound and the narrow layout
    // keeps the most waves resident; it also has the best lane utilisation for every n_acc.  The wider
    // instantiations stay selectable (SNPM_FORCE_BPL) for experiments.
    (void)n_acc;
    if (ctx->force_bpl == 8 || ctx->force_bpl == 16) return ctx->force_bpl;
    return 4;
}

FastGeom fast_geom(snpm_ctx *ctx, int64_t n_acc, int64_t n, int occ_blocks_hint, int bpl, int tile_rows = TILE_ROWS,
                   int wpb_fixed = 0, int kernel_parts_mult = 1)
{
    FastGeom g;
    g.bpl = bpl;
    g.tile_rows = tile_rows;
    const int64_t span = (int64_t)WAVE * bpl;
    g.n_wc = std::max<int64_t>(1, (n_acc + span - 1) / span);
    if (ctx->force_wpb >= 1 && ctx->force_wpb <= MAX_WAVES_PER_BLOCK) {
        g.wpb = (int)std::min<int64_t>(ctx->force_wpb, g.n_wc);
    } else if (wpb_fixed > 0) {
        g.wpb = (int)std::min<int64_t>(wpb_fixed, g.n_wc);
    } else if (g.n_wc <= 8) {
        g.wpb = (int)g.n_wc;
    } else {
        // Waves per block.  Two measured effects (round 1, tools/bench_shape.sh): waves of a block that
        // fall outside the panel only idle at the barriers, but they hold wave slots (cost ~ the idle
        // fraction); blocks whose wave count is not a multiple of the 4 SIMDs load them unevenly
        // (5- and 7-wave blocks ran ~10-15 % slower than 4/8-wave blocks of the same shape, 6-wave ~5 %).
        double best = -1.0;
        int best_w = 8;
        for (int w = 8; w >= 4; --w) {
            const int64_t blocks = (g.n_wc + w - 1) / w;
            const double active = (double)g.n_wc / (double)(blocks * w);
            const double balance = (w % 4 == 0) ? 1.0 : ((w % 2 == 0) ? 0.95 : 0.85);
            if (active * balance > best) { best = active * balance; best_w = w; }
        }
        g.wpb = best_w;
    }
    g.n_colblocks = (g.n_wc + g.wpb - 1) / g.wpb;
    int occ = occ_blocks_hint > 0 ? occ_blocks_hint : 2;
    // Resident blocks per CU of the int8 kernel.  On long scans full occupancy is not the optimum for 4- and
    // 5-wave blocks (measured, fast mode, panels of 64 GB: 5-wave blocks 3 per CU 80.0 % of HBM peak vs 78.1 % at
    // 4 per CU on 1252 x 50M, 78.5 vs 76.1 % on 2500 x 25M; 4-wave blocks 4-5 per CU 80 % vs 77.5 % at 6 on
    // 5000 x 12.5M), while 6- to 8-wave blocks and short scans (1135 x 11M, 14 GB) are 1-3 % better at full
    // occupancy.  The part count stays a multiple of the CU count either way (uneven counts cost 5-10 %).
    const int64_t pitch_bytes = ((n_acc + 255) / 256) * 256;
    if (bpl == 4 && occ_blocks_hint > 0 && !ctx->full_occupancy && (g.wpb == 4 || g.wpb == 5) &&
        n * pitch_bytes >= (int64_t(32) << 30))
        occ = std::min(occ, std::max(3, 18 / g.wpb));
    // Full 8-wave blocks (n_acc within 8 waves of a multiple of 2048): TWO resident blocks per CU instead of the three that fit
    // -- 16 row loads in flight per SIMD instead of 24 -- measured better or equal on every shape of that kind from 20 GB up
    // (round 3, profiles/r03b_ab_occ_cap*.txt: 10 000 x 20M 0.791 -> 0.808 of HBM peak, 8192 x 24M 0.767 -> 0.787, 16 384 x 12M
    // 0.758 -> 0.779, 6144 x 30M 0.822 -> 0.833, 20 480 x 9M 0.805 -> 0.821, 4096 x 40M 0.796 -> 0.804, 2048 x 50M equal), while
    // 5- and 7-wave blocks lose 10-25 % with it (1252 / 2500 / 5000 / 12 500 accessions) and keep their own cap above.
    if (bpl == 4 && occ_blocks_hint > 0 && !ctx->full_occupancy && g.wpb == 8 &&
        n * pitch_bytes >= (int64_t(4) << 30))
        occ = std::min(occ, 2);
    if (ctx->occ_cap > 0) occ = std::min(occ, ctx->occ_cap);
    // one-wave blocks of the int8 kernel (panels of up to 256 accessions): four times as many parts as resident blocks
    // (256 x 100M rows 0.597 -> 0.754 of HBM peak with the 128-row tiles; two-wave blocks and wider: no gain)
    const int narrow_mult = (bpl == 4 && g.wpb == 1 && occ_blocks_hint > 0 && ctx->parts_mult == 1) ? 4 : 1;
    const int64_t n_tiles = std::max<int64_t>(1, (n + tile_rows - 1) / tile_rows);
    // kernel_parts_mult: k_fast_packed_q4 runs best with MORE parts than resident blocks (run_fast) -- as long as a part keeps
    // eight tiles or so: every part costs a slot of partial sums to write and to add up, which on short scans outweighs the
    // gain (1135 accessions x 11M rows with 16 parts per block: kernel 1.07 -> 1.09 ms, the step 1.11 -> 1.25 ms; 32 / 16 / 8 / 4
    // tiles per part by the time of the whole step: profiles/r03j_ab_part_min_tiles.txt)
    int kmult = 1;
    if (ctx->parts_mult == 1 && kernel_parts_mult > 1 && g.wpb != 5) {       // (the one 5-wave block shape, 4097-5120 accessions: 2 / 4 parts per block lose 10 / 2 %, 8 gain 1 %)
        const int64_t base_parts = std::max<int64_t>(1, (int64_t)ctx->n_cu * occ * narrow_mult / g.n_colblocks);
        kmult = (int)std::max<int64_t>(1, std::min<int64_t>(kernel_parts_mult, n_tiles / (base_parts * std::max(1, ctx->part_min_tiles))));
    }
    int64_t resident = (int64_t)ctx->n_cu * occ * std::max(1, ctx->parts_mult) * narrow_mult * kmult;
    int64_t n_parts = std::max<int64_t>(1, resident / g.n_colblocks);
    n_parts = std::min(n_parts, n_tiles);                    // part p scores tiles p, p+P, p+2P, ...
    if (ctx->debug_max_parts > 0) n_parts = std::min<int64_t>(n_parts, ctx->debug_max_parts);   // tests: long parts
    n_parts = std::min<int64_t>(n_parts, 65535);             // grid.y
    g.n_parts = n_parts;
    const int64_t tiles_per_part = (n_tiles + n_parts - 1) / n_parts;
    g.part_rows = tiles_per_part * tile_rows;                // rows per part (upper bound)
    g.n_epochs = std::max<int64_t>(1, (tiles_per_part + EPOCH_TILES - 1) / EPOCH_TILES);
    g.n_slots = g.n_epochs * g.n_parts;
    g.n_groups = (g.n_slots + REDUCE_GROUP - 1) / REDUCE_GROUP;
    return g;
}

template <int BPL, bool SKIP, bool GATHER, bool NT>
int launch_fast_t(snpm_query *q, const FastGeom &g)
{
    snpm_ctx *ctx = q->panel->ctx;
    snpm_panel *p = q->panel;
    dim3 grid((unsigned)g.n_colblocks, (unsigned)g.n_parts);
    dim3 block(WAVE * g.wpb);
    ProfScope ps(ctx, PK_FAST);
    if (BPL == 4 && g.tile_rows == LONG_TILE_ROWS)          // long scans: tiles of LONG_TILE_ROWS rows (fast_tile_rows)
        hipLaunchKernelGGL((k_fast<BPL, SKIP, GATHER, NT, false, (BPL == 4 ? LONG_TILE_ROWS : TILE_ROWS)>), grid, block, 0, ctx->stream, p->d,
                           p->pitch, q->d_row_idx, q->row0, q->n, q->d_lut, (double *)ctx->ws_part_score.p, (uint32_t *)ctx->ws_part_miss.p,
                           p->ld, (const int64_t *)nullptr);
    else
        hipLaunchKernelGGL((k_fast<BPL, SKIP, GATHER, NT>), grid, block, 0, ctx->stream, p->d, p->pitch, q->d_row_idx, q->row0,
                           q->n, q->d_lut, (double *)ctx->ws_part_score.p, (uint32_t *)ctx->ws_part_miss.p, p->ld,
                           (const int64_t *)nullptr, (BPL == 4 && g.tile_rows < TILE_ROWS) ? g.tile_rows : 0);
    HIPCHK(ctx, hipGetLastError());
    return SNPM_OK;
}

template <int BPL, bool NT>
int launch_fast_b(snpm_query *q, const FastGeom &g, bool skip, bool gather)
{
    if (skip)
        return gather ? launch_fast_t<BPL, true, true, NT>(q, g) : launch_fast_t<BPL, true, false, NT>(q, g);
    return gather ? launch_fast_t<BPL, false, true, NT>(q, g) : launch_fast_t<BPL, false, false, NT>(q, g);
}

// block shape of k_fast_packed_q4 (a wave covers 1024 accessions): see run_fast
// Waves per block of k_fast_packed_q4 (a wave covers 1024 accessions; a block builds its four-row tables once for all its waves, and
// the waves of the last block that lie past the panel only help with that).  Round 3 sweep (profiles/r03g_ab_q4_wpb*.txt): panels of up
// to eight waves run as ONE block of exactly that many waves (6144 accessions 0.476 -> 0.556 of HBM peak on packed bytes, 7000
// 0.53 -> 0.588, 8192 0.61 -> 0.65); wider panels take the block size among 4, 7 and 8 waves that launches the fewest waves (ties: the
// larger block): 13 312 -> 7-wave blocks 0.519 -> 0.576, 14 336 -> 7 (0.546 -> 0.598), 15 360 / 16 384 / 24 576 -> 8 (0.577 -> 0.616,
// 0.60 -> 0.63), 9216 / 10 000 / 11 264 / 12 288 stay on 4-wave blocks (5- and 6-wave blocks lose 10-30 % there).
static int q4_waves_per_block(int64_t n_acc)
{
    const int64_t n_wc = (n_acc + 1023) / 1024;
    if (n_wc <= 8) return (int)n_wc;
    int best = 4;
    int64_t best_waves = (n_wc + 3) / 4 * 4;
    for (int w : {7, 8}) {
        const int64_t waves = (n_wc + w - 1) / w * w;
        if (waves <= best_waves) {
            best = w;
            best_waves = waves;
        }
    }
    return best;
}

// rows per LDS tile of k_fast_packed_q4 by block size (see the kernel): blocks of fewer than four waves take smaller tiles so that
// LDS does not bound the resident waves of a CU (SNPM_Q4_TILE_ROWS = 16 / 32 / 64 forces one size)
static int q4_tile_rows(const snpm_ctx *ctx, int wpb)
{
    if (ctx->q4_tile_rows == 16 || ctx->q4_tile_rows == 32 || ctx->q4_tile_rows == 64) return ctx->q4_tile_rows;
    return wpb >= 4 ? 64 : (wpb >= 2 ? 32 : 16);
}

template <bool SKIP, bool GATHER, bool NT, int TR>
int launch_p16_t(snpm_query *q, const FastGeom &g, int *occ_out, int threads)
{
    snpm_ctx *ctx = q->panel->ctx;
    snpm_panel *p = q->panel;
    if (occ_out) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_fast_packed_q4<SKIP, GATHER, NT, false, TR>, threads, 0) != hipSuccess) nb = 0;
        *occ_out = nb;
        return SNPM_OK;
    }
    dim3 grid((unsigned)g.n_colblocks, (unsigned)g.n_parts);
    dim3 block(WAVE * g.wpb);
    ProfScope ps(ctx, PK_FAST);
    hipLaunchKernelGGL((k_fast_packed_q4<SKIP, GATHER, NT, false, TR>), grid, block, 0, ctx->stream, p->d, p->kpitch, q->d_row_idx,
                       q->row0, q->n, q->d_lut, (double *)ctx->ws_part_score.p, (uint32_t *)ctx->ws_part_miss.p, p->ld, p->n_acc,
                       p->desc);
    HIPCHK(ctx, hipGetLastError());
    return SNPM_OK;
}

int launch_p16(snpm_query *q, const FastGeom &g, bool skip, bool gather, bool nt, int *occ_out, int threads)
{
    const int tr = q4_tile_rows(q->panel->ctx, threads / WAVE);
#define P16_CASE(S, G, N)                                                                           \
    if (skip == S && gather == G && nt == N) {                                                      \
        if (tr == 16) return launch_p16_t<S, G, N, 16>(q, g, occ_out, threads);                     \
        if (tr == 32) return launch_p16_t<S, G, N, 32>(q, g, occ_out, threads);                     \
        return launch_p16_t<S, G, N, 64>(q, g, occ_out, threads);                                   \
    }
    P16_CASE(false, false, false) P16_CASE(false, false, true) P16_CASE(false, true, false) P16_CASE(false, true, true)
    P16_CASE(true, false, false)  P16_CASE(true, false, true)  P16_CASE(true, true, false)  P16_CASE(true, true, true)
#undef P16_CASE
    return SNPM_ERR_STATE;
}

// hard-call samples on packed panels (k_fast_bits)
template <bool SKIP, bool GATHER, bool NT>
int launch_bits_t(snpm_query *q, const FastGeom &g, int *occ_out, int threads)
{
    snpm_ctx *ctx = q->panel->ctx;
    snpm_panel *p = q->panel;
    if (occ_out) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_fast_bits<SKIP, GATHER, NT>, threads, 0) != hipSuccess) nb = 0;
        *occ_out = nb;
        return SNPM_OK;
    }
    ProfScope ps(ctx, PK_FAST);
    // grid = (parts, column blocks): the part is the fast block index (XCD balance, see the kernel)
    hipLaunchKernelGGL((k_fast_bits<SKIP, GATHER, NT>), dim3((unsigned)g.n_parts, (unsigned)g.n_colblocks), dim3(WAVE * g.wpb), 0,
                       ctx->stream, p->d, p->kpitch, q->d_row_idx, q->row0, q->n, (const uint8_t *)q->d_wbits,
                       (double *)ctx->ws_part_score.p, (uint32_t *)ctx->ws_part_miss.p, p->ld, p->n_acc, p->desc);
    HIPCHK(ctx, hipGetLastError());
    return SNPM_OK;
}

int launch_bits(snpm_query *q, const FastGeom &g, bool skip, bool gather, bool nt, int *occ_out, int threads)
{
#define BITS_CASE(S, G, N) if (skip == S && gather == G && nt == N) return launch_bits_t<S, G, N>(q, g, occ_out, threads)
    BITS_CASE(false, false, false); BITS_CASE(false, false, true); BITS_CASE(false, true, false); BITS_CASE(false, true, true);
    BITS_CASE(true, false, false);  BITS_CASE(true, false, true);  BITS_CASE(true, true, false);  BITS_CASE(true, true, true);
#undef BITS_CASE
    return SNPM_ERR_STATE;
}

template <int BPL, bool NT>
int occ_b(bool skip, bool gather, int threads)
{
    if (skip) return gather ? occupancy_of<BPL, true, true, NT>(threads) : occupancy_of<BPL, true, false, NT>(threads);
    return gather ? occupancy_of<BPL, false, true, NT>(threads) : occupancy_of<BPL, false, false, NT>(threads);
}

int ensure_lut(snpm_query *q, int skip)
{
    snpm_ctx *ctx = q->panel->ctx;
    if (q->lut_skip == skip) return SNPM_OK;
    if (q->n > 0) {
        ProfScope ps(ctx, PK_LUT);
        const int thr = 256;
        hipLaunchKernelGGL(k_build_lut, dim3((unsigned)((q->n + thr - 1) / thr)), dim3(thr), 0, ctx->stream, q->d_w,
                           q->d_lut, q->n, skip, (int *)nullptr);
        HIPCHK(ctx, hipGetLastError());
    }
    q->lut_skip = skip;
    return SNPM_OK;
}

