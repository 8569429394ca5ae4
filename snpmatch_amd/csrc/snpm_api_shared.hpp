// snpm_api_shared.hpp -- host side of the shared-row scan of a batch (kernels and the idea: snpm_k_shared.hpp).
// Part of the one translation unit of libsnpmatch_hip.so: included by snpm_api.hip at this place (anonymous namespace), not on its own.
//
// shared_rows_try() takes a SegJob whose inputs are on the device (concatenated row lists + weights, sample = segment) and either
// scores EVERY sample into j.d_score / j.d_ninfo with the certificate's list of unproven (sample, accession) pairs left for
// seg_finish -- the same state seg_launch leaves -- or declines (rows not strictly increasing per sample, weights outside [0, 1],
// call codes > 2 in the panel, too little overlap between the samples) and touches nothing the per-sample pass needs.
struct SharedStats {
    int64_t union_rows = 0;         // distinct panel rows of the batch
    double density = 0.0;           // N / (union_rows * n_samples): share of (sample, union row) slots that hold a call
    int taken = 0;
    int reason = 0;                 // why not: 1 policy off, 2 too few samples / rows, 3 unsorted rows, 4 weights outside [0, 1], 5 codes > 2 in the panel,
                                    //          6 overlap below the threshold, 7 panel / batch too large for 32-bit indices, 8 a row index outside the panel
    int tiles = 0, groups = 0, accgroups = 0, passes = 0, digits = 0;
    const long long *d_meta = nullptr;  // taken: meta[1] bit 1 is raised by the expansion when a weight lies outside [0, 1] (read with the results)
};

static int shared_rows_try(snpm_ctx *ctx, SegJob &j, bool forced, SharedStats &st)
{
    snpm_panel *p = j.p;
    st = SharedStats();
    const int64_t n_seg = j.n_seg, N = j.n_total;
    if (n_seg < (forced ? 1 : ctx->shared_min_samples) || N < 1) { st.reason = 2; return SNPM_OK; }
    if (p->n_snp > 0x7fffffffLL || N > 0xfffffff0LL || n_seg > 65535 /* a grid dimension */ || !j.d_row_idx) { st.reason = 7; return SNPM_OK; }
    int rc;
    int64_t maxlen = 0, kmax = 1;
    for (int64_t s = 0; s < n_seg; ++s) {
        const int64_t len = j.seg_off[s + 1] - j.seg_off[s];
        maxlen = std::max(maxlen, len);
        kmax = std::max<int64_t>(kmax, (len + j.chunk - 1) / j.chunk);
    }
    // digits of the fixed-point weights: the fewest that keep a sample's quantisation (2^-F per matched SNP) below 2^-20 -- the
    // certificate then flags about one (sample, accession) pair in a million for it -- unless the caller fixed the number
    int digits = ctx->shared_digits;
    if (digits == 0) digits = maxlen <= (int64_t(1) << 18) ? 5 : (maxlen <= (int64_t(1) << 26) ? 6 : 7);
    st.digits = digits;
    const int rps = digits + 1;
    // ---- union of the samples' rows: bitmap of panel rows -> ranks -> row list
    const int64_t n_words = (p->n_snp + 31) / 32;
    const int64_t n_blocks = (n_words + SH_WORDS_PER_BLOCK - 1) / SH_WORDS_PER_BLOCK;
    const int64_t u_max = std::min<int64_t>(N, p->n_snp);
    const size_t urow_entries = (size_t)u_max + (size_t)(SH_PAD_STEPS + SH_DEPTH + 1) * SH_STEP_ROWS;   // zero rows behind the union: the steps that pad it to a multiple of SH_DEPTH and the kernel's read-ahead
    // [meta | seg_off | per-sample flag "a weight is not an integer"] and the bitmap behind it: ONE memset clears both
    const size_t meta_bytes = (64 + ((size_t)n_seg + 1) * 8 + (size_t)n_seg * 4 + 255) & ~size_t(255);
    if ((rc = ensure(ctx, ctx->ws_sh_bitmap, meta_bytes + (size_t)n_words * 4))) return rc;
    if ((rc = ensure(ctx, ctx->ws_sh_wordbase, (size_t)n_words * 4))) return rc;
    if ((rc = ensure(ctx, ctx->ws_sh_blocks, (size_t)n_blocks * 8 + 256))) return rc;
    if ((rc = ensure(ctx, ctx->ws_sh_urows, urow_entries * 4))) return rc;
    long long *d_meta = (long long *)ctx->ws_sh_bitmap.p;
    int64_t *d_seg_off = (int64_t *)((char *)ctx->ws_sh_bitmap.p + 64);
    int *d_nonint = (int *)((char *)ctx->ws_sh_bitmap.p + 64 + ((size_t)n_seg + 1) * 8);
    uint32_t *d_bitmap = (uint32_t *)((char *)ctx->ws_sh_bitmap.p + meta_bytes);
    uint32_t *d_wordbase = (uint32_t *)ctx->ws_sh_wordbase.p;
    uint32_t *d_blocksum = (uint32_t *)ctx->ws_sh_blocks.p, *d_blockbase = d_blocksum + n_blocks;
    int32_t *d_urows = (int32_t *)ctx->ws_sh_urows.p;
    HIPCHK(ctx, hipMemsetAsync(ctx->ws_sh_bitmap.p, 0, meta_bytes + (size_t)n_words * 4, ctx->stream));
    const int urow_pad = (int)(urow_entries - (size_t)u_max);            // zero rows behind the union: written by k_sh_scan, which knows where it ends
    const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>((maxlen + 255) / 256, 64));
    // the sample offsets: left in pinned memory, read there by the first kernels (k_sh_probe, k_sh_mark; the latter stores the device copy)
    if ((rc = ensure_pinned(ctx, 64 + ((size_t)n_seg + 1) * 8))) return rc;
    long long *h_meta = (long long *)ctx->h_pinned;
    const int64_t *h_seg_off = (const int64_t *)((char *)ctx->h_pinned + 64);
    memcpy((char *)ctx->h_pinned + 64, j.seg_off, ((size_t)n_seg + 1) * 8);
    if (!forced && p->n_snp >= 4096 && ctx->shared_probe) {
        // automatic policy: look at 1/32 of the panel first (the batch's calls below row n_snp / 32); scattered marker sets are
        // declined here, before the full pass over every row and weight of the batch
        const int64_t row_limit = (p->n_snp / 32 + 31) / 32 * 32;
        const int64_t words_probe = row_limit / 32;
        const int64_t blocks_probe = (words_probe + SH_WORDS_PER_BLOCK - 1) / SH_WORDS_PER_BLOCK;
        hipLaunchKernelGGL(k_sh_probe, dim3((unsigned)n_seg), dim3(256), 0, ctx->stream, j.d_row_idx, h_seg_off, row_limit,
                           p->n_snp, d_bitmap, d_meta);
        hipLaunchKernelGGL(k_sh_count, dim3((unsigned)blocks_probe), dim3(256), 0, ctx->stream, (const uint32_t *)d_bitmap, words_probe, d_blocksum);
        hipLaunchKernelGGL(k_sh_scan, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t *)d_blocksum, blocks_probe, d_blockbase, d_meta,
                           (int32_t *)nullptr, 0, (const int *)nullptr, (int *)nullptr);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(h_meta, d_meta, 32, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        const int64_t u_probe = h_meta[0], n_probe = h_meta[3];
        if (u_probe > 0 && n_probe >= 1024) {
            const double est = (double)n_probe / ((double)u_probe * (double)n_seg);
            st.union_rows = u_probe * 32;                       // an estimate
            st.density = est;
            if (est < 0.8 * ctx->shared_min_density_of(p->packed != 0, n_seg)) { st.reason = 6; return SNPM_OK; }
        }
    }
    // (the pair counter is cleared by k_sh_scan: no memset of its own, and nothing but launches follows the host's wait for the union)
    if ((rc = ensure(ctx, ctx->ws_pairs, 16 + (size_t)SEG_PAIR_CAP * 2 * sizeof(int32_t)))) return rc;
    // one pass over the row lists: marks and index checks (the weights are vetted by k_sh_expand while it converts them)
    {
        ProfScope ps(ctx, PK_LUT);
        const unsigned mgx = (unsigned)std::max<int64_t>(1, std::min<int64_t>((maxlen + 1023) / 1024, 16384));        // four entries per lane
        hipLaunchKernelGGL(k_sh_mark, dim3(mgx, (unsigned)n_seg), dim3(256), 0, ctx->stream, j.d_row_idx, h_seg_off, d_seg_off, p->n_snp,
                           d_bitmap, d_meta);
        hipLaunchKernelGGL(k_sh_count, dim3((unsigned)n_blocks), dim3(256), 0, ctx->stream, (const uint32_t *)d_bitmap, n_words, d_blocksum);
        hipLaunchKernelGGL(k_sh_scan, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t *)d_blocksum, n_blocks, d_blockbase, d_meta, d_urows,
                           urow_pad, (!p->packed && p->d_other) ? (const int *)p->d_other : (const int *)nullptr, seg_pair_count(ctx));
        hipLaunchKernelGGL(k_sh_fill, dim3((unsigned)n_blocks), dim3(256), 0, ctx->stream, (const uint32_t *)d_bitmap, n_words,
                           (const uint32_t *)d_blockbase, d_wordbase, d_urows);
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipMemcpyAsync(h_meta, d_meta, 24, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t U = h_meta[0];
    const int bad = (int)h_meta[1];
    const int other = (int)(h_meta[2] & 0xffffffffLL);
    st.union_rows = U;
    st.density = U > 0 ? (double)N / ((double)U * (double)n_seg) : 0.0;
    if (bad & 4) { st.reason = 8; return SNPM_OK; }
    if (bad & 1) { st.reason = 3; return SNPM_OK; }
    if (other) { st.reason = 5; return SNPM_OK; }
    if (U < 1) { st.reason = 2; return SNPM_OK; }
    if (!forced && st.density < ctx->shared_min_density_of(p->packed != 0, n_seg)) { st.reason = 6; return SNPM_OK; }

    // ---- geometry
    // K steps of 8 union rows, padded to a multiple of SH_DEPTH with rows no sample has a call at (zero digits): the kernel's body is unconditional
    const int64_t n_steps = ((U + SH_STEP_ROWS - 1) / SH_STEP_ROWS + SH_DEPTH - 1) / SH_DEPTH * SH_DEPTH;
    const int64_t steps_ld = n_steps + 2 * SH_DEPTH;
    const int64_t ld_pos = n_steps * SH_STEP_ROWS;
    const int n_accgroups = (int)((p->n_acc + SH_WAVE_ACCS - 1) / SH_WAVE_ACCS);
    const int64_t ldn = (int64_t)n_accgroups * SH_WAVE_ACCS;
    const int64_t groups_total = (n_seg * rps + SH_GROUP_ROWS - 1) / SH_GROUP_ROWS;   // sample s owns matrix rows s * rps .. + rps - 1
    const size_t bytes_per_group = (size_t)steps_ld * SH_A_STEP_BYTES;
    const size_t budget = ctx->shared_ws_bytes;
    int64_t groups_per_pass = std::max<int64_t>(1, std::min<int64_t>(groups_total, (int64_t)(budget / std::max<size_t>(bytes_per_group, 1))));
    // row tiles: blocks of one tile run on one XCD (n_cu / 8 CUs, one block each at a time); among T = 8 m tiles take the m with the
    // shortest schedule (rounds x (steps per tile + a round's fixed cost) + the finish over T tiles) whose partial sums stay small
    auto plan_tiles = [&](int64_t groups, int &n_tiles, int64_t &steps_per_tile, int &blocks_per_tile) {
        blocks_per_tile = (int)((groups * n_accgroups + 3) / 4);
        const int cu_per_xcd = std::max(1, ctx->n_cu / 8);
        double best = 1e300;
        int best_m = 1;
        for (int m = 1; m <= 16; ++m) {
            const int64_t T = 8 * m;
            if (T * 16 > n_steps && m > 1) break;                                   // at least 16 steps per tile
            const size_t part_bytes = (size_t)T * groups * SH_GROUP_ROWS * ldn * 4;
            if (part_bytes > (size_t(192) << 20) && m > 1) break;
            const int64_t rounds = ((int64_t)m * blocks_per_tile + cu_per_xcd - 1) / cu_per_xcd;
            // in units of one K step of one wave (~0.25 us): a round also pays ~42k cycles per wave outside its steps (its start, the
            // dispatch ramp, 64 KB of sums written when all waves finish together; fitted from two tile counts:
            // profiles/r05_shared_tiles.txt), and k_sh_finish reads every tile's sums back
            const double cost = (double)rounds * ((double)((n_steps + T - 1) / T) + 100.0) + 1.1 * (double)part_bytes / 1048576.0;
            if (cost < best * 0.97) { best = cost; best_m = m; }
        }
        int64_t T = 8 * best_m;
        {
            // filler tiles: the CUs of an XCD that the last round of aligned tiles leaves idle take the blocks of further tiles
            // (a filler tile then lies on two or three XCDs: 64 samples 27 x 32 -> 27 x 36 wave tiles on 1024 slots)
            const int64_t rounds = ((int64_t)best_m * blocks_per_tile + cu_per_xcd - 1) / cu_per_xcd;
            const int64_t idle = rounds * cu_per_xcd - (int64_t)best_m * blocks_per_tile;
            const bool one_launch = !(ctx->shared_parts > 1 && ctx->aux_stream);
            if (one_launch && ctx->shared_fill) T += 8 * idle / blocks_per_tile;
        }
        if (T * 16 > n_steps) T = std::max<int64_t>(1, n_steps / 16);
        steps_per_tile = ((n_steps + T - 1) / T + SH_DEPTH - 1) / SH_DEPTH * SH_DEPTH;
        n_tiles = (int)((n_steps + steps_per_tile - 1) / steps_per_tile);
    };
    int n_tiles = 1, blocks_per_tile = 1;
    int64_t steps_per_tile = n_steps;
    plan_tiles(groups_per_pass, n_tiles, steps_per_tile, blocks_per_tile);
    // the int32 accumulators of a tile hold sums of digits (|d| <= 128) over its rows
    if (steps_per_tile * SH_STEP_ROWS > (int64_t(1) << 23) && ctx->shared_force_tiles <= 0) { st.reason = 7; return SNPM_OK; }
    if (ctx->shared_force_tiles > 0) {
        n_tiles = (int)std::min<int64_t>(ctx->shared_force_tiles, n_steps);
        steps_per_tile = ((n_steps + n_tiles - 1) / n_tiles + SH_DEPTH - 1) / SH_DEPTH * SH_DEPTH;
        n_tiles = (int)((n_steps + steps_per_tile - 1) / steps_per_tile);
    }
    const int64_t samples_per_pass = std::max<int64_t>(1, groups_per_pass * SH_GROUP_ROWS / rps);
    if ((rc = ensure(ctx, ctx->ws_sh_A, ((size_t)groups_per_pass * steps_ld + SH_PAD_STEPS) * SH_A_STEP_BYTES))) return rc;
    if ((rc = ensure(ctx, ctx->ws_sh_pos, (size_t)std::min<int64_t>(samples_per_pass, n_seg) * ld_pos * 4))) return rc;
    if ((rc = ensure(ctx, ctx->ws_sh_partial, (size_t)n_tiles * groups_per_pass * SH_GROUP_ROWS * ldn * 4))) return rc;
    // ---- certificate: pair list (the per-sample reference-order bound is a closed form here: sh_eseg_of)
    j.d_seg_off = d_seg_off;
    j.kmax = kmax;
    j.cap = (int)std::max<int64_t>(64, std::min<int64_t>(SEG_PAIR_CAP, (int64_t(8) << 20) / kmax));
    if (j.certify && (rc = ensure(ctx, ctx->ws_pair_sums, (size_t)j.cap * (size_t)kmax * sizeof(double)))) return rc;
    // ---- passes over groups of samples
    st.tiles = n_tiles; st.accgroups = n_accgroups; st.groups = (int)groups_total;
    for (int64_t s_base = 0; s_base < n_seg; s_base += samples_per_pass) {
        const int64_t s_pass = std::min<int64_t>(samples_per_pass, n_seg - s_base);
        const int64_t groups = (s_pass * rps + SH_GROUP_ROWS - 1) / SH_GROUP_ROWS;
        int tiles = n_tiles, bpt = blocks_per_tile;
        int64_t spt = steps_per_tile;
        if (groups != groups_per_pass) {
            bpt = (int)((groups * n_accgroups + 3) / 4);
        }
        // (32 .. 512 blocks per sample: the same 38 us, it moves 250 MB)
        hipLaunchKernelGGL(k_sh_pos, dim3(gx, (unsigned)s_pass), dim3(256), 0, ctx->stream, j.d_row_idx, (const int64_t *)d_seg_off, s_base,
                           (const uint32_t *)d_bitmap, (const uint32_t *)d_wordbase, (uint32_t *)ctx->ws_sh_pos.p, ld_pos);
        // The pass in parts of whole row tiles (SNPM_SHARED_PARTS > 1, an experiment that measured SLOWER and is off by default): the
        // digits of part i + 1 are laid out (k_sh_expand, on the auxiliary stream: memory latency) while part i is contracted
        // (k_sh_mfma, one wave per SIMD: the matrix cores).  One part = everything on the main stream.
        const int m_tiles = (tiles + 7) / 8;                                // tiles come in sets of 8 (one per XCD)
        const int n_parts = (ctx->shared_parts > 0 && ctx->aux_stream) ? std::max(1, std::min(ctx->shared_parts, m_tiles)) : 1;
        const bool overlap = n_parts > 1;
        if (overlap) {
            HIPCHK(ctx, hipEventRecord(ctx->aux_ev[0], ctx->stream));       // k_sh_pos (and the previous pass's readers of A) before the first expansion
            HIPCHK(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->aux_ev[0], 0));
        }
        for (int part = 0; part < n_parts; ++part) {
            const int t0 = 8 * (int)((int64_t)m_tiles * part / n_parts), t1 = std::min(tiles, 8 * (int)((int64_t)m_tiles * (part + 1) / n_parts));
            if (t1 <= t0) continue;
            const int64_t k0 = (int64_t)t0 * spt, k1 = std::min<int64_t>(n_steps, (int64_t)t1 * spt);
            hipStream_t es = overlap ? ctx->aux_stream : ctx->stream;
            {
                ProfScope ps(ctx, PK_LUT);
                const dim3 eg((unsigned)s_pass, (unsigned)std::min<int64_t>(65535, (k1 - k0 + 127) / 128));
#define LAUNCH_EXPAND(D)                                                                                                     \
    hipLaunchKernelGGL((k_sh_expand<D>), eg, dim3(256), 0, es, (const uint32_t *)ctx->ws_sh_pos.p, ld_pos, j.d_w, s_pass, j.skip,  \
                       k0, k1 - k0, steps_ld, (sh_v4i *)ctx->ws_sh_A.p, d_nonint + s_base, d_meta)
                switch (digits) {
                case 3: LAUNCH_EXPAND(3); break;
                case 4: LAUNCH_EXPAND(4); break;
                case 5: LAUNCH_EXPAND(5); break;
                case 6: LAUNCH_EXPAND(6); break;
                default: LAUNCH_EXPAND(7); break;
                }
#undef LAUNCH_EXPAND
                HIPCHK(ctx, hipGetLastError());
            }
            if (overlap) {
                HIPCHK(ctx, hipEventRecord(ctx->aux_ev[1 + part], ctx->aux_stream));
                HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->aux_ev[1 + part], 0));
            }
            {
                ProfScope ps(ctx, PK_FAST);
                const int aligned = ((t1 - t0) / 8) * 8;                             // dealt one per XCD in turn
                const int fill_per_xcd = ((t1 - t0 - aligned) * bpt + 7) / 8;         // blocks of the remaining tiles, spread over the XCDs
                const unsigned nblk = (unsigned)(8 * ((aligned / 8) * bpt + fill_per_xcd));
#define LAUNCH_SH(PK)                                                                                                          \
    hipLaunchKernelGGL((k_sh_mfma<PK>), dim3(nblk), dim3(256), 0, ctx->stream, p->d, p->kpitch, p->desc, (const int32_t *)d_urows,     \
                       (const sh_v4i *)ctx->ws_sh_A.p, n_steps, steps_ld, spt, t0, t1, (int)groups, n_accgroups, bpt, aligned,          \
                       fill_per_xcd, (int *)ctx->ws_sh_partial.p, ldn)
                if (p->packed) LAUNCH_SH(true); else LAUNCH_SH(false);
#undef LAUNCH_SH
                HIPCHK(ctx, hipGetLastError());
            }
        }
        // (the reference-order bound of a sample is a closed form of its length and its "not an integer" flag, complete once every
        // expansion of the pass has run: k_sh_finish evaluates it where it needs it)
        {
            ProfScope ps(ctx, PK_REDUCE);
            hipLaunchKernelGGL(k_sh_finish, dim3((unsigned)p->n_acc, (unsigned)((s_pass + 63) / 64)), dim3(256), 0, ctx->stream,
                               (const int *)ctx->ws_sh_partial.p, tiles, (int)groups, ldn, digits, (const int64_t *)d_seg_off, s_base,
                               s_pass, p->n_acc, j.chunk, j.certify ? (const int *)d_nonint : (const int *)nullptr,
                               ctx->debug_reeval, j.d_score, j.d_ninfo, j.ldo, seg_pairs(ctx), seg_pair_count(ctx), j.cap);
            HIPCHK(ctx, hipGetLastError());
        }
        ++st.passes;
    }
    st.taken = 1;
    st.d_meta = d_meta;
    return SNPM_OK;
}
