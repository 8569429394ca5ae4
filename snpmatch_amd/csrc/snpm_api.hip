// snpm_api.hip -- C ABI of libsnpmatch_hip.so (see include/snpmatch_hip.h for the contract and the
// reference interfaces each entry point replaces).  gfx950 only; no CPU fallback lives here: every
// compute entry point launches HIP kernels and fails with SNPM_ERR_HIP when no device is usable.
#include "snpmatch_hip.h"

#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "snpm_kernels.hpp"

using namespace snpm;

// ------------------------------------------------------------------------------------------------
namespace {

thread_local std::string g_init_error;

enum ProfKind { PK_FAST = 0, PK_STRICT, PK_REDUCE, PK_SCAN, PK_LIK, PK_SYNTH, PK_LUT, PK_COUNT };
const char *kProfNames[PK_COUNT] = {"fast", "strict", "reduce", "scan", "likelihood", "synth", "lut"};

struct Buf {
    void *p = nullptr;
    size_t cap = 0;
};

}  // namespace

struct snpm_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;       // stream kernels are launched on (own or caller-provided)
    hipStream_t copy_stream = nullptr;  // H2D staging side stream
    std::string err;
    int n_cu = 256;
    // pinned staging (double-buffered)
    static constexpr size_t kStageBytes = 32u << 20;
    void *stage[2] = {nullptr, nullptr};
    hipEvent_t stage_done[2] = {nullptr, nullptr};
    hipEvent_t compute_mark = nullptr;  // "everything queued on the compute stream so far": uploads wait for it
    bool stage_busy[2] = {false, false};
    // pinned host buffer for small result readbacks (exactness check)
    void *h_pinned = nullptr;
    size_t h_pinned_cap = 0;
    // grow-only device workspaces
    Buf ws_grp_score, ws_grp_miss, ws_stage_dev, ws_flags2;
    Buf ws_part_score, ws_part_miss, ws_seg_score, ws_seg_miss, ws_seg_off, ws_cols, ws_tmp_score, ws_tmp_ninfo, ws_flags;
    Buf ws_lik_y, ws_lik_n, ws_lik_l, ws_lik_r;
    Buf ws_wprops, ws_epart;            // partial sums of k_wprops / k_eref
    Buf ws_once;                        // packed results of snpm_genotype_once
    Buf ws_seg_desc, ws_eseg, ws_pairs, ws_pair_sums, ws_bscore, ws_bninfo, ws_blut, ws_brows, ws_brows32, ws_bw, ws_bcodes;   // segmented / batched scoring
    int64_t *h_desc = nullptr;          // pinned host image of ws_seg_desc
    size_t h_desc_cap = 0;
    int stage_which = 0;                // next staging slab of stage_bytes
    // panel loader (snpm_loader.hpp): its own pinned slabs, filled by a persistent pool of host threads
    static constexpr int kLdStages = 3;
    void *ld_stage[kLdStages] = {nullptr, nullptr, nullptr};
    hipEvent_t ld_done[kLdStages] = {nullptr, nullptr, nullptr};
    bool ld_busy[kLdStages] = {false, false, false};
    int ld_next = 0;
    size_t ld_cap = 0;                  // bytes per loader slab (>= one staged row)
    size_t ld_want = size_t(64) << 20;  // SNPM_STAGE_MB
    void *pool = nullptr;               // HostPool
    int host_pack = 1;                  // SNPM_HOST_PACK=0: packed panels cross PCIe as int8 and are packed on the device
    int ld_avx2 = 1, ld_nt = 1;         // SNPM_NO_AVX2 / SNPM_NO_NT: scalar packer, plain memcpy into the slabs (the AVX2 forms need the CPU to have it)
    int odirect = -1;                   // SNPM_ODIRECT: 1 always, 0 never, -1 (default) contiguous reads of >= 1 GiB
    snpm_panel *last_touched = nullptr; // the panel the compute stream's most recent work reads or writes
    hipEvent_t batch_ev = nullptr;      // "this sub-batch's inputs have arrived" (copy stream -> compute stream)
    // device buffers of freed queries, kept for the next query (hipMalloc / hipFree cost more than a small query's run)
    struct Cached { void *p; size_t cap; };
    std::vector<Cached> qcache;
    // profiling
    bool prof_on = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    std::vector<std::pair<size_t, size_t>> prof_pairs[PK_COUNT];
    // tunables (environment)
    int force_bpl = 0;
    int force_wpb = 0;
    int seg_blocks_per_cu = 0;          // SNPM_SEG_BLOCKS_PER_CU: parts of a segmented pass per CU and column block (0: 32 on int8 panels, 8 on packed ones)
    int part_min_tiles = 8;             // SNPM_PART_MIN_TILES: tiles a part keeps when k_fast_packed_q4 takes more parts than resident blocks
    int q4_tile_rows = 0;               // SNPM_Q4_TILE_ROWS: rows per LDS tile of k_fast_packed_q4 (0: by block size)
    int parts_mult = 1;
    int use_acc_major = 1;  // keep an accession-major packed copy (+25 % memory) for contiguous re-evaluation reads
    int64_t acc_major_min_rows = 1000000;   // ... once a query is long enough for the strided path to hurt
    int strict4 = 1;           // dense strict scoring with 4 columns per lane (SNPM_STRICT4=0: one column per lane)
    int debug_max_parts = 0;   // SNPM_DEBUG_MAX_PARTS=k: cap the parts of the fast pass (tests of multi-epoch parts)
    int packed_split = 1;   // SNPM_PACKED_SPLIT=0: packed panels keep whole rows at a 256-B pitch (round 3's layout)
    int debug_reeval = 0;   // SNPM_DEBUG_REEVAL=k: also re-evaluate accessions 0..k-1 (to time that path)
    int stage_threads = 8;  // host threads filling the pinned staging slabs (default: the cores of this process, at most 16)
    int nt_loads = 1;      // panel bytes are read once: non-temporal loads (+5-8% measured)
    int64_t f1_slab_bytes = int64_t(2) << 30;   // SNPM_F1_SLAB_BYTES: compacted-weight scratch of the in-silico crosses
    int occ_cap = 0;          // SNPM_OCC_CAP=n: at most n resident blocks per CU in the fast pass (experiments)
    int full_occupancy = 0;   // SNPM_FULL_OCCUPANCY=1: as many resident blocks as the occupancy API allows
    int bits_path = 1;     // SNPM_BITS=0: hard-call samples on packed panels go through k_fast_packed_q4 like any other
    int64_t pitch_align = 256;          // SNPM_PITCH_ALIGN: bytes a panel row is padded to (a multiple of 64; experiments)
    bool pitch_align_forced = false;    // set by SNPM_PITCH_ALIGN: no per-width choice
    int64_t long_scan_rows = 2000000;   // SNPM_LONG_SCAN_ROWS: queries of at least this many rows walk LONG_TILE_ROWS-row tiles (int8 fast pass); -1: never
    // live panels of this context: snpm_destroy releases their device memory and orphans them (and their
    // queries), so that a panel / query handle freed AFTER its context is a harmless host-side delete
    std::vector<snpm_panel *> panels;
    std::vector<snpm_carry *> carries;
    std::vector<snpm_group *> groups;   // groups this context joined as a rank (snpm_group_create_rank): told when the context goes away
};

struct snpm_panel {
    snpm_ctx *ctx = nullptr;
    int64_t n_snp = 0, n_acc = 0;
    int64_t n_acc_total = 0;            // accessions of the panel the REFERENCE would see (= n_acc unless this is one shard of a wider
                                        // panel, snpm_panel_set_total_accessions): 1 selects numpy's vector summation (k_strict_single)
    int64_t pitch = 0;                  // bytes per SNP row (int8: >= n_acc; packed: >= n_acc / 4), multiple of ctx->pitch_align (256)
                                        // -- or, for a SPLIT packed panel, main part + tail part (e.g. 256 + 32): what a row costs in HBM
    int64_t kpitch = 0;                 // what the kernels stride rows by: pitch, or the main part's pitch (a multiple of 256, may be 0)
    int64_t tail_pitch = 0, tail_off = 0;   // split packed panels: bytes per row of the tail matrix / its offset from d (else 0)
    int64_t desc = 0;                   // the kernels' layout descriptor (snpm_k_common.hpp; 0 for int8 panels)
    int64_t ld = 0;                     // accessions per row rounded up to 256: leading dimension of result arrays
    int packed = 0;                     // 0 = int8 (one byte per call), 1 = 2 bits per call (4 accessions per byte)
    int8_t *d = nullptr;
    int *d_other = nullptr;             // int8 panels: 1 once an upload stored a call code > 2 ("other": informative, matches no
                                        // class), else 0; lives behind the rows in the same allocation.  k_strict4 reads it.
    hipEvent_t uploaded = nullptr;      // last upload / fill enqueued on copy_stream
    bool upload_pending = false;
    hipEvent_t used = nullptr;          // recorded on the compute stream when it moved on to another panel: everything that
    bool used_valid = false;            // reads this panel is complete once it fires (uploads into the panel wait for it)
    // accession-major packed copy (2 bits per call), built on first use by the exactness re-evaluation
    uint8_t *dT = nullptr;
    int64_t pitchT = 0;
    int dT_state = 0;                   // 0 = not built / stale, 1 = valid, -1 = unusable (code 3 present or no memory)
    std::vector<snpm_query *> queries;  // live queries against this panel (orphaned when the panel goes away)
};

constexpr int REEVAL_CAP = 64;          // flagged accessions the sparse re-evaluation tier takes; more -> dense tier

struct snpm_query {
    snpm_panel *panel = nullptr;
    int64_t n = 0, row0 = 0;
    int64_t *d_row_idx = nullptr;       // NULL = dense
    double *d_w = nullptr;              // [n,3]
    double *d_lut = nullptr;            // [n,4]
    int lut_skip = -1;                  // which skip_hets variant d_lut currently holds
    double *d_score = nullptr;          // results: own buffers [pitch] or caller-bound [n_acc]
    int64_t *d_ninfo = nullptr;
    double *own_score = nullptr;
    int64_t *own_ninfo = nullptr;
    double wsum = 0;                    // sum over rows of max_c |W[r,c]| (k_wprops)
    bool all_integer = false;
    bool hard01 = false;                // every weight is 0 or 1: scores are counts (k_fast_bits on packed panels)
    uint8_t *d_wbits = nullptr;         // [n + pad] ref | het << 1 | alt << 2 per matched row (only when hard01)
    // certificate state, all on the device: one small block {double eref; int count; int pad; int32 cols[REEVAL_CAP]}
    void *d_cert = nullptr;
    int64_t eref_chunk = -1, eref_after = -1;   // what d_cert->eref currently holds
    bool count_valid = false;           // the last run was a certified one (count / cols are meaningful)
    std::vector<snpm_ctx::Cached> owned;   // every device buffer of this query with its capacity
    const char *last_kernel = "";       // scoring kernel of the last run (reports)
    int reeval_path = 0;                // sparse re-evaluation reads: 1 = accession-major copy, 2 = SNP-major (strided)
    bool transient_panel = false;       // slab-streamed scoring: never build a transposed copy of a transient slab
    double *cert_eref() const { return (double *)d_cert; }
    int *cert_count() const { return (int *)((char *)d_cert + 8); }
    int32_t *cert_cols() const { return (int32_t *)((char *)d_cert + 16); }
};

// running totals of a job scored SNP slab after SNP slab (snpm_query_run_carry)
struct snpm_carry {
    snpm_ctx *ctx = nullptr;
    int64_t n_acc = 0, ld = 0;
    double *d_score = nullptr;          // [ld] totals so far (fast-pass totals, or the reference's chain in strict mode)
    int64_t *d_ninfo = nullptr;         // [ld]
    double *d_E = nullptr;              // [0] sum of the slabs' error bounds (device)
    int32_t *d_cols = nullptr;          // column-list mode (second pass): [n_cols] accessions
    int *d_ncols = nullptr;
    int64_t n_cols = -1;                // -1: all accessions
    int64_t n_rows = 0, n_slabs = 0;
    long double wsum = 0;               // of all slabs (bound of the slab-total additions)
    int mode = -1;                      // mode of the first slab; later slabs must agree
    bool all_integer = true;            // every slab so far had integer weights only (then any summation order is exact)
    bool finished = false;
    double *own_score = nullptr;        // d_score / d_ninfo point here unless the caller bound its own buffers
    int64_t *own_ninfo = nullptr;
    int64_t len = 0;                    // entries of d_score / d_ninfo (ld for own buffers, n_acc for bound ones)
};

// ------------------------------------------------------------------------------------------------
namespace {

int set_err(snpm_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    else
        g_init_error = buf;
    return code;
}

#define HIPCHK(ctx, expr)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return set_err((ctx), (e_ == hipErrorOutOfMemory) ? SNPM_ERR_OOM : SNPM_ERR_HIP,           \
                           "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define CHECK_ARG(ctx, cond, msg)                                   \
    do {                                                            \
        if (!(cond)) return set_err((ctx), SNPM_ERR_BADARG, "%s", msg); \
    } while (0)


// Set by an atexit handler registered at the first snpm_init, i.e. after the HIP runtime registered its own
// teardown: exit handlers run in reverse order of registration, so the flag is up before the runtime's static
// objects go away.  From then on free / destroy only drop host-side bookkeeping (the process is exiting and
// the driver reclaims device memory); no HIP call is made on a runtime that may be half torn down.
std::atomic<bool> g_exiting{false};
void mark_exiting() { g_exiting.store(true); }

bool hip_alive() { return !g_exiting.load(); }

void group_forget_ctx(snpm_group *g, snpm_ctx *ctx, bool use_hip);      // snpm_group.hpp

// an entry point that allocates host memory (std::vector, std::string, new) ends with SNPM_GUARD(ctx): no C++
// exception crosses the C ABI (ctypes would call std::terminate)
#define SNPM_GUARD(CTX)                                                                              \
    catch (const std::bad_alloc &) { return set_err((CTX), SNPM_ERR_OOM, "out of host memory"); }    \
    catch (const std::exception &e_) { return set_err((CTX), SNPM_ERR_STATE, "internal error: %s", e_.what()); } \
    catch (...) { return set_err((CTX), SNPM_ERR_STATE, "internal error"); }

// handles whose context (panel: or panel, query) is gone: every entry point except the matching free refuses them
#define CHECK_PANEL(P)                                                                               \
    do {                                                                                             \
        if (!(P)) return set_err(nullptr, SNPM_ERR_BADARG, "panel is NULL");                         \
        if (!(P)->ctx) return set_err(nullptr, SNPM_ERR_STATE, "panel outlived its context (snpm_destroy was called)"); \
    } while (0)
#define CHECK_QUERY(Q)                                                                               \
    do {                                                                                             \
        if (!(Q)) return set_err(nullptr, SNPM_ERR_BADARG, "query is NULL");                         \
        if (!(Q)->panel || !(Q)->panel->ctx)                                                         \
            return set_err(nullptr, SNPM_ERR_STATE, "query outlived its panel or context");          \
    } while (0)

#define CHECK_CARRY(C)                                                                               \
    do {                                                                                             \
        if (!(C)) return set_err(nullptr, SNPM_ERR_BADARG, "carry is NULL");                         \
        if (!(C)->ctx) return set_err(nullptr, SNPM_ERR_STATE, "carry outlived its context");        \
    } while (0)

int ensure(snpm_ctx *ctx, Buf &b, size_t bytes)
{
    if (bytes <= b.cap && b.p) return SNPM_OK;
    if (b.p) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = std::max<size_t>(bytes, 256);
    HIPCHK(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return SNPM_OK;
}

// Query buffers come from / go back to a small per-context cache.  Everything that touches them is ordered on
// ctx->stream, so a buffer can be handed to the next query without a synchronisation.
constexpr size_t kQueryCacheEntries = 24;
constexpr size_t kQueryCacheMaxBytes = size_t(64) << 20;

hipError_t query_alloc(snpm_query *q, void **out, size_t bytes)
{
    snpm_ctx *ctx = q->panel->ctx;
    bytes = std::max<size_t>(bytes, 256);
    size_t best = ctx->qcache.size();
    for (size_t i = 0; i < ctx->qcache.size(); ++i)
        if (ctx->qcache[i].cap >= bytes && ctx->qcache[i].cap <= 2 * bytes &&
            (best == ctx->qcache.size() || ctx->qcache[i].cap < ctx->qcache[best].cap))
            best = i;
    if (best < ctx->qcache.size()) {
        q->owned.push_back(ctx->qcache[best]);
        *out = ctx->qcache[best].p;
        ctx->qcache.erase(ctx->qcache.begin() + (long)best);
        return hipSuccess;
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipSuccess) q->owned.push_back({*out, bytes});
    return e;
}

void query_release(snpm_query *q, void *ptr)
{
    if (!ptr) return;
    snpm_ctx *ctx = q->panel->ctx;
    for (size_t i = 0; i < q->owned.size(); ++i) {
        if (q->owned[i].p != ptr) continue;
        const snpm_ctx::Cached c = q->owned[i];
        q->owned.erase(q->owned.begin() + (long)i);
        if (c.cap <= kQueryCacheMaxBytes && ctx->qcache.size() < kQueryCacheEntries) {
            ctx->qcache.push_back(c);
        } else {
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipFree(c.p);
        }
        return;
    }
}

struct ProfScope {
    snpm_ctx *ctx;
    int kind;
    size_t a = 0, b = 0;
    bool on = false;
    ProfScope(snpm_ctx *c, int k) : ctx(c), kind(k)
    {
        if (!ctx->prof_on) return;
        if (ctx->ev_used + 2 > ctx->ev_pool.size()) {
            for (int i = 0; i < 64; ++i) {
                hipEvent_t e;
                if (hipEventCreate(&e) != hipSuccess) return;
                ctx->ev_pool.push_back(e);
            }
        }
        a = ctx->ev_used++;
        b = ctx->ev_used++;
        on = true;
        (void)hipEventRecord(ctx->ev_pool[a], ctx->stream);
    }
    ~ProfScope()
    {
        if (!on) return;
        (void)hipEventRecord(ctx->ev_pool[b], ctx->stream);
        ctx->prof_pairs[kind].push_back({a, b});
    }
};

int wait_upload(snpm_panel *p)
{
    // the compute stream is about to read (or fill) this panel.  If its previous work belonged to ANOTHER panel, mark the end of
    // that work now: an upload into that panel then waits for this event instead of for everything queued later (stage_rows)
    snpm_ctx *ctx = p->ctx;
    if (ctx->last_touched && ctx->last_touched != p) {
        snpm_panel *o = ctx->last_touched;
        if (!o->used) HIPCHK(ctx, hipEventCreateWithFlags(&o->used, hipEventDisableTiming));
        HIPCHK(ctx, hipEventRecord(o->used, ctx->stream));
        o->used_valid = true;
    }
    ctx->last_touched = p;
    // make the compute stream wait for any pending staging copies into this panel
    if (p->upload_pending) {
        HIPCHK(p->ctx, hipStreamWaitEvent(p->ctx->stream, p->uploaded, 0));
        p->upload_pending = false;      // the compute stream is ordered after the upload from here on
    }                                   // (snpm_set_stream drains the copy stream before switching streams)
    return SNPM_OK;
}

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
                           q->n, q->d_lut, (double *)ctx->ws_part_score.p, (uint32_t *)ctx->ws_part_miss.p, p->ld);
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
    FastGeom g = fast_geom(ctx, p->n_acc, q->n, occ, bpl, tile_rows, wpb_fixed, kmult);
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
    HIPCHK(ctx, hipMemsetAsync(q->cert_count(), 0, sizeof(int), ctx->stream));
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
        if (n_groups > 0) {
            hipLaunchKernelGGL(k_reduce_groups, dim3(cb, (unsigned)n_groups), dim3(thr), 0, ctx->stream,
                               (const double *)ctx->ws_part_score.p, (const uint32_t *)ctx->ws_part_miss.p, g.n_slots,
                               p->ld, p->n_acc, (double *)ctx->ws_grp_score.p, (uint32_t *)ctx->ws_grp_miss.p);
            HIPCHK(ctx, hipGetLastError());
        }
        hipLaunchKernelGGL(k_reduce, dim3(cb), dim3(thr), 0, ctx->stream, (const double *)ctx->ws_grp_score.p,
                           (const uint32_t *)ctx->ws_grp_miss.p, n_groups, p->ld, p->n_acc, q->n, q->d_score,
                           q->d_ninfo, (certify && cert.flag) ? (const double *)q->cert_eref() : (const double *)nullptr,
                           certify ? efast_bound(q, g) : 0.0, ctx->debug_reeval, q->cert_cols(), q->cert_count(),
                           REEVAL_CAP);
        HIPCHK(ctx, hipGetLastError());
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

// ---- reference-order (strict) launches ---------------------------------------------------------------------
// Dense: every accession, segments [seg0, seg0 + n_seg) -> ctx->ws_seg_* [n_seg, ld].  Segments are explicit
// (d_seg_off: windows) or `chunk`-row pieces of the query.  gate (may be NULL): device count; the launch is a
// no-op unless *gate > REEVAL_CAP.
// A panel of ONE accession: the reference's per-call sums are numpy's vector sums (snpm_kernels_single.hpp)
static inline bool single_accession(const snpm_panel *p) { return p->n_acc_total == 1; }

// k_strict_single in place of the strict kernel of a site; tier / pairs / count as in the kernel's header
static int launch_strict_single(snpm_ctx *ctx, const snpm_panel *p, const int64_t *d_row_idx, int64_t row0, const double *d_w,
                                int skip, const int64_t *seg_off, int64_t chunk, int64_t n, int64_t seg0, int64_t n_seg,
                                const int32_t *pairs, const int *count, int cap, int tier, int64_t kmax, dim3 grid,
                                double *out_score, uint32_t *out_miss, int64_t ld)
{
    const bool gather = d_row_idx != nullptr;
#define LAUNCH_SINGLE(S, G)                                                                                          \
    hipLaunchKernelGGL((k_strict_single<S, G>), grid, dim3(SINGLE_THREADS), 0, ctx->stream, p->d, p->kpitch, p->desc,  \
                       d_row_idx, row0, d_w, seg_off, chunk, n, seg0, n_seg, pairs, count, cap, tier, kmax, out_score,  \
                       out_miss, ld)
    if (skip) {
        if (gather) LAUNCH_SINGLE(true, true); else LAUNCH_SINGLE(true, false);
    } else {
        if (gather) LAUNCH_SINGLE(false, true); else LAUNCH_SINGLE(false, false);
    }
#undef LAUNCH_SINGLE
    HIPCHK(ctx, hipGetLastError());
    return SNPM_OK;
}

int launch_strict_dense(snpm_query *q, int skip, const int64_t *d_seg_off, int64_t chunk, int64_t seg0, int64_t n_seg,
                        const int *gate)
{
    snpm_ctx *ctx = q->panel->ctx;
    snpm_panel *p = q->panel;
    const int64_t ncols = p->n_acc, ld = p->ld;
    const bool gather = q->d_row_idx != nullptr;
    const int64_t *seg_off = d_seg_off ? d_seg_off + seg0 : nullptr;
    if (n_seg == 0) return SNPM_OK;
    if (single_accession(p)) {
        ProfScope ps(ctx, PK_STRICT);
        return launch_strict_single(ctx, p, q->d_row_idx, q->row0, q->d_w, skip, seg_off, chunk, q->n, seg0, n_seg, nullptr, gate,
                                    REEVAL_CAP, gate ? 2 : 0, 0, dim3((unsigned)std::min<int64_t>(n_seg, 65535)),
                                    (double *)ctx->ws_seg_score.p, (uint32_t *)ctx->ws_seg_miss.p, ld);
    }
    if (ctx->strict4) {
        // 4 columns per lane (one dword of an int8 panel, one byte of a packed panel)
        const int64_t lanes = (ncols + 3) / 4;
        const int t4 = lanes >= 256 ? 256 : (lanes > 64 ? 128 : 64);
        // a gated launch (the certificate's dense tier) usually has nothing to do: a bounded grid that walks the segments
        dim3 grid4((unsigned)(gate ? std::min<int64_t>(n_seg, 2048) : n_seg), (unsigned)((lanes + t4 - 1) / t4));
        ProfScope ps(ctx, PK_STRICT);
#define LAUNCH_STRICT4(S, G)                                                                                     \
    do {                                                                                                         \
        if (p->packed)                                                                                           \
            hipLaunchKernelGGL((k_strict4<S, G, true>), grid4, dim3(t4), 0, ctx->stream, p->d, p->kpitch, q->d_row_idx, q->row0, \
                               q->d_w, seg_off, chunk, q->n, seg0, n_seg, ncols, (double *)ctx->ws_seg_score.p,  \
                               (uint32_t *)ctx->ws_seg_miss.p, ld, (const int *)nullptr, gate, REEVAL_CAP, p->desc); \
        else                                                                                                     \
            hipLaunchKernelGGL((k_strict4<S, G, false>), grid4, dim3(t4), 0, ctx->stream, p->d, p->pitch, q->d_row_idx, q->row0, \
                               q->d_w, seg_off, chunk, q->n, seg0, n_seg, ncols, (double *)ctx->ws_seg_score.p,  \
                               (uint32_t *)ctx->ws_seg_miss.p, ld, (const int *)p->d_other, gate, REEVAL_CAP);   \
    } while (0)
        if (skip) {
            if (gather) LAUNCH_STRICT4(true, true); else LAUNCH_STRICT4(true, false);
        } else {
            if (gather) LAUNCH_STRICT4(false, true); else LAUNCH_STRICT4(false, false);
        }
#undef LAUNCH_STRICT4
        HIPCHK(ctx, hipGetLastError());
        return SNPM_OK;
    }
    const int thr = ncols > 128 ? 256 : (ncols > 64 ? 128 : 64);
    dim3 grid((unsigned)(gate ? std::min<int64_t>(n_seg, 2048) : n_seg), (unsigned)((ncols + thr - 1) / thr));
    ProfScope ps(ctx, PK_STRICT);
#define LAUNCH_STRICT(S, G)                                                                                       \
    hipLaunchKernelGGL((k_strict<S, G>), grid, dim3(thr), 0, ctx->stream, p->d, p->kpitch, p->desc, q->d_row_idx, q->row0,  \
                       q->d_w, seg_off, chunk, q->n, seg0, n_seg, (const int32_t *)nullptr, ncols,                \
                       (double *)ctx->ws_seg_score.p, (uint32_t *)ctx->ws_seg_miss.p, ld, gate, REEVAL_CAP)
    if (skip) {
        if (gather) LAUNCH_STRICT(true, true); else LAUNCH_STRICT(true, false);
    } else {
        if (gather) LAUNCH_STRICT(false, true); else LAUNCH_STRICT(false, false);
    }
#undef LAUNCH_STRICT
    HIPCHK(ctx, hipGetLastError());
    return SNPM_OK;
}

// The reference's whole chunk loop for every accession: strict chunk sums in groups of segments (bounded
// workspace), each group's sums added onto the running totals in order (ScoreList += chunk, core/snpmatch.py:224).
// carry_*: totals of earlier SNP slabs (or NULL).  gate as above.
constexpr size_t kStrictGroupBytes = size_t(512) << 20;

int run_strict_chain(snpm_query *q, int skip, int64_t chunk, const int *gate, const double *carry_score,
                     const int64_t *carry_ninfo, double *dst_score, int64_t *dst_ninfo)
{
    snpm_ctx *ctx = q->panel->ctx;
    snpm_panel *p = q->panel;
    const int64_t n_seg = (q->n + chunk - 1) / chunk;
    const int64_t per_seg = p->ld * (int64_t)(sizeof(double) + sizeof(uint32_t));
    const int64_t group = std::max<int64_t>(1, std::min<int64_t>(std::max<int64_t>(n_seg, 1), (int64_t)kStrictGroupBytes / per_seg));
    int rc = ensure(ctx, ctx->ws_seg_score, (size_t)group * p->ld * sizeof(double));
    if (rc) return rc;
    rc = ensure(ctx, ctx->ws_seg_miss, (size_t)group * p->ld * sizeof(uint32_t));
    if (rc) return rc;
    const int thr = 256;
    const unsigned cb = (unsigned)((p->n_acc + thr - 1) / thr);
    bool first = true;
    for (int64_t g0 = 0; g0 < n_seg || first; g0 += group) {
        const int64_t ns = std::max<int64_t>(0, std::min<int64_t>(group, n_seg - g0));
        rc = launch_strict_dense(q, skip, nullptr, chunk, g0, ns, gate);
        if (rc) return rc;
        const int64_t rows = std::min<int64_t>(q->n, (g0 + ns) * chunk) - std::min<int64_t>(q->n, g0 * chunk);
        ProfScope ps(ctx, PK_SCAN);
        hipLaunchKernelGGL(k_scan, dim3(cb), dim3(thr), 0, ctx->stream, (const double *)ctx->ws_seg_score.p,
                           (const uint32_t *)ctx->ws_seg_miss.p, rows, ns, p->ld, p->n_acc, dst_score, dst_ninfo,
                           first ? carry_score : (const double *)dst_score, first ? carry_ninfo : (const int64_t *)dst_ninfo,
                           gate, REEVAL_CAP);
        HIPCHK(ctx, hipGetLastError());
        first = false;
    }
    return SNPM_OK;
}

// Sparse tier: reference-order chunk sums of the accessions listed on the device (d_cols, *d_ncols <= REEVAL_CAP;
// the kernels do nothing for other counts) -> ws_seg_score [n_seg, REEVAL_CAP] -> chain of additions ->
// ws_tmp_score [REEVAL_CAP].  carry (may be NULL): compact totals of earlier slabs, continued by the chain.
int run_strict_sparse(snpm_query *q, int skip, int64_t chunk, const int32_t *d_cols, const int *d_ncols, const double *carry,
                      const int64_t *d_seg_off = nullptr, int64_t n_seg_explicit = 0)
{
    snpm_ctx *ctx = q->panel->ctx;
    snpm_panel *p = q->panel;
    const int64_t n_seg = d_seg_off ? n_seg_explicit : (q->n + chunk - 1) / chunk;
    const int64_t ld = REEVAL_CAP;
    int rc = ensure(ctx, ctx->ws_seg_score, (size_t)std::max<int64_t>(n_seg, 1) * ld * sizeof(double));
    if (rc) return rc;
    rc = ensure(ctx, ctx->ws_seg_miss, (size_t)std::max<int64_t>(n_seg, 1) * ld * sizeof(uint32_t));
    if (rc) return rc;
    rc = ensure(ctx, ctx->ws_tmp_score, (size_t)ld * sizeof(double));
    if (rc) return rc;
    const bool gather = q->d_row_idx != nullptr;
    const int64_t total = n_seg * ld;
    if (total > 0) {
        dim3 sgrid((unsigned)std::min<int64_t>((total + 255) / 256, (int64_t)ctx->n_cu * 8));     // grid-stride over (segment, column)
        const bool use_T = !q->transient_panel && q->n >= ctx->acc_major_min_rows && p->dT_state == 1;
        ProfScope ps(ctx, PK_STRICT);
        if (single_accession(p)) {
            // the only column that can be flagged is column 0; its segment sums land in slot 0 of the compact [n_seg, ld] rows
            q->reeval_path = 2;
            rc = launch_strict_single(ctx, p, q->d_row_idx, q->row0, q->d_w, skip, d_seg_off, chunk, q->n, 0, n_seg, nullptr, d_ncols,
                                      REEVAL_CAP, 1, 0, dim3((unsigned)std::min<int64_t>(n_seg, 65535)),
                                      (double *)ctx->ws_seg_score.p, (uint32_t *)ctx->ws_seg_miss.p, ld);
            if (rc) return rc;
        } else if (use_T) {
            q->reeval_path = 1;
#define LAUNCH_SPARSE_T(S, G)                                                                                      \
    hipLaunchKernelGGL((k_strict_sparse_T<S, G>), sgrid, dim3(256), 0, ctx->stream, p->dT, p->pitchT, q->d_row_idx, \
                       q->row0, q->d_w, d_seg_off, chunk, q->n, n_seg, d_cols, d_ncols, REEVAL_CAP, \
                       (double *)ctx->ws_seg_score.p, (uint32_t *)ctx->ws_seg_miss.p, ld)
            if (skip) {
                if (gather) LAUNCH_SPARSE_T(true, true); else LAUNCH_SPARSE_T(true, false);
            } else {
                if (gather) LAUNCH_SPARSE_T(false, true); else LAUNCH_SPARSE_T(false, false);
            }
#undef LAUNCH_SPARSE_T
        } else {
            q->reeval_path = 2;
#define LAUNCH_SPARSE(S, G)                                                                                        \
    hipLaunchKernelGGL((k_strict_sparse<S, G>), sgrid, dim3(256), 0, ctx->stream, p->d, p->kpitch, p->desc, q->d_row_idx, \
                       q->row0, q->d_w, d_seg_off, chunk, q->n, n_seg, d_cols, d_ncols, REEVAL_CAP, \
                       (double *)ctx->ws_seg_score.p, (uint32_t *)ctx->ws_seg_miss.p, ld)
            if (skip) {
                if (gather) LAUNCH_SPARSE(true, true); else LAUNCH_SPARSE(true, false);
            } else {
                if (gather) LAUNCH_SPARSE(false, true); else LAUNCH_SPARSE(false, false);
            }
#undef LAUNCH_SPARSE
        }
        HIPCHK(ctx, hipGetLastError());
    }
    ProfScope ps(ctx, PK_SCAN);
    hipLaunchKernelGGL(k_scan_few, dim3(1), dim3(256), 0, ctx->stream, (const double *)ctx->ws_seg_score.p, n_seg, ld,
                       d_ncols, REEVAL_CAP, (double *)ctx->ws_tmp_score.p, carry);
    HIPCHK(ctx, hipGetLastError());
    return SNPM_OK;
}

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
    const int64_t want_blocks = std::max<int64_t>(1, (int64_t)ctx->n_cu * per_cu / std::max<int64_t>(1, pl.g0.n_colblocks));
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
        dim3 grid((unsigned)std::min<int64_t>(std::max<int64_t>(j.kmax, 1), 256), (unsigned)std::min(j.cap, 512));   // a wave per (pair, chunk); both axes walk
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

}  // namespace

#include "snpm_loader.hpp"

// ================================================================================================
extern "C" {

int snpm_version(void) { return 100; }

// HIP version the library was built against, as hipcc's headers encode it (major * 10000000 + minor * 100000 + patch): the
// binding compares its major number with the HIP runtime it is about to share with PyTorch
int snpm_hip_build_version(void) { return HIP_VERSION; }

#ifndef SNPM_BUILD_ID
#define SNPM_BUILD_ID "unknown"
#endif
const char *snpm_build_id(void) { return SNPM_BUILD_ID; }

int snpm_device_count(int *count)
{
    if (!count) return set_err(nullptr, SNPM_ERR_BADARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return set_err(nullptr, SNPM_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    }
    *count = n;
    return SNPM_OK;
}

const char *snpm_last_error(const snpm_ctx *ctx) { return ctx ? ctx->err.c_str() : g_init_error.c_str(); }

int snpm_init(int device_id, snpm_ctx **out)
try {
    if (!out) return set_err(nullptr, SNPM_ERR_BADARG, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return set_err(nullptr, SNPM_ERR_HIP, "no usable HIP device (hipGetDeviceCount: %s, count %d)",
                       hipGetErrorString(e), n);
    if (device_id < 0 || device_id >= n)
        return set_err(nullptr, SNPM_ERR_BADARG, "device_id %d out of range (0..%d)", device_id, n - 1);
    HIPCHK(nullptr, hipSetDevice(device_id));
    static std::atomic<bool> exit_hook{false};
    if (!exit_hook.exchange(true)) std::atexit(mark_exiting);      // after the HIP runtime's own registrations
    snpm_ctx *ctx = new snpm_ctx();
    ctx->device = device_id;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return set_err(nullptr, SNPM_ERR_HIP, "hipStreamCreate failed");
    }
    ctx->stream = ctx->own_stream;
    if (const char *s = getenv("SNPM_FORCE_BPL")) ctx->force_bpl = atoi(s);
    if (const char *s = getenv("SNPM_PARTS_MULT")) ctx->parts_mult = std::max(1, atoi(s));
    if (const char *s = getenv("SNPM_FORCE_WPB")) ctx->force_wpb = atoi(s);
    if (const char *s = getenv("SNPM_Q4_TILE_ROWS")) ctx->q4_tile_rows = atoi(s);
    if (const char *s = getenv("SNPM_PART_MIN_TILES")) ctx->part_min_tiles = std::max(1, atoi(s));
    if (const char *s = getenv("SNPM_SEG_BLOCKS_PER_CU")) ctx->seg_blocks_per_cu = std::max(1, atoi(s));
    if (const char *s = getenv("SNPM_NT")) ctx->nt_loads = atoi(s);
    if (const char *s = getenv("SNPM_BITS")) ctx->bits_path = atoi(s);
    if (const char *s = getenv("SNPM_PITCH_ALIGN")) {
        ctx->pitch_align = std::max<int64_t>(64, (atoll(s) + 63) / 64 * 64);
        ctx->pitch_align_forced = true;
    }
    if (const char *s = getenv("SNPM_LONG_SCAN_ROWS")) ctx->long_scan_rows = atoll(s) < 0 ? INT64_MAX : atoll(s);
    if (const char *s = getenv("SNPM_FULL_OCCUPANCY")) ctx->full_occupancy = atoi(s);
    if (const char *s = getenv("SNPM_OCC_CAP")) ctx->occ_cap = atoi(s);
    if (const char *s = getenv("SNPM_F1_SLAB_BYTES")) ctx->f1_slab_bytes = std::max<int64_t>(1, atoll(s));
    if (const char *s = getenv("SNPM_ACC_MAJOR")) ctx->use_acc_major = atoi(s);
    if (const char *s = getenv("SNPM_ACC_MAJOR_MIN_ROWS")) ctx->acc_major_min_rows = atoll(s);
    if (const char *s = getenv("SNPM_DEBUG_REEVAL")) ctx->debug_reeval = atoi(s);
    if (const char *s = getenv("SNPM_PACKED_SPLIT")) ctx->packed_split = atoi(s) != 0;
    if (const char *s = getenv("SNPM_STRICT4")) ctx->strict4 = atoi(s);
    if (const char *s = getenv("SNPM_DEBUG_MAX_PARTS")) ctx->debug_max_parts = atoi(s);
    ctx->stage_threads = default_stage_threads();
    if (const char *s = getenv("SNPM_STAGE_THREADS")) ctx->stage_threads = std::max(1, atoi(s));
    if (const char *s = getenv("SNPM_STAGE_MB")) ctx->ld_want = (size_t)std::max(1, atoi(s)) << 20;
    if (const char *s = getenv("SNPM_HOST_PACK")) ctx->host_pack = atoi(s);
    if (const char *s = getenv("SNPM_ODIRECT")) ctx->odirect = atoi(s);
    if (getenv("SNPM_NO_AVX2")) ctx->ld_avx2 = 0;
    ctx->ld_nt = (cpu_has_avx2() && ctx->ld_avx2 && !getenv("SNPM_NO_NT")) ? 1 : 0;
    *out = ctx;
    return SNPM_OK;
} SNPM_GUARD(nullptr)

// device buffers of a query go back to the context (cache) or to the driver; the handle stays valid for its free
static void orphan_query(snpm_query *q, bool use_hip)
{
    snpm_ctx *ctx = q->panel ? q->panel->ctx : nullptr;
    if (use_hip && ctx) {
        while (!q->owned.empty()) query_release(q, q->owned.back().p);
    } else {
        q->owned.clear();
    }
    q->d_row_idx = nullptr; q->d_w = nullptr; q->d_lut = nullptr; q->d_score = nullptr; q->d_ninfo = nullptr;
    q->own_score = nullptr; q->own_ninfo = nullptr; q->d_wbits = nullptr; q->d_cert = nullptr;
    q->panel = nullptr;
}

// device memory of a panel is released, its queries are orphaned; the handle stays valid for snpm_panel_free
static void orphan_panel(snpm_panel *p, bool use_hip)
{
    for (snpm_query *q : p->queries) orphan_query(q, use_hip);
    p->queries.clear();
    if (use_hip) {
        if (p->d) (void)hipFree(p->d);
        if (p->dT) (void)hipFree(p->dT);
        if (p->uploaded) (void)hipEventDestroy(p->uploaded);
        if (p->used) (void)hipEventDestroy(p->used);
    }
    if (p->ctx && p->ctx->last_touched == p) p->ctx->last_touched = nullptr;
    p->used = nullptr;
    p->used_valid = false;
    p->d = nullptr;
    p->d_other = nullptr;
    p->dT = nullptr;
    p->uploaded = nullptr;
    p->ctx = nullptr;
}

int snpm_destroy(snpm_ctx *ctx)
{
    if (!ctx) return SNPM_OK;
    const bool use_hip = hip_alive();
    if (use_hip) {
        (void)hipSetDevice(ctx->device);
        (void)hipDeviceSynchronize();
    }
    // a group this context joined as a rank loses its member now (communicator and buffers released); its handle stays valid
    for (snpm_group *g : ctx->groups) group_forget_ctx(g, ctx, use_hip);
    ctx->groups.clear();
    // panels and queries created from this context may be freed later (or never): they become orphans now
    for (snpm_panel *p : ctx->panels) orphan_panel(p, use_hip);
    ctx->panels.clear();
    for (snpm_carry *c : ctx->carries) {
        if (use_hip) {
            (void)hipFree(c->own_score);
            (void)hipFree(c->own_ninfo);
            (void)hipFree(c->d_E);
        }
        c->own_score = nullptr; c->own_ninfo = nullptr;
        c->d_score = nullptr; c->d_ninfo = nullptr; c->d_E = nullptr; c->d_cols = nullptr; c->d_ncols = nullptr;
        c->ctx = nullptr;
    }
    ctx->carries.clear();
    if (use_hip) {
        for (auto &c : ctx->qcache) (void)hipFree(c.p);
        Buf *bufs[] = {&ctx->ws_stage_dev, &ctx->ws_flags2, &ctx->ws_grp_score, &ctx->ws_grp_miss, &ctx->ws_part_score, &ctx->ws_part_miss, &ctx->ws_seg_score, &ctx->ws_seg_miss, &ctx->ws_seg_off,
                       &ctx->ws_cols, &ctx->ws_tmp_score, &ctx->ws_tmp_ninfo, &ctx->ws_flags, &ctx->ws_lik_y,
                       &ctx->ws_lik_n, &ctx->ws_lik_l, &ctx->ws_lik_r, &ctx->ws_wprops, &ctx->ws_epart,
                       &ctx->ws_seg_desc, &ctx->ws_eseg, &ctx->ws_pairs, &ctx->ws_pair_sums, &ctx->ws_bscore, &ctx->ws_bninfo,
                       &ctx->ws_blut, &ctx->ws_brows, &ctx->ws_brows32, &ctx->ws_bw, &ctx->ws_bcodes};
        for (Buf *b : bufs)
            if (b->p) (void)hipFree(b->p);
        if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
        if (ctx->h_desc) (void)hipHostFree(ctx->h_desc);
        if (ctx->batch_ev) (void)hipEventDestroy(ctx->batch_ev);
        for (int i = 0; i < 2; ++i) {
            if (ctx->stage[i]) (void)hipHostFree(ctx->stage[i]);
            if (ctx->stage_done[i]) (void)hipEventDestroy(ctx->stage_done[i]);
        }
        for (int i = 0; i < snpm_ctx::kLdStages; ++i) {
            if (ctx->ld_stage[i]) (void)hipHostFree(ctx->ld_stage[i]);
            if (ctx->ld_done[i]) (void)hipEventDestroy(ctx->ld_done[i]);
        }
        for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
        if (ctx->compute_mark) (void)hipEventDestroy(ctx->compute_mark);
        if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
        if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    }
    ctx->qcache.clear();
    host_pool_destroy(ctx);
    delete ctx;
    return SNPM_OK;
}

int snpm_set_stream(snpm_ctx *ctx, void *hip_stream)
{
    if (!ctx) return SNPM_ERR_BADARG;
    // work queued on the old stream (and the cached query buffers it may still use) finishes before the switch
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));   // uploads in flight were only ordered against the old stream
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return SNPM_OK;
}

int snpm_device_mem_info(snpm_ctx *ctx, int64_t *free_bytes, int64_t *total_bytes)
{
    if (!ctx) return SNPM_ERR_BADARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    size_t f = 0, t = 0;
    HIPCHK(ctx, hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return SNPM_OK;
}

int snpm_synchronize(snpm_ctx *ctx)
{
    if (!ctx) return SNPM_ERR_BADARG;
    HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SNPM_OK;
}

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
static int enqueue_reevaluation(snpm_query *q, int skip, int64_t chunk)
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
    int rc = run_strict_sparse(q, skip, chunk, q->cert_cols(), q->cert_count(), nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(k_patch, dim3(1), dim3(REEVAL_CAP), 0, ctx->stream, (const double *)ctx->ws_tmp_score.p,
                       (const int32_t *)q->cert_cols(), (const int *)q->cert_count(), REEVAL_CAP, q->d_score);
    HIPCHK(ctx, hipGetLastError());
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
        if (tot_score) HIPCHK(ctx, hipMemcpyAsync(tot_score, q->d_score, na * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (tot_ninfo) {
            HIPCHK(ctx, hipMemcpyAsync(tot_ninfo, q->d_ninfo, na * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        }
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
        if (score)
            HIPCHK(ctx, hipMemcpyAsync(score, ctx->ws_tmp_score.p, (size_t)n_win * na * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (ninfo)
            HIPCHK(ctx, hipMemcpyAsync(ninfo, ctx->ws_tmp_ninfo.p, (size_t)n_win * na * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    // slab-streamed jobs do not wait here: the next slab is loaded while this one is scored; the per-window rows arrive in the
    // caller's (pinned) buffers by the time snpm_carry_finish / snpm_synchronize returns
    if (!carry) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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
    rc = run_strict_sparse(q, skip, maxlen, q->cert_cols(), q->cert_count(), nullptr, j.d_seg_off, n_win);
    if (rc) return rc;
    hipLaunchKernelGGL(k_patch, dim3(1), dim3(REEVAL_CAP), 0, ctx->stream, (const double *)ctx->ws_tmp_score.p,
                       (const int32_t *)q->cert_cols(), (const int *)q->cert_count(), REEVAL_CAP, q->d_score);
    HIPCHK(ctx, hipGetLastError());
    rc = ensure_pinned(ctx, 64);
    if (rc) return rc;
    int *h_cnt = (int *)ctx->h_pinned;
    HIPCHK(ctx, hipMemcpyAsync(h_cnt, seg_pair_count(ctx), sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h_cnt + 1, q->cert_count(), sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (score) HIPCHK(ctx, hipMemcpyAsync(score, ctx->ws_bscore.p, (size_t)n_win * na * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (ninfo) HIPCHK(ctx, hipMemcpyAsync(ninfo, ctx->ws_bninfo.p, (size_t)n_win * na * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    if (tot_score) HIPCHK(ctx, hipMemcpyAsync(tot_score, q->d_score, na * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (tot_ninfo) HIPCHK(ctx, hipMemcpyAsync(tot_ninfo, q->d_ninfo, na * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const int n_pairs = h_cnt[0], n_tot = h_cnt[1];
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
                            int mode, double *score, int64_t *ninfo, double *lik, double *lrt, int64_t *info)
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
    if ((rc = ensure(ctx, ctx->ws_bscore, B * na * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, ctx->ws_bninfo, B * na * sizeof(int64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->ws_flags2, sizeof(int)))) return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->ws_flags2.p, 0, sizeof(int), ctx->stream));
    int64_t *d_rows = (int64_t *)ctx->ws_brows.p;
    const double *d_w = nullptr;
    if (device_inputs) {
        HIPCHK(ctx, hipMemcpyAsync(d_rows, row_idx, (size_t)N * sizeof(int64_t), hipMemcpyDeviceToDevice, ctx->stream));
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
    if (device_inputs) HIPCHK(ctx, hipMemsetAsync(d_rows + N, 0, PREFETCH_PAD_ROWS * sizeof(int64_t), ctx->stream));
    else HIPCHK(ctx, hipMemsetAsync(d_rows, 0, ((size_t)N + PREFETCH_PAD_ROWS) * sizeof(int64_t), ctx->stream));
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
    j.d_score = (double *)ctx->ws_bscore.p;
    j.d_ninfo = (int64_t *)ctx->ws_bninfo.p;
    j.ldo = p->n_acc;
    // rows [r0, r1) of the concatenated inputs are on the device (or on their way, ordered before what follows):
    // sanitise the row list, build the LUT rows
    auto prepare_rows = [&](int64_t r0, int64_t r1, const int32_t *rows32) -> int {
        if (r1 <= r0) return SNPM_OK;
        const int64_t n = r1 - r0;
        hipLaunchKernelGGL(k_check_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_rows + r0, rows32, n,
                           p->n_snp, (int *)ctx->ws_flags2.p);
        ProfScope ps(ctx, PK_LUT);
        hipLaunchKernelGGL(k_build_lut, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_w + 3 * r0,
                           (double *)ctx->ws_blut.p + 4 * r0, n, skip, (int *)ctx->ws_flags2.p);
        HIPCHK(ctx, hipGetLastError());
        return SNPM_OK;
    };
    // every sample through the reference-order chain (requested, or more uncertain pairs than the sparse tier takes)
    auto strict_every_sample = [&]() -> int {
        for (int64_t b = 0; b < n_samples; ++b) {
            snpm_query *q = nullptr;
            const int64_t o = sample_off[b], nb = sample_off[b + 1] - o;
            int r = snpm_query_create_device(p, d_rows + o, 0, nb, d_w + 3 * o, &q);
            if (r) return r;
            r = run_strict_chain(q, skip, chunk, nullptr, nullptr, nullptr, (double *)ctx->ws_bscore.p + b * na,
                                 (int64_t *)ctx->ws_bninfo.p + b * na);
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
    SegPlan pl;
    if (!strict_all) {
        rc = seg_plan(ctx, j, pl);
        if (rc) return rc;
    }
    const double t_planned = now();
    if (device_inputs) {
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
            rc = prepare_rows(r0, r1, rows32);
            if (!rc && !strict_all) rc = seg_launch(ctx, j, pl, s0, s1);
            if (rc) return rc;
            t_launch += now() - ts1;
            s0 = s1;
        }
    }
    const double t_enqueued = now();
    int n_pairs = 0;
    if (strict_all) {
        rc = strict_every_sample();
        if (rc) return rc;
    } else {
        rc = seg_finish(ctx, j);
        if (rc) return rc;
        if (j.certify) {
            rc = ensure_pinned(ctx, 64);
            if (rc) return rc;
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, seg_pair_count(ctx), sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            n_pairs = *(const int *)ctx->h_pinned;
            if (n_pairs > j.cap) {
                strict_all = true;
                rc = strict_every_sample();
                if (rc) return rc;
            }
        }
    }
    if (lik) {
        if ((rc = ensure(ctx, ctx->ws_lik_l, B * na * sizeof(double)))) return rc;
        if ((rc = ensure(ctx, ctx->ws_lik_r, B * na * sizeof(double)))) return rc;
        int dom = 0;
        rc = snpm_likelihood_device(ctx, ctx->ws_bscore.p, ctx->ws_bninfo.p, n_samples, p->n_acc, 1, __builtin_nan(""),
                                    ctx->ws_lik_l.p, ctx->ws_lik_r.p, &dom);
        if (rc) return rc;
        if (dom) return set_err(ctx, SNPM_ERR_DOMAIN, "provided y is greater than n");
        HIPCHK(ctx, hipMemcpyAsync(lik, ctx->ws_lik_l.p, B * na * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(lrt, ctx->ws_lik_r.p, B * na * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (score) HIPCHK(ctx, hipMemcpyAsync(score, ctx->ws_bscore.p, B * na * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (ninfo) HIPCHK(ctx, hipMemcpyAsync(ninfo, ctx->ws_bninfo.p, B * na * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = ensure_pinned(ctx, 64))) return rc;
    int *h_bad = (int *)ctx->h_pinned + 8;
    HIPCHK(ctx, hipMemcpyAsync(h_bad, ctx->ws_flags2.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (trace)
        fprintf(stderr, "[snpm batch] plan %.3f ms, enqueue %.3f ms (staging %.3f, launches %.3f), finish+likelihood+copy back %.3f ms\n",
                t_planned - t_begin, t_enqueued - t_planned, t_stage, t_launch, now() - t_enqueued);
    if (*h_bad & 4) return set_err(ctx, SNPM_ERR_BADARG, "SNP weights must be finite (a NaN or infinite weight was given)");
    if (*h_bad) return set_err(ctx, SNPM_ERR_BADARG, "a row index lies outside the panel (n_snp %lld)", (long long)p->n_snp);
    if (info) { info[0] = n_pairs; info[1] = (strict_all && mode != SNPM_MODE_STRICT) ? 1 : 0; }
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

// ---------------------------------------------------------------------------------------------- profiling
int snpm_profile_enable(snpm_ctx *ctx, int on)
{
    if (!ctx) return SNPM_ERR_BADARG;
    ctx->prof_on = on != 0;
    return SNPM_OK;
}

int snpm_profile_reset(snpm_ctx *ctx)
{
    if (!ctx) return SNPM_ERR_BADARG;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < PK_COUNT; ++k) ctx->prof_pairs[k].clear();
    ctx->ev_used = 0;
    return SNPM_OK;
}

#include "snpm_once.hpp"

int snpm_profile_read(snpm_ctx *ctx, const char *kernel, int64_t *launches, double *total_ms)
{
    if (!ctx) return SNPM_ERR_BADARG;
    CHECK_ARG(ctx, kernel != nullptr, "kernel name is NULL");
    int kind = -1;
    for (int k = 0; k < PK_COUNT; ++k)
        if (strcmp(kernel, kProfNames[k]) == 0) kind = k;
    CHECK_ARG(ctx, kind >= 0, "unknown kernel name");
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    double tot = 0;
    for (auto &pr : ctx->prof_pairs[kind]) {
        float ms = 0;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_pool[pr.first], ctx->ev_pool[pr.second]));
        tot += ms;
    }
    if (launches) *launches = (int64_t)ctx->prof_pairs[kind].size();
    if (total_ms) *total_ms = tot;
    return SNPM_OK;
}

}  // extern "C"

// ================================================================================================ multi-GPU groups
#include "snpm_group.hpp"
