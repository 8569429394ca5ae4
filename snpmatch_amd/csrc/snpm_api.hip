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
    Buf ws_tickets;                     // k_reduce_all: one ticket per column block, zero between launches
    int even_tiles = 1;                 // SNPM_EVEN_TILES=0: short int8 queries keep 128-row tiles (round 4) instead of tiles that divide evenly over the parts
    int fused_reduce = 0;               // SNPM_FUSED_REDUCE=1: k_reduce_groups + k_reduce as ONE launch with a ticket per column block (measured SLOWER: 12.8 us against 5.4 + 4.7 us on a 200k-SNP sample, profiles/r05_once_timeline.txt)
    int once_tail = 1;                  // SNPM_ONCE_TAIL=0: snpm_genotype_once ends with k_scan_few + k_once_finish instead of k_once_tail
    Buf ws_once, ws_once_table;         // packed results of snpm_genotype_once; the weight table of its coded form
    std::vector<double> once_table;     // host image of ws_once_table
    std::vector<uint8_t> once_code_flags;   // per code: bit 0 a fractional / huge entry, 1 neither 0 nor 1, 2 NaN / infinite, 3 past the table
    std::vector<double> once_code_abs;  // per code: |entry| (0 for codes past the table and non-finite entries)
    Buf ws_once_state;                  // {ticket, bad-input bits} of k_once_prep / k_once_finish: zero between calls
    bool once_state_clean = false;
    int once_fused = 1;                 // SNPM_ONCE_FUSED=0: snpm_genotype_once keeps the unfused kernels and copies of its first version
    int once_zero_copy = 1;             // SNPM_ONCE_ZEROCOPY=0: the fused form sends the slab through the copy engine (two pieces behind the fill) instead of reading it in place
    Buf ws_seg_desc, ws_eseg, ws_pairs, ws_pair_sums, ws_bscore, ws_bninfo, ws_bout, ws_blut, ws_brows, ws_brows32, ws_bw, ws_bcodes;   // segmented / batched scoring
    // shared-row scan of a batch (snpm_api_shared.hpp): union of the samples' rows, the int8 digit matrix, partial digit sums
    Buf ws_sh_bitmap, ws_sh_wordbase, ws_sh_blocks, ws_sh_urows, ws_sh_meta, ws_sh_A, ws_sh_pos, ws_sh_partial;
    int batch_shared = -1;              // SNPM_BATCH_SHARED / snpm_batch_configure: -1 auto (batches whose inputs are on the device), 0 never, 1 whenever the batch allows it
    int shared_digits = 0;              // base-256 digits of the fixed-point weights (3..7: 2^-(8 (digits - 1) + 6) per matched SNP of quantisation); 0 = by the longest sample
    int shared_min_samples = 4;         // auto: smaller batches keep the per-sample pass (2 / 3 / 4 / 6 samples of 194k SNPs on one marker set: 0.28 / 0.29 / 0.30 / 0.32 ms
                                        // against 0.32 / 0.35 / 0.39 / 0.53: the gain below 4 is within what a lower overlap takes back)
    double shared_min_density = -1.0;   // auto threshold; negative: by panel format (shared_min_density_of)
    size_t shared_ws_bytes = size_t(2) << 30;   // SNPM_SHARED_WS_MB: digit matrix per pass over groups of samples
    int shared_force_tiles = 0;         // SNPM_SHARED_TILES: row tiles of k_sh_mfma (tests, experiments)
    int shared_fill = 1;                // SNPM_SHARED_FILL=0: no filler tiles on the CUs the XCD-aligned row tiles leave idle
    int shared_probe = 1;               // SNPM_SHARED_PROBE=0: the automatic policy decides after the full pass over the batch only
    int shared_parts = 1;               // SNPM_SHARED_PARTS=n: the pass in n parts, the digit layout of a part (auxiliary stream) beside the previous part's contraction.
                                        // Measured SLOWER (64 x 200k x 1135: 1.08 ms on one stream, 1.28 / 1.33 / 1.91 ms with 2 / 4 / 8 parts: the layout's waves take issue slots
                                        // and L1 from the one-wave-per-SIMD contraction, and a part no longer fills the chip), kept for experiments
    hipStream_t aux_stream = nullptr;   // created with the context
    hipEvent_t aux_ev[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // the automatic choice: calls per (sample, union row) slot from which the contraction is the cheaper pass -- measured on 64
    // samples x 200k SNPs x 1135 accessions: the contraction costs ~2.8 ns per union row, the per-sample pass 0.27 ns (int8) /
    // 0.16 ns (packed) per call
    // break-even of the two passes at 64 samples x 200k markers on 1135 accessions (profiles/r05_shared_density.txt): int8 at 0.10
    // calls per (sample, union row) slot, packed at 0.20
    double shared_min_density_of(bool packed, int64_t n_samples) const
    {
        if (shared_min_density >= 0.0) return shared_min_density;
        const double th = packed ? 0.21 : 0.11;
        return n_samples < 8 ? (th > 0.5 ? th : 0.5) : th;      // a handful of samples: only when they really are on one marker set
    }
    int64_t shared_last[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // snpm_batch_last_stats
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
    bool cert_count_clean = false;      // the flag count is already zero (k_once_prep cleared it): run_fast skips its fill, once
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

#include "snpm_api_launch.hpp"

#include "snpm_api_cert.hpp"

#include "snpm_api_strict.hpp"

#include "snpm_api_seg.hpp"

#include "snpm_api_shared.hpp"
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
    // auxiliary compute stream + its events (the shared-row scan lays out a part's digits beside the previous part's contraction);
    // without them that pass simply runs on one stream
    if (hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        ctx->aux_stream = nullptr;
    } else {
        for (auto &ev : ctx->aux_ev)
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                (void)hipStreamDestroy(ctx->aux_stream);
                ctx->aux_stream = nullptr;
                break;
            }
    }
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
    if (const char *s = getenv("SNPM_ONCE_FUSED")) ctx->once_fused = atoi(s) != 0;
    if (const char *s = getenv("SNPM_ONCE_ZEROCOPY")) ctx->once_zero_copy = atoi(s) != 0;
    if (const char *s = getenv("SNPM_DEBUG_MAX_PARTS")) ctx->debug_max_parts = atoi(s);
    if (const char *s = getenv("SNPM_FUSED_REDUCE")) ctx->fused_reduce = atoi(s) != 0;
    if (const char *s = getenv("SNPM_EVEN_TILES")) ctx->even_tiles = atoi(s) != 0;
    if (const char *s = getenv("SNPM_ONCE_TAIL")) ctx->once_tail = atoi(s) != 0;
    if (const char *s = getenv("SNPM_BATCH_SHARED")) ctx->batch_shared = atoi(s) < 0 ? -1 : (atoi(s) ? 1 : 0);
    if (const char *s = getenv("SNPM_SHARED_DIGITS")) ctx->shared_digits = atoi(s) <= 0 ? 0 : std::min(7, std::max(3, atoi(s)));
    if (const char *s = getenv("SNPM_SHARED_MIN_DENSITY")) ctx->shared_min_density = atof(s);
    if (const char *s = getenv("SNPM_SHARED_FILL")) ctx->shared_fill = atoi(s) != 0;
    if (const char *s = getenv("SNPM_SHARED_WS_MB")) ctx->shared_ws_bytes = (size_t)std::max(1, atoi(s)) << 20;
    if (const char *s = getenv("SNPM_SHARED_TILES")) ctx->shared_force_tiles = std::max(0, atoi(s));
    if (const char *s = getenv("SNPM_SHARED_PROBE")) ctx->shared_probe = atoi(s) != 0;
    if (const char *s = getenv("SNPM_SHARED_PARTS")) ctx->shared_parts = std::max(1, std::min(8, atoi(s)));
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
                       &ctx->ws_seg_desc, &ctx->ws_eseg, &ctx->ws_pairs, &ctx->ws_pair_sums, &ctx->ws_bscore, &ctx->ws_bninfo, &ctx->ws_bout,
                       &ctx->ws_blut, &ctx->ws_brows, &ctx->ws_brows32, &ctx->ws_bw, &ctx->ws_bcodes,
                       &ctx->ws_sh_bitmap, &ctx->ws_sh_wordbase, &ctx->ws_sh_blocks, &ctx->ws_sh_urows, &ctx->ws_sh_meta, &ctx->ws_sh_A,
                       &ctx->ws_sh_pos, &ctx->ws_sh_partial, &ctx->ws_tickets};
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
        for (auto &ev : ctx->aux_ev)
            if (ev) (void)hipEventDestroy(ev);
        if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
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

#include "snpm_api_panel.hpp"

#include "snpm_api_query.hpp"

#include "snpm_api_carry.hpp"

#include "snpm_api_oneshot.hpp"
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
