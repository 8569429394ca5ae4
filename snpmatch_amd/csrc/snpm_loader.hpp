// snpm_loader.hpp -- the pinned-host staging path that replaces the reference's HDF5 row reads
// (core/snpmatch.py:222 `g.g.snps[idx, :]`, pygwas/genotype.py:548-550): DB rows from host memory or from a flat
// file -> pinned slabs (filled by a persistent pool of host threads) -> hipMemcpyAsync on the side stream -> a device
// kernel that writes the panel rows.  Included by snpm_api.hip (same translation unit).
//
// Round 3:
//   * packed panels are packed to 2 bits per call ON THE HOST while the slab is filled (AVX2 where the CPU has it): a
//     quarter of the bytes cross PCIe, the device kernel only re-pitches;
//   * a persistent thread pool (no thread creation per slab), three slabs of 64 MiB in flight;
//   * snpm_panel_load_file_rows: arbitrary row lists, strided files (a column range of a wider matrix: this rank's
//     accession shard), O_DIRECT for contiguous reads of files that are not in the page cache;
//   * the copy stream waits for the compute work queued on THE PANEL BEING WRITTEN, not for everything queued on the
//     context: loading one panel overlaps scoring another (slab-streamed DBs: two half-buffers).
#pragma once

#include <condition_variable>
#include <functional>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "snpm_h5.hpp"
#include "snpm_hostpool.hpp"

namespace {

HostPool *host_pool(snpm_ctx *ctx)
{
    if (!ctx->pool) ctx->pool = new HostPool(std::max(0, ctx->stage_threads - 1));     // + the calling thread
    return (HostPool *)ctx->pool;
}

void host_pool_destroy(snpm_ctx *ctx)
{
    delete (HostPool *)ctx->pool;
    ctx->pool = nullptr;
}

int default_stage_threads()
{
    cpu_set_t set;
    int n = 8;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    return std::max(2, std::min(n, 16));
}

// copy n bytes with the pool (one memcpy stream is ~10-15 GB/s, well under what PCIe gen5 x16 takes)
void parallel_copy(snpm_ctx *ctx, int8_t *dst, const int8_t *src, size_t n)
{
    const size_t piece = size_t(2) << 20;
    const int tasks = (int)std::min<size_t>(4096, (n + piece - 1) / piece);
    if (tasks <= 1) {
        memcpy(dst, src, n);
        return;
    }
    const size_t per = ((n + tasks - 1) / tasks + 63) & ~size_t(63);
    host_pool(ctx)->run(tasks, [=](int t) {
        const size_t o = (size_t)t * per;
        if (o < n) copy_to_slab(ctx->ld_nt != 0, dst + o, src + o, std::min(per, n - o));
    });
}

// ---------------------------------------------------------------------------------------------- sources of DB rows
struct RowSource {
    // host memory: row r at host + r * host_pitch
    const int8_t *host = nullptr;
    int64_t host_pitch = 0;
    // or a file: row r (or row_idx[r]) of a matrix of file_pitch bytes per row starting at byte file_offset; the panel's columns
    // begin at byte col0 of the row
    int fd = -1;
    bool direct = false;                // fd was opened with O_DIRECT (contiguous reads only)
    int64_t file_offset = 0, file_pitch = 0, col0 = 0, file_row0 = 0;
    const int64_t *row_idx = nullptr;
    const char *path = "";
    bool prepacked = false;             // the file's rows are already 2 bits per call (a packed .snpm): col0 / file_pitch in BYTES of
                                        // packed row, the panel's first accession is a multiple of 4
    // or a 2-D int8 dataset of an open HDF5 file (the reference's DB format): chunks are decompressed by the filling threads
    snpm_h5 *h5 = nullptr;
    const void *h5_dataset = nullptr;
    int64_t h5_chunk_rows = 0;
};

// Fill `n` staged rows (tight: n_acc bytes per row, or (n_acc + 3) / 4 bytes when `pack`) starting at source row
// `first`.  Work is cut into runs of rows handed to the pool.  Returns SNPM_OK or an error; *bad |= 1 on a call > 2 (pack).
int fill_rows(snpm_ctx *ctx, const RowSource &src, int64_t n_acc, bool pack, int8_t *dst, int64_t first, int64_t n, int *bad)
{
    const pack_row_fn pack_row = pick_pack_row(ctx->ld_avx2 != 0);
    const bool nt = ctx->ld_nt != 0;
    if (src.prepacked) {
        // rows of a packed file travel as they are: (n_acc + 3) / 4 bytes per row; the unused fields of the last byte (accessions of
        // a neighbouring shard, or padding) become "missing"
        const int64_t pk = (n_acc + 3) / 4;
        const uint8_t tail = (n_acc & 3) ? (uint8_t)(0xFFu << (2 * (n_acc & 3))) : 0;
        const int64_t run_p = std::max<int64_t>(1, (int64_t)((size_t(1) << 20) / (size_t)pk));
        const int tasks_p = (int)std::min<int64_t>(1 << 20, (n + run_p - 1) / run_p);
        std::atomic<int> err_p{0};
        const bool contiguous = !src.row_idx && src.file_pitch == pk && src.col0 == 0;
        host_pool(ctx)->run(tasks_p, [&](int t) {
            const int64_t k0 = (int64_t)t * run_p, k1 = std::min<int64_t>(n, k0 + run_p);
            if (k0 >= k1 || err_p.load(std::memory_order_relaxed)) return;
            if (contiguous) {
                const off_t off = (off_t)(src.file_offset + (src.file_row0 + first + k0) * pk);
                const size_t len = (size_t)(k1 - k0) * pk;
                int8_t *buf = t_scratch.get(len + 8192);
                const int8_t *data = buf;
                int e = buf ? 0 : ENOMEM;
                if (!e) e = src.direct ? pread_direct(src.fd, buf, len, off, &data) : pread_full(src.fd, buf, len, off);
                if (e) { err_p.store(e); return; }
                copy_to_slab(nt, dst + k0 * pk, data, len);
            } else {
                for (int64_t k = k0; k < k1; ++k) {
                    const int64_t r = src.row_idx ? src.row_idx[first + k] : src.file_row0 + first + k;
                    const int e = pread_full(src.fd, dst + k * pk, (size_t)pk, (off_t)(src.file_offset + r * src.file_pitch + src.col0));
                    if (e) { err_p.store(e); return; }
                }
            }
            if (tail)
                for (int64_t k = k0; k < k1; ++k) ((uint8_t *)dst)[k * pk + pk - 1] |= tail;
        });
        const int e = err_p.load();
        if (e) return set_err(ctx, SNPM_ERR_BADARG, "short read from %s (%s)", src.path, e > 0 ? strerror(e) : "end of file");
        return SNPM_OK;
    }
    const int64_t out_pitch = pack ? (n_acc + 3) / 4 : n_acc;
    int64_t run = std::max<int64_t>(1, (int64_t)((size_t(1) << 20) / (size_t)n_acc));      // ~1 MiB of source per task
    int64_t lead = 0;                   // rows of the first task (HDF5: up to the next chunk boundary, so that a chunk is decompressed once)
    if (src.h5 && !src.row_idx && src.h5_chunk_rows > 0) {
        run = src.h5_chunk_rows;
        lead = (run - (src.file_row0 + first) % run) % run;
    } else if (src.h5) {
        run = 256;                      // a row list: runs of the list, each thread keeps its last chunk
    }
    const int tasks = (int)std::min<int64_t>(1 << 20, (lead > 0 ? 1 : 0) + (std::max<int64_t>(n - lead, 0) + run - 1) / run);
    std::atomic<int> bad_any{0}, err_any{0};
    std::mutex msg_mu;
    std::string h5_msg;
    const bool contiguous_file = src.fd >= 0 && !src.row_idx && src.file_pitch == n_acc && src.col0 == 0;
    host_pool(ctx)->run(tasks, [&](int t) {
        int64_t k0, k1;
        if (lead > 0) {
            k0 = t == 0 ? 0 : lead + (int64_t)(t - 1) * run;
            k1 = t == 0 ? std::min(lead, n) : std::min<int64_t>(n, k0 + run);
        } else {
            k0 = (int64_t)t * run;
            k1 = std::min<int64_t>(n, k0 + run);
        }
        if (k0 >= k1 || err_any.load(std::memory_order_relaxed)) return;
        int b = 0;
        if (src.h5) {
            const size_t len = (size_t)(k1 - k0) * n_acc;
            int8_t *to = pack ? t_scratch.get(len + 64) : dst + k0 * n_acc;
            int rc = to ? snpm_h5_rows_raw(src.h5, src.h5_dataset, src.row_idx ? src.row_idx + first + k0 : nullptr,
                                           src.file_row0 + first + k0, k1 - k0, src.col0, n_acc, to, n_acc)
                        : SNPM_ERR_OOM;
            if (rc) {
                std::lock_guard<std::mutex> lk(msg_mu);
                if (h5_msg.empty()) h5_msg = snpm_h5_thread_error();
                err_any.store(EIO);
                return;
            }
            if (pack)
                for (int64_t k = k0; k < k1; ++k) b |= pack_row(to + (k - k0) * n_acc, n_acc, (uint8_t *)dst + k * out_pitch);
        } else if (src.host) {
            if (!pack && src.host_pitch == n_acc) {
                copy_to_slab(nt, dst + k0 * n_acc, src.host + (first + k0) * n_acc, (size_t)(k1 - k0) * n_acc);
            } else {
                for (int64_t k = k0; k < k1; ++k) {
                    const int8_t *row = src.host + (first + k) * src.host_pitch;
                    if (pack) b |= pack_row(row, n_acc, (uint8_t *)dst + k * out_pitch);
                    else copy_to_slab(nt, dst + k * n_acc, row, (size_t)n_acc);
                }
            }
        } else if (contiguous_file) {
            const off_t off = (off_t)(src.file_offset + (src.file_row0 + first + k0) * n_acc);
            const size_t len = (size_t)(k1 - k0) * n_acc;
            {
                // through a per-thread scratch buffer (it stays in this core's cache), then packed or copied with non-temporal
                // stores into the slab: faster than pread() straight into the pinned slab (measured 16.8 -> 35 GB/s warm)
                int8_t *buf = t_scratch.get(len + 8192);
                const int8_t *data = buf;
                int e = buf ? 0 : ENOMEM;
                if (!e) e = src.direct ? pread_direct(src.fd, buf, len, off, &data) : pread_full(src.fd, buf, len, off);
                if (e) {
                    err_any.store(e);
                } else if (pack) {
                    for (int64_t k = k0; k < k1; ++k) b |= pack_row(data + (k - k0) * n_acc, n_acc, (uint8_t *)dst + k * out_pitch);
                } else {
                    copy_to_slab(nt, dst + k0 * n_acc, data, len);
                }
            }
        } else {
            // a row list, or a column range of a wider matrix: one read per row
            int8_t *buf = pack ? t_scratch.get((size_t)n_acc + 64) : nullptr;
            for (int64_t k = k0; k < k1; ++k) {
                const int64_t r = src.row_idx ? src.row_idx[first + k] : src.file_row0 + first + k;
                const off_t off = (off_t)(src.file_offset + r * src.file_pitch + src.col0);
                int8_t *to = pack ? buf : dst + k * n_acc;
                const int e = to ? pread_full(src.fd, to, (size_t)n_acc, off) : ENOMEM;
                if (e) {
                    err_any.store(e);
                    return;
                }
                if (pack) b |= pack_row(buf, n_acc, (uint8_t *)dst + k * out_pitch);
            }
        }
        if (b) bad_any.store(1);
    });
    if (bad_any.load()) *bad |= 1;
    const int e = err_any.load();
    if (e && src.h5) return set_err(ctx, SNPM_ERR_BADARG, "%s", h5_msg.c_str());
    if (e) return set_err(ctx, SNPM_ERR_BADARG, "short read from %s (%s)", src.path, e > 0 ? strerror(e) : "end of file");
    return SNPM_OK;
}

// ---------------------------------------------------------------------------------------------- the staging pipeline
// host-packed rows (tight, src_pitch bytes) -> packed panel rows (256-B pitch, pad bytes 0xFF = four missing calls)
__global__ void k_repitch_packed(const uint8_t *__restrict__ src, int64_t src_pitch, int64_t nrows, uint8_t *__restrict__ db,
                                 int64_t pitch, int64_t desc, int64_t row0)
{
    // db / pitch / desc: the packed panel (row-major or split rows, snpm_k_common.hpp); one thread per destination dword
    const int64_t dwords_per_row = (pitch + snpm::pk_tail_pitch(desc)) / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * dwords_per_row) return;
    const int64_t r = i / dwords_per_row, d = i - r * dwords_per_row;
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t b = d * 4 + j;
        out |= (uint32_t)(b < src_pitch ? src[r * src_pitch + b] : 0xffu) << (8 * j);
    }
    *reinterpret_cast<uint32_t *>(db + snpm::pk_off(pitch, desc, row0 + r, d * 4)) = out;
}

// packed staged rows (tight, src_pitch bytes) -> int8 panel rows (256-B pitch): one thread per destination dword = one source byte;
// bytes past the row's accessions are 0xFF
__global__ void k_unpack_repitch(const uint8_t *__restrict__ src, int64_t src_pitch, int64_t nrows, int64_t n_acc,
                                 uint32_t *__restrict__ dst, int64_t dst_pitch)
{
    const int64_t dwords_per_row = dst_pitch / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * dwords_per_row) return;
    const int64_t r = i / dwords_per_row, d = i - r * dwords_per_row;
    const uint32_t b = d < src_pitch ? src[r * src_pitch + d] : 0xffu;
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t v = (b >> (2 * j)) & 3u;
        const uint32_t c = (d * 4 + j < n_acc && v != 3u) ? v : 0xffu;
        out |= c << (8 * j);
    }
    dst[i] = out;
}

int ensure_loader(snpm_ctx *ctx, size_t min_slab)
{
    const size_t want = std::max(ctx->ld_want, min_slab);
    if (ctx->ld_cap >= want && ctx->ld_stage[0]) return SNPM_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
    for (int i = 0; i < snpm_ctx::kLdStages; ++i) {
        if (ctx->ld_stage[i]) (void)hipHostFree(ctx->ld_stage[i]);
        ctx->ld_stage[i] = nullptr;
        ctx->ld_busy[i] = false;
    }
    ctx->ld_cap = 0;
    for (int i = 0; i < snpm_ctx::kLdStages; ++i) {
        hipError_t e = hipHostMalloc(&ctx->ld_stage[i], want, hipHostMallocDefault);
        if (e != hipSuccess) return set_err(ctx, SNPM_ERR_OOM, "hipHostMalloc of a %zu-byte staging slab failed: %s", want, hipGetErrorString(e));
        if (!ctx->ld_done[i]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ld_done[i], hipEventDisableTiming));
    }
    ctx->ld_cap = want;
    return ensure(ctx, ctx->ws_stage_dev, (size_t)snpm_ctx::kLdStages * want);
}

// rows [row0, row0 + nrows) of the panel <- the source.  Returns once the last slab is enqueued on the copy stream
// (packed panels: once it has arrived -- the check for calls a packed panel cannot hold is synchronous).
int stage_rows(snpm_panel *p, int64_t row0, int64_t nrows, const RowSource &src)
{
    snpm_ctx *ctx = p->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int64_t n_acc = p->n_acc;
    // staged rows are tight: n_acc bytes, or (n_acc + 3) / 4 when they are packed on the host (packed panels) or come from a packed file
    const bool host_pack = src.prepacked || (p->packed && ctx->host_pack);
    const int64_t spitch = host_pack ? (n_acc + 3) / 4 : n_acc;
    int rc = ensure_loader(ctx, (size_t)spitch);
    if (rc) return rc;
    rc = ensure(ctx, ctx->ws_flags2, sizeof(int));
    if (rc) return rc;
    // Rows about to be overwritten may still be read by scoring kernels queued on the compute stream.  The copy stream waits
    // for the work queued on THIS panel: if it was the last panel the compute stream touched, everything queued so far (an event
    // recorded now); otherwise the event recorded when the compute stream moved on to another panel (wait_upload).
    if (ctx->last_touched == p || !p->used) {
        if (!ctx->compute_mark) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->compute_mark, hipEventDisableTiming));
        HIPCHK(ctx, hipEventRecord(ctx->compute_mark, ctx->stream));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->compute_mark, 0));
    } else if (p->used_valid) {
        HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream, p->used, 0));
    }
    if (p->packed && !host_pack) HIPCHK(ctx, hipMemsetAsync(ctx->ws_flags2.p, 0, sizeof(int), ctx->copy_stream));
    const int64_t slab_rows = std::max<int64_t>(1, (int64_t)(ctx->ld_cap / (size_t)spitch));
    int bad = 0;
    for (int64_t r = 0; r < nrows && !bad; r += slab_rows) {
        const int64_t nr = std::min(slab_rows, nrows - r);
        const int which = ctx->ld_next;
        ctx->ld_next = (ctx->ld_next + 1) % snpm_ctx::kLdStages;
        if (ctx->ld_busy[which]) {
            HIPCHK(ctx, hipEventSynchronize(ctx->ld_done[which]));
            ctx->ld_busy[which] = false;
        }
        int8_t *st = (int8_t *)ctx->ld_stage[which];
        rc = fill_rows(ctx, src, n_acc, host_pack, st, r, nr, &bad);
        if (rc) return rc;
        if (bad) break;
        int8_t *scratch = (int8_t *)ctx->ws_stage_dev.p + (size_t)which * ctx->ld_cap;
        HIPCHK(ctx, hipMemcpyAsync(scratch, st, (size_t)nr * spitch, hipMemcpyHostToDevice, ctx->copy_stream));
        if (!p->packed && host_pack) {              // a packed file into an int8 panel: unpacked on the device
            const int64_t total = nr * (p->pitch / 4);
            hipLaunchKernelGGL(k_unpack_repitch, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->copy_stream,
                               (const uint8_t *)scratch, spitch, nr, n_acc, (uint32_t *)(p->d + (row0 + r) * p->pitch), p->pitch);
        } else if (!p->packed) {
            const int64_t total = nr * (p->pitch / 4);
            hipLaunchKernelGGL(k_repitch_canon, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->copy_stream,
                               scratch, spitch, nr, n_acc, (uint32_t *)(p->d + (row0 + r) * p->pitch), p->pitch, p->d_other);
        } else if (host_pack) {             // (packed panels: p->pitch = bytes per row, kpitch / desc = how they lie in memory)
            const int64_t total = nr * (p->pitch / 4);
            hipLaunchKernelGGL(k_repitch_packed, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->copy_stream,
                               (const uint8_t *)scratch, spitch, nr, (uint8_t *)p->d, p->kpitch, p->desc, row0 + r);
        } else {
            const int64_t total = nr * p->pitch;
            hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->copy_stream, scratch,
                               spitch, nr, n_acc, (uint8_t *)p->d, p->kpitch, p->desc, row0 + r, (int *)ctx->ws_flags2.p);
        }
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipEventRecord(ctx->ld_done[which], ctx->copy_stream));
        ctx->ld_busy[which] = true;
    }
    if (p->packed && !host_pack) {
        rc = ensure_pinned(ctx, 64);
        if (rc) return rc;
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, ctx->ws_flags2.p, sizeof(int), hipMemcpyDeviceToHost, ctx->copy_stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
        for (int i = 0; i < snpm_ctx::kLdStages; ++i) ctx->ld_busy[i] = false;
        bad = *(const int *)ctx->h_pinned;
    }
    HIPCHK(ctx, hipEventRecord(p->uploaded, ctx->copy_stream));
    p->upload_pending = true;
    p->dT_state = 0;                    // the accession-major copy is stale
    if (bad)
        return set_err(ctx, SNPM_ERR_BADARG, "a packed panel holds only the codes -1 (any negative), 0, 1, 2; "
                                             "use the int8 panel for other values");
    return SNPM_OK;
}

// fraction of 64 sample pages of [off, off + bytes) that sit in the page cache (mincore on a read-only mapping; -1: unknown)
double cached_fraction(int fd, int64_t off, int64_t bytes)
{
    const int64_t page = 4096;
    const int64_t a = off & ~(page - 1);
    const size_t len = (size_t)(off + bytes - a);
    void *m = mmap(nullptr, len, PROT_READ, MAP_SHARED, fd, (off_t)a);
    if (m == MAP_FAILED) return -1.0;
    int hit = 0, seen = 0;
    for (int i = 0; i < 64; ++i) {
        const size_t o = ((len / 64) * (size_t)i) & ~(size_t)(page - 1);
        unsigned char v = 0;
        if (o < len && mincore((char *)m + o, 1, &v) == 0) {
            ++seen;
            hit += v & 1;
        }
    }
    munmap(m, len);
    return seen ? (double)hit / seen : -1.0;
}

// open a DB file for the loader.  O_DIRECT when asked for (SNPM_ODIRECT=1) or, by default, for contiguous reads of >= 1 GiB of
// a file that is NOT in the page cache (measured on the GPU box's disk, 20 GB: cold 16.3 GB/s direct vs 4.5 GB/s buffered; warm
// 16.4 GB/s direct vs 28.6 GB/s buffered) and only where the file system takes the flag
int open_source(snpm_ctx *ctx, const char *path, bool contiguous, int64_t off, int64_t bytes, RowSource *src)
{
    int fd = -1;
    bool want_direct = contiguous && (ctx->odirect == 1 || (ctx->odirect < 0 && bytes >= (int64_t(1) << 30)));
    if (want_direct && ctx->odirect < 0) {
        const int probe = open(path, O_RDONLY);
        if (probe >= 0) {
            if (cached_fraction(probe, off, bytes) >= 0.5) want_direct = false;
            close(probe);
        }
    }
    if (want_direct) {
        fd = open(path, O_RDONLY | O_DIRECT);
        if (fd >= 0) src->direct = true;
    }
    if (fd < 0) {
        fd = open(path, O_RDONLY);
        src->direct = false;
    }
    if (fd < 0) return set_err(ctx, SNPM_ERR_BADARG, "cannot open %s: %s", path, strerror(errno));
    src->fd = fd;
    src->path = path;
    return SNPM_OK;
}

}  // namespace

extern "C" {

int snpm_panel_upload_rows(snpm_panel *p, int64_t row0, int64_t nrows, const int8_t *host, int64_t host_pitch)
try {
    CHECK_PANEL(p);
    snpm_ctx *ctx = p->ctx;
    CHECK_ARG(ctx, row0 >= 0 && nrows >= 0 && row0 + nrows <= p->n_snp, "upload rows outside the panel");
    CHECK_ARG(ctx, nrows == 0 || host != nullptr, "host pointer is NULL");
    CHECK_ARG(ctx, host_pitch >= p->n_acc, "host_pitch smaller than n_acc");
    RowSource src;
    src.host = host;
    src.host_pitch = host_pitch;
    return stage_rows(p, row0, nrows, src);
} SNPM_GUARD((p ? p->ctx : nullptr))

// Rows [row0, row0 + nrows) of the panel from a file that holds an int8 matrix with file_pitch bytes per row (>= col0 + n_acc)
// starting at byte file_offset: file row row_idx[i] (row_idx != NULL) or file_row0 + i, columns [col0, col0 + n_acc).
int snpm_panel_load_file_rows(snpm_panel *p, const char *path, int64_t file_offset, int64_t file_pitch, int64_t col0,
                              const int64_t *row_idx, int64_t file_row0, int64_t row0, int64_t nrows)
try {
    CHECK_PANEL(p);
    snpm_ctx *ctx = p->ctx;
    CHECK_ARG(ctx, path != nullptr && file_offset >= 0, "bad file arguments");
    CHECK_ARG(ctx, row0 >= 0 && nrows >= 0 && row0 + nrows <= p->n_snp, "load rows outside the panel");
    CHECK_ARG(ctx, col0 >= 0 && file_pitch >= col0 + p->n_acc, "file_pitch smaller than col0 + n_acc");
    CHECK_ARG(ctx, row_idx != nullptr || file_row0 >= 0, "negative file row");
    struct stat st;
    if (stat(path, &st) != 0) return set_err(ctx, SNPM_ERR_BADARG, "cannot open %s: %s", path, strerror(errno));
    if (row_idx) {
        for (int64_t i = 0; i < nrows; ++i)
            if (row_idx[i] < 0 || file_offset + row_idx[i] * file_pitch + col0 + p->n_acc > (int64_t)st.st_size)
                return set_err(ctx, SNPM_ERR_BADARG, "file row %lld at %lld lies outside %s", (long long)row_idx[i], (long long)i, path);
    } else if (nrows > 0 && file_offset + (file_row0 + nrows - 1) * file_pitch + col0 + p->n_acc > (int64_t)st.st_size) {
        return set_err(ctx, SNPM_ERR_BADARG, "short read from %s (end of file)", path);
    }
    RowSource src;
    src.file_offset = file_offset;
    src.file_pitch = file_pitch;
    src.col0 = col0;
    src.row_idx = row_idx;
    src.file_row0 = row_idx ? 0 : file_row0;
    const bool contiguous = !row_idx && file_pitch == p->n_acc && col0 == 0;
    int rc = open_source(ctx, path, contiguous, file_offset + file_row0 * file_pitch, nrows * p->n_acc, &src);
    if (rc) return rc;
#ifdef POSIX_FADV_SEQUENTIAL
    if (contiguous && !src.direct)
        (void)posix_fadvise(src.fd, (off_t)(file_offset + file_row0 * file_pitch), (off_t)(nrows * file_pitch), POSIX_FADV_SEQUENTIAL);
#endif
    rc = stage_rows(p, row0, nrows, src);
    close(src.fd);
    return rc;
} SNPM_GUARD((p ? p->ctx : nullptr))

int snpm_panel_load_h5(snpm_panel *p, snpm_h5 *file, const char *dataset, int64_t col0, const int64_t *row_idx, int64_t file_row0,
                       int64_t row0, int64_t nrows)
try {
    CHECK_PANEL(p);
    snpm_ctx *ctx = p->ctx;
    CHECK_ARG(ctx, file != nullptr && dataset != nullptr, "HDF5 file / dataset name is NULL");
    CHECK_ARG(ctx, row0 >= 0 && nrows >= 0 && row0 + nrows <= p->n_snp, "load rows outside the panel");
    int64_t n_rows = 0, n_cols = 0, chunk_rows = 0;
    const void *ds = snpm_h5_int8_matrix(file, dataset, &n_rows, &n_cols, &chunk_rows);
    if (!ds) return set_err(ctx, SNPM_ERR_BADARG, "%s", snpm_h5_last_error(file));
    CHECK_ARG(ctx, col0 >= 0 && col0 + p->n_acc <= n_cols, "columns outside the HDF5 dataset");
    if (row_idx) {
        for (int64_t i = 0; i < nrows; ++i)
            if (row_idx[i] < 0 || row_idx[i] >= n_rows)
                return set_err(ctx, SNPM_ERR_BADARG, "row %lld at %lld lies outside the HDF5 dataset (%lld rows)", (long long)row_idx[i],
                               (long long)i, (long long)n_rows);
    } else {
        CHECK_ARG(ctx, file_row0 >= 0 && file_row0 + nrows <= n_rows, "rows outside the HDF5 dataset");
    }
    RowSource src;
    src.h5 = file;
    src.h5_dataset = ds;
    src.h5_chunk_rows = chunk_rows;
    src.col0 = col0;
    src.row_idx = row_idx;
    src.file_row0 = row_idx ? 0 : file_row0;
    return stage_rows(p, row0, nrows, src);
} SNPM_GUARD((p ? p->ctx : nullptr))

// the host packer on its own (no ctx, no GPU): nrows rows of n_acc int8 calls (row stride src_pitch) -> rows of (n_acc + 3) / 4
// bytes (row stride dst_pitch); *bad (may be NULL) = 1 when a call > 2 was met.  force_scalar != 0 skips the AVX2 form.
int snpm_pack_rows_host(const int8_t *src, int64_t src_pitch, int64_t nrows, int64_t n_acc, uint8_t *dst, int64_t dst_pitch,
                        int force_scalar, int *bad)
{
    if (!src || !dst || nrows < 0 || n_acc < 1 || src_pitch < n_acc || dst_pitch < (n_acc + 3) / 4) return SNPM_ERR_BADARG;
    const pack_row_fn fn = force_scalar ? pack_row_scalar : pick_pack_row();
    int b = 0;
    for (int64_t r = 0; r < nrows; ++r) b |= fn(src + r * src_pitch, n_acc, dst + r * dst_pitch);
    if (bad) *bad = b ? 1 : 0;
    return SNPM_OK;
}

// The same from a PACKED flat file (a .snpm written with 2 bits per call: 4 accessions per byte, field f of byte b = accession
// 4 b + f; file_pitch BYTES per row): panel row row0 + i receives accessions [acc0, acc0 + n_acc) (acc0 a multiple of 4) of file
// row row_idx[i] or file_row0 + i.  A quarter of the bytes leave the disk and cross PCIe, whatever the panel's format.
int snpm_panel_load_file_rows_packed(snpm_panel *p, const char *path, int64_t file_offset, int64_t file_pitch, int64_t acc0,
                                     const int64_t *row_idx, int64_t file_row0, int64_t row0, int64_t nrows)
try {
    CHECK_PANEL(p);
    snpm_ctx *ctx = p->ctx;
    CHECK_ARG(ctx, path != nullptr && file_offset >= 0, "bad file arguments");
    CHECK_ARG(ctx, row0 >= 0 && nrows >= 0 && row0 + nrows <= p->n_snp, "load rows outside the panel");
    CHECK_ARG(ctx, acc0 >= 0 && (acc0 & 3) == 0, "the first accession of a shard of a packed file must be a multiple of 4");
    const int64_t pk = (p->n_acc + 3) / 4, col0 = acc0 / 4;
    CHECK_ARG(ctx, file_pitch >= col0 + pk, "file_pitch smaller than the bytes of the requested accessions");
    CHECK_ARG(ctx, row_idx != nullptr || file_row0 >= 0, "negative file row");
    struct stat st;
    if (stat(path, &st) != 0) return set_err(ctx, SNPM_ERR_BADARG, "cannot open %s: %s", path, strerror(errno));
    if (row_idx) {
        for (int64_t i = 0; i < nrows; ++i)
            if (row_idx[i] < 0 || file_offset + row_idx[i] * file_pitch + col0 + pk > (int64_t)st.st_size)
                return set_err(ctx, SNPM_ERR_BADARG, "file row %lld at %lld lies outside %s", (long long)row_idx[i], (long long)i, path);
    } else if (nrows > 0 && file_offset + (file_row0 + nrows - 1) * file_pitch + col0 + pk > (int64_t)st.st_size) {
        return set_err(ctx, SNPM_ERR_BADARG, "short read from %s (end of file)", path);
    }
    RowSource src;
    src.prepacked = true;
    src.file_offset = file_offset;
    src.file_pitch = file_pitch;
    src.col0 = col0;
    src.row_idx = row_idx;
    src.file_row0 = row_idx ? 0 : file_row0;
    const bool contiguous = !row_idx && file_pitch == pk && col0 == 0;
    int rc = open_source(ctx, path, contiguous, file_offset + file_row0 * file_pitch, nrows * pk, &src);
    if (rc) return rc;
    rc = stage_rows(p, row0, nrows, src);
    close(src.fd);
    return rc;
} SNPM_GUARD((p ? p->ctx : nullptr))

// the contiguous form: tightly packed rows of exactly n_acc bytes starting at file_offset (the data section of snps.npy)
int snpm_panel_load_file(snpm_panel *p, const char *path, int64_t file_offset, int64_t row0, int64_t nrows)
{
    CHECK_PANEL(p);
    return snpm_panel_load_file_rows(p, path, file_offset, p->n_acc, 0, nullptr, 0, row0, nrows);
}

}  // extern "C"
