// snpm_hostpool.hpp -- the pure-host building blocks of the panel loader (snpm_loader.hpp): a persistent thread pool, copies
// into pinned slabs with non-temporal stores, the 2-bit packer (scalar and AVX2), exact / O_DIRECT file reads.  No HIP in here:
// tests/loader_tsan_driver.cpp builds this header with ThreadSanitizer and AddressSanitizer (tests/test_host_sanitizers_cpu.py).
#pragma once

#include <fcntl.h>
#include <stdint.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace {

// ---------------------------------------------------------------------------------------------- host thread pool
class HostPool {
public:
    // Workers SPIN for a short while (SNPM_POOL_SPIN_US, default 400 us) for the next run() before they block on the condition
    // variable, and so does the caller for the last task of a run: a sleeping thread takes 30-50 us to wake, which was most of
    // the 0.1 ms a 200k-SNP sample's staging fill took (snpm_genotype_once: calls follow each other within that window when
    // samples are scored in a row).  Long waits still sleep.
    explicit HostPool(int n)
    {
        if (const char *e = getenv("SNPM_POOL_SPIN_US")) spin_us_ = atoi(e) > 0 ? atoi(e) : 0;
        for (int i = 0; i < n; ++i) threads_.emplace_back([this] { loop(); });
    }
    ~HostPool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_.store(true, std::memory_order_release);
        }
        cv_work_.notify_all();
        for (auto &t : threads_) t.join();
    }
    int size() const { return (int)threads_.size(); }
    // fn(task) for task in [0, n_tasks), tasks handed out dynamically; returns when all are done.  The calling thread works too.
    void run(int n_tasks, const std::function<void(int)> &fn)
    {
        if (n_tasks <= 0) return;
        if (n_tasks == 1 || threads_.empty()) {
            for (int t = 0; t < n_tasks; ++t) fn(t);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn;
            n_tasks_ = n_tasks;
            next_ = 0;
            pending_.store(n_tasks, std::memory_order_relaxed);
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_work_.notify_all();
        work();
        if (!spin_until([this] { return pending_.load(std::memory_order_acquire) == 0; })) {
            std::unique_lock<std::mutex> lk(m_);
            cv_done_.wait(lk, [this] { return pending_.load(std::memory_order_acquire) == 0; });
        }
        std::lock_guard<std::mutex> lk(m_);            // (also: the last worker has left its notify)
        fn_ = nullptr;
    }

private:
    template <class Pred>
    bool spin_until(Pred done) const
    {
        if (spin_us_ <= 0) return done();
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            for (int i = 0; i < 64; ++i) {
                if (done()) return true;
#if defined(__x86_64__)
                _mm_pause();
#endif
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us_)) return done();
        }
    }
    void work()
    {
        for (;;) {
            int t;
            const std::function<void(int)> *fn;
            {
                std::lock_guard<std::mutex> lk(m_);
                if (!fn_ || next_ >= n_tasks_) return;
                t = next_++;
                fn = fn_;
            }
            (*fn)(t);
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(m_);
                cv_done_.notify_all();
            }
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            auto news = [&] { return stop_.load(std::memory_order_acquire) || gen_.load(std::memory_order_acquire) != seen; };
            if (!spin_until(news)) {
                std::unique_lock<std::mutex> lk(m_);
                cv_work_.wait(lk, news);
            }
            if (stop_.load(std::memory_order_acquire)) return;
            seen = gen_.load(std::memory_order_acquire);
            work();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_work_, cv_done_;
    const std::function<void(int)> *fn_ = nullptr;
    int n_tasks_ = 0, next_ = 0;
    std::atomic<int> pending_{0};
    std::atomic<uint64_t> gen_{0};
    std::atomic<bool> stop_{false};
    int spin_us_ = 400;
};

// Copy into a pinned staging slab with non-temporal stores: the slab is read next by the DMA engine, not by this core, so the
// lines need neither a read-for-ownership nor a place in the cache (the fill is bound by the memory bandwidth of the NUMA
// node: 3 instead of 4 transfers per payload byte).
#if defined(__x86_64__)
__attribute__((target("avx2"))) void copy_nt_avx2(int8_t *dst, const int8_t *src, size_t n)
{
    size_t head = (32 - ((uintptr_t)dst & 31)) & 31;
    if (head > n) head = n;
    if (head) {
        memcpy(dst, src, head);
        dst += head;
        src += head;
        n -= head;
    }
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(src + i)), b = _mm256_loadu_si256((const __m256i *)(src + i + 32));
        const __m256i c = _mm256_loadu_si256((const __m256i *)(src + i + 64)), d = _mm256_loadu_si256((const __m256i *)(src + i + 96));
        _mm256_stream_si256((__m256i *)(dst + i), a);
        _mm256_stream_si256((__m256i *)(dst + i + 32), b);
        _mm256_stream_si256((__m256i *)(dst + i + 64), c);
        _mm256_stream_si256((__m256i *)(dst + i + 96), d);
    }
    if (i < n) memcpy(dst + i, src + i, n - i);
    _mm_sfence();
}
#endif

void copy_to_slab(bool nt, int8_t *dst, const int8_t *src, size_t n)
{
#if defined(__x86_64__)
    if (nt && n >= 4096) {
        copy_nt_avx2(dst, src, n);
        return;
    }
#endif
    (void)nt;
    memcpy(dst, src, n);
}

bool cpu_has_avx2()
{
#if defined(__x86_64__)
    return __builtin_cpu_supports("avx2") != 0;
#else
    return false;
#endif
}

// ---------------------------------------------------------------------------------------------- 2-bit packing on the host
// n int8 calls -> (n + 3) / 4 bytes: field f of byte b holds call 4 b + f as 0 ref, 1 alt, 2 het, 3 missing (any negative);
// fields past n are 3.  Returns nonzero when a call > 2 was seen (a packed panel cannot store it).
int pack_row_scalar(const int8_t *src, int64_t n, uint8_t *dst)
{
    int bad = 0;
    int64_t i = 0;
    for (; i + 4 <= n; i += 4) {
        uint32_t out = 0;
        for (int f = 0; f < 4; ++f) {
            const int v = src[i + f];
            bad |= (v > 2);
            out |= (uint32_t)(v < 0 ? 3 : (v & 3)) << (2 * f);
        }
        dst[i >> 2] = (uint8_t)out;
    }
    if (i < n) {
        uint32_t out = 0;
        for (int f = 0; f < 4; ++f) {
            const int v = (i + f < n) ? src[i + f] : -1;
            bad |= (v > 2);
            out |= (uint32_t)(v < 0 ? 3 : (v & 3)) << (2 * f);
        }
        dst[i >> 2] = (uint8_t)out;
    }
    return bad;
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) int pack_row_avx2(const int8_t *src, int64_t n, uint8_t *dst)
{
    const __m256i zero = _mm256_setzero_si256(), two = _mm256_set1_epi8(2), three = _mm256_set1_epi8(3);
    const __m256i pick = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                          0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    __m256i badv = zero;
    int64_t i = 0;
    for (; i + 32 <= n; i += 32) {
        __m256i x = _mm256_loadu_si256((const __m256i *)(src + i));
        badv = _mm256_or_si256(badv, _mm256_cmpgt_epi8(x, two));
        x = _mm256_and_si256(_mm256_or_si256(x, _mm256_cmpgt_epi8(zero, x)), three);      // negative -> 3
        x = _mm256_or_si256(x, _mm256_srli_epi32(x, 6));                                  // byte 0 |= byte 1 << 2, byte 2 |= byte 3 << 2
        x = _mm256_or_si256(x, _mm256_srli_epi32(x, 12));                                 // byte 0 |= byte 2 << 4
        x = _mm256_shuffle_epi8(x, pick);                                                 // low byte of every dword
        const __m128i lo = _mm256_castsi256_si128(x), hi = _mm256_extracti128_si256(x, 1);
        _mm_storel_epi64((__m128i *)(dst + (i >> 2)), _mm_unpacklo_epi32(lo, hi));
    }
    int bad = !_mm256_testz_si256(badv, badv);
    if (i < n) bad |= pack_row_scalar(src + i, n - i, dst + (i >> 2));
    return bad;
}
#endif

typedef int (*pack_row_fn)(const int8_t *, int64_t, uint8_t *);

pack_row_fn pick_pack_row(bool avx2 = true)
{
#if defined(__x86_64__)
    if (avx2 && cpu_has_avx2()) return pack_row_avx2;
#endif
    (void)avx2;
    return pack_row_scalar;
}

struct ThreadScratch {
    int8_t *p = nullptr;
    size_t cap = 0;
    ~ThreadScratch() { free(p); }
    int8_t *get(size_t bytes)
    {
        if (bytes > cap) {
            free(p);
            p = nullptr;
            cap = 0;
            void *q = nullptr;
            if (posix_memalign(&q, 4096, (bytes + 4095) & ~size_t(4095)) != 0) return nullptr;
            p = (int8_t *)q;
            cap = bytes;
        }
        return p;
    }
};
thread_local ThreadScratch t_scratch;

// read exactly [off, off + len) of the file into dst; 0 or an errno (-1: end of file)
int pread_full(int fd, int8_t *dst, size_t len, off_t off)
{
    size_t o = 0;
    while (o < len) {
        const ssize_t k = pread(fd, dst + o, len - o, off + (off_t)o);
        if (k < 0 && errno == EINTR) continue;
        if (k < 0) return errno;
        if (k == 0) return -1;
        o += (size_t)k;
    }
    return 0;
}

// O_DIRECT: read the 4096-aligned span that covers [off, off + len) into `buf` (4096-aligned, >= len + 8192 bytes);
// *data = where byte `off` landed.  A short count at the end of the file is fine as long as the wanted bytes arrived.
int pread_direct(int fd, int8_t *buf, size_t len, off_t off, const int8_t **data)
{
    const off_t a = off & ~(off_t)4095;
    const size_t want = (size_t)(off - a) + len;
    const size_t span = (want + 4095) & ~size_t(4095);
    size_t o = 0;
    while (o < want) {
        const ssize_t k = pread(fd, buf + o, span - o, a + (off_t)o);
        if (k < 0 && errno == EINTR) continue;
        if (k < 0) return errno;
        if (k == 0) return -1;
        o += (size_t)k;
        if (o < want && (o & 4095)) return EIO;          // a partial block in the middle of the file: give up on this mode
    }
    *data = buf + (off - a);
    return 0;
}

}  // namespace
