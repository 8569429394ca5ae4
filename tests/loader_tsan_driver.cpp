// Driver for the sanitizer builds of the loader's host-side building blocks (csrc/snpm_hostpool.hpp): the persistent thread pool
// under many short generations with data every task writes and the caller reads back, the 2-bit packer (AVX2 against scalar) on
// odd lengths and unaligned buffers, the non-temporal copy on every alignment, exact and O_DIRECT reads of a file at odd offsets.
// Built twice by tests/test_host_sanitizers_cpu.py: -fsanitize=thread and -fsanitize=address,undefined.
#include "snpm_hostpool.hpp"

#include <cstdio>
#include <random>
#include <string>

static int fails = 0;
#define EXPECT(cond)                                                    \
    do {                                                                \
        if (!(cond)) { ++fails; printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)

int main(int argc, char **argv)
{
    std::mt19937_64 rng(7);
    // ---- pool: 3000 generations of 1..70 tasks; every task writes its own slots, the caller sums them after run() returns
    {
        HostPool pool(7);
        std::vector<long> out(70);
        long total = 0, want = 0;
        for (int gen = 0; gen < 3000; ++gen) {
            const int n = 1 + (int)(rng() % 70);
            std::fill(out.begin(), out.end(), 0);
            pool.run(n, [&](int t) { out[(size_t)t] = (long)gen * 100 + t; });
            for (int t = 0; t < n; ++t) {
                total += out[(size_t)t];
                want += (long)gen * 100 + t;
            }
        }
        EXPECT(total == want);
        // nested data parallel copy through the pool-style splitting used by parallel_copy
        std::vector<int8_t> src(9u << 20), dst(9u << 20);
        for (auto &b : src) b = (int8_t)rng();
        const size_t per = ((src.size() + 15) / 16 + 63) & ~size_t(63);
        pool.run(16, [&](int t) {
            const size_t o = (size_t)t * per;
            if (o < src.size()) copy_to_slab(true, dst.data() + o, src.data() + o, std::min(per, src.size() - o));
        });
        EXPECT(memcmp(src.data(), dst.data(), src.size()) == 0);
    }
    { HostPool empty(0); int hit = 0; empty.run(5, [&](int) { ++hit; }); EXPECT(hit == 5); }
    // ---- packer: AVX2 == scalar on every length 1..300 and a few long rows, unaligned sources, all code values
    {
        const pack_row_fn fast = pick_pack_row(true);
        std::vector<int8_t> row(70000 + 3);
        std::vector<uint8_t> a(70000 / 4 + 8), b(70000 / 4 + 8);
        for (int rep = 0; rep < 400; ++rep) {
            const int64_t n = rep < 300 ? rep + 1 : (int64_t)(1000 + rng() % 69000);
            const int shift = (int)(rng() % 3);
            const bool with_bad = rep % 7 == 0;
            for (int64_t i = 0; i < n; ++i) row[(size_t)(i + shift)] = (int8_t)((int)(rng() % (with_bad ? 9 : 6)) - 3);     // -3 .. 2 (or .. 5)
            std::fill(a.begin(), a.end(), 0xAB);
            std::fill(b.begin(), b.end(), 0xAB);
            const int ba = fast(row.data() + shift, n, a.data() + 1), bb = pack_row_scalar(row.data() + shift, n, b.data() + 1);
            EXPECT((ba != 0) == (bb != 0));
            EXPECT(memcmp(a.data(), b.data(), a.size()) == 0);
            bool any_bad = false;
            for (int64_t i = 0; i < n; ++i) any_bad |= row[(size_t)(i + shift)] > 2;
            EXPECT(any_bad == (bb != 0));
            for (int64_t i = 0; i < n; ++i) {
                const int v = row[(size_t)(i + shift)], f = (b[1 + (size_t)(i >> 2)] >> (2 * (i & 3))) & 3;
                EXPECT(f == (v < 0 ? 3 : (v & 3)));
            }
            for (int64_t i = n; i < ((n + 3) / 4) * 4; ++i) EXPECT(((b[1 + (size_t)(i >> 2)] >> (2 * (i & 3))) & 3) == 3);
        }
    }
    // ---- non-temporal copy: every source / destination alignment, lengths around the 4096-byte switch
    {
        std::vector<int8_t> src(20000), dst(20000 + 64);
        for (auto &v : src) v = (int8_t)rng();
        for (int da = 0; da < 33; da += 3)
            for (size_t n : {size_t(1), size_t(31), size_t(4095), size_t(4096), size_t(4097), size_t(12345)})
                for (int nt = 0; nt < 2; ++nt) {
                    std::fill(dst.begin(), dst.end(), 0x55);
                    copy_to_slab(nt != 0 && cpu_has_avx2(), dst.data() + da, src.data() + (da % 5), n);
                    EXPECT(memcmp(dst.data() + da, src.data() + (da % 5), n) == 0);
                    EXPECT(dst[(size_t)da + n] == 0x55 && (da == 0 || dst[(size_t)da - 1] == 0x55));
                }
    }
    // ---- file reads: exact reads and O_DIRECT reads (where the file system takes the flag) at odd offsets, end of file
    if (argc > 1) {
        const std::string path = std::string(argv[1]) + "/reads.bin";
        std::vector<int8_t> data(3 * 1024 * 1024 + 777);
        for (auto &v : data) v = (int8_t)rng();
        FILE *fh = fopen(path.c_str(), "wb");
        EXPECT(fh && fwrite(data.data(), 1, data.size(), fh) == data.size());
        if (fh) fclose(fh);
        const int fd = open(path.c_str(), O_RDONLY);
        const int fdd = open(path.c_str(), O_RDONLY | O_DIRECT);
        EXPECT(fd >= 0);
        ThreadScratch sc;
        for (int rep = 0; rep < 200; ++rep) {
            const size_t off = (size_t)(rng() % (data.size() - 1)), len = 1 + (size_t)(rng() % std::min<size_t>(data.size() - off, 300000));
            std::vector<int8_t> got(len);
            EXPECT(pread_full(fd, got.data(), len, (off_t)off) == 0 && memcmp(got.data(), data.data() + off, len) == 0);
            if (fdd >= 0) {
                int8_t *buf = sc.get(len + 8192);
                const int8_t *at = nullptr;
                EXPECT(buf && pread_direct(fdd, buf, len, (off_t)off, &at) == 0 && memcmp(at, data.data() + off, len) == 0);
            }
        }
        std::vector<int8_t> got(100);
        EXPECT(pread_full(fd, got.data(), 100, (off_t)(data.size() - 50)) == -1);        // runs past the end of the file
        if (fd >= 0) close(fd);
        if (fdd >= 0) close(fdd);
        printf("o_direct %s\n", fdd >= 0 ? "used" : "refused by the file system");
    }
    printf("fails=%d\ndone\n", fails);
    return fails ? 1 : 0;
}
