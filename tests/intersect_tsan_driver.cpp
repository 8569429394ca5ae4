// ThreadSanitizer driver of the pooled galloping intersection (snpmatch_amd/csrc/snpm_host.cpp: snpm_intersect_sorted_search
// cuts the searched list into ranges that run on the persistent pool's threads).  Built and run by
// tests/test_host_sanitizers_cpu.py; callers on two threads at once take turns on the pool, as ctypes callers without the GIL do.
#include <cstdint>
#include <cstdio>
#include <random>
#include <thread>
#include <vector>

#include "snpmatch_hip.h"

static int one_round(uint64_t seed, int na, int nb)
{
    std::mt19937_64 rng(seed);
    std::vector<int64_t> a, b;
    int64_t x = 0;
    for (int i = 0; i < na; ++i) { x += 1 + (int64_t)(rng() % 4); a.push_back(x); }
    x = 0;
    for (int i = 0; i < nb; ++i) { x += 1 + (int64_t)(rng() % 9); b.push_back(x); }
    const size_t cap = (size_t)(na < nb ? na : nb);
    std::vector<int64_t> ia(cap), ib(cap), ja(cap), jb(cap);
    int64_t k1 = -1, k2 = -1;
    if (snpm_intersect_sorted(a.data(), na, b.data(), nb, ia.data(), ib.data(), &k1) != SNPM_OK) return 1;
    if (snpm_intersect_sorted_search(a.data(), na, b.data(), nb, ja.data(), jb.data(), &k2) != SNPM_OK) return 1;
    if (k1 != k2) return 1;
    for (int64_t t = 0; t < k1; ++t)
        if (ia[(size_t)t] != ja[(size_t)t] || ib[(size_t)t] != jb[(size_t)t]) return 1;
    return 0;
}

int main()
{
    int fails = 0;
    const int shapes[][2] = {{200000, 16384}, {10, 20000}, {50000, 8192}, {8192, 8192}, {100000, 4096}};
    for (int r = 0; r < 5; ++r) fails += one_round(100 + r, shapes[r][0], shapes[r][1]);
    // two callers at once
    int f1 = 0, f2 = 0;
    std::thread t1([&] { for (int r = 0; r < 4; ++r) f1 += one_round(200 + r, 150000, 12000); });
    std::thread t2([&] { for (int r = 0; r < 4; ++r) f2 += one_round(300 + r, 90000, 9000); });
    t1.join();
    t2.join();
    fails += f1 + f2;
    printf("fails=%d\ndone\n", fails);
    return fails ? 1 : 0;
}
