// snpm_host.cpp -- entry points of libsnpmatch_hip.so that are pure host code (no HIP, no context): caller-side index
// preparation of the scoring path.  Compiled into the library by build_lib.sh and, together with
// snpm_vcf.cpp, into an AddressSanitizer / UBSan driver by the CPU test-suite (tests/test_host_sanitizers_cpu.py).
#include <sched.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "snpmatch_hip.h"
#include "snpm_hostpool.hpp"

namespace {
// one persistent pool for the context-free host entry points (threads are made on first use; their workers spin briefly for
// the next call -- the five chromosomes of a sample follow each other within microseconds -- before they sleep)
HostPool &shared_pool()
{
    unsigned hw = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) hw = (unsigned)CPU_COUNT(&set);
    static HostPool pool((int)std::max(0, std::min<int>((int)(hw ? hw : 1), 16) - 1));
    return pool;
}
}  // namespace

extern "C" {

// Sorted-merge intersection of two strictly increasing int64 arrays (position lists of one chromosome):
// ia/ib receive the indices of the common values, *n_out their number.  Pure host code (no ctx).
// Replaces the two np.in1d calls per chromosome of get_common_positions (core/snp_genotype.py:66-67).
int snpm_intersect_sorted(const int64_t *a, int64_t na, const int64_t *b, int64_t nb, int64_t *ia, int64_t *ib,
                          int64_t *n_out)
{
    if (na < 0 || nb < 0 || !n_out || ((na > 0 && nb > 0) && (!a || !b || !ia || !ib))) return SNPM_ERR_BADARG;
    for (int64_t i = 1; i < na; ++i)
        if (a[i] <= a[i - 1]) return SNPM_ERR_STATE;       // not strictly increasing: caller uses its generic path
    for (int64_t j = 1; j < nb; ++j)
        if (b[j] <= b[j - 1]) return SNPM_ERR_STATE;
    int64_t i = 0, j = 0, k = 0;
    while (i < na && j < nb) {
        if (a[i] < b[j]) ++i;
        else if (a[i] > b[j]) ++j;
        else { ia[k] = i; ib[k] = j; ++k; ++i; ++j; }
    }
    *n_out = k;
    return SNPM_OK;
}

// The same intersection for a short list b against a long list a (a 200k-SNP sample against an 11M-SNP DB
// chromosome set): galloping search of every b[j] from the previous hit, O(nb log(na / nb)).  a is NOT
// re-checked (the DB positions are verified once by the caller); b is.
// One galloping walk over b[j0, j1): hits go to ia / ib starting at slot j0 (a range can hold at most j1 - j0 of them);
// returns their number.  lo = where the walk starts in a (a[lo - 1] < b[j0] or lo == 0).
static int64_t gallop_range(const int64_t *a, int64_t na, const int64_t *b, int64_t j0, int64_t j1, int64_t lo, int64_t *ia,
                            int64_t *ib)
{
    int64_t k = j0;
    for (int64_t j = j0; j < j1 && lo < na; ++j) {
        const int64_t v = b[j];
        int64_t step = 1, hi = lo;                   // a[lo..] >= everything matched so far
        while (hi < na && a[hi] < v) { lo = hi + 1; hi += step; step <<= 1; }
        if (hi > na) hi = na;
        while (lo < hi) {                            // first index in [lo, hi) with a[idx] >= v
            const int64_t mid = lo + (hi - lo) / 2;
            if (a[mid] < v) lo = mid + 1; else hi = mid;
        }
        if (lo < na && a[lo] == v) { ia[k] = lo; ib[k] = j; ++k; ++lo; }
    }
    return k - j0;
}

int snpm_intersect_sorted_search(const int64_t *a, int64_t na, const int64_t *b, int64_t nb, int64_t *ia, int64_t *ib,
                                 int64_t *n_out)
{
    if (na < 0 || nb < 0 || !n_out || ((na > 0 && nb > 0) && (!a || !b || !ia || !ib))) return SNPM_ERR_BADARG;
    for (int64_t j = 1; j < nb; ++j)
        if (b[j] <= b[j - 1]) return SNPM_ERR_STATE;
    // Every probe of a long DB list is a cache miss of its own (a 200k-SNP sample against 11M DB positions: ~55 rows apart),
    // so the walk is cut into ranges of b that run on their own threads (each finds its start by a binary search) and the
    // per-range hit lists are closed up afterwards: the same pairs in the same order as the single walk.
    HostPool &pool = shared_pool();
    const int n_thr = (int)std::max<int64_t>(1, std::min<int64_t>(4 * (int64_t)(pool.size() + 1), nb / 2048));   // ranges, handed out dynamically
    // The ranges write their hits at slot j0 of ia / ib, i.e. they need room for nb entries; the contract promises the caller
    // min(na, nb).  With na < nb (a list shorter than the one searched for) the single walk runs: it fills ia / ib from slot 0.
    if (n_thr <= 1 || na < nb) {
        *n_out = gallop_range(a, na, b, 0, nb, 0, ia, ib);
        return SNPM_OK;
    }
    std::vector<int64_t> cnt((size_t)n_thr, 0), start((size_t)n_thr + 1, 0);
    for (int t = 0; t <= n_thr; ++t) start[(size_t)t] = nb * t / n_thr;
    auto work = [&](int t) {
        const int64_t j0 = start[(size_t)t], j1 = start[(size_t)t + 1];
        const int64_t lo = (j0 < j1) ? (int64_t)(std::lower_bound(a, a + na, b[j0]) - a) : 0;
        cnt[(size_t)t] = gallop_range(a, na, b, j0, j1, lo, ia, ib);
    };
    {
        // the pool runs one job at a time, and ctypes calls arrive without the GIL: callers on several Python threads take turns
        static std::mutex one_job;
        std::lock_guard<std::mutex> turn(one_job);
        pool.run(n_thr, work);
    }
    int64_t k = cnt[0];
    for (int t = 1; t < n_thr; ++t) {
        const int64_t j0 = start[(size_t)t];
        if (k != j0) {
            memmove(ia + k, ia + j0, (size_t)cnt[(size_t)t] * sizeof(int64_t));
            memmove(ib + k, ib + j0, (size_t)cnt[(size_t)t] * sizeof(int64_t));
        }
        k += cnt[(size_t)t];
    }
    *n_out = k;
    return SNPM_OK;
}

}  // extern "C"
