// snpm_host.cpp -- entry points of libsnpmatch_hip.so that are pure host code (no HIP, no context): caller-side index
// preparation of the scoring path.  Compiled into the library by build_lib.sh and, together with
// snpm_vcf.cpp, into an AddressSanitizer / UBSan driver by the CPU test-suite (tests/test_host_sanitizers_cpu.py).
#include <cstdint>

#include "snpmatch_hip.h"

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
int snpm_intersect_sorted_search(const int64_t *a, int64_t na, const int64_t *b, int64_t nb, int64_t *ia, int64_t *ib,
                                 int64_t *n_out)
{
    if (na < 0 || nb < 0 || !n_out || ((na > 0 && nb > 0) && (!a || !b || !ia || !ib))) return SNPM_ERR_BADARG;
    for (int64_t j = 1; j < nb; ++j)
        if (b[j] <= b[j - 1]) return SNPM_ERR_STATE;
    int64_t lo = 0, k = 0;
    for (int64_t j = 0; j < nb && lo < na; ++j) {
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
    *n_out = k;
    return SNPM_OK;
}

}  // extern "C"
