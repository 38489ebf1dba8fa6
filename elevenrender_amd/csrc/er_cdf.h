// er_cdf.h -- HDRI::binarySearch (reference src/HDRI.cpp:85-98) with far fewer dependent loads.
//
// The reference walks a 21-level binary search over the luminance CDF (2048x1024 HDRI): 21
// dependent 4-byte gathers per opaque bounce, the longest dependency chain of the shading
// step.  Its result is a pure function of (cdf, value), and every comparison it makes is
// decided by two indices: k = first index with cdf[k] >= value and k2 = first index with
// cdf[k2] > value (cdf is non-decreasing).  So:
//   1. a guide table (one entry per 1/B of the unit interval, built once on the host) brackets
//      k within a few entries -> 2-4 dependent loads instead of 21;
//   2. the reference's loop is then replayed in integer arithmetic, substituting
//      "m < k", "k <= m < k2", "m >= k2" for the three float comparisons.
// The returned index is identical to the reference's for every float `value` (tests/test_cdf.py
// checks it against the oracle's restatement, including values that equal CDF entries).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ER_CDF_HD __host__ __device__ inline
#else
#define ER_CDF_HD inline
#endif

// guide[j] = first index i in [0, length] with cdf[i] >= j / buckets (length if none); buckets is a power of two.
ER_CDF_HD int er_cdf_search(const float* cdf, int length, const uint32_t* guide, int buckets, float value) {
    int b = (int)(value * (float)buckets);          // exact: value has 24 significant bits, buckets = 2^n
    if (b < 0) b = 0;
    if (b > buckets - 1) b = buckets - 1;
    // k lies in [lo, hi]; the last bucket is left open (accumulated CDFs can end slightly above 1)
    int lo = (int)guide[b], hi = b == buckets - 1 ? length : (int)guide[b + 1];
    while (lo < hi) {                               // lower bound
        int m = lo + ((hi - lo) >> 1);
        if (cdf[m] < value) lo = m + 1; else hi = m;
    }
    const int k = lo;
    int k2 = k;
    if (k < length && cdf[k] == value) {            // run of entries equal to value: gallop to its end
        int step = 1, a = k, z = k + 1;
        while (z < length && cdf[z] == value) { a = z; z = z + step; step <<= 1; }
        if (z > length) z = length;
        // cdf[a] == value, and (z == length or cdf[z] > value): first index > value lies in (a, z]
        int l2 = a + 1, h2 = z;
        while (l2 < h2) {
            int m = l2 + ((h2 - l2) >> 1);
            if (cdf[m] > value) h2 = m; else l2 = m + 1;
        }
        k2 = l2;
    }
    // the reference's loop, comparisons replaced by index tests
    int from = 0, to = length - 1;
    while (to - from > 0) {
        int m = from + (to - from) / 2;
        if (m >= k && m < k2) return m;             // value == arr[m]
        if (m >= k2) to = m - 1;                    // value <  arr[m]
        else from = m + 1;                          // value >  arr[m]
    }
    return to;
}
