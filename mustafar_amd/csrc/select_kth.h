// select_kth.h -- k-th smallest fp16 magnitude of a 128-element row, bit-sliced (torch.kthvalue(|x|, k), reference
// models/llama_mustafar_kernel.py:103).  Plain C++ shared by the device code (compress.hip: one lane = one row, the row in
// 64 registers) and a host test (tests/test_select_kth.py compiles it with g++ against numpy).
//
// The search by value (thr |= bit iff fewer than k magnitudes are < thr | bit) costs a pass over the 64 registers per bit:
// 15 x 64 x 4 = 3 840 operations per row-lane, two thirds of everything the compression kernel does.  Sliced by bit the same
// search is 15 x ~21 operations on 128-bit sets:
//   planes   the 64 words (two 16-bit magnitudes each) are transposed as two 32 x 32 bit matrices (5 butterfly stages each):
//            word 31 - b of a transposed block = bit b of its 32 input words.  Plane k of the row = bit k of the low halfs and
//            bit 16 + k of the high halfs = four 32-bit words.  (Which element sits at which bit of a plane does not matter:
//            every plane uses the same order, and only counts and intersections are taken.)
//   search   S = candidates (all), r = k.  From bit 14 down: Z = S \ plane (candidates whose bit is 0, the smaller ones);
//            r <= |Z| ? the k-th smallest is among them, S = Z, the bit of the result is 0 : r -= |Z|, S = S & plane, bit = 1.
// The result is the same value the search by value finds: the v with |{m < v}| < k <= |{m <= v}|.
#ifndef MUSTAFAR_SELECT_KTH_H
#define MUSTAFAR_SELECT_KTH_H
#include <stdint.h>

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#define MUSTAFAR_HD __host__ __device__ __forceinline__
#else
#define MUSTAFAR_HD inline
#endif

// In-place transpose of a 32 x 32 bit matrix about its anti-diagonal: out[31 - b] bit (31 - i) = in[i] bit b.
MUSTAFAR_HD void bit_transpose32(uint32_t (&a)[32])
{
    uint32_t m = 0x0000ffffu;
#pragma unroll
    for (int j = 16; j != 0; j >>= 1) {
#pragma unroll
        for (int k = 0; k < 32; k = (k + j + 1) & ~j) {
            const uint32_t t = (a[k] ^ (a[k + j] >> j)) & m;
            a[k] ^= t;
            a[k + j] ^= t << j;
        }
        m ^= m << (j >> 1);
    }
}

// raw: the row as 64 words, element 2j in the low half of word j, element 2j + 1 in the high half (sign bits ignored).
// Returns the k-th smallest magnitude (1 <= kth <= 128) as a 15-bit integer.
MUSTAFAR_HD uint32_t kth_magnitude128(const uint32_t (&raw)[64], int kth)
{
    uint32_t a[32], b[32];
#pragma unroll
    for (int j = 0; j < 32; j++) { a[j] = raw[j]; b[j] = raw[32 + j]; }
    bit_transpose32(a);
    bit_transpose32(b);
    uint32_t s0 = ~0u, s1 = ~0u, s2 = ~0u, s3 = ~0u;
    int ns = 128, r = kth;
    uint32_t thr = 0;
#pragma unroll
    for (int k = 14; k >= 0; k--) {
        const uint32_t t0 = s0 & a[31 - k], t1 = s1 & a[15 - k], t2 = s2 & b[31 - k], t3 = s3 & b[15 - k];   // candidates with bit k set
        const int nt = __builtin_popcount(t0) + __builtin_popcount(t1) + __builtin_popcount(t2) + __builtin_popcount(t3);
        const int nz = ns - nt;
        const bool low = r <= nz;   // the k-th smallest has bit k clear
        s0 = low ? s0 ^ t0 : t0;
        s1 = low ? s1 ^ t1 : t1;
        s2 = low ? s2 ^ t2 : t2;
        s3 = low ? s3 ^ t3 : t3;
        ns = low ? nz : nt;
        r = low ? r : r - nz;
        thr |= low ? 0u : (1u << k);
    }
    return thr;
}
#endif
