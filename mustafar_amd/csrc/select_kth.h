// select_kth.h -- k-th smallest fp16 magnitude of a 128-element row, bit-sliced (torch.kthvalue(|x|, k), reference
// models/llama_mustafar_kernel.py:103).  Plain C++ shared by the device code (compress.hip: one lane = one row, the row in
// 64 registers) and a host test (tests/test_select_kth.py compiles it with g++ against numpy).
//
// The search by value (thr |= bit iff fewer than k magnitudes are < thr | bit) costs a pass over the 64 registers per bit:
// 15 x 64 x 4 = 3 840 operations per row-lane, two thirds of everything the compression kernel does.  Sliced by bit the same
// search is 15 x ~21 operations on 128-bit sets:
//   planes   the high bytes of the 128 magnitudes (four to a word: 32 words) and, later, the low bytes are transposed as 32 x 32 bit
//            matrices (5 butterfly stages each): word 31 - b of a transposed block = bit b of its 32 input words.  Plane k of the
//            row = bit k % 8 of the four byte positions = four 32-bit words.  (Which element sits at which bit of a plane does not
//            matter: every plane uses the same order, and only counts and intersections are taken.)
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
#if defined(__HIP_DEVICE_COMPILE__)
    // the stages that move whole bytes (j = 16, j = 8) are one byte permute per word instead of shift + select:
    //   j = 16: a' = (a & 0xffff0000) | (b >> 16),   b' = (b & 0x0000ffff) | (a << 16)
    //   j = 8 : a' = (a & 0xff00ff00) | ((b >> 8) & 0x00ff00ff),   b' = (b & 0x00ff00ff) | ((a << 8) & 0xff00ff00)
    // (v_perm_b32 D, S0, S1, sel: selector values 0..3 = bytes of S1, 4..7 = bytes of S0)
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t x = a[k], y = a[k + 16];
        a[k] = __builtin_amdgcn_perm(x, y, 0x07060302u);        // bytes: x3 x2 y3 y2
        a[k + 16] = __builtin_amdgcn_perm(x, y, 0x05040100u);   // bytes: x1 x0 y1 y0
    }
#pragma unroll
    for (int k = 0; k < 32; k = (k + 8 + 1) & ~8) {
        const uint32_t x = a[k], y = a[k + 8];
        a[k] = __builtin_amdgcn_perm(x, y, 0x07030501u);        // bytes: x3 y3 x1 y1
        a[k + 8] = __builtin_amdgcn_perm(x, y, 0x06020400u);    // bytes: x2 y2 x0 y0
    }
    m = 0x0f0f0f0fu;
#pragma unroll
    for (int j = 4; j != 0; j >>= 1) {
#else
#pragma unroll
    for (int j = 16; j != 0; j >>= 1) {
#endif
#pragma unroll
        for (int k = 0; k < 32; k = (k + j + 1) & ~j) {
            const uint32_t t = (a[k] ^ (a[k + j] >> j)) & m;
            a[k] ^= t;
            a[k + j] ^= t << j;
        }
        m ^= m << (j >> 1);
    }
}

// The bytes 1 and 3 (HI) or 0 and 2 (!HI) of a and b: the high / low bytes of four consecutive magnitudes in one word.
template <bool HI>
MUSTAFAR_HD uint32_t gather_bytes(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(b, a, HI ? 0x07050301u : 0x06040200u);   // (selector 0..3: bytes of the second operand, 4..7: of the first)
#else
    const int sh = HI ? 8 : 0;
    return ((a >> sh) & 0xffu) | (((a >> (16 + sh)) & 0xffu) << 8) | (((b >> sh) & 0xffu) << 16) | (((b >> (16 + sh)) & 0xffu) << 24);
#endif
}

// One phase of the search: bits KHI .. KLO of the magnitudes (KHI - KLO < 8), whose planes are p[31 - 8 q - (k - KLO) ... ] for the
// byte positions q = 0..3 of the gathered words.
template <int KHI, int KLO, int SHIFT>
MUSTAFAR_HD void kth_phase(const uint32_t (&p)[32], uint32_t& s0, uint32_t& s1, uint32_t& s2, uint32_t& s3, int& ns, int& r, uint32_t& thr)
{
#pragma unroll
    for (int k = KHI; k >= KLO; k--) {
        const int b = k - SHIFT;   // bit of the byte
        const uint32_t t0 = s0 & p[31 - b], t1 = s1 & p[23 - b], t2 = s2 & p[15 - b], t3 = s3 & p[7 - b];   // candidates with bit k set
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
}

// raw: the row as 64 words, element 2j in the low half of word j, element 2j + 1 in the high half (sign bits ignored).
// Returns the k-th smallest magnitude (1 <= kth <= 128) as a 15-bit integer.
// Round 4b: the planes are built in two rounds of ONE 32 x 32 transpose each -- the high bytes of the 128 magnitudes (bits 14..8), then,
// when those planes are dead, the low bytes (bits 7..0) -- so that 32 words are live next to the row instead of 64: with it the
// compression kernel fits four waves per SIMD.  Which element sits at which bit of a plane is the same in both rounds.
// SERIAL (device): a scheduling barrier between the rounds, for a caller that needs the second round's words NOT gathered while the
// first round's planes are live (prune_magnitude_kernel fits 128 registers with it; compress_block_kernel is faster without).
template <bool SERIAL = false>
MUSTAFAR_HD uint32_t kth_magnitude128(const uint32_t (&raw)[64], int kth)
{
    uint32_t p[32];
    uint32_t s0 = ~0u, s1 = ~0u, s2 = ~0u, s3 = ~0u;
    int ns = 128, r = kth;
    uint32_t thr = 0;
#pragma unroll
    for (int j = 0; j < 32; j++) p[j] = gather_bytes<true>(raw[2 * j], raw[2 * j + 1]);
    bit_transpose32(p);
    kth_phase<14, 8, 8>(p, s0, s1, s2, s3, ns, r, thr);
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (SERIAL) __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int j = 0; j < 32; j++) p[j] = gather_bytes<false>(raw[2 * j], raw[2 * j + 1]);
    bit_transpose32(p);
    kth_phase<7, 0, 0>(p, s0, s1, s2, s3, ns, r, thr);
    return thr;
}
#endif
