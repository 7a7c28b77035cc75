// compress.hip -- per-token magnitude pruning and bitmap/offset compression of K and V for gfx950.
//
// Replaces, behind the C ABI of include/mustafar_hip.h:
//   dh_prune_key / dh_prune_value            models/llama_mustafar_kernel.py:77-153   (torch.kthvalue + 3 passes)
//   calculate_bitmap_{key,value}_batched     kernel/compression.py:8-115              (Triton, 64-lane programs)
//   compress_{key,value}_batched             kernel/compression.py:117-247            (Triton)
//   the torch glue between them              kernel/compression.py:294-309            (cumsum, cat, .item())
//
// wave64 makes the format's 64-element tile exactly one wavefront: __ballot() of (x != 0) IS the tile's
// bitmap (bit-reversed, the format is MSB-first), s_bcnt1 its population count and v_mbcnt the in-tile
// exclusive prefix that the Triton code computes with tl.cumsum.  All integer work; results are bit-exact.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mustafar_hip.h"
#include "select_kth.h"

namespace {

constexpr int kD       = 128;
constexpr int kThreads = 256;
constexpr int kWaves   = 4;

__device__ __forceinline__ bool nonzero_h(uint16_t v) { return (v & 0x7fffu) != 0; }   // -0.0 is zero, NaN is not

// ------------------------------------------------------------------------------------------------ prune
// thr = k-th smallest magnitude of a row of 128 halfs (torch.kthvalue(|x|, k), model :103); keep |x| >= thr (:107),
// pruned entries become x * 0 = sign-preserving zero (:110).  fp16 magnitudes are monotone as unsigned integers for
// non-NaN values, so thr is found bit by bit from the MSB: thr |= b  iff  fewer than k magnitudes are < (thr | b).
//
// Lane = row: one wave takes 64 rows, transposed through LDS half a row at a time (row stride 33 dwords: conflict-free both ways), and every
// lane finds the threshold of its own row held in 64 VGPRs (select_kth.h: bit planes + 128-bit candidate sets, ~1 300
// operations per row-lane; round 1 searched by value, 15 steps x 64 words x 4 operations, and before that a wave per row
// spent ~12 SALU + 4 VALU per step and per ROW on ballots and was instruction-bound at 1.7 TB/s).  Keep / prune is a SWAR
// compare: with the guard bit H = 0x8000 set in each half, (m | H) - (thr | thr << 16) keeps H in a half iff that
// magnitude >= thr (no borrow crosses the halves).
constexpr int kPruneRows  = 64;            // rows per wave
constexpr int kPruneLd    = kD / 4 + 1;    // LDS row stride in dwords: HALF a row at a time (round 4b: 8.4 KB per wave instead of 16.6 KB,
                                           // 16 waves per CU instead of 9 -- the kernel is a chain of load, transpose, search, transpose, store)

__global__ __launch_bounds__(64, 4) void prune_magnitude_kernel(const uint32_t* __restrict__ x, uint32_t* __restrict__ out,
                                                                int64_t n_rows, int kth)
{
    __shared__ uint32_t s_rows[kPruneRows * kPruneLd];
    const int lane = threadIdx.x;
    const uint32_t H = 0x80008000u, ONES = 0x00010001u;
    const int r0 = lane >> 3, c4 = lane & 7;   // a pass covers rows 8p .. 8p + 7: eight lanes x 16 B = one 128-byte half row each
    {   // (one group of 64 rows per workgroup -- a grid-stride loop here makes the compiler hoist ~40 loop-invariant addresses and spill them)
        const int64_t row0 = (int64_t)blockIdx.x * kPruneRows;
        const int rows = (int)((n_rows - row0) < kPruneRows ? (n_rows - row0) : kPruneRows);
        const uint4* src = reinterpret_cast<const uint4*>(x + row0 * (kD / 2));
        uint32_t raw[kD / 2];   // the lane's row, signs included (the search does not read the sign planes)
#pragma unroll
        for (int half = 0; half < 2; half++) {
            uint4 v[kPruneRows / 8];
            int piece = half * 8 + c4;
            asm volatile("" : "+v"(piece));   // (keeps the second half's loads behind the first half's transpose: 32 registers in flight, not 64)
#pragma unroll
            for (int p = 0; p < kPruneRows / 8; p++) {
                const int r = min(p * 8 + r0, rows - 1);   // (a short last group re-reads its last row: nothing of it is stored)
                v[p] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(src) + (uint32_t)(r * 16 + piece) * 16u);
            }
            __syncthreads();
#pragma unroll
            for (int p = 0; p < kPruneRows / 8; p++) {
                uint32_t* d = s_rows + (p * 8 + r0) * kPruneLd + c4 * 4;
                d[0] = v[p].x; d[1] = v[p].y; d[2] = v[p].z; d[3] = v[p].w;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kD / 4; j++) raw[half * (kD / 4) + j] = s_rows[lane * kPruneLd + j];
        }
        // k-th smallest magnitude, sliced by bit (select_kth.h)
        const uint32_t thr = kth_magnitude128<true>(raw, kth);
        // keep |x| >= thr, else the sign bit alone
        const uint32_t tt = thr | (thr << 16);
        uint4* dst = reinterpret_cast<uint4*>(out + row0 * (kD / 2));
#pragma unroll
        for (int half = 0; half < 2; half++) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kD / 4; j++) {
                const uint32_t w = raw[half * (kD / 4) + j];
                const uint32_t k = (((w | H) - tt) >> 15) & ONES;   // 1 per half that stays
                const uint32_t keep = (k << 16) - k;                 // 0xffff per half that stays
                s_rows[lane * kPruneLd + j] = w & (keep | H);
            }
            __syncthreads();
            int piece = half * 8 + c4;
            asm volatile("" : "+v"(piece));
#pragma unroll
            for (int p = 0; p < kPruneRows / 8; p++) {
                const int r = p * 8 + r0;
                const uint32_t* d = s_rows + r * kPruneLd + c4 * 4;
                if (r < rows) *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(dst) + (uint32_t)(r * 16 + piece) * 16u) = make_uint4(d[0], d[1], d[2], d[3]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ compression
// Two passes over the dense block, each ONE read of it (round 1 took three reads and one write: prune, bitmaps, pack):
//   pass 1  tile_meta_kernel : raw rows -> [per-row prune threshold, in registers] -> bitmaps, padded counts, the prefix
//                              of the counts inside the 64-token block, the block's total.  Nothing pruned is written.
//           block_scan_kernel: one workgroup per head turns the block totals into every block's base (+ head totals).
//   pass 2  tile_pack_kernel : raw rows + the bitmaps -> packed stream; adds the block's base to the block-local prefix
//                              (the offsets the SpMV kernels read).  A kept value IS the raw value, so the bitmap alone
//                              says what to pack: no pruned copy of the block exists anywhere.
// Destination rows: the reference's contiguous result tensors (bmp [B', 2t], accum [B', 2t + 1], stride = row length,
// tile0 = 0) or rows of a cache view with spare capacity, written behind the tiles already in use (in-place append).
struct Rows {
    int64_t bmp_stride;   // elements between the heads' bitmap rows
    int64_t idx_stride;   // elements between the heads' offset rows
    int64_t tile0;        // first tile of a row this call writes (2 * old_tokens)
};

// One side (K or V) of a pass-1 / pass-2 launch; a launch carries up to two sides in grid.z (the trigger compresses the
// 256 oldest window tokens of K and V with three launches in all).
struct Side {
    const uint16_t* x;          // rows of 128 halfs; head h starts at x + h * head_stride (a window buffer has spare rows)
    int64_t head_stride;        // elements
    int64_t* bmp;               // [B', rows.bmp_stride]
    int32_t* accum;             // [B', rows.idx_stride]
    int32_t* blk;               // [B', ntb] block totals (pass 1) -> block bases (scan); pass 2: nullptr = accum is final
    int64_t* totals;            // [B'] (+1 with the exclusive prefix) stream length of every head in halfs
    const int64_t* head_off;    // pass 2: start of every head's stream in halfs (fresh result tensor) ...
    const uint32_t* nz_offset;  //         ... or in uint4 units (cache view)
    uint16_t* nz;
    Rows rows;
    int64_t region_halfs;       // > 0: a head's stream may not grow beyond this (cache view); checked by the scan
    int kth;                    // > 0: prune first (k-th smallest magnitude, model :97-110); 0: rows are already pruned
    int key;                    // tile geometry: 1 = K (64 tokens of a channel), 0 = V (64 channels of a token)
};

// v_writelane_b32 with the lane as an immediate (two SGPR operands would break the constant-bus rule; this clang has no
// builtin for the instruction).
template <int LANE>
__device__ __forceinline__ void write_lane(uint32_t& dst, uint32_t src)
{
    asm("v_writelane_b32 %0, %1, %2" : "+v"(dst) : "s"(src), "i"(LANE));
}

// K geometry of pass 1: tile d = flag d of the 64 rows -> one ballot each; the mask of tile d goes to lane d % 64 (first
// 64 tiles: a, the others: b).  Compile-time recursion over the flag registers (the lane operand must be an immediate).
template <int J>
__device__ __forceinline__ void key_masks(const uint32_t (&f)[kD / 2], uint32_t& a_lo, uint32_t& a_hi, uint32_t& b_lo, uint32_t& b_hi)
{
    if constexpr (J < kD / 2) {
        const uint64_t m0 = __builtin_bitreverse64(__ballot((f[J] & 1u) != 0));    // element 2J
        const uint64_t m1 = __builtin_bitreverse64(__ballot((f[J] >> 16) != 0));   // element 2J + 1
        if constexpr (2 * J < 64) {
            write_lane<2 * J>(a_lo, (uint32_t)m0);     write_lane<2 * J>(a_hi, (uint32_t)(m0 >> 32));
            write_lane<2 * J + 1>(a_lo, (uint32_t)m1); write_lane<2 * J + 1>(a_hi, (uint32_t)(m1 >> 32));
        } else {
            write_lane<2 * J - 64>(b_lo, (uint32_t)m0); write_lane<2 * J - 64>(b_hi, (uint32_t)(m0 >> 32));
            write_lane<2 * J - 63>(b_lo, (uint32_t)m1); write_lane<2 * J - 63>(b_hi, (uint32_t)(m1 >> 32));
        }
        key_masks<J + 1>(f, a_lo, a_hi, b_lo, b_hi);
    }
}

// Pass 1.  One wave per 64-token block, lane = row (token): the row sits in 64 VGPRs (16 loads of 16 bytes; the lanes of
// one instruction touch 64 different lines, the next 7 instructions hit them in L1).  No LDS: 6-8 waves per SIMD hide the
// loads, where the LDS-transposed prune kernel runs 9 waves per CU.
//   K tile d   = element d of the 64 rows   -> one ballot; the 128 masks are handed to lanes d % 64 by v_writelane
//   V tiles    = the lane's own row halves  -> the lane packs its 2 x 64 flags itself (4 VALU ops per register)
// Either way lane l ends up owning tiles l and 64 + l of the block: counts, two wave scans, two coalesced stores.
__global__ __launch_bounds__(64, 6) void tile_meta_kernel(Side s0, Side s1, int ntb)
{
    const bool z = blockIdx.z != 0;
    const uint16_t* x = z ? s1.x : s0.x;
    const int64_t head_stride = z ? s1.head_stride : s0.head_stride;
    int64_t* bmp = z ? s1.bmp : s0.bmp;
    int32_t* accum = z ? s1.accum : s0.accum;
    int32_t* blk = z ? s1.blk : s0.blk;
    const Rows rows = z ? s1.rows : s0.rows;
    const int kth = z ? s1.kth : s0.kth;
    const int key = z ? s1.key : s0.key;
    const int lane = threadIdx.x, tb = blockIdx.x, h = blockIdx.y;
    const uint32_t H = 0x80008000u, ONES = 0x00010001u;

    const uint4* src = reinterpret_cast<const uint4*>(x + h * head_stride + ((int64_t)tb * 64 + lane) * kD);
    uint32_t w[kD / 2];   // magnitudes with the guard bits set
#pragma unroll
    for (int p = 0; p < kD / 8; p++) {
        const uint4 v = src[p];
        w[4 * p] = v.x; w[4 * p + 1] = v.y; w[4 * p + 2] = v.z; w[4 * p + 3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < kD / 2; j++)   // guard bits IN PLACE (a plain `| H` gets fresh registers: 128 live instead of 64)
        asm("v_or_b32 %0, %1, %0" : "+v"(w[j]) : "s"(H));
    uint32_t thr = 0;   // 0 keeps everything (rows already pruned)
    if (kth > 0) {      // bit-by-bit search of the k-th smallest magnitude, as prune_magnitude_kernel
#pragma unroll 1
        for (int bit = 14; bit >= 0; bit--) {
            const uint32_t c = thr | (1u << bit);
            const uint32_t cc = c | (c << 16);
            uint32_t ge = 0;
#pragma unroll
            for (int j = 0; j < kD / 2; j++) ge += ((w[j] - cc) >> 15) & ONES;
            const int below = kD - (int)((ge & 0xffffu) + (ge >> 16));
            if (below < kth) thr = c;
        }
    }
    // f[j]: bit 0 / bit 16 set iff element 2j / 2j + 1 is kept (|x| >= thr) and non-zero (-0.0 is zero)
    const uint32_t tt = thr | (thr << 16);
#pragma unroll
    for (int j = 0; j < kD / 2; j++) {
        const uint32_t keep = (w[j] - tt) >> 15;                      // guard bit survives iff magnitude >= thr
        const uint32_t nz = ((w[j] & 0x7fff7fffu) + 0x7fff7fffu) >> 15;   // carry into the guard position iff magnitude != 0
        w[j] = keep & nz & ONES;
    }
    uint32_t a_lo = 0, a_hi = 0, b_lo = 0, b_hi = 0;   // masks of the lane's two tiles (MSB = element 0)
    if (key) {
        key_masks<0>(w, a_lo, a_hi, b_lo, b_hi);
    } else {
#pragma unroll
        for (int j = 0; j < kD / 2; j++) {
            const uint32_t two = ((w[j] << 1) | (w[j] >> 16)) & 3u;   // (element 2j, element 2j + 1)
            const int e = 2 * (j & 31);                                // first of the two elements inside its tile
            uint32_t& word = (j < 32) ? (e < 32 ? a_hi : a_lo) : (e < 32 ? b_hi : b_lo);
            word |= two << (30 - (e & 31));
        }
    }
    const uint64_t ma = ((uint64_t)a_hi << 32) | a_lo, mb = ((uint64_t)b_hi << 32) | b_lo;
    int32_t ca = ((__popcll(ma) + 7) & ~7) >> 1, cb = ((__popcll(mb) + 7) & ~7) >> 1;   // compression.py:46-48
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {   // inclusive scans over the lanes
        const int32_t ua = __shfl_up(ca, o), ub = __shfl_up(cb, o);
        if (lane >= o) { ca += ua; cb += ub; }
    }
    const int32_t tot_a = __shfl(ca, 63);
    int64_t* bmp_row = bmp + h * rows.bmp_stride + rows.tile0 + (int64_t)tb * kD;
    int32_t* acc_row = accum + h * rows.idx_stride + rows.tile0 + (int64_t)tb * kD;
    bmp_row[lane] = (int64_t)ma;
    bmp_row[64 + lane] = (int64_t)mb;
    acc_row[lane + 1] = ca;                  // prefix INSIDE the block; pass 2 adds the block's base
    acc_row[64 + lane + 1] = tot_a + cb;
    if (lane == 63) blk[(int64_t)h * ntb + tb] = tot_a + cb;
}

// ------------------------------------------------------------------------------------------------ scan
// One workgroup per (head, side): block totals -> exclusive bases (torch.cumsum + cat, compression.py:294-298, at block
// granularity; the prefix inside the blocks is already there).  Append mode (rows.tile0 > 0): the head's entry tile0
// already holds the total of the tiles in use (model :352-360) and is the first base.  Also the head's new stream length
// (compression.py:302) and, for a cache view, the check that it still fits the head's region.
__global__ __launch_bounds__(kThreads) void block_scan_kernel(Side s0, Side s1, int ntb, int32_t* __restrict__ overflow)
{
    __shared__ int32_t s_wave[kWaves];
    const bool z = blockIdx.y != 0;
    int32_t* blk = z ? s1.blk : s0.blk;
    int32_t* accum = z ? s1.accum : s0.accum;
    int64_t* totals = z ? s1.totals : s0.totals;
    const Rows rows = z ? s1.rows : s0.rows;
    const int64_t region = z ? s1.region_halfs : s0.region_halfs;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t* bt = blk + (int64_t)blockIdx.x * ntb;
    int32_t* a = accum + (int64_t)blockIdx.x * rows.idx_stride + rows.tile0;
    int32_t carry = rows.tile0 ? a[0] : 0;
    for (int base = 0; base < ntb; base += kThreads) {
        const int i = base + threadIdx.x;
        const int32_t own = (i < ntb) ? bt[i] : 0;
        int32_t v = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {   // inclusive scan inside the wave
            const int32_t u = __shfl_up(v, o);
            if (lane >= o) v += u;
        }
        if (lane == 63) s_wave[wave] = v;
        __syncthreads();
        int32_t add = carry;
        for (int w = 0; w < wave; w++) add += s_wave[w];
        const int32_t tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        if (i < ntb) bt[i] = v - own + add;   // exclusive: the base of block i
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (rows.tile0 == 0) a[0] = 0;
        totals[blockIdx.x] = 2 * (int64_t)carry;   // halfs in this head's stream (compression.py:302)
        if (region > 0 && 2 * (int64_t)carry > region && overflow) atomicOr(overflow, 1);
    }
}

// head_off[h] = exclusive prefix of totals (compression.py:303-304), head_off[B'] = grand total.  In place, one workgroup.
// `mirror` (or null; round 6): the same B' + 1 values stored a second time, at SYSTEM scope, into device-visible HOST memory -- the caller of
// mustafar_compress_bitmap_mirrored polls it instead of copying head_off back behind the stream (each value one aligned 8-byte store: a
// reader sees an entry's old or new value, never a torn one, whatever order the entries arrive in).
// `flag` (with a mirror only): a device word the launches in front of this one have finished with; mirror[B' + 1] = its value, zero-extended.
__global__ __launch_bounds__(kThreads) void head_offsets_kernel(int64_t* __restrict__ head_off, int Bp, int64_t* mirror = nullptr,
                                                                const int32_t* flag = nullptr)
{
    __shared__ int64_t s_wave[kWaves];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t carry = 0;
    for (int base = 0; base < Bp; base += kThreads) {
        const int i = base + threadIdx.x;
        const int64_t own = (i < Bp) ? head_off[i] : 0;
        int64_t v = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int64_t u = __shfl_up(v, o);
            if (lane >= o) v += u;
        }
        if (lane == 63) s_wave[wave] = v;
        __syncthreads();
        int64_t add = carry;
        for (int w = 0; w < wave; w++) add += s_wave[w];
        const int64_t tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        if (i < Bp) {
            head_off[i] = v - own + add;
            if (mirror) __hip_atomic_store(mirror + i, v - own + add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        head_off[Bp] = carry;
        if (mirror) {
            __hip_atomic_store(mirror + Bp, carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (flag) __hip_atomic_store(mirror + Bp + 1, (int64_t)(uint32_t)*flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ------------------------------------------------------------------------------------------------ pack
// Non-zeros of a tile in ascending element order at stream offset 2*accum[tile] (compression.py:164-174), then zeros up
// to ceil8(nnz) (the reference relies on a pre-zeroed buffer, :309; here the wave writes them).  `m` = the tile's bitmap
// as stored (MSB = element 0), wave-uniform; v = the lane's raw element.
__device__ __forceinline__ void pack_tile(uint16_t* __restrict__ dst, uint64_t m, uint16_t v, int lane)
{
    const uint64_t mr = __builtin_bitreverse64(m);   // bit l <=> element l
    const int nnz = __popcll(mr);
    const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mr >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mr, 0u));
    if ((mr >> lane) & 1ull) dst[rank] = v;
    const int padded = (nnz + 7) & ~7;
    if (lane >= nnz && lane < padded) dst[lane] = 0;
}

// The 64x128 K block staged through LDS (row stride 65 dwords: conflict-free column reads).
constexpr int kRowWords = kD / 2 + 1;

__device__ __forceinline__ void load_block_transposable(uint32_t* s_blk, const uint16_t* __restrict__ xb)
{
    // 64 rows x 256 B; thread i copies 16-byte pieces i, i+256, ... (coalesced), scattered into padded rows
    const uint4* src = reinterpret_cast<const uint4*>(xb);
    for (int p = threadIdx.x; p < 64 * (kD / 8); p += kThreads) {
        const uint4 v = src[p];
        const int row = p / (kD / 8), c4 = p % (kD / 8);
        uint32_t* dst = s_blk + row * kRowWords + c4 * 4;
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
}

__device__ __forceinline__ uint16_t block_elem(const uint32_t* s_blk, int token, int d)
{
    const uint32_t w = s_blk[token * kRowWords + (d >> 1)];
    return (uint16_t)((d & 1) ? (w >> 16) : (w & 0xffffu));
}

// Pass 2.  grid: x = token block, y = head, z = side; 256 threads.  `overflow` set (a head outgrew its region of a cache
// view) -> nothing is written: the caller re-houses the cache and repeats the append.
__global__ __launch_bounds__(kThreads) void tile_pack_kernel(Side s0, Side s1, int ntb, const int32_t* __restrict__ overflow)
{
    __shared__ uint32_t s_blk[64 * kRowWords];
    __shared__ uint64_t s_bmp[kD];
    __shared__ int32_t s_start[kD];   // stream start of every tile of the block, half2 units, before the base is added
    if (overflow && *overflow) return;
    const bool z = blockIdx.z != 0;
    const uint16_t* x = z ? s1.x : s0.x;
    const int64_t head_stride = z ? s1.head_stride : s0.head_stride;
    const int64_t* bmp = z ? s1.bmp : s0.bmp;
    int32_t* accum = z ? s1.accum : s0.accum;
    const int32_t* blk = z ? s1.blk : s0.blk;
    const int64_t* head_off = z ? s1.head_off : s0.head_off;
    const uint32_t* nz_offset = z ? s1.nz_offset : s0.nz_offset;
    uint16_t* nz = z ? s1.nz : s0.nz;
    const Rows rows = z ? s1.rows : s0.rows;
    const int key = z ? s1.key : s0.key;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tb = blockIdx.x, h = blockIdx.y;
    const uint16_t* xb = x + h * head_stride + (int64_t)tb * 64 * kD;
    const int32_t base = blk ? blk[(int64_t)h * ntb + tb] : 0;
    int32_t* acc_row = accum + (int64_t)h * rows.idx_stride + rows.tile0 + (int64_t)tb * kD;
    if (key) load_block_transposable(s_blk, xb);
    if (threadIdx.x < kD) {
        const int d = threadIdx.x;
        s_bmp[d] = (uint64_t)bmp[h * rows.bmp_stride + rows.tile0 + (int64_t)tb * kD + d];
        if (blk) {
            const int32_t incl = acc_row[d + 1];      // block-local inclusive prefix left by pass 1
            if (d + 1 < kD) s_start[d + 1] = incl;
            if (d == 0) s_start[0] = 0;
            acc_row[d + 1] = base + incl;              // the offsets of the format (exclusive prefix over the whole head)
        } else {
            s_start[d] = acc_row[d];                   // two-call form: the fix-up pass has already added the bases
        }
    }
    __syncthreads();
    uint16_t* nz_h = nz + (head_off ? head_off[h] : 8 * (int64_t)nz_offset[h]);
    if (key) {
        for (int d = wave; d < kD; d += kWaves)
            pack_tile(nz_h + 2 * (int64_t)(base + s_start[d]), s_bmp[d], block_elem(s_blk, lane, d), lane);
    } else {
        for (int r = wave; r < 64; r += kWaves) {
            pack_tile(nz_h + 2 * (int64_t)(base + s_start[r]), s_bmp[r], xb[r * kD + lane], lane);
            pack_tile(nz_h + 2 * (int64_t)(base + s_start[64 + r]), s_bmp[64 + r], xb[r * kD + 64 + lane], lane);
        }
    }
}

// ------------------------------------------------------------------------------------------------ one pass
// Prune + compress + pack of a 64-token block in ONE read of it (the fused forms: mustafar_cache_append_kv).  Both passes
// above are bound by vector instructions, not bytes (pass 1 ~5 800 per block, pass 2 ~3 200: 136 us per side at c3 against
// 40 us of traffic), and pass 2 re-derives from memory what pass 1 had in registers.  Here one wave keeps the 64 raw rows
// in registers from the load to the packed stream:
//   threshold   lane = row, sliced by bit (select_kth.h): the bytes of the row are transposed into bit planes and the
//               k-th smallest magnitude is read off 128-bit candidate sets
//   K           tile d = element d of the 64 rows: one ballot; first pass: the mask and the tile's start go to lane d % 64
//               (v_writelane) and the lengths add up; second pass, half a block at a time: the lane's own element goes to
//               rank(lane) of the tile in an LDS image of that half of the block's stream
//   V           tiles = the lane's own row halves: the lane packs its 128 flags into masks, a scan over the lanes gives the
//               starts; then it walks its flags again and appends the kept values itself, first half-row, then second
//   base        the stream position of the block = sum of the lengths of the blocks in front of it in the head.  Every
//               block publishes its length as soon as it is known -- one 8-byte {length, valid} word, one sc1 store -- and
//               reads the words of its predecessors (sc1 loads, polled).  Nobody waits before publishing, so the chain is one
//               hop long whatever the number of blocks; workgroups are dispatched in linear order and a predecessor has a
//               smaller linear id, so what a block waits for is running or done.  The poll is bounded all the same.
//   flush       each half of the LDS image (one contiguous range of the output) leaves as 16-byte stores
constexpr int kSpinMax = 1 << 22;

__device__ __forceinline__ void gran_publish(uint64_t* g, uint32_t len)
{
    __hip_atomic_store(g, (1ull << 32) | len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t gran_load(const uint64_t* g) { return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Keep test of a word (two halfs), SWAR: with the guard bit H = 0x8000 set in each half, (m | H) - tt keeps H in a half iff that
// magnitude >= the threshold (no borrow crosses the halves).  tt holds max(thr, 1) in both halfs: a kept element is one with
// magnitude >= thr that is not zero (-0.0 is zero), and for thr >= 1 the first implies the second, for thr = 0 "magnitude >= 1" IS
// the second -- two operations per word instead of eight.  Bit 15 / bit 31 of the result = element 2j / 2j + 1 is kept.
__device__ __forceinline__ uint32_t keep_bits(uint32_t raw, uint32_t tt) { return (raw | 0x80008000u) - tt; }
__device__ __forceinline__ bool kept_lo(uint32_t d) { return (d & 0x8000u) != 0; }
__device__ __forceinline__ bool kept_hi(uint32_t d) { return (int32_t)d < 0; }
// K geometry, first pass over registers J .. 63 of the block: masks, starts and the block's length, nothing written to LDS.
// tt = the row's threshold in both halfs (the keep flags are recomputed from it).  `run` = stream position (half2 units, inside the block) of the next tile:
// wave-uniform.  half_run: `run` in front of tile 64 (where the second half of the block's image begins).
template <int J>
__device__ __forceinline__ void key_count(const uint32_t tt, const uint32_t (&raw)[kD / 2], uint32_t& run, uint32_t& half_run, uint32_t& a_lo, uint32_t& a_hi,
                                          uint32_t& b_lo, uint32_t& b_hi, uint32_t& sa, uint32_t& sb)
{
    if constexpr (J < kD / 2) {
        if constexpr (J == kD / 4) half_run = run;
        const uint32_t fj = keep_bits(raw[J], tt);
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const bool kept = half ? kept_hi(fj) : kept_lo(fj);
            const uint64_t m = __ballot(kept);                       // bit l <=> token l of the block has element 2J + half
            const int padded = (__popcll(m) + 7) & ~7;
            const uint64_t mr = __builtin_bitreverse64(m);            // as stored: MSB = element 0
            constexpr int D = 2 * J;
            if (half == 0) {
                if constexpr (D < 64) { write_lane<D>(a_lo, (uint32_t)mr); write_lane<D>(a_hi, (uint32_t)(mr >> 32)); write_lane<D>(sa, run); }
                else { write_lane<D - 64>(b_lo, (uint32_t)mr); write_lane<D - 64>(b_hi, (uint32_t)(mr >> 32)); write_lane<D - 64>(sb, run); }
            } else {
                if constexpr (D < 64) { write_lane<D + 1>(a_lo, (uint32_t)mr); write_lane<D + 1>(a_hi, (uint32_t)(mr >> 32)); write_lane<D + 1>(sa, run); }
                else { write_lane<D - 63>(b_lo, (uint32_t)mr); write_lane<D - 63>(b_hi, (uint32_t)(mr >> 32)); write_lane<D - 63>(sb, run); }
            }
            run += padded >> 1;
        }
        key_count<J + 1>(tt, raw, run, half_run, a_lo, a_hi, b_lo, b_hi, sa, sb);
    }
}
// Second pass over registers J .. JE - 1: the lane's own element goes to rank(lane) of its tile in the LDS image of this HALF of
// the block's stream (s_img, 8 KB: 64 tiles of at most 64 halfs); `run` restarts at 0 with the half.
template <int J, int JE>
__device__ __forceinline__ void key_fill(const uint32_t tt, const uint32_t (&raw)[kD / 2], uint16_t* s_img, int lane, uint32_t& run)
{
    if constexpr (J < JE) {
        const uint32_t fj = keep_bits(raw[J], tt);
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const bool kept = half ? kept_hi(fj) : kept_lo(fj);
            const uint64_t m = __ballot(kept);
            const int cnt = __popcll(m), padded = (cnt + 7) & ~7;
            const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            uint16_t* tile = s_img + 2 * run;
            if (kept) tile[rank] = (uint16_t)(half ? raw[J] >> 16 : raw[J]);
            if (lane >= cnt && lane < padded) tile[lane] = 0;         // compression.py:309 relies on a pre-zeroed buffer
            run += padded >> 1;
        }
        key_fill<J + 1, JE>(tt, raw, s_img, lane, run);
    }
}

// grid: x = token block, y = head, z = side; ONE wave.  gran: [sides][B'][ntb] words, zero at launch.  overflow: bit 0 = a
// head outgrew its region (nothing of the offending blocks is written; the caller re-houses and repeats), bit 1 = a poll ran out.
// Round 4b: four waves per SIMD instead of two (16 per CU cover for each other's loads, polls and stores: c3 158 -> 103 us) --
//   * the block's stream leaves in TWO halves (tiles 0..63, tiles 64..127) through an 8 KB LDS image instead of one of 16 KB: the
//     lengths are counted before anything is packed;
//   * the keep flags are recomputed from the threshold wherever they are needed (6 operations per word) instead of kept in 64 registers;
//   * the threshold search holds 32 plane words next to the row, not 64 (select_kth.h);
//   125 vector registers, no spills (a build that reaches the register budget by spilling is 30 % SLOWER than the one it came from).
#ifndef MUSTAFAR_CB_WAVES
#define MUSTAFAR_CB_WAVES 4
#endif
__device__ __forceinline__ void flush_image(const uint16_t* s_img, uint16_t* dst_halfs, uint32_t n_half2, int lane)
{
    uint4* dst = reinterpret_cast<uint4*>(dst_halfs);
    const int n16 = (int)(n_half2 >> 2);   // every tile is padded to 8 halfs = 16 bytes
    for (int p = lane; p < n16; p += 64) dst[p] = reinterpret_cast<const uint4*>(s_img)[p];
}
__global__ __launch_bounds__(64, MUSTAFAR_CB_WAVES) void compress_block_kernel(Side s0, Side s1, int ntb, uint64_t* __restrict__ gran, int32_t* __restrict__ overflow,
                                                                             int skip_publish_tb)   // (tests: this block keeps its length to itself -> its successors time out)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_img[64 * 64];   // one half of the block's stream, worst case (nothing pruned)
    const bool z = blockIdx.z != 0;
    const uint16_t* x = z ? s1.x : s0.x;
    const int64_t head_stride = z ? s1.head_stride : s0.head_stride;
    int64_t* bmp = z ? s1.bmp : s0.bmp;
    int32_t* accum = z ? s1.accum : s0.accum;
    int64_t* totals = z ? s1.totals : s0.totals;
    const uint32_t* nz_offset = z ? s1.nz_offset : s0.nz_offset;
    uint16_t* nz = z ? s1.nz : s0.nz;
    const Rows rows = z ? s1.rows : s0.rows;
    const int64_t region = z ? s1.region_halfs : s0.region_halfs;
    const int kth = z ? s1.kth : s0.kth;
    const int key = z ? s1.key : s0.key;
    const int lane = threadIdx.x, tb = blockIdx.x, h = blockIdx.y;

    const uint4* src = reinterpret_cast<const uint4*>(x + h * head_stride + ((int64_t)tb * 64 + lane) * kD);
    uint32_t raw[kD / 2];
#pragma unroll
    for (int p = 0; p < kD / 8; p++) {
        const uint4 v = src[p];
        raw[4 * p] = v.x; raw[4 * p + 1] = v.y; raw[4 * p + 2] = v.z; raw[4 * p + 3] = v.w;
    }
    // k-th smallest magnitude of the lane's row, sliced by bit (select_kth.h: ~1 300 operations against the 3 840 of the
    // search by value in tile_meta_kernel); 0 keeps everything (rows already pruned)
    const uint32_t thr = kth > 0 ? kth_magnitude128(raw, kth) : 0u;
    const uint32_t thr1 = thr > 1u ? thr : 1u;          // (keep_bits: "kept" = magnitude >= thr and not zero)
    uint32_t tt = thr1 | (thr1 << 16);
    uint32_t a_lo = 0, a_hi = 0, b_lo = 0, b_hi = 0;   // masks of tiles lane and 64 + lane (MSB = element 0)
    uint32_t sa = 0, sb = 0;                            // their starts inside the block, half2 units
    uint32_t total = 0, total_a = 0;                    // the block's length / the length of its first 64 tiles, half2 units (wave-uniform)
    int na = 0, nb = 0;
    // (the keep bits of a word are recomputed from tt wherever they are needed: kept in registers they cost 64 of them)
    // ---- lengths first: nothing is packed before the block's length is published
    if (key) {
        key_count<0>(tt, raw, total, total_a, a_lo, a_hi, b_lo, b_hi, sa, sb);
    } else {
#pragma unroll
        for (int j = 0; j < kD / 2; j++) {
            const uint32_t wj = keep_bits(raw[j], tt);
            const uint32_t two = ((wj >> 14) & 2u) | (wj >> 31);   // (element 2j, element 2j + 1)
            const int e = 2 * (j & 31);
            uint32_t& word = (j < 32) ? (e < 32 ? a_hi : a_lo) : (e < 32 ? b_hi : b_lo);
            word |= two << (30 - (e & 31));
        }
        na = __popc(a_lo) + __popc(a_hi);
        nb = __popc(b_lo) + __popc(b_hi);
        const int32_t pa = ((na + 7) & ~7) >> 1, pb = ((nb + 7) & ~7) >> 1;   // compression.py:46-48
        int32_t ca = pa, cb = pb;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {   // inclusive scans over the lanes
            const int32_t ua = __shfl_up(ca, o), ub = __shfl_up(cb, o);
            if (lane >= o) { ca += ua; cb += ub; }
        }
        total_a = (uint32_t)__shfl(ca, 63);
        total = total_a + (uint32_t)__shfl(cb, 63);
        sa = (uint32_t)(ca - pa);
        sb = total_a + (uint32_t)(cb - pb);
    }
    // ---- publish the block's length
    uint64_t* g = gran + ((int64_t)(z ? gridDim.y : 0) + h) * ntb;
    if (lane == 0 && tb != skip_publish_tb) gran_publish(g + tb, total);
    // ---- first half of the image (tiles 0..63) while the blocks in front finish counting
    if (key) {
        uint32_t run = 0;
        asm volatile("" : "+v"(tt));   // (the flags are recomputed, not kept from the pass above)
        key_fill<0, kD / 4>(tt, raw, s_img, lane, run);
    } else {
        // the lane appends the kept values of its first tile in element order, then the zeros up to the padded length
        asm volatile("" : "+v"(tt));   // (the flags are recomputed, not kept from the counting loop above)
        uint32_t cur = 2 * sa;         // halfs
#pragma unroll
        for (int j = 0; j < kD / 4; j++) {
            const uint32_t wj = keep_bits(raw[j], tt);
            if (kept_lo(wj)) s_img[cur] = (uint16_t)raw[j];
            cur += (wj >> 15) & 1u;
            if (kept_hi(wj)) s_img[cur] = (uint16_t)(raw[j] >> 16);
            cur += wj >> 31;
        }
        const int pa2 = (na + 7) & ~7;
#pragma unroll
        for (int i = 0; i < 7; i++) if (na + i < pa2) s_img[cur + i] = 0;
    }
    // ---- collect the lengths in front of the block
    int32_t* head_acc = accum + (int64_t)h * rows.idx_stride + rows.tile0;
    uint32_t sum = 0;
    bool failed = false;
    for (int b0 = 0; b0 < tb; b0 += 64) {
        const int b = b0 + lane;
        if (b < tb) {
            uint64_t v = gran_load(g + b);
            for (int spin = 0; !(v >> 32) && spin < kSpinMax; spin++) {
                __builtin_amdgcn_s_sleep(2);
                v = gran_load(g + b);
            }
            failed |= !(v >> 32);
            sum += (uint32_t)v;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const uint32_t base = (rows.tile0 ? (uint32_t)head_acc[0] : 0u) + sum;   // (append: entry tile0 holds the head's length so far, model :352-360)
    if (__ballot(failed)) {
        if (lane == 0 && overflow) atomicOr(overflow, 2);
        return;
    }
    // ---- metadata of the block's 128 tiles: bitmaps, offsets (exclusive prefix over the whole head, compression.py:294-298)
    int64_t* bmp_row = bmp + h * rows.bmp_stride + rows.tile0 + (int64_t)tb * kD;
    int32_t* acc_row = head_acc + (int64_t)tb * kD;
    bmp_row[lane] = (int64_t)(((uint64_t)a_hi << 32) | a_lo);
    bmp_row[64 + lane] = (int64_t)(((uint64_t)b_hi << 32) | b_lo);
    acc_row[lane] = (int32_t)(base + sa);          // (entry 0 of a block = entry 128 of the block in front: same value from both)
    acc_row[64 + lane] = (int32_t)(base + sb);
    if (lane == 0) acc_row[kD] = (int32_t)(base + total);
    const int64_t end_halfs = 2 * (int64_t)(base + total);
    if (tb == ntb - 1 && lane == 0) totals[h] = end_halfs;   // the head's new stream length (compression.py:302)
    if (region > 0 && end_halfs > region) {   // a cache view whose region the head has outgrown: nothing beyond it is written
        if (lane == 0 && overflow) atomicOr(overflow, 1);
        return;
    }
    uint16_t* dst = nz + 8 * (int64_t)nz_offset[h] + 2 * (int64_t)base;
    flush_image(s_img, dst, total_a, lane);
    // ---- second half (tiles 64..127): the same 8 KB (LDS operations of a wave execute in order: the reads of the flush are done)
    if (key) {
        uint32_t run = 0;
        asm volatile("" : "+v"(tt));
        key_fill<kD / 4, kD / 2>(tt, raw, s_img, lane, run);
    } else {
        asm volatile("" : "+v"(tt));
        uint32_t cur = 2 * (sb - total_a);
#pragma unroll
        for (int j = kD / 4; j < kD / 2; j++) {
            const uint32_t wj = keep_bits(raw[j], tt);
            if (kept_lo(wj)) s_img[cur] = (uint16_t)raw[j];
            cur += (wj >> 15) & 1u;
            if (kept_hi(wj)) s_img[cur] = (uint16_t)(raw[j] >> 16);
            cur += wj >> 31;
        }
        const int pb2 = (nb + 7) & ~7;
#pragma unroll
        for (int i = 0; i < 7; i++) if (nb + i < pb2) s_img[cur + i] = 0;
    }
    flush_image(s_img, dst + 2 * (int64_t)total_a, total - total_a, lane);
}

// Move the rows [drop, len) of every head's window to the front (model :392-393 slices and clones; here in place).
// One workgroup per (head, side).  The ranges may overlap (more rows stay than leave: a residual_length above 256), so the move goes in
// ascending pieces of 4 x kThreads x 16 bytes, each read in full before it is written: a piece's destination lies below everything a
// later piece reads.
__device__ __forceinline__ void slide_rows(uint16_t* win, int len, int drop)
{
    const int n16 = (len - drop) * (kD / 8);   // 16-byte pieces to move
    const uint4* src = reinterpret_cast<const uint4*>(win + (int64_t)drop * kD);
    uint4* dst = reinterpret_cast<uint4*>(win);
    for (int base = 0; base < n16; base += 4 * kThreads) {   // (workgroup-uniform trip count: every thread reaches the barriers)
        uint4 v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int p = base + threadIdx.x + i * kThreads;
            if (p < n16) v[i] = src[p];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int p = base + threadIdx.x + i * kThreads;
            if (p < n16) dst[p] = v[i];
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(kThreads) void window_drop_front_kernel(uint16_t* k_win, uint16_t* v_win, int64_t head_stride, int len, int drop)
{
    slide_rows((blockIdx.y ? v_win : k_win) + blockIdx.x * head_stride, len, drop);
}

// The tail of a trigger that grew the cache by an extent (cache.py: append_extent_pairs): the window slide of both sides and, by
// one workgroup, the extent's two views into their slots of the device tables -- one launch per layer, no host copy.
__global__ __launch_bounds__(kThreads) void trigger_finish_kernel(uint16_t* k_win, uint16_t* v_win, int64_t head_stride, int len, int drop,
                                                                  mustafar_cache_view k_view, mustafar_cache_view v_view,
                                                                  mustafar_cache_view* k_slot, mustafar_cache_view* v_slot)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const mustafar_cache_view& view = blockIdx.y ? v_view : k_view;
        mustafar_cache_view* slot = blockIdx.y ? v_slot : k_slot;
        if (slot) *slot = view;
    }
    if (drop <= 0 || len <= drop) return;
    slide_rows((blockIdx.y ? v_win : k_win) + blockIdx.x * head_stride, len, drop);
}

// Re-housing of a cache (cache.py: an arena moved into larger rows / regions): the three arrays of every head in ONE launch.
// grid: x = 16-KiB pieces of a head's largest array, y = head, z = array (0 bitmaps, 1 offsets, 2 stream); z = 0 also writes
// the destination's nz_offset (equally spaced regions).
__global__ __launch_bounds__(kThreads) void cache_rehouse_kernel(const uint64_t* s_bmp, const uint32_t* s_idx, const uint16_t* s_nz,
                                                                 const uint32_t* s_off, int64_t s_bmp_stride, int64_t s_idx_stride,
                                                                 uint64_t* d_bmp, uint32_t* d_idx, uint16_t* d_nz, uint32_t* d_off,
                                                                 int64_t d_bmp_stride, int64_t d_idx_stride, uint32_t d_nz_stride,
                                                                 int64_t tiles, int64_t nz_halfs)
{
    const int h = blockIdx.y, which = blockIdx.z;
    static_assert(kThreads == 256, "a workgroup of this kernel moves pieces of 4 x kThreads = 1024 16-byte units (the host sizes grid.x for that)");
    const int64_t piece = (int64_t)blockIdx.x * (4 * kThreads) + threadIdx.x;   // 16-byte units, 4 per thread and piece
    if (which == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) d_off[h] = (uint32_t)h * d_nz_stride;
        const uint4* src = reinterpret_cast<const uint4*>(s_bmp + h * s_bmp_stride);
        uint4* dst = reinterpret_cast<uint4*>(d_bmp + h * d_bmp_stride);
        const int64_t n = tiles / 2;                                  // (tiles is a multiple of 128; rows start 16-byte aligned)
#pragma unroll
        for (int i = 0; i < 4; i++) { const int64_t p = piece + i * 256; if (p < n) dst[p] = src[p]; }
    } else if (which == 1) {
        const uint32_t* src = s_idx + h * s_idx_stride;               // (rows of 2 cap + 1 words: no alignment to speak of)
        uint32_t* dst = d_idx + h * d_idx_stride;
        const int64_t n = tiles + 1;
#pragma unroll
        for (int i = 0; i < 16; i++) { const int64_t p = (int64_t)blockIdx.x * 4096 + threadIdx.x + i * 256; if (p < n) dst[p] = src[p]; }
    } else {
        const uint4* src = reinterpret_cast<const uint4*>(s_nz) + (int64_t)s_off[h];
        uint4* dst = reinterpret_cast<uint4*>(d_nz) + (int64_t)h * d_nz_stride;
        const int64_t n = nz_halfs / 8;
#pragma unroll
        for (int i = 0; i < 4; i++) { const int64_t p = piece + i * 256; if (p < n) dst[p] = src[p]; }
    }
}

// consolidate() on the device (round 5; cache.py): the appended 256-token extents of a cache -- listed in its DEVICE table of views --
// copied behind the base tokens of a destination view that mustafar_cache_rehouse has just filled with the base: bitmaps as they are,
// offsets shifted by the head's stream length in front of the extent (the model's append, llama_mustafar_kernel.py:352-360), streams
// behind that length.  The lengths are read on the device: dst's offset entry at the end of the base (written by the re-housing launch in
// front of this one) and every earlier extent's last offset entry.  grid: x = 16-KiB pieces of an extent's largest array, y = head,
// z = 3 * extent + array (0 bitmaps, 1 offsets, 2 stream).
constexpr int kExtTiles = 256 * kD / 64;   // tiles of one extent
__global__ __launch_bounds__(kThreads) void cache_consolidate_kernel(const mustafar_cache_view* __restrict__ ext, int base_tiles, uint64_t* d_bmp,
                                                                     uint32_t* d_idx, uint16_t* d_nz, int64_t d_bmp_stride, int64_t d_idx_stride,
                                                                     uint32_t d_nz_stride)
{
    __shared__ uint32_t s_before;
    const int h = blockIdx.y, e = blockIdx.z / 3, which = blockIdx.z % 3;
    const mustafar_cache_view v = ext[e];
    const int64_t tile0 = (int64_t)base_tiles + (int64_t)e * kExtTiles;
    const uint32_t* e_idx = v.idx + (int64_t)h * v.idx_head_stride;
    uint32_t before = 0;   // the head's stream length (half2 units) in front of this extent
    if (which != 0) {
        uint32_t part = 0;
        for (int k = threadIdx.x; k < e; k += kThreads) part += ext[k].idx[(int64_t)h * ext[k].idx_head_stride + kExtTiles];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if (threadIdx.x == 0) s_before = d_idx[(int64_t)h * d_idx_stride + base_tiles];
        __syncthreads();
        if ((threadIdx.x & 63) == 0 && part) atomicAdd(&s_before, part);
        __syncthreads();
        before = s_before;
    }
    static_assert(kThreads == 256, "a workgroup of this kernel moves pieces of 4 x kThreads = 1024 16-byte units (the host sizes grid.x for that)");
    const int64_t piece = (int64_t)blockIdx.x * (4 * kThreads) + threadIdx.x;   // 16-byte units, 4 per thread and piece
    if (which == 0) {
        const uint4* src = reinterpret_cast<const uint4*>(v.bmp + (int64_t)h * v.bmp_head_stride);
        uint4* dst = reinterpret_cast<uint4*>(d_bmp + h * d_bmp_stride + tile0);
        const int64_t n = kExtTiles / 2;
#pragma unroll
        for (int i = 0; i < 4; i++) { const int64_t p = piece + i * kThreads; if (p < n) dst[p] = src[p]; }
    } else if (which == 1) {
        // entry 0 of an extent IS entry kExtTiles of the one in front of it (or the base's last entry, which other workgroups read as
        // `before`): it is written once, by its owner -- an extent writes entries 1 .. kExtTiles (round 6; before, neighbours wrote the
        // shared entry twice)
        uint32_t* dst = d_idx + h * d_idx_stride + tile0;
        for (int64_t p = (int64_t)blockIdx.x * (16 * kThreads) + threadIdx.x; p < (int64_t)(blockIdx.x + 1) * (16 * kThreads) && p <= kExtTiles; p += kThreads)
            if (p > 0) dst[p] = e_idx[p] + before;   // (an extent's own offsets start at 0)
    } else {
        const uint4* src = reinterpret_cast<const uint4*>(v.nz) + (int64_t)h * v.nz_head_stride;
        uint4* dst = reinterpret_cast<uint4*>(d_nz) + (int64_t)h * d_nz_stride + before / 4;   // (lengths are multiples of four half2: 16 bytes)
        const int64_t n = e_idx[kExtTiles] / 4;
        const int64_t room = (int64_t)d_nz_stride - before / 4;   // (the host sizes the regions from the heads' summed lengths; never write past one)
#pragma unroll
        for (int i = 0; i < 4; i++) { const int64_t p = piece + i * kThreads; if (p < n && p < room) dst[p] = src[p]; }
    }
}

// MUSTAFAR_COMPRESS=twopass keeps the round-2 two-pass form of the fused calls (pass 1 + scan + pass 2); default: one pass.
// (round 6: both per host THREAD -- a second thread's append neither sees nor resets the form a first thread chose for its repeat, and the
// test hook is consumed by the thread that armed it)
thread_local int g_compress_form = 0;       // mustafar_compress_set_form: 0 = the process default (environment), 1 = one pass, 2 = two passes
thread_local int g_skip_publish_tb = -1;    // mustafar_compress_test_skip_publish: this thread's NEXT one-pass launch's block of this index does not publish its length (tests)
inline bool one_pass_compress()
{
    static const int mode = [] { const char* e = getenv("MUSTAFAR_COMPRESS"); return (e && !strcmp(e, "twopass")) ? 0 : 1; }();
    return g_compress_form ? g_compress_form == 1 : mode != 0;
}
inline int take_skip_publish() { const int k = g_skip_publish_tb; g_skip_publish_tb = -1; return k; }

inline Rows fresh_rows(int t) { const int64_t tiles = (int64_t)t * kD / 64; return Rows{tiles, tiles + 1, 0}; }

// Rows of a cache view behind `old_tokens`; false if the view cannot take t more tokens.
inline bool view_rows(const mustafar_cache_view* v, int old_tokens, int t, Rows& rows)
{
    if (!v || !v->bmp || !v->idx || !v->nz_offset || old_tokens < 0 || (old_tokens & 63)) return false;
    const int64_t tiles = (int64_t)(old_tokens + t) * kD / 64;
    rows = Rows{v->bmp_head_stride ? v->bmp_head_stride : tiles, v->idx_head_stride ? v->idx_head_stride : tiles + 1,
                (int64_t)old_tokens * kD / 64};
    return rows.bmp_stride >= tiles && rows.idx_stride >= tiles + 1;
}

// Pass 1 (+ scan) of one or two sides.  `blk`: [sides][B'][ntb] ints of scratch.
int launch_meta(hipStream_t st, Side* s, int sides, int Bp, int t, int32_t* blk, int32_t* overflow, bool exclusive_prefix)
{
    const int ntb = t / 64;
    for (int i = 0; i < sides; i++) s[i].blk = blk + (int64_t)i * Bp * ntb;
    const Side& s1 = s[sides - 1];
    tile_meta_kernel<<<dim3(ntb, Bp, sides), 64, 0, st>>>(s[0], s1, ntb);
    block_scan_kernel<<<dim3(Bp, sides), kThreads, 0, st>>>(s[0], s1, ntb, overflow);
    if (exclusive_prefix)
        for (int i = 0; i < sides; i++) head_offsets_kernel<<<1, kThreads, 0, st>>>(s[i].totals, Bp);
    return (int)hipGetLastError();
}

int launch_pack(hipStream_t st, const Side* s, int sides, int Bp, int t, const int32_t* overflow)
{
    tile_pack_kernel<<<dim3(t / 64, Bp, sides), kThreads, 0, st>>>(s[0], s[sides - 1], t / 64, overflow);
    return (int)hipGetLastError();
}

// grid: x = token block, y = head; 128 threads.  Two-call form only (mustafar_compress_bitmap_* must leave complete
// offsets behind: the caller sizes the packed buffer between the calls); the fused form adds the bases in pass 2.
__global__ __launch_bounds__(kD) void block_fixup_kernel(const int32_t* __restrict__ blk_base, int32_t* __restrict__ accum,
                                                         Rows rows)
{
    const int tb = blockIdx.x, h = blockIdx.y;
    const int32_t base = blk_base[(int64_t)h * gridDim.x + tb];
    accum[h * rows.idx_stride + rows.tile0 + (int64_t)tb * kD + threadIdx.x + 1] += base;
}

int bitmap_common(bool key, void* stream, const void* x, int Bp, int t, int D, int64_t* bmp, int32_t* accum,
                  int64_t* totals, Rows rows, bool exclusive_prefix, int64_t* mirror = nullptr)
{
    if (D != kD || Bp < 1 || t < 64 || (t & 63) || !x || !bmp || !accum || !totals) return MUSTAFAR_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int ntb = t / 64;
    // block totals / bases [B'][ntb]: stream-ordered scratch, freed behind the last kernel that reads it
    int32_t* blk = nullptr;
    bool pooled = true;
    if (hipMallocAsync(reinterpret_cast<void**>(&blk), sizeof(int32_t) * (size_t)Bp * ntb, st) != hipSuccess) {
        (void)hipGetLastError();   // no stream-ordered allocator here: plain allocation, freed after the stream drains
        pooled = false;
        if (hipMalloc(reinterpret_cast<void**>(&blk), sizeof(int32_t) * (size_t)Bp * ntb) != hipSuccess) return (int)hipGetLastError();
    }
    Side s{static_cast<const uint16_t*>(x), (int64_t)t * kD, bmp, accum, nullptr, totals, nullptr, nullptr, nullptr, rows, 0, 0, key ? 1 : 0};
    int err = launch_meta(st, &s, 1, Bp, t, blk, nullptr, false);
    if (!err) {
        // (the head offsets need the block scan's totals only: in front of the fix-up, so that a caller polling the mirror has them while
        // the fix-up still runs)
        if (exclusive_prefix) head_offsets_kernel<<<1, kThreads, 0, st>>>(totals, Bp, mirror);
        block_fixup_kernel<<<dim3(ntb, Bp), kD, 0, st>>>(blk, accum, rows);
        err = (int)hipGetLastError();
    }
    if (pooled) {
        (void)hipFreeAsync(blk, st);
    } else {
        (void)hipStreamSynchronize(st);
        (void)hipFree(blk);
    }
    return err;
}

int pack_common(bool key, void* stream, const void* x, int Bp, int t, int D, const int64_t* bmp, const int32_t* accum,
                const int64_t* head_off, const uint32_t* nz_offset, void* nz_flat, Rows rows)
{
    if (D != kD || Bp < 1 || t < 64 || (t & 63) || !x || !bmp || !accum || (!head_off && !nz_offset)) return MUSTAFAR_EINVAL;
    if (!nz_flat) return 0;   // nothing to write: every tile of every head is empty
    Side s{static_cast<const uint16_t*>(x), (int64_t)t * kD, const_cast<int64_t*>(bmp), const_cast<int32_t*>(accum), nullptr, nullptr,
           head_off, nz_offset, static_cast<uint16_t*>(nz_flat), rows, 0, 0, key ? 1 : 0};
    return launch_pack(static_cast<hipStream_t>(stream), &s, 1, Bp, t, nullptr);
}

// ---- the reference's two conversion calls WITHOUT a host read (round 5: mustafar_convert_onepass) ------------------------------------
// compress_block_kernel (one read of the rows: bitmaps, offsets, streams) writes every head's stream into a region of worst-case size;
// then the regions are packed into the reference's layout -- head h's stream right behind head h - 1's -- by one copy launch that
// finds its offsets on the device.  The two-pass form (tile_meta -> block_scan -> host read of the sizes -> tile_pack) stays behind
// mustafar_compress_bitmap_* / _pack_*.
__global__ __launch_bounds__(64) void convert_init_kernel(uint32_t* __restrict__ nz_offset, int Bp, uint32_t region_uint4)
{
    const int h = blockIdx.x * 64 + threadIdx.x;
    if (h < Bp) nz_offset[h] = (uint32_t)h * region_uint4;
}
// grid: x = 16-KiB pieces of a region, y = head.  head_off: exclusive prefix of the heads' stream lengths (halfs), [B' + 1].
__global__ __launch_bounds__(kThreads) void convert_pack_kernel(const uint16_t* __restrict__ regions, int64_t region_halfs,
                                                                const int64_t* __restrict__ head_off, uint16_t* __restrict__ packed)
{
    const int h = blockIdx.y;
    const int64_t o0 = head_off[h], len16 = (head_off[h + 1] - o0) / 8;   // 16-byte pieces (every tile's stream is padded to eight halfs)
    const uint4* src = reinterpret_cast<const uint4*>(regions + (int64_t)h * region_halfs);
    uint4* dst = reinterpret_cast<uint4*>(packed + o0);                    // (o0 is a multiple of 8 halfs: 16-byte aligned)
    const int64_t p0 = (int64_t)blockIdx.x * 1024;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int64_t p = p0 + threadIdx.x + i * kThreads;
        if (p < len16) dst[p] = src[p];
    }
}

}  // namespace

extern "C" {

int mustafar_prune_magnitude(void* stream, const void* x, void* out, int64_t n_rows, int D, int kth)
{
    if (D != kD || kth < 1 || kth > D || n_rows < 0 || !x || !out) return MUSTAFAR_EINVAL;
    if (n_rows == 0) return 0;
    const int64_t blocks = (n_rows + kPruneRows - 1) / kPruneRows;
    if (blocks > 0x7fffffff) return MUSTAFAR_EINVAL;
    const unsigned grid = (unsigned)blocks;
    prune_magnitude_kernel<<<grid, 64, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const uint32_t*>(x), static_cast<uint32_t*>(out), n_rows, kth);
    return (int)hipGetLastError();
}

int mustafar_compress_bitmap_key(void* stream, const void* x, int Bp, int t, int D, int64_t* bmp, int32_t* accum,
                                 int64_t* head_off)
{ return bitmap_common(true, stream, x, Bp, t, D, bmp, accum, head_off, fresh_rows(t), true); }

int mustafar_compress_bitmap_value(void* stream, const void* x, int Bp, int t, int D, int64_t* bmp, int32_t* accum,
                                   int64_t* head_off)
{ return bitmap_common(false, stream, x, Bp, t, D, bmp, accum, head_off, fresh_rows(t), true); }

int mustafar_compress_bitmap_mirrored(void* stream, const void* x, int Bp, int t, int D, int key, int64_t* bmp, int32_t* accum, int64_t* head_off,
                                      int64_t* host_mirror)
{ return bitmap_common(key != 0, stream, x, Bp, t, D, bmp, accum, head_off, fresh_rows(t), true, host_mirror); }

int mustafar_compress_pack_key(void* stream, const void* x, int Bp, int t, int D, const int64_t* bmp,
                               const int32_t* accum, const int64_t* head_off, void* nz_flat)
{ return pack_common(true, stream, x, Bp, t, D, bmp, accum, head_off, nullptr, nz_flat, fresh_rows(t)); }

int mustafar_compress_pack_value(void* stream, const void* x, int Bp, int t, int D, const int64_t* bmp,
                                 const int32_t* accum, const int64_t* head_off, void* nz_flat)
{ return pack_common(false, stream, x, Bp, t, D, bmp, accum, head_off, nullptr, nz_flat, fresh_rows(t)); }

// ---- in-place append into a cache view (model :339-390 without the re-copies) ----------------------------------------
int64_t mustafar_convert_scratch_bytes(int Bp, int t)
{
    if (Bp < 1 || t < 64 || (t & 63)) return 0;
    return (int64_t)Bp * (t / 64) * (int64_t)sizeof(uint64_t) + (((int64_t)Bp * 4 + 15) & ~(int64_t)15);
}

int mustafar_convert_onepass_mirrored(void* stream, const void* x, int Bp, int t, int D, int key, int64_t* bmp, int32_t* accum, int64_t* head_off,
                                      void* regions, int32_t* overflow_flag, void* scratch, int64_t* host_mirror)
{
    if (D != kD || Bp < 1 || t < 64 || (t & 63) || !x || !bmp || !accum || !head_off || !regions || !overflow_flag || !scratch ||
        (int64_t)Bp * t * (kD / 8) > 0xffffffffll)   // (stream starts in 16-byte units are 32-bit, as in the format)
        return MUSTAFAR_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int ntb = t / 64;
    const int64_t region_halfs = (int64_t)t * kD;                 // worst case: nothing pruned, every tile full
    uint64_t* gran = static_cast<uint64_t*>(scratch);
    uint32_t* nz_off = reinterpret_cast<uint32_t*>(gran + (int64_t)Bp * ntb);
    int err = (int)hipMemsetAsync(gran, 0, (size_t)Bp * ntb * sizeof(uint64_t), st);   // the length words: not yet valid
    if (err) return err;
    convert_init_kernel<<<(Bp + 63) / 64, 64, 0, st>>>(nz_off, Bp, (uint32_t)(region_halfs / 8));
    const Side s{static_cast<const uint16_t*>(x), (int64_t)t * kD, bmp, accum, nullptr, head_off, nullptr, nz_off, static_cast<uint16_t*>(regions),
                 fresh_rows(t), region_halfs, 0, key ? 1 : 0};
    compress_block_kernel<<<dim3(ntb, Bp, 1), 64, 0, st>>>(s, s, ntb, gran, overflow_flag, take_skip_publish());
    head_offsets_kernel<<<1, kThreads, 0, st>>>(head_off, Bp, host_mirror, overflow_flag);
    return (int)hipGetLastError();
}

int mustafar_convert_onepass(void* stream, const void* x, int Bp, int t, int D, int key, int64_t* bmp, int32_t* accum, int64_t* head_off,
                             void* regions, int32_t* overflow_flag, void* scratch)
{ return mustafar_convert_onepass_mirrored(stream, x, Bp, t, D, key, bmp, accum, head_off, regions, overflow_flag, scratch, nullptr); }

int mustafar_convert_pack(void* stream, const void* regions, int Bp, int t, int D, const int64_t* head_off, void* packed)
{
    if (D != kD || Bp < 1 || t < 64 || (t & 63) || !regions || !head_off) return MUSTAFAR_EINVAL;
    if (!packed) return 0;   // nothing to write: every tile of every head is empty
    const int64_t region_halfs = (int64_t)t * kD;
    convert_pack_kernel<<<dim3((unsigned)((region_halfs / 8 + 1023) / 1024), Bp), kThreads, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const uint16_t*>(regions), region_halfs, head_off, static_cast<uint16_t*>(packed));
    return (int)hipGetLastError();
}

int mustafar_cache_append_bitmap_key(void* stream, const void* x, int Bp, int t, int D, const mustafar_cache_view* dst,
                                     int old_tokens, int64_t* head_total)
{
    Rows rows;
    if (!view_rows(dst, old_tokens, t, rows)) return MUSTAFAR_EINVAL;
    return bitmap_common(true, stream, x, Bp, t, D, reinterpret_cast<int64_t*>(dst->bmp), reinterpret_cast<int32_t*>(dst->idx),
                         head_total, rows, false);
}

int mustafar_cache_append_bitmap_value(void* stream, const void* x, int Bp, int t, int D, const mustafar_cache_view* dst,
                                       int old_tokens, int64_t* head_total)
{
    Rows rows;
    if (!view_rows(dst, old_tokens, t, rows)) return MUSTAFAR_EINVAL;
    return bitmap_common(false, stream, x, Bp, t, D, reinterpret_cast<int64_t*>(dst->bmp), reinterpret_cast<int32_t*>(dst->idx),
                         head_total, rows, false);
}

int mustafar_cache_append_pack_key(void* stream, const void* x, int Bp, int t, int D, const mustafar_cache_view* dst,
                                   int old_tokens)
{
    Rows rows;
    if (!view_rows(dst, old_tokens, t, rows) || !dst->nz) return MUSTAFAR_EINVAL;
    return pack_common(true, stream, x, Bp, t, D, reinterpret_cast<const int64_t*>(dst->bmp), reinterpret_cast<const int32_t*>(dst->idx), nullptr,
                       dst->nz_offset, dst->nz, rows);
}

int mustafar_cache_append_pack_value(void* stream, const void* x, int Bp, int t, int D, const mustafar_cache_view* dst,
                                     int old_tokens)
{
    Rows rows;
    if (!view_rows(dst, old_tokens, t, rows) || !dst->nz) return MUSTAFAR_EINVAL;
    return pack_common(false, stream, x, Bp, t, D, reinterpret_cast<const int64_t*>(dst->bmp), reinterpret_cast<const int32_t*>(dst->idx), nullptr,
                       dst->nz_offset, dst->nz, rows);
}

// ---- fused forms (one launch per pass for K and V together; no allocation, no host read: graph-capturable) ----------
int64_t mustafar_compress_scratch_bytes(int Bp, int t)
{
    if (Bp < 1 || t < 64 || (t & 63)) return 0;
    return 2 * (int64_t)Bp * (t / 64) * (int64_t)sizeof(uint64_t);   // one length word per block and side (two-pass form: an int)
}

int mustafar_cache_append_kv(void* stream, const void* k_x, const void* v_x, int64_t head_stride, int Bp, int t, int D, int kth_k,
                             int kth_v, const mustafar_cache_view* k_dst, const mustafar_cache_view* v_dst, int old_tokens,
                             int64_t* k_head_total, int64_t* v_head_total, int64_t k_region_halfs, int64_t v_region_halfs,
                             int32_t* overflow_flag, void* scratch)
{
    Rows kr, vr;
    if (D != kD || Bp < 1 || t < 64 || (t & 63) || !k_x || !v_x || head_stride < (int64_t)t * kD || kth_k < 0 || kth_k > kD ||
        kth_v < 0 || kth_v > kD || !k_head_total || !v_head_total || !scratch || !view_rows(k_dst, old_tokens, t, kr) ||
        !view_rows(v_dst, old_tokens, t, vr) || !k_dst->nz || !v_dst->nz)
        return MUSTAFAR_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    Side s[2] = {
        {static_cast<const uint16_t*>(k_x), head_stride, reinterpret_cast<int64_t*>(k_dst->bmp), reinterpret_cast<int32_t*>(k_dst->idx), nullptr,
         k_head_total, nullptr, k_dst->nz_offset, static_cast<uint16_t*>(k_dst->nz), kr, k_region_halfs, kth_k, 1},
        {static_cast<const uint16_t*>(v_x), head_stride, reinterpret_cast<int64_t*>(v_dst->bmp), reinterpret_cast<int32_t*>(v_dst->idx), nullptr,
         v_head_total, nullptr, v_dst->nz_offset, static_cast<uint16_t*>(v_dst->nz), vr, v_region_halfs, kth_v, 0}};
    if (one_pass_compress()) {
        // the one-pass form can fail at run time (a head outgrowing its region; a block giving up on its predecessors' lengths):
        // without the flag either would pass silently, so the flag is required here
        if (!overflow_flag) return MUSTAFAR_EINVAL;
        const int ntb = t / 64;
        const int err = (int)hipMemsetAsync(scratch, 0, 2 * (size_t)Bp * ntb * sizeof(uint64_t), st);   // the length words: not yet valid
        if (err) return err;
        compress_block_kernel<<<dim3(ntb, Bp, 2), 64, 0, st>>>(s[0], s[1], ntb, static_cast<uint64_t*>(scratch), overflow_flag, take_skip_publish());
        return (int)hipGetLastError();
    }
    const int err = launch_meta(st, s, 2, Bp, t, static_cast<int32_t*>(scratch), overflow_flag, false);
    if (err) return err;
    return launch_pack(st, s, 2, Bp, t, overflow_flag);
}

// The trigger of ALL layers of a model in two calls (round 4; cache.py: append_extent_pairs): `items` is a HOST array, one entry per
// layer.  (1) compress: one memset of the whole scratch, then the layers' compression launches back to back -- no host work in
// between, nothing read back: the caller reads every layer's flag and lengths with ONE copy behind the call.
int mustafar_trigger_compress_batch(void* stream, int n, const mustafar_trigger_item* items, int64_t head_stride, int Bp, int t, int D,
                                    int kth_k, int kth_v, int64_t k_region_halfs, int64_t v_region_halfs, void* scratch)
{
    if (n < 1 || !items || D != kD || Bp < 1 || t < 64 || (t & 63) || head_stride < (int64_t)t * kD || kth_k < 0 || kth_k > kD || kth_v < 0 ||
        kth_v > kD || !scratch)
        return MUSTAFAR_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int ntb = t / 64;
    const int64_t per_item = mustafar_compress_scratch_bytes(Bp, t);
    for (int i = 0; i < n; i++) {   // every item is checked before anything is launched
        Rows r;
        const mustafar_trigger_item& it = items[i];
        if (!it.k_window || !it.v_window || !it.k_head_total || !it.v_head_total || !it.overflow_flag || !view_rows(&it.k_dst, 0, t, r) ||
            !view_rows(&it.v_dst, 0, t, r) || !it.k_dst.nz || !it.v_dst.nz)
            return MUSTAFAR_EINVAL;
    }
    const bool one_pass = one_pass_compress();
    const int skip0 = take_skip_publish();   // (tests: the first layer's launch)
    if (one_pass) {
        const int err = (int)hipMemsetAsync(scratch, 0, (size_t)(per_item * n), st);   // the length words of every layer: not yet valid
        if (err) return err;
    }
    for (int i = 0; i < n; i++) {
        const mustafar_trigger_item& it = items[i];
        Rows kr, vr;
        (void)view_rows(&it.k_dst, 0, t, kr);
        (void)view_rows(&it.v_dst, 0, t, vr);
        Side s[2] = {
            {static_cast<const uint16_t*>(it.k_window), head_stride, reinterpret_cast<int64_t*>(it.k_dst.bmp), reinterpret_cast<int32_t*>(it.k_dst.idx),
             nullptr, it.k_head_total, nullptr, it.k_dst.nz_offset, static_cast<uint16_t*>(it.k_dst.nz), kr, k_region_halfs, kth_k, 1},
            {static_cast<const uint16_t*>(it.v_window), head_stride, reinterpret_cast<int64_t*>(it.v_dst.bmp), reinterpret_cast<int32_t*>(it.v_dst.idx),
             nullptr, it.v_head_total, nullptr, it.v_dst.nz_offset, static_cast<uint16_t*>(it.v_dst.nz), vr, v_region_halfs, kth_v, 0}};
        unsigned char* sc = static_cast<unsigned char*>(scratch) + per_item * i;
        if (one_pass) {
            compress_block_kernel<<<dim3(ntb, Bp, 2), 64, 0, st>>>(s[0], s[1], ntb, reinterpret_cast<uint64_t*>(sc), it.overflow_flag, i == 0 ? skip0 : -1);
        } else {
            int err = launch_meta(st, s, 2, Bp, t, reinterpret_cast<int32_t*>(sc), it.overflow_flag, false);
            if (!err) err = launch_pack(st, s, 2, Bp, t, it.overflow_flag);
            if (err) return err;
        }
    }
    return (int)hipGetLastError();
}

// (2) finish: per layer ONE launch that lists the extent in the two device tables (k_table_slot / v_table_slot; NULL: not listed) and
// slides both windows by `drop` rows (model :392-393).  Called once the caller has seen every flag clear.
int mustafar_trigger_finish_batch(void* stream, int n, const mustafar_trigger_item* items, int64_t head_stride, int Bp, int len, int drop)
{
    if (n < 1 || !items || Bp < 1 || drop < 0 || len < drop || head_stride < (int64_t)len * kD) return MUSTAFAR_EINVAL;
    for (int i = 0; i < n; i++)
        if (!items[i].k_window || !items[i].v_window) return MUSTAFAR_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int i = 0; i < n; i++) {
        const mustafar_trigger_item& it = items[i];
        trigger_finish_kernel<<<dim3(Bp, 2), kThreads, 0, st>>>(static_cast<uint16_t*>(it.k_window), static_cast<uint16_t*>(it.v_window), head_stride,
                                                                len, drop, it.k_dst, it.v_dst, it.k_table_slot, it.v_table_slot);
    }
    return (int)hipGetLastError();
}

int mustafar_cache_rehouse(void* stream, const mustafar_cache_view* src, const mustafar_cache_view* dst, int Bp, int tokens,
                           int64_t stream_halfs)
{
    const int64_t tiles = (int64_t)tokens * kD / 64;
    if (!src || !dst || Bp < 1 || tokens < 0 || (tokens & 63) || stream_halfs < 0 || (stream_halfs & 7) || !src->bmp || !src->idx ||
        !src->nz || !src->nz_offset || !dst->bmp || !dst->idx || !dst->nz || !dst->nz_offset || dst->nz_head_stride <= 0 ||
        stream_halfs > 8 * dst->nz_head_stride || (src->bmp_head_stride ? src->bmp_head_stride : tiles) < tiles ||
        (dst->bmp_head_stride ? dst->bmp_head_stride : tiles) < tiles || (src->idx_head_stride ? src->idx_head_stride : tiles + 1) < tiles + 1 ||
        (dst->idx_head_stride ? dst->idx_head_stride : tiles + 1) < tiles + 1)
        return MUSTAFAR_EINVAL;
    const int64_t largest = stream_halfs * 2 > tiles * 8 ? stream_halfs * 2 : tiles * 8;   // bytes of a head's largest array
    const unsigned gx = (unsigned)((largest + 16383) / 16384 > 0 ? (largest + 16383) / 16384 : 1);
    cache_rehouse_kernel<<<dim3(gx, Bp, 3), kThreads, 0, static_cast<hipStream_t>(stream)>>>(
        src->bmp, src->idx, static_cast<const uint16_t*>(src->nz), src->nz_offset, src->bmp_head_stride ? src->bmp_head_stride : tiles,
        src->idx_head_stride ? src->idx_head_stride : tiles + 1, dst->bmp, dst->idx, static_cast<uint16_t*>(dst->nz), dst->nz_offset,
        dst->bmp_head_stride ? dst->bmp_head_stride : tiles, dst->idx_head_stride ? dst->idx_head_stride : tiles + 1,
        (uint32_t)dst->nz_head_stride, tiles, stream_halfs);
    return (int)hipGetLastError();
}

int mustafar_compress_set_form(int form)
{
    if (form < 0 || form > 2) return MUSTAFAR_EINVAL;
    g_compress_form = form;
    return 0;
}

int mustafar_compress_get_form(void) { return g_compress_form; }

int mustafar_compress_test_skip_publish(int block)
{
    // a test hook in the product library: armed only in a process that asked for test hooks before its first use (tests/conftest.py)
    static const bool hooks = [] { const char* e = getenv("MUSTAFAR_TEST_HOOKS"); return e && *e == '1'; }();
    if (!hooks) return MUSTAFAR_EINVAL;
    g_skip_publish_tb = block < 0 ? -1 : block;
    return 0;
}

int mustafar_cache_consolidate_extents(void* stream, const mustafar_cache_view* dst, const mustafar_cache_view* extents, int n_extents, int Bp,
                                       int base_tokens, int64_t max_extent_halfs)
{
    if (!dst || !extents || n_extents < 1 || Bp < 1 || base_tokens < 0 || (base_tokens & 63) || max_extent_halfs < 0 || !dst->bmp || !dst->idx ||
        !dst->nz || dst->nz_head_stride <= 0 || 3 * (int64_t)n_extents > 65535)
        return MUSTAFAR_EINVAL;
    const int64_t tiles = ((int64_t)base_tokens + 256ll * n_extents) * kD / 64;
    if ((dst->bmp_head_stride ? dst->bmp_head_stride : tiles) < tiles || (dst->idx_head_stride ? dst->idx_head_stride : tiles + 1) < tiles + 1)
        return MUSTAFAR_EINVAL;
    int64_t largest = max_extent_halfs * 2 > (int64_t)kExtTiles * 8 ? max_extent_halfs * 2 : (int64_t)kExtTiles * 8;   // bytes of an extent's largest array (per head)
    const unsigned gx = (unsigned)((largest + 16383) / 16384);
    cache_consolidate_kernel<<<dim3(gx, Bp, 3 * n_extents), kThreads, 0, static_cast<hipStream_t>(stream)>>>(
        extents, (int)((int64_t)base_tokens * kD / 64), dst->bmp, dst->idx, static_cast<uint16_t*>(dst->nz),
        dst->bmp_head_stride ? dst->bmp_head_stride : tiles, dst->idx_head_stride ? dst->idx_head_stride : tiles + 1, (uint32_t)dst->nz_head_stride);
    return (int)hipGetLastError();
}

int mustafar_window_drop_front(void* stream, void* k_window, void* v_window, int64_t head_stride, int Bp, int len, int drop)
{
    if (!k_window || !v_window || Bp < 1 || drop < 0 || len < drop || head_stride < (int64_t)len * kD) return MUSTAFAR_EINVAL;
    if (len == drop || drop == 0) return 0;
    window_drop_front_kernel<<<dim3(Bp, 2), kThreads, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<uint16_t*>(k_window), static_cast<uint16_t*>(v_window), head_stride, len, drop);
    return (int)hipGetLastError();
}

}  // extern "C"
