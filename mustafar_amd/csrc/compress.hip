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

#include "../../include/mustafar_hip.h"

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
// Lane = row: one wave takes 64 rows, transposed through LDS (row stride 65 dwords: conflict-free both ways), and every
// lane runs the 15-step search on its own row held in 64 VGPRs.  A count is a SWAR pass over the 64 packed pairs:
// with the guard bit H = 0x8000 set in each half, (m | H) - (c | c << 16) keeps H in a half iff that magnitude >= c
// (no borrow crosses the halves), and the set guard bits are tallied two 16-bit counters at a time.  4 plain VALU ops
// per pair per step (sub, shift, and, add); the wave-per-row form spent ~12 SALU + 4 VALU per step and per ROW on ballots and was
// instruction-bound at 1.7 TB/s.
constexpr int kPruneRows  = 64;            // rows per wave
constexpr int kPruneLd    = kD / 2 + 1;    // LDS row stride in dwords

__global__ __launch_bounds__(64) void prune_magnitude_kernel(const uint32_t* __restrict__ x, uint32_t* __restrict__ out,
                                                             int64_t n_rows, int kth)
{
    __shared__ uint32_t s_rows[kPruneRows * kPruneLd];
    const int lane = threadIdx.x;
    const uint32_t H = 0x80008000u, ONES = 0x00010001u;
    for (int64_t row0 = (int64_t)blockIdx.x * kPruneRows; row0 < n_rows; row0 += (int64_t)gridDim.x * kPruneRows) {
        const int rows = (int)((n_rows - row0) < kPruneRows ? (n_rows - row0) : kPruneRows);
        // coalesced 16-byte loads: pass p covers rows 4p .. 4p+3 (16 lanes x 16 B per row)
        const uint4* src = reinterpret_cast<const uint4*>(x + row0 * (kD / 2));
        __syncthreads();
#pragma unroll 4
        for (int p = 0; p < kPruneRows / 4; p++) {
            const int r = p * 4 + (lane >> 4), c4 = lane & 15;
            uint4 v = {0u, 0u, 0u, 0u};
            if (r < rows) v = src[r * 16 + c4];
            uint32_t* d = s_rows + r * kPruneLd + c4 * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
        uint32_t wh[kD / 2];   // magnitudes with the guard bits set (the signs stay behind in LDS)
#pragma unroll
        for (int j = 0; j < kD / 2; j++) wh[j] = s_rows[lane * kPruneLd + j] | H;
        // bit-by-bit search of the k-th smallest magnitude
        uint32_t thr = 0;
#pragma unroll 1
        for (int bit = 14; bit >= 0; bit--) {
            const uint32_t c = thr | (1u << bit);
            const uint32_t cc = c | (c << 16);
            uint32_t ge = 0;   // two 16-bit counters: magnitudes >= c among the low / high halves
#pragma unroll
            for (int j = 0; j < kD / 2; j++) ge += ((wh[j] - cc) >> 15) & ONES;
            const int below = kD - (int)((ge & 0xffffu) + (ge >> 16));
            if (below < kth) thr = c;
        }
        // keep |x| >= thr, else the sign bit alone
        const uint32_t tt = thr | (thr << 16);
#pragma unroll
        for (int j = 0; j < kD / 2; j++) {
            const uint32_t k = ((wh[j] - tt) >> 15) & ONES;   // 1 per half that stays
            const uint32_t keep = (k << 16) - k;               // 0xffff per half that stays
            s_rows[lane * kPruneLd + j] &= keep | H;
        }
        __syncthreads();
        uint4* dst = reinterpret_cast<uint4*>(out + row0 * (kD / 2));
#pragma unroll 4
        for (int p = 0; p < kPruneRows / 4; p++) {
            const int r = p * 4 + (lane >> 4), c4 = lane & 15;
            const uint32_t* d = s_rows + r * kPruneLd + c4 * 4;
            if (r < rows) dst[r * 16 + c4] = make_uint4(d[0], d[1], d[2], d[3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ bitmaps
// Destination rows of the passes: the reference's contiguous result tensors (bmp [B', 2t], accum [B', 2t + 1], stride =
// row length, tile0 = 0) or rows of a cache view with spare capacity, written behind the tiles already in use
// (in-place append: mustafar_cache_append_*).
struct Rows {
    int64_t bmp_stride;   // elements between the heads' bitmap rows
    int64_t idx_stride;   // elements between the heads' offset rows
    int64_t tile0;        // first tile of a row this call writes (2 * old_tokens)
};

// grid: x = token block (64 tokens), y = head.  Writes the 128 bitmaps of the block and the raw padded
// counts (half2 units) into accum[h][tile + 1]; the scan kernel turns them into the exclusive prefix.

// The 128 raw counts of a token block (threads 0..127 hold one each) -> inclusive prefix inside the block, written to
// accum[...][tile + 1], and the block's total to blk_total.  Two waves: wave scan + the first wave's total.
__device__ __forceinline__ void block_prefix_store(int32_t cnt, int32_t* __restrict__ accum_row, int64_t tile_in_row,
                                                   int32_t* __restrict__ blk_total_slot, int32_t* s_tot)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t v = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int32_t u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    if (threadIdx.x == 63) *s_tot = v;
    __syncthreads();
    if (wave == 1) v += *s_tot;
    if (threadIdx.x < kD) accum_row[tile_in_row + 1] = v;
    if (threadIdx.x == kD - 1) *blk_total_slot = v;
}

// V: tile (tb, half, r) = channels half*64..+63 of token tb*64+r (compression.py:87-97); lane = channel.
__global__ __launch_bounds__(kThreads) void bitmap_value_kernel(const uint16_t* __restrict__ x, int t, int64_t* __restrict__ bmp,
                                                                int32_t* __restrict__ accum, int32_t* __restrict__ blk_total,
                                                                Rows rows)
{
    __shared__ uint64_t s_bmp[kD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tb = blockIdx.x, h = blockIdx.y;
    const uint16_t* xb = x + ((int64_t)h * t + (int64_t)tb * 64) * kD;
    for (int r = wave; r < 64; r += kWaves) {
        const uint16_t a = xb[r * kD + lane], b = xb[r * kD + 64 + lane];
        const uint64_t m0 = __builtin_bitreverse64(__ballot(nonzero_h(a)));
        const uint64_t m1 = __builtin_bitreverse64(__ballot(nonzero_h(b)));
        if (lane == 0) {
            s_bmp[r]      = m0;
            s_bmp[64 + r] = m1;
        }
    }
    __syncthreads();
    __shared__ int32_t s_tot;
    int32_t cnt = 0;
    const int64_t tile = (int64_t)tb * kD + (threadIdx.x & (kD - 1));
    if (threadIdx.x < kD) {
        const uint64_t m = s_bmp[threadIdx.x];
        bmp[h * rows.bmp_stride + rows.tile0 + tile] = (int64_t)m;
        cnt = ((__popcll(m) + 7) & ~7) >> 1;   // compression.py:46-48
    }
    block_prefix_store(cnt, accum + h * rows.idx_stride + rows.tile0, tile, blk_total + (int64_t)h * gridDim.x + tb, &s_tot);
}

// K: tile (tb, d) = tokens tb*64..+63 of channel d (compression.py:32-36 on the transposed input); lane = token.
// The 64x128 block is staged through LDS (row stride 65 dwords: conflict-free column reads).
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

__global__ __launch_bounds__(kThreads) void bitmap_key_kernel(const uint16_t* __restrict__ x, int t, int64_t* __restrict__ bmp,
                                                                int32_t* __restrict__ accum, int32_t* __restrict__ blk_total,
                                                                Rows rows)
{
    __shared__ uint32_t s_blk[64 * kRowWords];
    __shared__ uint64_t s_bmp[kD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tb = blockIdx.x, h = blockIdx.y;
    load_block_transposable(s_blk, x + ((int64_t)h * t + (int64_t)tb * 64) * kD);
    __syncthreads();
    for (int d = wave; d < kD; d += kWaves) {
        const uint64_t m = __builtin_bitreverse64(__ballot(nonzero_h(block_elem(s_blk, lane, d))));
        if (lane == 0) s_bmp[d] = m;
    }
    __syncthreads();
    __shared__ int32_t s_tot;
    int32_t cnt = 0;
    const int64_t tile = (int64_t)tb * kD + (threadIdx.x & (kD - 1));
    if (threadIdx.x < kD) {
        const uint64_t m = s_bmp[threadIdx.x];
        bmp[h * rows.bmp_stride + rows.tile0 + tile] = (int64_t)m;
        cnt = ((__popcll(m) + 7) & ~7) >> 1;
    }
    block_prefix_store(cnt, accum + h * rows.idx_stride + rows.tile0, tile, blk_total + (int64_t)h * gridDim.x + tb, &s_tot);
}

// ------------------------------------------------------------------------------------------------ scan
// accum[h][0] = 0, accum[h][i+1] = sum of raw counts [0..i]  (torch.cumsum + cat, compression.py:294-298), in three
// parallel pieces: the bitmap kernels leave the inclusive prefix INSIDE each 64-token block and the block totals;
// block_scan_kernel (one workgroup per head) turns the totals into every block's base; block_fixup_kernel (one
// workgroup per block) adds the base.  Append mode (rows.tile0 > 0): the head's entry tile0 already holds the total of
// the tiles in use (model :352-360) and is the first base.
__global__ __launch_bounds__(kThreads) void block_scan_kernel(int32_t* __restrict__ blk_total, int ntb,
                                                              int32_t* __restrict__ accum, int64_t* __restrict__ totals,
                                                              Rows rows)
{
    __shared__ int32_t s_wave[kWaves];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t* bt = blk_total + (int64_t)blockIdx.x * ntb;
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
        const int32_t blk = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        if (i < ntb) bt[i] = v - own + add;   // exclusive: the base of block i
        carry += blk;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (rows.tile0 == 0) a[0] = 0;
        totals[blockIdx.x] = 2 * (int64_t)carry;   // halfs in this head's stream (compression.py:302)
    }
}

// grid: x = token block, y = head; 128 threads.
__global__ __launch_bounds__(kD) void block_fixup_kernel(const int32_t* __restrict__ blk_base, int32_t* __restrict__ accum,
                                                         Rows rows)
{
    const int tb = blockIdx.x, h = blockIdx.y;
    const int32_t base = blk_base[(int64_t)h * gridDim.x + tb];
    accum[h * rows.idx_stride + rows.tile0 + (int64_t)tb * kD + threadIdx.x + 1] += base;
}

// head_off[h] = exclusive prefix of totals (compression.py:303-304), head_off[B'] = grand total.  In place.
__global__ void head_offsets_kernel(int64_t* __restrict__ head_off, int Bp)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int64_t run = 0;
        for (int h = 0; h < Bp; h++) {
            const int64_t tot = head_off[h];
            head_off[h] = run;
            run += tot;
        }
        head_off[Bp] = run;
    }
}

// ------------------------------------------------------------------------------------------------ pack
// Non-zeros of a tile in ascending element order at stream offset 2*accum[tile] (compression.py:164-174),
// then zeros up to ceil8(nnz) (the reference relies on a pre-zeroed buffer, :309; here the wave writes them).
__device__ __forceinline__ void pack_tile(uint16_t* __restrict__ dst, uint16_t v, int lane)
{
    const uint64_t m = __ballot(nonzero_h(v));   // bit l <=> element l
    const int nnz = __popcll(m);
    const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if ((m >> lane) & 1ull) dst[rank] = v;
    const int padded = (nnz + 7) & ~7;
    if (lane >= nnz && lane < padded) dst[lane] = 0;
}

// Start of head h's stream in halfs: head_off[h] (fresh result tensor) or 8 * nz_offset[h] (cache view).
__device__ __forceinline__ int64_t head_base(const int64_t* __restrict__ head_off, const uint32_t* __restrict__ nz_offset, int h)
{
    return head_off ? head_off[h] : 8 * (int64_t)nz_offset[h];
}

__global__ __launch_bounds__(kThreads) void pack_value_kernel(const uint16_t* __restrict__ x, int t,
                                                              const int32_t* __restrict__ accum,
                                                              const int64_t* __restrict__ head_off,
                                                              const uint32_t* __restrict__ nz_offset,
                                                              uint16_t* __restrict__ nz, Rows rows)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tb = blockIdx.x, h = blockIdx.y;
    const uint16_t* xb = x + ((int64_t)h * t + (int64_t)tb * 64) * kD;
    const int32_t* acc = accum + (int64_t)h * rows.idx_stride + rows.tile0 + (int64_t)tb * kD;
    uint16_t* nz_h = nz + head_base(head_off, nz_offset, h);
    for (int r = wave; r < 64; r += kWaves) {
        pack_tile(nz_h + 2 * (int64_t)acc[r], xb[r * kD + lane], lane);
        pack_tile(nz_h + 2 * (int64_t)acc[64 + r], xb[r * kD + 64 + lane], lane);
    }
}

__global__ __launch_bounds__(kThreads) void pack_key_kernel(const uint16_t* __restrict__ x, int t,
                                                            const int32_t* __restrict__ accum,
                                                            const int64_t* __restrict__ head_off,
                                                            const uint32_t* __restrict__ nz_offset,
                                                            uint16_t* __restrict__ nz, Rows rows)
{
    __shared__ uint32_t s_blk[64 * kRowWords];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tb = blockIdx.x, h = blockIdx.y;
    load_block_transposable(s_blk, x + ((int64_t)h * t + (int64_t)tb * 64) * kD);
    __syncthreads();
    const int32_t* acc = accum + (int64_t)h * rows.idx_stride + rows.tile0 + (int64_t)tb * kD;
    uint16_t* nz_h = nz + head_base(head_off, nz_offset, h);
    for (int d = wave; d < kD; d += kWaves)
        pack_tile(nz_h + 2 * (int64_t)acc[d], block_elem(s_blk, lane, d), lane);
}

int bitmap_common(bool key, void* stream, const void* x, int Bp, int t, int D, int64_t* bmp, int32_t* accum,
                  int64_t* totals, Rows rows, bool exclusive_prefix)
{
    if (D != kD || Bp < 1 || t < 64 || (t & 63) || !x || !bmp || !accum || !totals) return MUSTAFAR_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int ntb = t / 64;
    const dim3 grid(ntb, Bp);
    auto xs = static_cast<const uint16_t*>(x);
    // block totals / bases [B'][ntb]: stream-ordered scratch, freed behind the last kernel that reads it
    int32_t* blk = nullptr;
    bool pooled = true;
    if (hipMallocAsync(reinterpret_cast<void**>(&blk), sizeof(int32_t) * (size_t)Bp * ntb, st) != hipSuccess) {
        (void)hipGetLastError();   // no stream-ordered allocator here: plain allocation, freed after the stream drains
        pooled = false;
        if (hipMalloc(reinterpret_cast<void**>(&blk), sizeof(int32_t) * (size_t)Bp * ntb) != hipSuccess) return (int)hipGetLastError();
    }
    if (key) bitmap_key_kernel<<<grid, kThreads, 0, st>>>(xs, t, bmp, accum, blk, rows);
    else     bitmap_value_kernel<<<grid, kThreads, 0, st>>>(xs, t, bmp, accum, blk, rows);
    block_scan_kernel<<<Bp, kThreads, 0, st>>>(blk, ntb, accum, totals, rows);
    block_fixup_kernel<<<grid, kD, 0, st>>>(blk, accum, rows);
    if (exclusive_prefix) head_offsets_kernel<<<1, 64, 0, st>>>(totals, Bp);
    const int err = (int)hipGetLastError();
    if (pooled) {
        (void)hipFreeAsync(blk, st);
    } else {
        (void)hipStreamSynchronize(st);
        (void)hipFree(blk);
    }
    return err;
}

int pack_common(bool key, void* stream, const void* x, int Bp, int t, int D, const int32_t* accum,
                const int64_t* head_off, const uint32_t* nz_offset, void* nz_flat, Rows rows)
{
    if (D != kD || Bp < 1 || t < 64 || (t & 63) || !x || !accum || (!head_off && !nz_offset)) return MUSTAFAR_EINVAL;
    if (!nz_flat) return 0;   // nothing to write: every tile of every head is empty
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(t / 64, Bp);
    auto xs = static_cast<const uint16_t*>(x);
    auto nz = static_cast<uint16_t*>(nz_flat);
    if (key) pack_key_kernel<<<grid, kThreads, 0, st>>>(xs, t, accum, head_off, nz_offset, nz, rows);
    else     pack_value_kernel<<<grid, kThreads, 0, st>>>(xs, t, accum, head_off, nz_offset, nz, rows);
    return (int)hipGetLastError();
}

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

}  // namespace

extern "C" {

int mustafar_prune_magnitude(void* stream, const void* x, void* out, int64_t n_rows, int D, int kth)
{
    if (D != kD || kth < 1 || kth > D || n_rows < 0 || !x || !out) return MUSTAFAR_EINVAL;
    if (n_rows == 0) return 0;
    const int64_t blocks = (n_rows + kPruneRows - 1) / kPruneRows;
    const unsigned grid = (unsigned)(blocks < 65536 ? blocks : 65536);
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

int mustafar_compress_pack_key(void* stream, const void* x, int Bp, int t, int D, const int64_t* /*bmp*/,
                               const int32_t* accum, const int64_t* head_off, void* nz_flat)
{ return pack_common(true, stream, x, Bp, t, D, accum, head_off, nullptr, nz_flat, fresh_rows(t)); }

int mustafar_compress_pack_value(void* stream, const void* x, int Bp, int t, int D, const int64_t* /*bmp*/,
                                 const int32_t* accum, const int64_t* head_off, void* nz_flat)
{ return pack_common(false, stream, x, Bp, t, D, accum, head_off, nullptr, nz_flat, fresh_rows(t)); }

// ---- in-place append into a cache view (model :339-390 without the re-copies) ----------------------------------------
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
    return pack_common(true, stream, x, Bp, t, D, reinterpret_cast<const int32_t*>(dst->idx), nullptr, dst->nz_offset, dst->nz, rows);
}

int mustafar_cache_append_pack_value(void* stream, const void* x, int Bp, int t, int D, const mustafar_cache_view* dst,
                                     int old_tokens)
{
    Rows rows;
    if (!view_rows(dst, old_tokens, t, rows) || !dst->nz) return MUSTAFAR_EINVAL;
    return pack_common(false, stream, x, Bp, t, D, reinterpret_cast<const int32_t*>(dst->idx), nullptr, dst->nz_offset, dst->nz, rows);
}

}  // extern "C"
