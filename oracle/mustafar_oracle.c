/*
 * mustafar_oracle.c -- CPU restatement of the Mustafar sparse-attention decode path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (mustafar_amd/) never
 * links, imports or calls anything in oracle/; it fails loudly without the HIP library.
 *
 * Parity status: the prune rule and the compressed format are PINNED by golden vectors
 * generated in-container from the reference's own code (oracle/gen_golden.py: the
 * reference's Triton compression kernels under TRITON_INTERPRET=1 and the reference's
 * dh_prune_* functions run on CPU).  The two SpMV functions are pinned only through those
 * format fixtures plus their mathematical definition: the reference CUDA kernels cannot be
 * built here (no nvcc; inline PTX cp.async/ldmatrix/mma.sync), and the reference has no
 * tests or golden vectors of its own -- see DESIGN.md "Oracle".
 *
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 * Plain C99, no dependencies; built by oracle/Makefile into oracle/liboracle.so.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ------------------------------------------------------------------ fp16 <-> fp32 */

static float h2f(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp  = (h >> 10) & 0x1fu;
    uint32_t man  = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400u));
            man &= 0x3ffu;
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

/* round-to-nearest-even, the behaviour of __float2half_rn (SpMM_Kernel.cuh:418, :673) */
static uint16_t f2h_rn(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t absx = x & 0x7fffffffu;
    if (absx >= 0x7f800000u) { /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((absx > 0x7f800000u) ? 0x200u : 0u));
    }
    if (absx >= 0x477ff000u) { /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7c00u);
    }
    if (absx < 0x33000001u) { /* < 2^-25 (or exactly 2^-25, ties to even 0) */
        return (uint16_t)sign;
    }
    int32_t  e = (int32_t)(absx >> 23) - 127;
    uint32_t m = (absx & 0x7fffffu) | 0x800000u;
    uint32_t shift;
    uint32_t hexp;
    if (e < -14) { /* subnormal half */
        shift = (uint32_t)(13 + (-14 - e));
        hexp  = 0;
    } else {
        shift = 13;
        hexp  = (uint32_t)(e + 15);
    }
    uint32_t hm   = m >> shift;
    uint32_t rem  = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hm & 1u))) hm++;
    /* hm carries the implicit bit for normals: fold exponent by addition so that a
       mantissa overflow bumps the exponent */
    uint32_t out;
    if (hexp == 0) out = hm;                       /* subnormal (hm may reach 0x400 = min normal) */
    else           out = ((hexp - 1) << 10) + hm;  /* hm in [0x400, 0x800] */
    return (uint16_t)(sign | out);
}

float    orc_h2f(uint16_t h) { return h2f(h); }
uint16_t orc_f2h(float f)    { return f2h_rn(f); }

/* ------------------------------------------------------------------ P1: magnitude prune
 * models/llama_mustafar_kernel.py:77-113 (dh_prune_key) and :117-153 (dh_prune_value),
 * identical bodies:
 *   num_to_keep = max(1, int(target_sparsity * D))          (:97)   -> argument `kth`
 *   thr  = kthvalue(|x|, num_to_keep) along D (k-th SMALLEST, 1-indexed)   (:103)
 *   mask = |x| >= thr                                       (:107)
 *   out  = x * mask                                         (:110)
 * `x * False` keeps the sign of x: pruned negatives become -0.0 (0x8000).  Non-finite
 * inputs are outside the contract (inf*0 = NaN payload is platform dependent).
 */
static int cmp_u16(const void* a, const void* b)
{
    uint16_t x = *(const uint16_t*)a, y = *(const uint16_t*)b;
    return (x > y) - (x < y);
}

int orc_prune_magnitude(const uint16_t* x, uint16_t* out, int64_t n_rows, int D, int kth)
{
    if (D <= 0 || kth < 1 || kth > D) return -1;
    uint16_t* mag = (uint16_t*)malloc((size_t)D * sizeof(uint16_t));
    if (!mag) return -2;
    for (int64_t r = 0; r < n_rows; r++) {
        const uint16_t* row = x + r * D;
        for (int i = 0; i < D; i++) mag[i] = row[i] & 0x7fffu; /* |x| as ordered integer */
        qsort(mag, (size_t)D, sizeof(uint16_t), cmp_u16);
        uint16_t thr = mag[kth - 1];
        for (int i = 0; i < D; i++) {
            uint16_t v = row[i];
            out[r * D + i] = ((v & 0x7fffu) >= thr) ? v : (uint16_t)(v & 0x8000u);
        }
    }
    free(mag);
    return 0;
}

/* ------------------------------------------------------------------ F1-F3, C1, C2: compression
 * kernel/compression.py.
 *   element != 0.0 test (:42, :103): false for +0.0 and -0.0, true for NaN.
 *   bitmap bit (63 - i) <=> element i of the tile non-zero (:43-44, shifts table :265-267).
 *   cnt = ((popc + 7) & ~7) >> 1  -- padded to x8 halfs, stored in half2 units (:46-48).
 *   K tile id = tokblk * D + d ; elements = tokens tokblk*64 .. +63 of channel d
 *       (:32-36 on the transposed input [B, D, t], :255).
 *   V tile id = tokblk * (D/64)*64 + col_tile*64 + r ; elements = channels col_tile*64..+63
 *       of token tokblk*64 + r (:87-97).
 *   accum_counts = [0, cumsum(cnt)] per head, int32 [B, tiles+1] (:294-298).
 *   packed stream: head h starts at half offset 2*sum_{h'<h} accum[h'][-1] (:302-304);
 *       tile stream at +2*accum[h][tile]; non-zeros in ascending element order (:168-174),
 *       padding slots zero (:309).
 */
static int is_nonzero_h(uint16_t v) { return (v & 0x7fffu) != 0; }

static uint16_t key_elem(const uint16_t* x, int t, int D, int b, int tile, int i)
{
    int d = tile % D, tokblk = tile / D;
    return x[((int64_t)b * t + (int64_t)tokblk * 64 + i) * D + d];
}
static uint16_t value_elem(const uint16_t* x, int t, int D, int b, int tile, int i)
{
    int tiles_per_row = D / 64, tiles_per_block = tiles_per_row * 64;
    int block_idx = tile / tiles_per_block, rem = tile % tiles_per_block;
    int col_tile = rem / 64, r = rem % 64;
    int64_t row = (int64_t)block_idx * 64 + r;
    return x[((int64_t)b * t + row) * D + col_tile * 64 + i];
}

typedef uint16_t (*elem_fn)(const uint16_t*, int, int, int, int, int);

static int bitmap_generic(elem_fn ef, const uint16_t* x, int B, int t, int D,
                          int64_t* bmp, int32_t* accum)
{
    if (t % 64 || D % 64 || B < 0) return -1;
    int tiles = t * D / 64;
    for (int b = 0; b < B; b++) {
        int32_t* acc = accum + (int64_t)b * (tiles + 1);
        acc[0] = 0;
        for (int tile = 0; tile < tiles; tile++) {
            uint64_t m = 0;
            int cnt = 0;
            for (int i = 0; i < 64; i++) {
                if (is_nonzero_h(ef(x, t, D, b, tile, i))) { m |= 1ull << (63 - i); cnt++; }
            }
            bmp[(int64_t)b * tiles + tile] = (int64_t)m;
            acc[tile + 1] = acc[tile] + (((cnt + 7) & ~7) >> 1);
        }
    }
    return 0;
}

static int pack_generic(elem_fn ef, const uint16_t* x, int B, int t, int D,
                        const int64_t* bmp, const int32_t* accum, uint16_t* nz_flat)
{
    if (t % 64 || D % 64) return -1;
    int tiles = t * D / 64;
    int64_t head_off = 0; /* halfs */
    for (int b = 0; b < B; b++) {
        const int32_t* acc = accum + (int64_t)b * (tiles + 1);
        int64_t head_len = 2 * (int64_t)acc[tiles];
        memset(nz_flat + head_off, 0, (size_t)head_len * sizeof(uint16_t));
        for (int tile = 0; tile < tiles; tile++) {
            uint64_t m = (uint64_t)bmp[(int64_t)b * tiles + tile];
            int64_t  o = head_off + 2 * (int64_t)acc[tile];
            int k = 0;
            for (int i = 0; i < 64; i++)
                if ((m >> (63 - i)) & 1ull) nz_flat[o + k++] = ef(x, t, D, b, tile, i);
        }
        head_off += head_len;
    }
    return 0;
}

int orc_bitmap_key(const uint16_t* x, int B, int t, int D, int64_t* bmp, int32_t* accum)
{ return bitmap_generic(key_elem, x, B, t, D, bmp, accum); }
int orc_bitmap_value(const uint16_t* x, int B, int t, int D, int64_t* bmp, int32_t* accum)
{ return bitmap_generic(value_elem, x, B, t, D, bmp, accum); }
int orc_pack_key(const uint16_t* x, int B, int t, int D, const int64_t* bmp,
                 const int32_t* accum, uint16_t* nz_flat)
{ return pack_generic(key_elem, x, B, t, D, bmp, accum, nz_flat); }
int orc_pack_value(const uint16_t* x, int B, int t, int D, const int64_t* bmp,
                   const int32_t* accum, uint16_t* nz_flat)
{ return pack_generic(value_elem, x, B, t, D, bmp, accum, nz_flat); }

/* ------------------------------------------------------------------ tile decompression
 * SpMM_Kernel.cuh:26-81 (stream addressing NZ[NZ_offset[g] + idx/4 + j], uint4 units) and
 * :109-151 (clz => MSB-first; the j-th non-zero of the stream lands at element `pos`).
 */
static void decompress_tile(uint64_t m, const uint16_t* stream, uint16_t dst[64])
{
    memset(dst, 0, 64 * sizeof(uint16_t));
    int j = 0;
    for (int pos = 0; pos < 64; pos++)
        if ((m >> (63 - pos)) & 1ull) dst[pos] = stream[j++];
}

/* ------------------------------------------------------------------ K1: key SpMV
 * mustafar_wrapper.cu:19-133 -> SpMM_API.cu:86-139 -> SpMM_Kernel.cuh:156-419.
 *   g = b / groups (:175); NZ_batch = NZ + NZ_offset[g] (uint4 units) (:177);
 *   idx_batch = idx + g*(1 + M*K/64) (:179); bmp_batch = bmp + g*(M*K/64) (:180);
 *   B_batch = B + b*K*N (:183); C_batch = C + b*M*N (:185);
 *   tile id = (m/64)*K + k (:248-254, :312); fp32 accumulate (mma f32), RN to fp16 (:418).
 *   C[b][n][m] = fp16( sum_k fp32(Khat_g[m][k]) * fp32(B[b][n][k]) ).
 * `Cd` (optional) receives the same sums accumulated in double, for tolerance budgeting.
 */
int orc_key_spmv(const int64_t* bmp, const uint16_t* NZ, const int32_t* idx,
                 const int32_t* NZ_offset, const uint16_t* Bm, uint16_t* C, double* Cd,
                 int M_Global, int N_Global, int K_Global, int Batch_Size, int groups)
{
    if (M_Global % 64 || K_Global % 64 || groups < 1 || Batch_Size % groups) return -1;
    const int M = M_Global, N = N_Global, K = K_Global;
    const int64_t tiles = (int64_t)M * K / 64;
    uint16_t tilebuf[64];
    float*  acc  = (float*)malloc(sizeof(float) * 64 * (size_t)N);
    double* accd = (double*)malloc(sizeof(double) * 64 * (size_t)N);
    if (!acc || !accd) { free(acc); free(accd); return -2; }
    for (int b = 0; b < Batch_Size; b++) {
        int g = b / groups;
        const uint16_t* NZ_batch  = NZ + (int64_t)(uint32_t)NZ_offset[g] * 8;
        const int32_t*  idx_batch = idx + (int64_t)g * (1 + tiles);
        const int64_t*  bmp_batch = bmp + (int64_t)g * tiles;
        const uint16_t* B_batch   = Bm + (int64_t)b * K * N;
        for (int tokblk = 0; tokblk < M / 64; tokblk++) {
            for (int i = 0; i < 64 * N; i++) { acc[i] = 0.f; accd[i] = 0.0; }
            for (int k = 0; k < K; k++) {
                int64_t tile = (int64_t)tokblk * K + k;
                const uint16_t* stream = NZ_batch + (int64_t)((uint32_t)idx_batch[tile] / 4) * 8;
                decompress_tile((uint64_t)bmp_batch[tile], stream, tilebuf);
                for (int n = 0; n < N; n++) {
                    float q = h2f(B_batch[(int64_t)n * K + k]);
                    for (int i = 0; i < 64; i++) {
                        float a = h2f(tilebuf[i]);
                        acc[n * 64 + i]  += a * q;
                        accd[n * 64 + i] += (double)a * (double)q;
                    }
                }
            }
            for (int n = 0; n < N; n++)
                for (int i = 0; i < 64; i++) {
                    int64_t o = ((int64_t)b * N + n) * M + (int64_t)tokblk * 64 + i;
                    C[o] = f2h_rn(acc[n * 64 + i]);
                    if (Cd) Cd[o] = accd[n * 64 + i];
                }
        }
    }
    free(acc); free(accd);
    return 0;
}

/* ------------------------------------------------------------------ V1: value SpMV
 * mustafar_wrapper.cu:139-263 -> SpMM_API.cu:193-254 -> SpMM_Kernel.cuh:421-676.
 *   M_Global = head_dim (must be 128 in the reference: TILE_M = 128), K_Global = T.
 *   same per-batch bases as the key kernel (:448-458);
 *   tile id = (k/64)*M + (m/64)*64 + (k%64) (:515-525, :575): 64 channels of one token;
 *   C[b][n][m] = fp16( sum_k fp32(Vhat_g[k][m]) * fp32(B[b][n][k]) ),  B = probs [Batch,N,K].
 */
int orc_value_spmv(const int64_t* bmp, const uint16_t* NZ, const int32_t* idx,
                   const int32_t* NZ_offset, const uint16_t* Bm, uint16_t* C, double* Cd,
                   int M_Global, int N_Global, int K_Global, int Batch_Size, int groups)
{
    if (M_Global % 64 || K_Global % 64 || groups < 1 || Batch_Size % groups) return -1;
    const int M = M_Global, N = N_Global, K = K_Global;
    const int64_t tiles = (int64_t)M * K / 64;
    uint16_t tilebuf[64];
    float*  acc  = (float*)malloc(sizeof(float) * (size_t)M * N);
    double* accd = (double*)malloc(sizeof(double) * (size_t)M * N);
    if (!acc || !accd) { free(acc); free(accd); return -2; }
    for (int b = 0; b < Batch_Size; b++) {
        int g = b / groups;
        const uint16_t* NZ_batch  = NZ + (int64_t)(uint32_t)NZ_offset[g] * 8;
        const int32_t*  idx_batch = idx + (int64_t)g * (1 + tiles);
        const int64_t*  bmp_batch = bmp + (int64_t)g * tiles;
        const uint16_t* B_batch   = Bm + (int64_t)b * K * N;
        for (int i = 0; i < M * N; i++) { acc[i] = 0.f; accd[i] = 0.0; }
        for (int k = 0; k < K; k++) {
            for (int half = 0; half < M / 64; half++) {
                int64_t tile = (int64_t)(k / 64) * M + (int64_t)half * 64 + (k % 64);
                const uint16_t* stream = NZ_batch + (int64_t)((uint32_t)idx_batch[tile] / 4) * 8;
                decompress_tile((uint64_t)bmp_batch[tile], stream, tilebuf);
                for (int n = 0; n < N; n++) {
                    float p = h2f(B_batch[(int64_t)n * K + k]);
                    for (int i = 0; i < 64; i++) {
                        float a = h2f(tilebuf[i]);
                        acc[n * M + half * 64 + i]  += a * p;
                        accd[n * M + half * 64 + i] += (double)a * (double)p;
                    }
                }
            }
        }
        for (int n = 0; n < N; n++)
            for (int m = 0; m < M; m++) {
                int64_t o = ((int64_t)b * N + n) * M + m;
                C[o] = f2h_rn(acc[n * M + m]);
                if (Cd) Cd[o] = accd[n * M + m];
            }
    }
    free(acc); free(accd);
    return 0;
}

/* ------------------------------------------------------------------ R1: split-K reduction
 * Reduction_Kernel.cuh:26-48: C[b][i] = fp16( sum_s fp32(W[b][s][i]) ), fp16 partials.
 * Dead in the reference (Split_K == 1, mustafar_wrapper.cu:128, :199); kept for the record.
 */
int orc_splitk_reduce(const uint16_t* W, uint16_t* C, int64_t elems, int Split_K, int Batch_Size)
{
    for (int b = 0; b < Batch_Size; b++)
        for (int64_t i = 0; i < elems; i++) {
            float s = 0.f;
            for (int k = 0; k < Split_K; k++) s += h2f(W[((int64_t)b * Split_K + k) * elems + i]);
            C[(int64_t)b * elems + i] = f2h_rn(s);
        }
    return 0;
}
