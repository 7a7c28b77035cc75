// spmv.hip -- batched sparse q.K^T and p.V over the bitmap-compressed KV cache, for gfx950 (MI355X).
//
// Replaces the reference's Key_Kernel / Value_Kernel / SplitK_Reduction
// (kernel/csrc/SpMM_Kernel.cuh:156-419, :421-676; Reduction_Kernel.cuh:26-48) and their launchers
// (kernel/csrc/SpMM_API.cu:86-139, :193-254).  Same inputs, same outputs, different algorithm:
//
//   * The reference decompresses every tile into a dense shared-memory tile and runs tensor-core MMA
//     against a query padded to 8 rows (7/8 of the MMA work is padding), once per q-head.
//   * Here nothing dense is ever materialised and no matrix core is used: the path is HBM-bound.
//     One wave64 (or two, one per half) owns a 64-token block of one kv-head = 128 consecutive tiles whose
//     packed non-zeros are ONE contiguous byte range of the stream.  The range is copied with 16-byte coalesced
//     loads into a wave-private LDS window (chunks of 32 tiles, next chunk prefetched in registers).  For each tile
//     the 64-bit bitmap lives in an SGPR pair (scalar load): bit-reversed, it is at once the lane mask of
//     the tile and the input of v_mbcnt, which gives every lane the rank of its element in the packed
//     stream; one ds_read_u16 fetches the value and v_fma_mix_f32 accumulates in fp32.
//       K: lane = token  (tile = 64 tokens of one channel),  coefficient = q[d]   -> no cross-lane sum.
//       V: lane = channel (tile = 64 channels of one token), coefficient = p[t]   -> per-lane sum over
//          tokens, then a workgroup LDS reduction and (Split_K > 1) fp32 partial slabs + a combine pass.
//     All q-heads of a GQA group are processed in the same pass, so the compressed bytes are read once
//     per kv-head rather than once per q-head.
//
// Numerics: products fp16 x fp16 are exact in fp32; accumulation is fp32 (order differs from the
// reference's MMA tree, as any two correct implementations do); final rounding RN to fp16 like
// __float2half_rn (SpMM_Kernel.cuh:418, :673).

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/mustafar_hip.h"

namespace {

typedef _Float16 h16;

// Wave timeline (tools/wave_trace.py; only in builds with -DMUSTAFAR_WAVE_TRACE): lane 0 of every SpMV wave records
// {start, end} of the 100 MHz wall clock, its HW_ID / XCC_ID registers and its grid position (record = 4 x u64).
#ifdef MUSTAFAR_WAVE_TRACE
__device__ unsigned long long* g_trace_buf = nullptr;
__device__ unsigned int g_trace_cap = 0;
struct WaveTrace {
    unsigned long long t0;
    unsigned int kernel;
    __device__ explicit WaveTrace(unsigned int k) : t0(__builtin_amdgcn_s_memrealtime()), kernel(k) {}
    __device__ void end() const
    {
        if (g_trace_buf == nullptr || (threadIdx.x & 63) != 0) return;
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        // no atomics (32k same-address atomics would serialise the kernel): the slot is the wave's grid position, key
        // launches in the lower half of the buffer, value launches in the upper half; a later launch overwrites
        const unsigned int half = g_trace_cap / 2;
        const unsigned int w = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (w >= half) return;
        const unsigned int slot = (kernel == 2 ? half : 0u) + w;
        const unsigned int hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        unsigned long long* r = g_trace_buf + 4ull * slot;
        r[0] = t0;
        r[1] = t1;
        r[2] = ((unsigned long long)xcc << 32) | hw;
        r[3] = ((unsigned long long)kernel << 56) | ((unsigned long long)(threadIdx.x >> 6) << 48) |
               ((unsigned long long)blockIdx.y << 24) | blockIdx.x;
    }
};
#define MUSTAFAR_TRACE_BEGIN(k) const WaveTrace wave_trace_(k)
#define MUSTAFAR_TRACE_END() wave_trace_.end()
// One-pass launches (tools/wave_trace_onepass.py): 16 x u64 per wave -- t[0] start, t[1] first key chunk staged, t[2] key phase
// done, t[3] softmax step done, t[4] first value chunk staged, t[5] value phase done (all of the wave's FIRST block), t[6] end;
// [8] HW_ID / XCC_ID, [9] grid position, [10] valid, [11] / [12] the SHADER clock (s_memtime) next to t[0] / t[6]: the wave's in-kernel clock is
// ([12] - [11]) / (t[6] - t[0]) x 100 MHz (round 6, tools/clock_probe.py; MI355X_MICROARCH.md, DVFS give-back (6)).  Slot = the wave's linear grid position.
struct PhaseTrace {
    unsigned long long t[7];
    unsigned long long c0;
    __device__ PhaseTrace() { t[0] = __builtin_amdgcn_s_memrealtime(); c0 = __builtin_amdgcn_s_memtime(); for (int i = 1; i < 7; i++) t[i] = 0; }
    __device__ void stamp(int i) { if (t[i] == 0) t[i] = __builtin_amdgcn_s_memrealtime(); }
    __device__ void end(unsigned int kernel)
    {
        if (g_trace_buf == nullptr || (threadIdx.x & 63) != 0) return;
        t[6] = __builtin_amdgcn_s_memrealtime();
        const unsigned long long c1 = __builtin_amdgcn_s_memtime();
        const unsigned int w = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (w >= g_trace_cap / 4) return;
        const unsigned int hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        unsigned long long* r = g_trace_buf + 16ull * w;
        for (int i = 0; i < 7; i++) r[i] = t[i];
        r[8] = ((unsigned long long)xcc << 32) | hw;
        r[9] = ((unsigned long long)kernel << 56) | ((unsigned long long)(threadIdx.x >> 6) << 48) | ((unsigned long long)blockIdx.y << 24) | blockIdx.x;
        r[10] = 1;
        r[11] = c0;
        r[12] = c1;
    }
};
#define MUSTAFAR_PTRACE_BEGIN() PhaseTrace phase_trace_
#define MUSTAFAR_PTRACE_STAMP(i) phase_trace_.stamp(i)
#define MUSTAFAR_PTRACE_END(k) phase_trace_.end(k)
#define MUSTAFAR_PTRACE_ARG , phase_trace_
#else
#define MUSTAFAR_TRACE_BEGIN(k)
#define MUSTAFAR_TRACE_END()
#define MUSTAFAR_PTRACE_BEGIN()
#define MUSTAFAR_PTRACE_STAMP(i)
#define MUSTAFAR_PTRACE_END(k)
#define MUSTAFAR_PTRACE_ARG
#endif

constexpr int kWaves      = 4;                    // waves per workgroup (key kernel; the value kernel defaults to kValueWaves = 8)
constexpr int kThreads    = 64 * kWaves;
constexpr int kChunkTiles = 32;                   // tiles staged per LDS chunk
constexpr int kChunkBytes = kChunkTiles * 128;    // worst case: 64 halfs per tile
constexpr int kStageBytes = kChunkBytes;          // lanes whose element is zero may read up to 126 B past the chunk's data: into
                                                  // the next wave's window or past the workgroup's LDS (reads as 0); never used.
                                                  // No pad: 4 x 4 KiB = 16 KiB exactly, one LDS allocation granule less per workgroup
constexpr int kD          = 128;                  // head_dim supported by this build
constexpr int kTilesPerTb = kD;                   // tiles per 64-token block (both formats)
// MFMA engine: row strides of the LDS coefficient tables.  Lane l reads 8 bytes of row l % 4 (ds_read_b64: bank =
// dword address mod 64); with the natural strides (256 B / 128 B) the four rows sit on the same banks and every read
// is a 4-way / 2-way conflict (measured 0.30 vs 1.1 wave-instructions/ns/CU).  One 16-byte granule of padding per row
// puts the four rows on disjoint bank pairs and keeps 16-byte alignment for the fill.
constexpr int kKeyTabStride = kD * 2 + 16;        // 272 B: [4 heads][128 channels] fp16
constexpr int kValTabStride = 64 * 2 + 16;        // 144 B: [4 heads][64 tokens] fp16

// Register image of one staged chunk (<= 4 KiB, 16-byte granules: 4 x uint4 per lane).
struct Stage {
    uint4 r0, r1, r2, r3;
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Issue the global loads of one chunk into registers.  Buffer loads with num_records = chunk length:
// lanes past the end get zeros from the range check and touch no memory, so there is no branch, no exec
// masking and nothing is ever read beyond the packed stream.  `src` and `len` are wave-uniform.
// SKIP: the upper half of the 4 KiB window is loaded (and, stage_commit, written to LDS) only for a chunk that reaches into it -- a
// chunk is ~2 KiB at 70 % sparsity.  Used by the matrix-pipe engine of the lean kernels, whose steps also read their coefficients from
// LDS (round 4b, kernel us c3 / c4 / c5: 32.5 / 57.1 / 110.0 -> 32.1 / 55.4 / 105.9).  NOT by the vector engines: nothing gained there
// (they are bound by vector issue), and the wave-uniform branch makes the compiler move scalar registers around while the scalar loads
// their asm statements issue are still in flight (wrong results; profiles/r04_probes.txt).
template <bool SKIP = false>
__device__ __forceinline__ Stage stage_issue(const unsigned char* __restrict__ src, uint32_t len, int lane)
{
#ifdef MUSTAFAR_PROBE_NOSTREAM   // timing-only build (tools/ab.py): zero records -> every stream load returns zeros without
    const __amdgpu_buffer_rsrc_t rsrc =   // touching memory; instruction stream, waits and LDS traffic unchanged, results wrong
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), (short)0, 0, 0x00020000);
#else
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), (short)0, (int)len, 0x00020000);
#endif
    const int off = lane * 16;
    Stage s;
    u32x4 t;
    // Cache policy of the stream loads: non-temporal (aux bit 1 = nt on gfx940+).  Every stream byte is read once per launch; with
    // the default policy that traffic pushed the metadata lines out of L2 between their vector prefetch and the scalar loads
    // that use them (c3 one-pass, pair grain: 52.3 -> 48.1 us with nt alone).  MUSTAFAR_STREAM_AUX: experiment knob.
#ifndef MUSTAFAR_STREAM_AUX
#define MUSTAFAR_STREAM_AUX 2
#endif
    // (skipping the 1-KiB pieces that lie wholly beyond the chunk's data -- a chunk is ~1.5 KiB at 70 % sparsity -- was
    // measured in round 2: the scalar branches cost the VALU engine 2-3 %, the matrix-pipe engine nothing either way)
    t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, MUSTAFAR_STREAM_AUX);        s.r0 = {t.x, t.y, t.z, t.w};
    t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 1024, 0, MUSTAFAR_STREAM_AUX); s.r1 = {t.x, t.y, t.z, t.w};
    if constexpr (SKIP) { s.r2 = {0u, 0u, 0u, 0u}; s.r3 = {0u, 0u, 0u, 0u}; }
    if (!SKIP || len > 2048u) {
        t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 2048, 0, MUSTAFAR_STREAM_AUX); s.r2 = {t.x, t.y, t.z, t.w};
        t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 3072, 0, MUSTAFAR_STREAM_AUX); s.r3 = {t.x, t.y, t.z, t.w};
    }
    return s;
}

// Copy the register image into the wave's LDS window (the zero tail is written too: the window is 4 KiB).
template <bool SKIP = false>
__device__ __forceinline__ void stage_commit(unsigned char* lds, const Stage& s, int lane, uint32_t len)
{
    uint4* w = reinterpret_cast<uint4*>(lds) + lane;
    w[0]   = s.r0;
    w[64]  = s.r1;
    if (!SKIP || len > 2048u) {   // (a skipped upper half keeps a previous chunk's bytes: no gather of this chunk addresses them)
        w[128] = s.r2;
        w[192] = s.r3;
    }
}

typedef h16 h16x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------------------
// Inner loop: hand-written for gfx950.  Measured issue rates on MI355X (tools/ubench/issue_rates.hip):
// v_mbcnt / v_lshl_add / v_cndmask / v_fma_mix / anything with an SGPR operand issue at ~1 wave-instruction
// per cycle per CU, SALU at ~1 per cycle per CU beside them.  At ~118 B of traffic per tile the HBM roofline
// leaves ~3 CU-cycles per tile, so the loop is written to the instruction, 8 tiles ("step") at a time:
//
//   VALU per tile : v_mbcnt_lo, v_mbcnt_hi (rank of the lane's element in the packed stream),
//                   v_lshl_add_u32 (LDS address), G x v_fma_mix_f32 (fp16 x fp16 -> fp32 accumulate, the
//                   coefficient is an SGPR half picked by op_sel)                                   = 3 + G
//   no mask op    : the FMAs run under EXEC = bit-reversed bitmap (one s_mov_b64 per tile), so lanes whose
//                   element is zero neither need a v_cndmask nor a clean gather result
//   SALU per tile : s_brev_b64, s_lshl2_add_u32 (tile's LDS byte offset), s_mov_b64 exec            = 3
//   per step      : 2 + G scalar loads (bitmaps x16, offsets x8, coefficients x4 per head) for the NEXT step,
//                   issued between the gather wait and the FMAs; two s_waitcnt lgkmcnt(0) (SMEM returns out
//                   of order and shares the counter with the LDS gathers, so every wait is a full drain).
//
// Every lgkm-counted operation of the loop (s_load, ds_read) lives inside asm statements so that the
// compiler's own counted waits never see a queue it does not know about; the chunk staging around it uses
// vmcnt (buffer loads) and LDS stores only.

typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));

// MUSTAFAR_META_TOUCH (experiment, off): see metab_issue_touch_at
#ifndef MUSTAFAR_META_TOUCH
#define MUSTAFAR_META_TOUCH 0
#endif
struct MetaB {         // tile metadata of one step (8 tiles) -- all SGPRs
    u32x16 bm;         // 8 bitmaps (lo, hi dwords)
    u32x8 ix;          // 8 stream offsets (half2 units)
#if MUSTAFAR_META_TOUCH
    uint32_t t0 = 0, t1 = 0;   // landing registers of metab_issue_touch_at's two touches (never read; they stay allocated until the wait)
#endif
};

// MUSTAFAR_META_EARLY (default): the NEXT step's bitmaps and offsets are requested in front of this step's gathers, so that the one
// wait of the step (every lgkmcnt wait is a full drain) covers LDS and scalar-memory latency at once; 0: requested behind the
// gather wait and waited for behind the FMAs (rounds 1-2: two waits per step, the second one all scalar-memory latency).
// 1 = the matrix-pipe pair form (its coefficients sit in LDS: c3 38.1 -> 37.6 us); 2 = experiments: the G <= 2 forms too (c2: key 10.7 ->
// 11.1 us, value 16.1 -> 16.5, one-pass 15.8 -> 16.3: their launches are latency chains of a few waves, the extra scalar spills cost
// more than the wait saved) and the GQA-4 vector engines (140+ scalar registers short: the compiler then spills registers that loads
// are still writing -- tools/check_smem_hazards.py).
#ifndef MUSTAFAR_META_EARLY
#define MUSTAFAR_META_EARLY 1
#endif

// Issue the scalar loads of the bitmaps / offsets of step S of a chunk (byte offsets are immediates).  Nothing may
// read `m` before a metab_wait() that follows.
#ifdef MUSTAFAR_PROBE_HOTMETA   // timing-only build: every wave reads the SAME 256 + 128 bytes of metadata (scalar-cache hits);
__device__ uint64_t g_hot_bmp[32] = {   // ~30 % dense masks; results are wrong, the instruction stream is unchanged
    0x1249249249249249ull, 0x2492492492492492ull, 0x4924924924924924ull, 0x9249249249249249ull, 0x1111111144444444ull, 0x0f0f00ff00f0f00full,
    0x1249249249249249ull, 0x2492492492492492ull, 0x4924924924924924ull, 0x9249249249249249ull, 0x1111111144444444ull, 0x0f0f00ff00f0f00full,
    0x1249249249249249ull, 0x2492492492492492ull, 0x4924924924924924ull, 0x9249249249249249ull, 0x1111111144444444ull, 0x0f0f00ff00f0f00full,
    0x1249249249249249ull, 0x2492492492492492ull, 0x4924924924924924ull, 0x9249249249249249ull, 0x1111111144444444ull, 0x0f0f00ff00f0f00full,
    0x1249249249249249ull, 0x2492492492492492ull, 0x4924924924924924ull, 0x9249249249249249ull, 0x1111111144444444ull, 0x0f0f00ff00f0f00full,
    0x1249249249249249ull, 0x2492492492492492ull};
__device__ uint32_t g_hot_idx[32] = {0, 12, 24, 36, 48, 60, 72, 84, 96, 108, 120, 132, 144, 156, 168, 180,
                                     192, 204, 216, 228, 240, 252, 264, 276, 288, 300, 312, 324, 336, 348, 360, 372};
#endif

template <int S>
__device__ __forceinline__ void metab_issue(MetaB& m, const uint64_t* __restrict__ bmp, const uint32_t* __restrict__ idx)
{
#ifdef MUSTAFAR_PROBE_HOTMETA
    bmp = g_hot_bmp;
    idx = g_hot_idx;
#endif
    asm volatile("s_load_dwordx16 %0, %2, %4\n\ts_load_dwordx8 %1, %3, %5"
                 : "=&s"(m.bm), "=&s"(m.ix)
                 : "s"(bmp), "s"(idx), "i"(S * 64), "i"(S * 32));
}
#ifdef MUSTAFAR_PROBE_NOMETAWAIT
// (rounds 3-5 had a timing-only build here whose next step did not wait for its metadata.  It consumed scalar registers that
// s_load_dwordx16 / x8 were still writing -- as STREAM OFFSETS of the gathers and, through the chunk bounds, of global loads -- and
// took a GPU box down with a memory-access fault twice (profiles/r05_probes.txt item 3).  Removed in round 6: it cannot be built.)
#error "MUSTAFAR_PROBE_NOMETAWAIT was removed: it uses registers a scalar load is still writing as addresses (GPU memory-access fault)"
#endif
#if MUSTAFAR_META_TOUCH
#define MUSTAFAR_MOPS_T , "+s"(m.t0), "+s"(m.t1)
#else
#define MUSTAFAR_MOPS_T
#endif
__device__ __forceinline__ void metab_wait(MetaB& m) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(m.bm), "+s"(m.ix) MUSTAFAR_MOPS_T); }
// Ordering point without an instruction: legal right after a wait that already drained the counter.
__device__ __forceinline__ void metab_ready(MetaB& m) { asm volatile("" : "+s"(m.bm), "+s"(m.ix) MUSTAFAR_MOPS_T); }

// Coefficients of step S: per head 8 halfs (4 dwords).  ONE buffer: the loads of step S are issued after the FMAs of
// step S - 1 have issued and land during the gather phase of step S (whose wait drains them too) -- a second buffer
// for the next step would push the loop past the 102 SGPRs a wave has.  Readable after gather_wait().
template <int G, int S>
__device__ __forceinline__ void coef_issue(u32x4 (&c)[G], const h16x2* const (&cp)[G])
{
    if constexpr (G == 4) {
        asm volatile("s_load_dwordx4 %0, %4, %8\n\ts_load_dwordx4 %1, %5, %8\n\t"
                     "s_load_dwordx4 %2, %6, %8\n\ts_load_dwordx4 %3, %7, %8"
                     : "=&s"(c[0]), "=&s"(c[1]), "=&s"(c[2]), "=&s"(c[3])
                     : "s"(cp[0]), "s"(cp[1]), "s"(cp[2]), "s"(cp[3]), "i"(S * 16));
    } else if constexpr (G == 2) {
        asm volatile("s_load_dwordx4 %0, %2, %4\n\ts_load_dwordx4 %1, %3, %4"
                     : "=&s"(c[0]), "=&s"(c[1])
                     : "s"(cp[0]), "s"(cp[1]), "i"(S * 16));
    } else {
        asm volatile("s_load_dwordx4 %0, %1, %2" : "=&s"(c[0]) : "s"(cp[0]), "i"(S * 16));
    }
}

struct Gathered {
    uint64_t m[8];   // bit i <=> element i of the tile non-zero (SGPR pairs; the format stores element i at bit 63 - i)
    uint32_t t[8];   // gathered halfs (VGPRs; meaningful only in lanes whose bit is set, and only after gather_wait)
};

#ifdef MUSTAFAR_PROBE_NOGATHER   // timing-only build: no LDS gather (the address stands in for the value); results wrong
#define MUSTAFAR_GATHER_LD(j) "v_mov_b32 %[t" #j "], %[t" #j "]\n\t"
#else
#define MUSTAFAR_GATHER_LD(j) "ds_read_u16 %[t" #j "], %[t" #j "]\n\t"
#endif
#define MUSTAFAR_GATHER(j)                                                  \
    "s_lshl2_add_u32 %[u" #j "], %[o" #j "], %[adj]\n\t"                     \
    "v_mbcnt_lo_u32_b32 %[t" #j "], %[l" #j "], 0\n\t"                        \
    "v_mbcnt_hi_u32_b32 %[t" #j "], %[h" #j "], %[t" #j "]\n\t"               \
    "v_lshl_add_u32 %[t" #j "], %[t" #j "], 1, %[u" #j "]\n\t"                \
    MUSTAFAR_GATHER_LD(j)
#define MUSTAFAR_GOPS(j) [l##j] "s"((uint32_t)g.m[j]), [h##j] "s"((uint32_t)(g.m[j] >> 32)), [o##j] "s"(m.ix[j])

// Gather the 8 tiles of a step from the wave's LDS window (no wait).
//   adj = (LDS byte address of the window) - 4 * (stream offset of its first byte)  ->  tile offset = 4*idx + adj
// (The tile offsets land in scalar registers of their own: formed in place, the offsets -- elements of the 8-register
// tuple the scalar load wrote -- were first copied out one s_mov_b32 each, 0.9 scalar instructions per tile.)
__device__ __forceinline__ void gather8(const MetaB& m, uint32_t adj, Gathered& g)
{
#pragma unroll
    for (int j = 0; j < 8; j++) g.m[j] = __builtin_bitreverse64(m.bm[2 * j] | ((uint64_t)m.bm[2 * j + 1] << 32));
    uint32_t u0, u1, u2, u3, u4, u5, u6, u7;
    asm volatile(MUSTAFAR_GATHER(0) MUSTAFAR_GATHER(1) MUSTAFAR_GATHER(2) MUSTAFAR_GATHER(3)
                 MUSTAFAR_GATHER(4) MUSTAFAR_GATHER(5) MUSTAFAR_GATHER(6) MUSTAFAR_GATHER(7)
                 : [t0] "=&v"(g.t[0]), [t1] "=&v"(g.t[1]), [t2] "=&v"(g.t[2]), [t3] "=&v"(g.t[3]), [t4] "=&v"(g.t[4]),
                   [t5] "=&v"(g.t[5]), [t6] "=&v"(g.t[6]), [t7] "=&v"(g.t[7]), [u0] "=&s"(u0), [u1] "=&s"(u1), [u2] "=&s"(u2),
                   [u3] "=&s"(u3), [u4] "=&s"(u4), [u5] "=&s"(u5), [u6] "=&s"(u6), [u7] "=&s"(u7)
                 : MUSTAFAR_GOPS(0), MUSTAFAR_GOPS(1), MUSTAFAR_GOPS(2), MUSTAFAR_GOPS(3), MUSTAFAR_GOPS(4), MUSTAFAR_GOPS(5),
                   MUSTAFAR_GOPS(6), MUSTAFAR_GOPS(7), [adj] "s"(adj)
                 : "scc");
}

// Drains the counter: the gathers AND the coefficient loads issued before them.
#ifdef MUSTAFAR_PROBE_NOLDSWAIT   // timing-only build: the FMAs do not wait for the gathers (nor the coefficients); results wrong
#define MUSTAFAR_GWAIT "s_nop 0"
#else
#define MUSTAFAR_GWAIT "s_waitcnt lgkmcnt(0)"
#endif
template <int G>
__device__ __forceinline__ void gather_wait(Gathered& g, u32x4 (&c)[G])
{
    if constexpr (G == 4)
        asm volatile(MUSTAFAR_GWAIT
                     : "+v"(g.t[0]), "+v"(g.t[1]), "+v"(g.t[2]), "+v"(g.t[3]), "+v"(g.t[4]), "+v"(g.t[5]), "+v"(g.t[6]), "+v"(g.t[7]),
                       "+s"(c[0]), "+s"(c[1]), "+s"(c[2]), "+s"(c[3]));
    else if constexpr (G == 2)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(g.t[0]), "+v"(g.t[1]), "+v"(g.t[2]), "+v"(g.t[3]), "+v"(g.t[4]), "+v"(g.t[5]), "+v"(g.t[6]), "+v"(g.t[7]),
                       "+s"(c[0]), "+s"(c[1]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(g.t[0]), "+v"(g.t[1]), "+v"(g.t[2]), "+v"(g.t[3]), "+v"(g.t[4]), "+v"(g.t[5]), "+v"(g.t[6]), "+v"(g.t[7]),
                       "+s"(c[0]));
}

#define MUSTAFAR_FMA(a, j, c, sel) "v_fma_mix_f32 %[" #a "], %[t" #j "], %[" #c "], %[" #a "] " sel "\n\t"
#define MUSTAFAR_LO "op_sel_hi:[1,1,0]"
#define MUSTAFAR_HI "op_sel:[0,1,0] op_sel_hi:[1,1,0]"
#define MUSTAFAR_FMA4(j, w, sel)                                                                     \
    "s_mov_b64 exec, %[m" #j "]\n\t" MUSTAFAR_FMA(a0, j, c0##w, sel) MUSTAFAR_FMA(a1, j, c1##w, sel)   \
    MUSTAFAR_FMA(a2, j, c2##w, sel) MUSTAFAR_FMA(a3, j, c3##w, sel)
#define MUSTAFAR_FMA2(j, w, sel) \
    "s_mov_b64 exec, %[m" #j "]\n\t" MUSTAFAR_FMA(a0, j, c0##w, sel) MUSTAFAR_FMA(a1, j, c1##w, sel)
#define MUSTAFAR_FMA1(j, w, sel) "s_mov_b64 exec, %[m" #j "]\n\t" MUSTAFAR_FMA(a0, j, c0##w, sel)
#define MUSTAFAR_FOPS                                                                                              \
    [m0] "s"(g.m[0]), [m1] "s"(g.m[1]), [m2] "s"(g.m[2]), [m3] "s"(g.m[3]), [m4] "s"(g.m[4]), [m5] "s"(g.m[5]),      \
    [m6] "s"(g.m[6]), [m7] "s"(g.m[7]), [t0] "v"(g.t[0]), [t1] "v"(g.t[1]), [t2] "v"(g.t[2]), [t3] "v"(g.t[3]),      \
    [t4] "v"(g.t[4]), [t5] "v"(g.t[5]), [t6] "v"(g.t[6]), [t7] "v"(g.t[7])
#define MUSTAFAR_COPS(h) [c##h##0] "s"(c[h][0]), [c##h##1] "s"(c[h][1]), [c##h##2] "s"(c[h][2]), [c##h##3] "s"(c[h][3])

// acc[h] += tile element x coefficient for the 8 gathered tiles.  The FMAs of tile j run under EXEC = its
// bitmap, so lanes whose element is zero are untouched: no v_cndmask and no clean gather result needed.
// EXEC contract (fma8 and gather8_clean): the statement loads EXEC with bitmaps and restores it with -1, declares no
// clobber, and therefore requires a FULL wave at entry: workgroups are multiples of 64 threads (static_asserts below)
// and every call site is wave-uniform.  tools/check_smem_hazards.py verifies on the generated ISA that each such
// statement restores EXEC before it ends and never sits inside a compiler-made divergent region.
static_assert(kThreads % 64 == 0, "the asm helpers restore EXEC to a full wave");
template <int G>
__device__ __forceinline__ void fma8(const u32x4 (&c)[G], const Gathered& g, float (&acc)[G])
{
    if constexpr (G == 4) {
        asm volatile(MUSTAFAR_FMA4(0, 0, MUSTAFAR_LO) MUSTAFAR_FMA4(1, 0, MUSTAFAR_HI) MUSTAFAR_FMA4(2, 1, MUSTAFAR_LO)
                     MUSTAFAR_FMA4(3, 1, MUSTAFAR_HI) MUSTAFAR_FMA4(4, 2, MUSTAFAR_LO) MUSTAFAR_FMA4(5, 2, MUSTAFAR_HI)
                     MUSTAFAR_FMA4(6, 3, MUSTAFAR_LO) MUSTAFAR_FMA4(7, 3, MUSTAFAR_HI) "s_mov_b64 exec, -1"
                     : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3])
                     : MUSTAFAR_FOPS, MUSTAFAR_COPS(0), MUSTAFAR_COPS(1), MUSTAFAR_COPS(2), MUSTAFAR_COPS(3));
    } else if constexpr (G == 2) {
        asm volatile(MUSTAFAR_FMA2(0, 0, MUSTAFAR_LO) MUSTAFAR_FMA2(1, 0, MUSTAFAR_HI) MUSTAFAR_FMA2(2, 1, MUSTAFAR_LO)
                     MUSTAFAR_FMA2(3, 1, MUSTAFAR_HI) MUSTAFAR_FMA2(4, 2, MUSTAFAR_LO) MUSTAFAR_FMA2(5, 2, MUSTAFAR_HI)
                     MUSTAFAR_FMA2(6, 3, MUSTAFAR_LO) MUSTAFAR_FMA2(7, 3, MUSTAFAR_HI) "s_mov_b64 exec, -1"
                     : [a0] "+v"(acc[0]), [a1] "+v"(acc[1])
                     : MUSTAFAR_FOPS, MUSTAFAR_COPS(0), MUSTAFAR_COPS(1));
    } else {
        asm volatile(MUSTAFAR_FMA1(0, 0, MUSTAFAR_LO) MUSTAFAR_FMA1(1, 0, MUSTAFAR_HI) MUSTAFAR_FMA1(2, 1, MUSTAFAR_LO)
                     MUSTAFAR_FMA1(3, 1, MUSTAFAR_HI) MUSTAFAR_FMA1(4, 2, MUSTAFAR_LO) MUSTAFAR_FMA1(5, 2, MUSTAFAR_HI)
                     MUSTAFAR_FMA1(6, 3, MUSTAFAR_LO) MUSTAFAR_FMA1(7, 3, MUSTAFAR_HI) "s_mov_b64 exec, -1"
                     : [a0] "+v"(acc[0])
                     : MUSTAFAR_FOPS, MUSTAFAR_COPS(0));
    }
}

// One staged chunk = 32 tiles = 4 steps of 8.  Per step:
//   coefficient loads(s) -> gather(s) -> wait (LDS latency of the last gather; the coefficients are in by then) ->
//   bitmap/offset loads(s+1) -> FMAs(s) -> wait (what is left of the scalar-load latency) -> ...
// i.e. the big scalar loads get the whole FMA phase (the only stretch without a wait) to land, the small ones the
// gather phase.
//   bmp/idx : wave-uniform pointers to the chunk's 32 bitmaps / stream offsets
//   cp[h]   : coefficient pairs of head h for the chunk's first tile (32 consecutive halfs are used)
template <int G>
__device__ __forceinline__ void chunk32(uint32_t adj, const uint64_t* __restrict__ bmp, const uint32_t* __restrict__ idx,
                                        const h16x2* const (&cp)[G], float (&acc)[G])
{
    constexpr bool kEarly = MUSTAFAR_META_EARLY > 1 && G <= 2;   // (experiment: see MUSTAFAR_META_EARLY)
    MetaB cur, nxt;
    u32x4 c[G];
    Gathered g;
    metab_issue<0>(cur, bmp, idx);
    coef_issue<G, 0>(c, cp);
    metab_wait(cur);
#define MUSTAFAR_STEP(S)                  \
    if constexpr (kEarly) metab_issue<S + 1>(nxt, bmp, idx); \
    gather8(cur, adj, g);                 \
    gather_wait<G>(g, c);                 \
    if constexpr (!kEarly) metab_issue<S + 1>(nxt, bmp, idx); \
    fma8<G>(c, g, acc);                   \
    if constexpr (kEarly) metab_ready(nxt); else metab_wait(nxt); \
    coef_issue<G, S + 1>(c, cp);          \
    cur = nxt;
    MUSTAFAR_STEP(0) MUSTAFAR_STEP(1) MUSTAFAR_STEP(2)
#undef MUSTAFAR_STEP
    gather8(cur, adj, g);
    gather_wait<G>(g, c);
    fma8<G>(c, g, acc);
}

// ---------------------------------------------------------------------------------------------------------
// Alternative FMA engine (opt-in, MUSTAFAR_FMA_ENGINE=mfma; OFF by default because the north_star rules MFMA out):
// the rank/gather part is unchanged, but the G = 4 multiply-accumulates of FOUR tiles are issued as ONE
// v_mfma_f32_4x4x4_16B_f16 -- 16 independent 4x4x4 blocks, block = 4 consecutive lanes.  With
//   B[k][j] = element of tile k in lane 4*blk + j   (the gathered halfs, packed two per VGPR)
//   A[i][k] = coefficient of head i for tile k      (lane 4*blk + i reads its head's row from a small LDS table)
// lane l receives D[i][l % 4] = sum_k A[i][k] * B[k][l % 4] = the four head accumulators of ITS element (layout
// verified by tools/ubench/mfma4x4_layout.hip).  No dense tile is built and nothing is reshaped into a GEMM; the
// matrix pipe is used as a 4-wide FMA unit, which takes ~3 of the 7 per-tile VALU issue slots away.
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GatheredClean {
    uint32_t t[8];    // gathered halfs (even tiles: bits 15:0, odd tiles: bits 31:16), exact zero where the tile has no element in this lane
    uint64_t a[2];    // A fragments (4 coefficient halfs of this lane's head) for tiles 0-3 and 4-7
};

// LDS byte address of the lane's element = 2 * rank + 4 * idx + adj, formed WITHOUT a scalar op per tile: the matrix-pipe
// engine is bound by the scalar pipe (5.9 SALU against 5.7 VALU wave-instructions per tile, one of each per SIMD per
// 4 cycles: profiles/r02_pmc_sq_c5_mfma_before.txt), so the tile's base moves to the vector side, where there is room --
//   v_mbcnt_lo  x = popcount(mask_lo below the lane) + adj/2      (adj/2 lives in a VGPR, set once per chunk)
//   v_mbcnt_hi  x += popcount(mask_hi below the lane)
//   v_lshl_add  x += idx << 1                                      (idx: the SGPR the scalar load wrote, read-only)
//   v_add       x += x                                             (full-rate VOP2)
// The arithmetic is mod 2^32 throughout (adj may be "negative"); adj is even (16-byte aligned window, offsets * 4).
#define MUSTAFAR_RANK(j)                                                    \
    "v_mbcnt_lo_u32_b32 %[x" #j "], %[l" #j "], %[vadj]\n\t"                 \
    "v_mbcnt_hi_u32_b32 %[x" #j "], %[h" #j "], %[x" #j "]\n\t"               \
    "v_lshl_add_u32 %[x" #j "], %[o" #j "], 1, %[x" #j "]\n\t"                \
    "v_add_u32 %[x" #j "], %[x" #j "], %[x" #j "]\n\t"                        \
    "v_mov_b32 %[t" #j "], 0\n\t"
// even tiles land in the low half of their register (upper half zero), odd tiles in the HIGH half: on gfx950 a d16_hi
// load zero-fills the other half (tools/ubench/probe2.hip), so a pair packs with one plain v_or_b32 (a full-rate VOP2;
// v_lshl_or_b32 / v_perm_b32 issue at half that rate)
#define MUSTAFAR_MLOAD(j) "s_mov_b64 exec, %[m" #j "]\n\tds_read_u16 %[t" #j "], %[x" #j "]\n\t"
#define MUSTAFAR_MLOAD_HI(j) "s_mov_b64 exec, %[m" #j "]\n\tds_read_u16_d16_hi %[t" #j "], %[x" #j "]\n\t"
#define MUSTAFAR_MOPS(j) [m##j] "s"(m##j), [l##j] "s"((uint32_t)m##j), [h##j] "s"((uint32_t)(m##j >> 32))

// COFF: byte offset of the step's first coefficient inside a head's row of the LDS coefficient table.
//   vadj = (LDS byte address of the window - 4 * (stream offset of its first byte)) / 2, the same in every lane
template <int COFF>
__device__ __forceinline__ void gather8_clean(const MetaB& m, uint32_t vadj, uint32_t ctab_lane, GatheredClean& g)
{
    const uint64_t m0 = __builtin_bitreverse64(m.bm[0] | ((uint64_t)m.bm[1] << 32));
    const uint64_t m1 = __builtin_bitreverse64(m.bm[2] | ((uint64_t)m.bm[3] << 32));
    const uint64_t m2 = __builtin_bitreverse64(m.bm[4] | ((uint64_t)m.bm[5] << 32));
    const uint64_t m3 = __builtin_bitreverse64(m.bm[6] | ((uint64_t)m.bm[7] << 32));
    const uint64_t m4 = __builtin_bitreverse64(m.bm[8] | ((uint64_t)m.bm[9] << 32));
    const uint64_t m5 = __builtin_bitreverse64(m.bm[10] | ((uint64_t)m.bm[11] << 32));
    const uint64_t m6 = __builtin_bitreverse64(m.bm[12] | ((uint64_t)m.bm[13] << 32));
    const uint64_t m7 = __builtin_bitreverse64(m.bm[14] | ((uint64_t)m.bm[15] << 32));
    uint32_t x0, x1, x2, x3, x4, x5, x6, x7;
    asm volatile(MUSTAFAR_RANK(0) MUSTAFAR_RANK(1) MUSTAFAR_RANK(2) MUSTAFAR_RANK(3)
                 MUSTAFAR_RANK(4) MUSTAFAR_RANK(5) MUSTAFAR_RANK(6) MUSTAFAR_RANK(7)
                 MUSTAFAR_MLOAD(0) MUSTAFAR_MLOAD_HI(1) MUSTAFAR_MLOAD(2) MUSTAFAR_MLOAD_HI(3)
                 MUSTAFAR_MLOAD(4) MUSTAFAR_MLOAD_HI(5) MUSTAFAR_MLOAD(6) MUSTAFAR_MLOAD_HI(7)
                 "s_mov_b64 exec, -1\n\t"
                 "ds_read_b64 %[a0], %[ct] offset:%[c0]\n\t"
                 "ds_read_b64 %[a1], %[ct] offset:%[c1]"
                 : [t0] "=&v"(g.t[0]), [t1] "=&v"(g.t[1]), [t2] "=&v"(g.t[2]), [t3] "=&v"(g.t[3]), [t4] "=&v"(g.t[4]),
                   [t5] "=&v"(g.t[5]), [t6] "=&v"(g.t[6]), [t7] "=&v"(g.t[7]), [a0] "=&v"(g.a[0]), [a1] "=&v"(g.a[1]),
                   [x0] "=&v"(x0), [x1] "=&v"(x1), [x2] "=&v"(x2), [x3] "=&v"(x3), [x4] "=&v"(x4), [x5] "=&v"(x5),
                   [x6] "=&v"(x6), [x7] "=&v"(x7)
                 : MUSTAFAR_MOPS(0), MUSTAFAR_MOPS(1), MUSTAFAR_MOPS(2), MUSTAFAR_MOPS(3), MUSTAFAR_MOPS(4), MUSTAFAR_MOPS(5),
                   MUSTAFAR_MOPS(6), MUSTAFAR_MOPS(7), [o0] "s"(m.ix[0]), [o1] "s"(m.ix[1]), [o2] "s"(m.ix[2]), [o3] "s"(m.ix[3]),
                   [o4] "s"(m.ix[4]), [o5] "s"(m.ix[5]), [o6] "s"(m.ix[6]), [o7] "s"(m.ix[7]), [vadj] "v"(vadj),
                   [ct] "v"(ctab_lane), [c0] "i"(COFF), [c1] "i"(COFF + 8));
}

__device__ __forceinline__ void gatherc_wait(GatheredClean& g)
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(g.t[0]), "+v"(g.t[1]), "+v"(g.t[2]), "+v"(g.t[3]), "+v"(g.t[4]), "+v"(g.t[5]), "+v"(g.t[6]),
                   "+v"(g.t[7]), "+v"(g.a[0]), "+v"(g.a[1]));
}

__device__ __forceinline__ h16x4 pack4(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3)
{
    const uint2 u = {t0 | t1, t2 | t3};   // t1 / t3 arrive in the high half (MUSTAFAR_MLOAD_HI)
    return __builtin_bit_cast(h16x4, u);
}

__device__ __forceinline__ void fma8_mfma(const GatheredClean& g, f32x4& acc)
{
    acc = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(h16x4, g.a[0]), pack4(g.t[0], g.t[1], g.t[2], g.t[3]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(h16x4, g.a[1]), pack4(g.t[4], g.t[5], g.t[6], g.t[7]), acc, 0, 0, 0);
}

// One staged chunk (32 tiles) on the MFMA engine; CBASE = byte offset of the chunk's first coefficient in a table row.
template <int CBASE>
__device__ __forceinline__ void chunk32_mfma(uint32_t adj_s, const uint64_t* __restrict__ bmp, const uint32_t* __restrict__ idx,
                                             uint32_t ctab_lane, f32x4& acc)
{
    MetaB cur, nxt;
    GatheredClean g;
    uint32_t adj;   // adj_s / 2 in a VGPR (one v_lshrrev per chunk)
    asm volatile("v_lshrrev_b32 %0, 1, %1" : "=v"(adj) : "s"(adj_s));
    metab_issue<0>(cur, bmp, idx);
    metab_wait(cur);
#define MUSTAFAR_STEP(S)                               \
    metab_issue<S + 1>(nxt, bmp, idx);                 \
    gather8_clean<CBASE + S * 16>(cur, adj, ctab_lane, g); \
    gatherc_wait(g);                                   \
    metab_ready(nxt);                                  \
    fma8_mfma(g, acc);                                 \
    cur = nxt;
    MUSTAFAR_STEP(0) MUSTAFAR_STEP(1) MUSTAFAR_STEP(2)
#undef MUSTAFAR_STEP
    gather8_clean<CBASE + 48>(cur, adj, ctab_lane, g);
    gatherc_wait(g);
    fma8_mfma(g, acc);
}

__device__ __forceinline__ uint32_t nzbits(uint4 v)
{
    return (v.x | v.y | v.z | v.w) & 0x7fff7fffu;   // -0.0 counts as zero
}

// Bit n (1 <= n < N) of the result is set iff pad row n of the dense operand holds a non-zero for any
// of the G heads over [col0, col0 + ncols) (ncols % 8 == 0).  The hook pads rows 1..7 with zeros
// (llama_mustafar_kernel.py:273, :313); the reference kernel computes them anyway, so a set bit makes the
// workgroup compute that row too, a clear bit makes it write exact zeros.  Uniform over the workgroup.
template <int G>
__device__ __forceinline__ uint32_t pad_row_mask(const h16* __restrict__ dense, int64_t row_len, int bh0, int N,
                                                 int col0, int ncols, uint32_t* sh_mask)
{
    if (threadIdx.x == 0) *sh_mask = 0u;
    __syncthreads();
    const int per_row = ncols / 8;
    uint32_t mine = 0;
    for (int u = threadIdx.x; u < G * (N - 1) * per_row; u += blockDim.x) {
        const int hn = u / per_row, k = u % per_row;
        const int h = hn / (N - 1), n = 1 + hn % (N - 1);
        const uint4 v = *reinterpret_cast<const uint4*>(dense + ((int64_t)(bh0 + h) * N + n) * row_len + col0 + k * 8);
        if (nzbits(v)) mine |= 1u << n;
    }
    if (mine) atomicOr(sh_mask, mine);
    __syncthreads();
    const uint32_t all = *sh_mask;
    __syncthreads();   // sh_mask lives in the stage area: nobody may stage before everybody has read it
    return all;
}

// The same question for a GROUP of consecutive token chunks (round 6, value_lean_kernel's pad workgroups): bit 8 c + n of the result is set iff pad row
// n (1 <= n < N) holds a non-zero for any of the G heads over chunk c of [col0, col0 + ncols) (chunks of chunk_cols columns, at most four; ncols % 8 == 0).
// Written for memory-level parallelism -- a thread's loads are independent and MUSTAFAR_PAD_UNROLL of them are in flight per trip; a row's slice is one contiguous run.
#ifndef MUSTAFAR_PROBE_N1AS8
#define MUSTAFAR_PROBE_N1AS8 0    // (timing probe, with MUSTAFAR_PROBE_NOPADWG: 8-row calls run the N = 1 instantiation on row 0 at the 8-row pitch of the dense operand; the output layout is wrong)
#endif
#ifndef MUSTAFAR_PAD_UNROLL
#define MUSTAFAR_PAD_UNROLL 4
#endif
#ifndef MUSTAFAR_PAD_GROUP
#define MUSTAFAR_PAD_GROUP 4      // token chunks per pad workgroup (at most 4: a byte of the mask per chunk)
#endif
template <int G, int N>
__device__ __forceinline__ uint32_t pad_group_mask(const h16* __restrict__ dense, int64_t row_len, int bh0, int col0, int ncols, int chunk_cols,
                                                   uint32_t* sh_mask)
{
    if (threadIdx.x == 0) *sh_mask = 0u;
    __syncthreads();
    const int V = ncols / 8;              // 16-byte vectors of a row's slice
    constexpr int R = G * (N - 1);        // rows
    const int nthr = (int)blockDim.x;
    int r = (int)threadIdx.x / V, k = (int)threadIdx.x - r * V;
    const int dr = nthr / V, dk = nthr - dr * V;
    uint32_t mine = 0;
    while (r < R) {
        u32x4 v[MUSTAFAR_PAD_UNROLL];
        int rr[MUSTAFAR_PAD_UNROLL], kk[MUSTAFAR_PAD_UNROLL];
#pragma unroll
        for (int i = 0; i < MUSTAFAR_PAD_UNROLL; i++) {
            rr[i] = r;
            kk[i] = k;
            v[i] = u32x4{0u, 0u, 0u, 0u};
            if (r < R) {
                const int h = r / (N - 1), n = 1 + r % (N - 1);
                v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(dense + ((int64_t)(bh0 + h) * N + n) * row_len + col0 + k * 8));
            }
            r += dr;
            k += dk;
            if (k >= V) { k -= V; r++; }
        }
#pragma unroll
        for (int i = 0; i < MUSTAFAR_PAD_UNROLL; i++)
            if ((v[i].x | v[i].y | v[i].z | v[i].w) & 0x7fff7fffu) mine |= 1u << (8 * ((kk[i] * 8) / chunk_cols) + 1 + rr[i] % (N - 1));   // (-0.0 counts as zero)
    }
    if (mine) atomicOr(sh_mask, mine);
    __syncthreads();
    const uint32_t all = *sh_mask;
    __syncthreads();   // sh_mask lives in the stage area: nobody may stage before everybody has read it
    return all;
}

// ------------------------------------------------------------------------------------------------ key
// The 5 stream offsets that bound the 4 chunks of a 64-token block (idx[0], idx[32], ..., idx[128]), fetched by
// lanes 0..4 with one vector load; bnd_get() broadcasts one of them into an SGPR (v_readlane).
__device__ __forceinline__ uint32_t bnd_load(const uint32_t* __restrict__ idx_t, int lane)
{
    return idx_t[(lane < 5 ? lane : 0) * kChunkTiles];
}
__device__ __forceinline__ uint32_t bnd_get(uint32_t bnd, int k) { return __builtin_amdgcn_readlane(bnd, k); }
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p)   // a pointer every lane holds alike, moved to scalar registers
{
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}

// Pull the metadata lines of one 64-token block (16 x 64 B of bitmaps, 9 of offsets) and, for the value kernel,
// its coefficient lines (2 x 64 B of probabilities per head) into L2 with ONE vector load, 4 bytes per line.
// The scalar loads of the inner loop then hit L2 instead of paying an HBM miss every other half-step (the
// scalar path has no prefetch of its own and every wait on it is a full drain).  The value is never used.
template <int NCOEF>
__device__ __forceinline__ uint32_t prefetch_meta(const uint64_t* __restrict__ bmp_t, const uint32_t* __restrict__ idx_t,
                                                  const h16x2* __restrict__ coef, uint32_t chead, int lane)
{
    const unsigned char* a = reinterpret_cast<const unsigned char*>(bmp_t) + lane * 64;
    if (lane >= 16) a = reinterpret_cast<const unsigned char*>(idx_t) + (lane - 16) * 64;
    if (NCOEF > 0 && lane >= 25) {
        const int k = lane - 25;   // head k / 2, line k % 2
        a = reinterpret_cast<const unsigned char*>(coef + (k >> 1) * chead) + (k & 1) * 64;
    }
    uint32_t v = 0;
    if (lane < 25 + 2 * NCOEF) v = *reinterpret_cast<const uint32_t*>(a);
    return v;
}
__device__ __forceinline__ void prefetch_done(uint32_t v) { asm volatile("" ::"v"(v)); }

// One 64-token block against coefficient row `qw` (head stride `chead` pairs) -> acc[h] for lane = token.
template <int G, bool MF, int CB, int CN>   // chunks [CB, CB + CN) of the token block
__device__ __forceinline__ void key_tokblk(unsigned char* smem, uint32_t lds_off,
                                           const uint64_t* __restrict__ bmp_t, const uint32_t* __restrict__ idx_t,
                                           const unsigned char* __restrict__ nz_h, const h16x2* __restrict__ qw,
                                           uint32_t chead, int lane, float (&acc)[G], uint32_t ctab_lane)
{
    f32x4 accv = {0.f, 0.f, 0.f, 0.f};   // MFMA engine: the four head accumulators as one register quad
    unsigned char* lds = smem + lds_off;
    const uint32_t lds_addr = (uint32_t)reinterpret_cast<uintptr_t>(lds);   // low half of a flat LDS pointer = LDS offset
    const uint32_t pf = prefetch_meta<0>(bmp_t, idx_t, qw, chead, lane);
    const uint32_t bnd = bnd_load(idx_t, lane);
    uint32_t i0 = bnd_get(bnd, CB);
    const uint32_t len0 = 4u * (bnd_get(bnd, CB + 1) - i0);
    Stage st = stage_issue(nz_h + 4ull * i0, len0, lane);
    stage_commit(lds, st, lane, len0);
    prefetch_done(pf);
#pragma unroll
    for (int c = CB; c < CB + CN; c++) {
        uint32_t n0 = 0, nlen = 0;
        if (c < CB + CN - 1) {
            n0 = bnd_get(bnd, c + 1);
            nlen = 4u * (bnd_get(bnd, c + 2) - n0);
            st = stage_issue(nz_h + 4ull * n0, nlen, lane);
        }
        __builtin_amdgcn_wave_barrier();
        {
            const h16x2* cp[G];
#pragma unroll
            for (int h = 0; h < G; h++) cp[h] = qw + h * chead + c * (kChunkTiles / 2);
            #ifdef MUSTAFAR_PROBE_HOTMETA
            const uint32_t adj = __builtin_amdgcn_readfirstlane(lds_addr);   // the fixed offsets of g_hot_idx stay inside the window
#else
            const uint32_t adj = __builtin_amdgcn_readfirstlane(lds_addr - 4u * i0);
#endif
            if constexpr (MF && G == 4) {
                if (c == 0)      chunk32_mfma<0>(adj, bmp_t, idx_t, ctab_lane, accv);
                else if (c == 1) chunk32_mfma<64>(adj, bmp_t + kChunkTiles, idx_t + kChunkTiles, ctab_lane, accv);
                else if (c == 2) chunk32_mfma<128>(adj, bmp_t + 2 * kChunkTiles, idx_t + 2 * kChunkTiles, ctab_lane, accv);
                else             chunk32_mfma<192>(adj, bmp_t + 3 * kChunkTiles, idx_t + 3 * kChunkTiles, ctab_lane, accv);
            } else {
                chunk32<G>(adj, bmp_t + c * kChunkTiles, idx_t + c * kChunkTiles, cp, acc);
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (c < CB + CN - 1) {
            stage_commit(lds, st, lane, nlen);
            i0 = n0;
        }
    }
    if constexpr (MF && G == 4) {
#pragma unroll
        for (int h = 0; h < 4; h++) acc[h] = accv[h];
    }
}

// ------------------------------------------------------------------------------------------------ dense window
// Window work of the fused decode path.  It depends only on q / p and the dense local window, not on the SpMV results,
// so it rides along in the two SpMV launches as a few extra workgroups -- whole grid rows in front of (or behind) the
// SpMV rows, see window_rows_last() -- instead of
// sitting in the two row kernels after them, where it was a chain of L2 round trips on the critical path.
union Vec8 {
    uint4 u;
    h16 h[8];
};

struct WinArgs {
    h16* win;             // [B', w_cap, 128] window buffer (nullptr: no window workgroups in this launch)
    const h16* fresh;     // [B', 128] newest row, stored at row w_len - 1 (or nullptr: already there)
    const int* w_extra;   // device step counter added to w_len (graph replay), or nullptr
    int w_len, w_cap;
    int rows;             // grid rows (blockIdx.y) taken by window workgroups; SpMV rows follow
    int nchunks;          // window token chunks per (kv-head, head batch)
};

__device__ __forceinline__ int window_len(const int* w_extra, int w_len, int w_cap)
{
    return w_extra ? min(w_len + *w_extra, w_cap) : w_len;
}

constexpr int kKeyWinChunk   = 64;    // window tokens per key-side window workgroup
constexpr int kValueWinChunk = 128;   // window tokens per value-side window workgroup (one partial slab each)

// scores[bh0 + h][T + w] = fp16( q[bh0 + h] . K_window[kvh][w] )   (llama_mustafar_kernel.py:270, :278).
// 256 threads, 64 tokens: 4 threads per token (32 channels = 4 x 16 B each, all loads in flight at once), q rows of
// the G heads staged in LDS, 2 shuffles to fold.
template <int G>
__device__ __forceinline__ void key_window_wg(unsigned char* smem, const h16* __restrict__ q, h16* wa_win,
                                              const h16* wa_fresh, int w_len, int w_cap, int nchunks,
                                              h16* __restrict__ scores, int T, int ld, int groups, int task)
{   // (the WinArgs fields arrive as scalars: a reference to the by-value kernel argument would pin it to a stack slot)
    const int hb_per_kv = groups / G;
    const int hb = task / nchunks, chunk = task % nchunks;
    const int kvh = hb / hb_per_kv, bh0 = kvh * groups + (hb % hb_per_kv) * G;
    if (chunk * kKeyWinChunk >= w_len) return;   // workgroup-uniform
    const int tid = threadIdx.x;
    h16* qs = reinterpret_cast<h16*>(smem);      // [G][128]
    if (tid < G * 16) reinterpret_cast<uint4*>(qs)[tid] = reinterpret_cast<const uint4*>(q + (int64_t)bh0 * kD)[tid];
    const int row = tid >> 2, part = tid & 3;
    const int w = chunk * kKeyWinChunk + row;
    const bool valid = w < w_len;
    const int wr = valid ? w : w_len - 1;
    const h16* fresh = wa_fresh ? wa_fresh + (int64_t)kvh * kD : nullptr;
    h16* win = wa_win + (int64_t)kvh * w_cap * kD;
    const h16* kr = ((fresh && wr == w_len - 1) ? fresh : win + (int64_t)wr * kD) + part * 32;
    Vec8 kv[4];
#pragma unroll
    for (int c = 0; c < 4; c++) kv[c].u = reinterpret_cast<const uint4*>(kr)[c];
    if (fresh && valid && w == w_len - 1 && hb % hb_per_kv == 0) {   // store the new key row (:270)
#pragma unroll
        for (int c = 0; c < 4; c++) reinterpret_cast<uint4*>(win + (int64_t)w * kD + part * 32)[c] = kv[c].u;
    }
    __syncthreads();
    float acc[G];
#pragma unroll
    for (int h = 0; h < G; h++) {
        acc[h] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            Vec8 qv;
            qv.u = reinterpret_cast<const uint4*>(qs + h * kD + part * 32)[c];
#pragma unroll
            for (int j = 0; j < 8; j++) acc[h] = __builtin_fmaf((float)kv[c].h[j], (float)qv.h[j], acc[h]);
        }
        acc[h] += __shfl_xor(acc[h], 1);
        acc[h] += __shfl_xor(acc[h], 2);
    }
    if (part == 0 && valid) {
#pragma unroll
        for (int h = 0; h < G; h++) scores[(int64_t)(bh0 + h) * ld + T + w] = (h16)acc[h];
    }
}

// slab[bh0 + h][c] = sum over the chunk's window tokens of p[bh0 + h][T + w] * V_window[kvh][w][c]   (:316).
// NW waves: 16 lanes x 16 bytes per row, NW * 4 rows per sweep, LDS fold; an empty chunk (beyond the current window
// length) writes zeros so that the finish pass can add every slab blindly.
template <int G, int NW>
__device__ __forceinline__ void value_window_wg(unsigned char* smem, const h16* __restrict__ probs, h16* wa_win,
                                                const h16* wa_fresh, int w_len, int w_cap, int nchunks,
                                                float* __restrict__ ws, int64_t slab_stride, int slab0, int T, int ld,
                                                int groups, int task)
{
    constexpr int kGrp = NW * 4, kRedLd = kD + 4;
    static_assert(kGrp * kRedLd * 4 <= NW * kStageBytes, "fold buffer must fit in the stage area");
    const int hb_per_kv = groups / G;
    const int hb = task / nchunks, chunk = task % nchunks;
    const int kvh = hb / hb_per_kv, bh0 = kvh * groups + (hb % hb_per_kv) * G;
    const int tid = threadIdx.x, sub = tid & 15, grp = tid >> 4;
    const int w0 = chunk * kValueWinChunk, w1 = min(w0 + kValueWinChunk, w_len);
    float* slab = ws + (int64_t)(slab0 + chunk) * slab_stride;
    float* red = reinterpret_cast<float*>(smem);   // [kGrp][kRedLd]
    const h16* fresh = wa_fresh ? wa_fresh + (int64_t)kvh * kD : nullptr;
    h16* win = wa_win + (int64_t)kvh * w_cap * kD;
    float acc[G][8];
#pragma unroll
    for (int h = 0; h < G; h++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[h][j] = 0.f;
    if (w0 >= w_len) {   // empty chunk: zeros, and out of the way at once
        for (int o = tid; o < G * kD; o += NW * 64) slab[(int64_t)bh0 * kD + o] = 0.f;
        return;
    }
    // two sweeps per pass, every load of the pass in flight before the first use: the workgroup sits in a slot the
    // SpMV workgroups are waiting for, so it is written for latency (within the register budget of 6 waves per SIMD)
    for (int wp = w0 + grp; wp < w1; wp += 2 * kGrp) {
        Vec8 vv[2];
        h16 pv[2][G];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int w = wp + u * kGrp;
            const int wr = w < w1 ? w : w0;
            const h16* vr = (fresh && wr == w_len - 1) ? fresh : win + (int64_t)wr * kD;
            vv[u].u = reinterpret_cast<const uint4*>(vr)[sub];
#pragma unroll
            for (int h = 0; h < G; h++) pv[u][h] = probs[(int64_t)(bh0 + h) * ld + T + wr];
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int w = wp + u * kGrp;
            if (w < w1) {
                if (fresh && w == w_len - 1 && hb % hb_per_kv == 0)   // store the new value row (:309)
                    reinterpret_cast<uint4*>(win + (int64_t)w * kD)[sub] = vv[u].u;
#pragma unroll
                for (int h = 0; h < G; h++) {
                    const float pw = (float)pv[u][h];
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[h][j] = __builtin_fmaf(pw, (float)vv[u].h[j], acc[h][j]);
                }
            }
        }
    }
#pragma unroll
    for (int h = 0; h < G; h++) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; j++) red[grp * kRedLd + sub * 8 + j] = acc[h][j];
        __syncthreads();
        if (tid < kD) {
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < kGrp; g++) sum += red[g * kRedLd + tid];
            slab[(int64_t)(bh0 + h) * kD + tid] = sum;
        }
    }
}

// grid: x = ceil(T/256) token super-blocks (SPLIT = 1: one wave per 64-token block) or ceil(T/128) (SPLIT = 2: two
// waves per token block, 64 channels each, partial scores folded through LDS -- twice the workgroups, half as long:
// used when the SPLIT = 1 grid would fit on the chip in a single round), y = kv-heads * (groups / G)
#ifndef MUSTAFAR_KEY_WAVES      // experiment knob (tools/build_variant.sh): minimum waves per SIMD the key kernel is compiled for
#define MUSTAFAR_KEY_BOUNDS __launch_bounds__(kThreads)
#else
#define MUSTAFAR_KEY_BOUNDS __launch_bounds__(kThreads, MUSTAFAR_KEY_WAVES)
#endif
template <int G, bool MF, int SPLIT>
__global__ MUSTAFAR_KEY_BOUNDS void key_spmv_kernel(
    const uint64_t* __restrict__ bmp, const unsigned char* __restrict__ nz, const uint32_t* __restrict__ idx,
    const uint32_t* __restrict__ nz_off, const h16* __restrict__ q, h16* __restrict__ out, int T, int N, int groups,
    int ldc, WinArgs wa, int64_t bmp_stride, int64_t idx_stride, uint32_t nz_stride)
{   // nz_stride != 0: head h's stream starts at uint4 index h * nz_stride (an arena: no load of nz_off in the wave's start-up chain)
    // ldc: row stride of `out` in halfs (T for the reference layout); bmp_stride / idx_stride: elements between the heads'
    // rows of `bmp` / `idx` (0 = the reference's contiguous layout, 2T and 2T + 1; larger for an arena with spare capacity)
    constexpr int kTabBytes = (MF && G == 4) ? 4 * kKeyTabStride : 0;   // MFMA engine: q rows of the 4 heads
    __shared__ __attribute__((aligned(16))) unsigned char smem[kWaves * kStageBytes + kTabBytes];
    MUSTAFAR_TRACE_BEGIN(1);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wrows = wa.rows < 0 ? -wa.rows : wa.rows;            // window rows lead (rows > 0) or trail (rows < 0) the grid
    const int wy = wa.rows < 0 ? (int)blockIdx.y - ((int)gridDim.y - wrows) : (int)blockIdx.y;
    if (wa.rows != 0 && wy >= 0 && wy < wrows) {   // fused decode only (N == 1): window scores
        const int task = wy * gridDim.x + blockIdx.x;
        if (task < (int)(gridDim.y - wrows) * wa.nchunks)
            key_window_wg<G>(smem, q, wa.win, wa.fresh, window_len(wa.w_extra, wa.w_len, wa.w_cap), wa.w_cap, wa.nchunks, out, T, ldc,
                             groups, task);
        MUSTAFAR_TRACE_END();
        return;
    }
    const int by = blockIdx.y - (wa.rows > 0 ? wa.rows : 0);
    const int hb_per_kv = groups / G;
    const int kvh = by / hb_per_kv;
    const int bh0 = kvh * groups + (by % hb_per_kv) * G;
    const int ntb = T >> 6;
    constexpr int kTbPerWg = kWaves / SPLIT;
    const int tb  = blockIdx.x * kTbPerWg + wave / SPLIT;
    const int part = wave % SPLIT;   // which half of the channels this wave covers (SPLIT = 2)
    const int64_t tiles = (int64_t)ntb * kTilesPerTb;

    const uint64_t* bmp_t = bmp + (int64_t)kvh * (bmp_stride ? bmp_stride : tiles) + (int64_t)tb * kTilesPerTb;
    const uint32_t* idx_t = idx + (int64_t)kvh * (idx_stride ? idx_stride : tiles + 1) + (int64_t)tb * kTilesPerTb;
    const unsigned char* nz_h = nz + 16ull * (nz_stride ? (uint64_t)kvh * nz_stride : (uint64_t)nz_off[kvh]);
    const uint32_t chead = (uint32_t)N * (kD / 2);

    uint32_t rows = 1u;   // bit n: row n has to be computed
    if (N > 1)
        rows |= pad_row_mask<G>(q, kD, bh0, N, 0, kD, reinterpret_cast<uint32_t*>(smem));

    // One wave per token block (or two).  A grid capped at a few resident rounds with the waves looping over their
    // blocks was measured too (round 2, both engines, caps 1024-4096 workgroups): 3-17 % slower at c5, no gain at c3 --
    // what a one-block wave pays is the memory latency of its start-up chain, which a looping wave pays per block as well.
    const int tok0 = blockIdx.x * kTbPerWg * 64;
    const int ntok = min(kTbPerWg * 64, T - tok0);
    uint32_t ctab_lane = 0;
    for (int n = 0; n < N; n++) {
        if constexpr (MF && G == 4) {
            if ((rows >> n) & 1u) {   // coefficient table: row n of the 4 heads, 16 bytes per thread
                unsigned char* tab = smem + kWaves * kStageBytes;
                __syncthreads();
                if (threadIdx.x < 64)
                    *reinterpret_cast<uint4*>(tab + (threadIdx.x >> 4) * kKeyTabStride + (threadIdx.x & 15) * 16) =
                        *reinterpret_cast<const uint4*>(q + ((int64_t)(bh0 + (threadIdx.x >> 4)) * N + n) * kD + (threadIdx.x & 15) * 8);
                __syncthreads();
                ctab_lane = (uint32_t)reinterpret_cast<uintptr_t>(tab) + (lane & 3) * kKeyTabStride;
            }
        }
        if ((rows >> n) & 1u) {
            float acc[G];
#pragma unroll
            for (int h = 0; h < G; h++) acc[h] = 0.f;
            const h16x2* qw = reinterpret_cast<const h16x2*>(q + ((int64_t)bh0 * N + n) * kD);
            if (tb < ntb) {
                if constexpr (SPLIT == 1) {
#ifdef MUSTAFAR_PROBE_REPEAT   // timing-only build: every wave processes its token block MUSTAFAR_PROBE_REPEAT times (warm-cache cost per block)
                    for (int rep = 0; rep < MUSTAFAR_PROBE_REPEAT; rep++)
#endif
                    key_tokblk<G, MF, 0, 4>(smem, wave * kStageBytes, bmp_t, idx_t, nz_h, qw, chead, lane, acc, ctab_lane);
                } else {
                    if (part == 0) key_tokblk<G, MF, 0, 2>(smem, wave * kStageBytes, bmp_t, idx_t, nz_h, qw, chead, lane, acc, ctab_lane);
                    else           key_tokblk<G, MF, 2, 2>(smem, wave * kStageBytes, bmp_t, idx_t, nz_h, qw, chead, lane, acc, ctab_lane);
                }
            }
            if constexpr (SPLIT == 2) {   // fold the two channel halves: odd wave -> its own (now dead) stage window -> even wave
                float* fold = reinterpret_cast<float*>(smem + wave * kStageBytes);
                __syncthreads();
                if (part == 1) {
#pragma unroll
                    for (int h = 0; h < G; h++) fold[h * 64 + lane] = acc[h];
                }
                __syncthreads();
                if (part == 0) {
                    const float* other = reinterpret_cast<const float*>(smem + (wave + 1) * kStageBytes);
#pragma unroll
                    for (int h = 0; h < G; h++) acc[h] += other[h * 64 + lane];
                }
            }
            if (tb < ntb && part == 0) {
#pragma unroll
                for (int h = 0; h < G; h++)
                    out[((int64_t)(bh0 + h) * N + n) * ldc + (int64_t)tb * 64 + lane] = (h16)acc[h];
            }
            if constexpr (SPLIT == 2) __syncthreads();   // the fold buffers are stage windows again in the next row
        } else {   // exact zeros, 16 bytes per lane
            const int per_row = ntok / 8;
            const uint4 z = {0u, 0u, 0u, 0u};
            for (int u = threadIdx.x; u < G * per_row; u += kThreads) {
                const int h = u / per_row, k = u % per_row;
                *reinterpret_cast<uint4*>(out + ((int64_t)(bh0 + h) * N + n) * ldc + tok0 + k * 8) = z;
            }
        }
    }
    MUSTAFAR_TRACE_END();
}

// ------------------------------------------------------------------------------------------------ value
// Accumulate token blocks tb_first, tb_first+4, ... < tb_end of one kv-head: lane = channel,
// acc0 = channels 0..63, acc1 = channels 64..127, coefficient row `pw` (head stride `chead` pairs).
template <int G, bool MF, int CB, int CN, int STRIDE, bool PTAB_READY = false>   // chunks [CB, CB + CN) of every STRIDE-th token block
__device__ __forceinline__ void value_tokblks(unsigned char* smem, uint32_t lds_off,
                                              const uint64_t* __restrict__ bmp_h, const uint32_t* __restrict__ idx_h,
                                              const unsigned char* __restrict__ nz_h, const h16x2* __restrict__ pw,
                                              uint32_t chead, int tb_first, int tb_end, int lane,
                                              float (&acc0)[G], float (&acc1)[G], unsigned char* ptab)
{
    if (tb_first >= tb_end) return;
    unsigned char* lds = smem + lds_off;
    const uint32_t lds_addr = (uint32_t)reinterpret_cast<uintptr_t>(lds);
    // MFMA engine: per-wave coefficient table [4 heads][64 tokens] of the current token block (4 x kValTabStride B), filled by
    // lanes 0..31 (16 B each) one token block ahead; lane l reads the row of head l % 4.
    f32x4 accv0 = {0.f, 0.f, 0.f, 0.f}, accv1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (MF && G == 4) {   // continue from the caller's sums (zero in the two-launch kernel, the running partial output in the one-pass launch)
#pragma unroll
        for (int h = 0; h < 4; h++) { accv0[h] = acc0[h]; accv1[h] = acc1[h]; }
    }
    uint32_t ctab_lane = 0;
    uint4 ptv = {0u, 0u, 0u, 0u};
    auto ptab_load = [&](int tb) -> uint4 {
        const h16* src = reinterpret_cast<const h16*>(pw) + (int64_t)((lane >> 3) & 3) * chead * 2 + (int64_t)tb * 64 + (lane & 7) * 8;
        return *reinterpret_cast<const uint4*>(src);
    };
    if constexpr (MF && G == 4) {
        ctab_lane = (uint32_t)reinterpret_cast<uintptr_t>(ptab) + (lane & 3) * kValTabStride;
        if constexpr (!PTAB_READY) {   // (PTAB_READY: the caller filled the table of its single block -- the one-pass launch)
            ptv = ptab_load(tb_first);
            if (lane < 32) *reinterpret_cast<uint4*>(ptab + (lane >> 3) * kValTabStride + (lane & 7) * 16) = ptv;
        }
    }
    uint32_t pf = prefetch_meta<G>(bmp_h + (int64_t)tb_first * kTilesPerTb, idx_h + (int64_t)tb_first * kTilesPerTb,
                                   pw + (uint32_t)tb_first * 32u, chead, lane);
    uint32_t bnd = bnd_load(idx_h + (int64_t)tb_first * kTilesPerTb, lane);
    uint32_t i0 = bnd_get(bnd, CB);
    const uint32_t len0 = 4u * (bnd_get(bnd, CB + 1) - i0);
    Stage st = stage_issue(nz_h + 4ull * i0, len0, lane);
    stage_commit(lds, st, lane, len0);
    for (int tb = tb_first; tb < tb_end; tb += STRIDE) {
        const uint64_t* bmp_t = bmp_h + (int64_t)tb * kTilesPerTb;
        const uint32_t* idx_t = idx_h + (int64_t)tb * kTilesPerTb;
        const bool more = tb + STRIDE < tb_end;
        prefetch_done(pf);
        // metadata lines and chunk bounds of the wave's next token block, in flight while this one is processed
        const int tbn = more ? tb + STRIDE : tb;
        pf = prefetch_meta<G>(bmp_h + (int64_t)tbn * kTilesPerTb, idx_h + (int64_t)tbn * kTilesPerTb,
                              pw + (uint32_t)tbn * 32u, chead, lane);
        const uint32_t bnd_next = bnd_load(more ? idx_t + STRIDE * kTilesPerTb : idx_t, lane);
        if constexpr (MF && G == 4 && !PTAB_READY) ptv = ptab_load(tbn);
#pragma unroll
        for (int c = CB; c < CB + CN; c++) {
            uint32_t n0 = 0, nlen = 0;
            const bool last = c == CB + CN - 1;
            const bool has_next = !last || more;
            if (has_next) {
                n0 = !last ? bnd_get(bnd, c + 1) : bnd_get(bnd_next, CB);
                const uint32_t n1 = !last ? bnd_get(bnd, c + 2) : bnd_get(bnd_next, CB + 1);
                nlen = 4u * (n1 - n0);
                st = stage_issue(nz_h + 4ull * n0, nlen, lane);
            }
            __builtin_amdgcn_wave_barrier();
            // chunk c: channel half = c >> 1, tokens (c & 1) * 32 .. +31 of the block
            const h16x2* cp[G];
#pragma unroll
            for (int h = 0; h < G; h++) cp[h] = pw + h * chead + ((uint32_t)tb * 64u + (c & 1) * 32u) / 2u;
            #ifdef MUSTAFAR_PROBE_HOTMETA
            const uint32_t adj = __builtin_amdgcn_readfirstlane(lds_addr);   // the fixed offsets of g_hot_idx stay inside the window
#else
            const uint32_t adj = __builtin_amdgcn_readfirstlane(lds_addr - 4u * i0);
#endif
            if constexpr (MF && G == 4) {
                if (c == 0)      chunk32_mfma<0>(adj, bmp_t, idx_t, ctab_lane, accv0);
                else if (c == 1) chunk32_mfma<64>(adj, bmp_t + kChunkTiles, idx_t + kChunkTiles, ctab_lane, accv0);
                else if (c == 2) chunk32_mfma<0>(adj, bmp_t + 2 * kChunkTiles, idx_t + 2 * kChunkTiles, ctab_lane, accv1);
                else             chunk32_mfma<64>(adj, bmp_t + 3 * kChunkTiles, idx_t + 3 * kChunkTiles, ctab_lane, accv1);
            } else {
                if (c < 2) chunk32<G>(adj, bmp_t + c * kChunkTiles, idx_t + c * kChunkTiles, cp, acc0);
                else       chunk32<G>(adj, bmp_t + c * kChunkTiles, idx_t + c * kChunkTiles, cp, acc1);
            }
            __builtin_amdgcn_wave_barrier();
            if (has_next) {
                stage_commit(lds, st, lane, nlen);
                i0 = n0;
            }
        }
        if constexpr (MF && G == 4) {
            if (more && lane < 32) *reinterpret_cast<uint4*>(ptab + (lane >> 3) * kValTabStride + (lane & 7) * 16) = ptv;   // this block's table is dead now
            __builtin_amdgcn_wave_barrier();
        }
        bnd = bnd_next;
    }
    prefetch_done(pf);
    if constexpr (MF && G == 4) {
#pragma unroll
        for (int h = 0; h < 4; h++) {
            acc0[h] = accv0[h];
            acc1[h] = accv1[h];
        }
    }
}

// grid: x = Split_K token chunks, y = kv-heads * (groups / G).  NW waves per workgroup, SPLIT of them share a token
// block (SPLIT = 2: one wave per 64-channel half, i.e. chunks {0,1} / {2,3}; the halves write disjoint accumulators,
// so the workgroup reduce below needs no change).  The finer grain matters when a wave only gets a few token blocks:
// the launch is sized to ONE round of resident workgroups (mustafar_value_pick_split_k) and ends when the longest
// wave does.
//   direct != 0 (one chunk): fp16 results go straight to `out`;
//   else fp32 partial slabs ws[(s*BH + bh)*N + n][128] + one row mask per workgroup in `flags`.
#ifndef MUSTAFAR_VALUE_MF_WAVES   // experiment knob (tools/build_variant.sh): minimum waves per SIMD of the matrix-pipe value kernels
#define MUSTAFAR_VALUE_BOUNDS __launch_bounds__(NW * 64)
#else
#define MUSTAFAR_VALUE_BOUNDS __launch_bounds__(NW * 64, MF ? MUSTAFAR_VALUE_MF_WAVES : 1)
#endif
#ifndef MUSTAFAR_PAD_LATE
#define MUSTAFAR_PAD_LATE 0   // (0 = the pad rows are read and reduced in front of row 0, rounds 1-5; 1 = requested in front, looked at behind row 0; 2 = read and looked at behind row 0)
#endif
template <int G, bool MF, int NW, int SPLIT, bool WIN = true>   // WIN = false: no window workgroups in the launch (the reference entry point): round 5, as key_lean_kernel
__global__ MUSTAFAR_VALUE_BOUNDS void value_spmv_kernel(
    const uint64_t* __restrict__ bmp, const unsigned char* __restrict__ nz, const uint32_t* __restrict__ idx,
    const uint32_t* __restrict__ nz_off, const h16* __restrict__ p, h16* __restrict__ out, float* __restrict__ ws,
    uint32_t* __restrict__ flags, int T, int N, int groups, int BH, int tb_per_wg, int direct, int ldb, WinArgs wa,
    int64_t bmp_stride, int64_t idx_stride, uint32_t nz_stride)
{   // bmp_stride / idx_stride: as in key_spmv_kernel;  ldb: row stride of `p` in halfs (T for the reference layout; must be even, % 8 == 0 for the MFMA engine)
    constexpr int kTabBytes = (MF && G == 4) ? NW * 4 * kValTabStride : 0;
    constexpr int kStride = NW / SPLIT;   // token blocks in flight per workgroup
    __shared__ __attribute__((aligned(16))) unsigned char smem[NW * kStageBytes + kTabBytes];
    MUSTAFAR_TRACE_BEGIN(2);
    static_assert(NW * kStageBytes >= NW * 2 * 4 * 64 * 4, "reduce buffer must fit in the stage area");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wrows = wa.rows < 0 ? -wa.rows : wa.rows;            // window rows lead (rows > 0) or trail (rows < 0) the grid
    const int wy = wa.rows < 0 ? (int)blockIdx.y - ((int)gridDim.y - wrows) : (int)blockIdx.y;
    if constexpr (WIN) {
        if (wa.rows != 0 && wy >= 0 && wy < wrows) {   // fused decode only (N == 1): window p.V -> slabs gridDim.x ..
            const int task = wy * gridDim.x + blockIdx.x;
            if (task < (int)(gridDim.y - wrows) * wa.nchunks)
                value_window_wg<G, NW>(smem, p, wa.win, wa.fresh, window_len(wa.w_extra, wa.w_len, wa.w_cap), wa.w_cap, wa.nchunks, ws,
                                       (int64_t)BH * kD, gridDim.x, T, ldb, groups, task);
            MUSTAFAR_TRACE_END();
            return;
        }
    }
    const int by = blockIdx.y - (WIN && wa.rows > 0 ? wa.rows : 0);
    const int hb_per_kv = groups / G;
    const int kvh = by / hb_per_kv;
    const int bh0 = kvh * groups + (by % hb_per_kv) * G;
    const int ntb = T >> 6;
    const int tb0 = blockIdx.x * tb_per_wg;
    const int tb_end = min(ntb, tb0 + tb_per_wg);
    const int64_t tiles = (int64_t)ntb * kTilesPerTb;

    const uint64_t* bmp_h = bmp + (int64_t)kvh * (bmp_stride ? bmp_stride : tiles);
    const uint32_t* idx_h = idx + (int64_t)kvh * (idx_stride ? idx_stride : tiles + 1);
    const unsigned char* nz_h = nz + 16ull * (nz_stride ? (uint64_t)kvh * nz_stride : (uint64_t)nz_off[kvh]);
    float* red = reinterpret_cast<float*>(smem);   // [NW][2*G][64], overlays the stage windows
    float* ws_slab = ws + (int64_t)blockIdx.x * BH * N * kD;
    const uint32_t chead = (uint32_t)N * ((uint32_t)ldb / 2u);

    uint32_t rows = 1u;
    // Pad rows (N > 1; the hook's seven zero rows, llama_mustafar_kernel.py:313): they must be READ to be known zero -- 28 MB per call at c3.
    // Round 6 measured WHERE in the workgroup's life they are read (MUSTAFAR_PAD_LATE; c3, N = 8, us per call, same box): in front of row 0
    // (rounds 1-5, the default) 38.6-40.0; requested in front and looked at behind row 0 (1) 42.6-43.9 -- sixteen more registers, and loads return
    // in order, so the first stream chunk waits for them all the same; read and looked at behind row 0 (2) 39.7-42.0, c5 109 against 103.  The call
    // moves 1.43 x the bytes of the N = 1 call and takes 1.49 x its time: it is the bytes, not their place (profiles/r06_probes.txt item 5).
    constexpr int kPadRegs = 4;   // 16-byte pieces per thread kept in flight: G x 7 rows x 256 tokens over 256 threads = 3.5
    uint4 padv[kPadRegs];
    const int pad_cols = (tb_end - tb0) * 64, pad_per_row = pad_cols / 8;
    const int npad = N > 1 ? G * (N - 1) * pad_per_row : 0;
    const bool pad_regs = N > 1 && npad <= kPadRegs * NW * 64 && MUSTAFAR_PAD_LATE == 1;
    if (pad_regs) {
#pragma unroll
        for (int i = 0; i < kPadRegs; i++) {
            const int u = min((int)threadIdx.x + i * NW * 64, npad - 1);   // (no divergent branch in front of the rows' asm statements: a lane beyond the slice repeats its last piece)
            const int hn = u / pad_per_row, k = u % pad_per_row;
            const int h = hn / (N - 1), n = 1 + hn % (N - 1);
            padv[i] = *reinterpret_cast<const uint4*>(p + ((int64_t)(bh0 + h) * N + n) * ldb + tb0 * 64 + k * 8);
        }
        __builtin_amdgcn_sched_barrier(0);   // (the requests stay in front of row 0: without it the compiler sinks them to their use behind it)
    } else if (N > 1 && MUSTAFAR_PAD_LATE != 2) {
        rows |= pad_row_mask<G>(p, ldb, bh0, N, tb0 * 64, pad_cols, reinterpret_cast<uint32_t*>(smem));
        if (!direct && threadIdx.x == 0) flags[blockIdx.x * gridDim.y + by] = rows;   // (no window rows when N > 1)
    }

    for (int n = 0; n < N; n++) {
        if (n == 1 && MUSTAFAR_PAD_LATE == 2) {   // (2: the slice is read AND looked at behind row 0 -- nothing in front of the workgroup's first stream request)
            rows |= pad_row_mask<G>(p, ldb, bh0, N, tb0 * 64, pad_cols, reinterpret_cast<uint32_t*>(smem));
            if (!direct && threadIdx.x == 0) flags[blockIdx.x * gridDim.y + by] = rows;
        }
        if (n == 1 && pad_regs) {   // (workgroup-uniform) row 0 is done: which pad rows hold a non-zero in this workgroup's columns?
            uint32_t* sh_mask = reinterpret_cast<uint32_t*>(smem);   // (behind the barrier that ended row 0: the stage windows are free)
            if (threadIdx.x == 0) *sh_mask = 0u;
            __syncthreads();
            uint32_t mine = 0;
#pragma unroll
            for (int i = 0; i < kPadRegs; i++) {
                const int u = threadIdx.x + i * NW * 64;
                if (u < npad && nzbits(padv[i])) mine |= 1u << (1 + (u / pad_per_row) % (N - 1));
            }
            if (mine) atomicOr(sh_mask, mine);
            __syncthreads();
            rows |= (uint32_t)__builtin_amdgcn_readfirstlane((int)*sh_mask);   // (wave-uniform, and known to the compiler as such: the rows' asm statements own EXEC)
            __syncthreads();
            if (!direct && threadIdx.x == 0) flags[blockIdx.x * gridDim.y + by] = rows;
        }
        const bool live = (rows >> n) & 1u;
        if (!live && !direct) continue;   // the combine pass skips this row of this slab
        float acc0[G], acc1[G];
#pragma unroll
        for (int h = 0; h < G; h++) acc0[h] = acc1[h] = 0.f;
        if (live) {
            const h16x2* pw = reinterpret_cast<const h16x2*>(p + ((int64_t)bh0 * N + n) * ldb);
            unsigned char* ptab = smem + NW * kStageBytes + wave * (4 * kValTabStride);
            const int tb_first = tb0 + wave / SPLIT;
            if constexpr (SPLIT == 1) {
                value_tokblks<G, MF, 0, 4, kStride>(smem, wave * kStageBytes, bmp_h, idx_h, nz_h, pw, chead, tb_first, tb_end, lane, acc0, acc1, ptab);
            } else {
                if (wave % SPLIT == 0) value_tokblks<G, MF, 0, 2, kStride>(smem, wave * kStageBytes, bmp_h, idx_h, nz_h, pw, chead, tb_first, tb_end, lane, acc0, acc1, ptab);
                else                   value_tokblks<G, MF, 2, 2, kStride>(smem, wave * kStageBytes, bmp_h, idx_h, nz_h, pw, chead, tb_first, tb_end, lane, acc0, acc1, ptab);
            }
        }
        __syncthreads();   // every wave is done with its stage window (and with the previous row's sums)
#pragma unroll
        for (int h = 0; h < G; h++) {
            red[(wave * 2 * G + h) * 64 + lane]     = acc0[h];
            red[(wave * 2 * G + G + h) * 64 + lane] = acc1[h];
        }
        __syncthreads();
        for (int o = threadIdx.x; o < 2 * G * 64; o += NW * 64) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w++) s += red[w * 2 * G * 64 + o];
            const int hh = o >> 6, l = o & 63;   // hh = half * G + h
            const int64_t row = (int64_t)(bh0 + hh % G) * N + n;
            if (direct) out[row * kD + (hh / G) * 64 + l] = (h16)s;
            else        ws_slab[row * kD + (hh / G) * 64 + l] = s;
        }
        __syncthreads();
    }
    MUSTAFAR_TRACE_END();
}

// out[bh, n, c] = fp16( sum_s ws[s, bh, n, c] ), pad rows only from the slabs whose row mask has them.
// (the role of the reference's SplitK_Reduction, Reduction_Kernel.cuh:26-48, with fp32 partials)
// One workgroup per head row bh (round 6; rounds 1-5: one per output row (bh, n) -- 8 x the workgroups for the hook's padded calls, seven of
// eight of them reading S flag words to learn they have nothing to add: 7.8 us against 4.6 for N = 1): thread = (channel, slab parity),
// independent loads, LDS fold of the halves; row 0 first, then -- N > 1 -- the OR of the head group's S row masks decides in one step whether
// any pad row holds anything (the hook's never do: 7 x 128 zeros are written), and only then the rows are folded one by one.
__global__ __launch_bounds__(256) void value_combine_kernel(const float* __restrict__ ws,
                                                            const uint32_t* __restrict__ flags, h16* __restrict__ out,
                                                            int BH, int N, int S, int groups, int G)
{
    __shared__ float part[kD];
    __shared__ uint32_t any_rows;
    const int bh = blockIdx.x;
    const int c = threadIdx.x & (kD - 1), par = threadIdx.x >> 7;
    const int hb_per_kv = groups / G;
    const int y  = (bh / groups) * hb_per_kv + (bh % groups) / G;   // blockIdx.y of the producer
    const int gy = (BH / groups) * hb_per_kv;
    const int64_t total = (int64_t)BH * N * kD;
    uint32_t mine = 0;
    if (N > 1) {
        if (threadIdx.x == 0) any_rows = 0u;
#ifndef MUSTAFAR_PROBE_NOPADWG   // (that timing probe leaves the masks unwritten: take them as "row 0 only")
        for (int k = threadIdx.x; k < S; k += 256) mine |= flags[k * gy + y];   // (requested in front of row 0's loads)
#endif
    }
    {
        const float* src = ws + (int64_t)bh * N * kD + c;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = par;
        for (; k + 6 < S; k += 8) {
            s0 += src[(int64_t)k * total];
            s1 += src[(int64_t)(k + 2) * total];
            s2 += src[(int64_t)(k + 4) * total];
            s3 += src[(int64_t)(k + 6) * total];
        }
        for (; k < S; k += 2) s0 += src[(int64_t)k * total];
        const float s = (s0 + s1) + (s2 + s3);
        if (par) part[c] = s;
        __syncthreads();
        if (!par) out[(int64_t)bh * N * kD + c] = (h16)(s + part[c]);
    }
    if (N == 1) return;
    if (mine & ~1u) atomicOr(&any_rows, mine);
    __syncthreads();
    const uint32_t live = any_rows;
    if (!(live & ~1u)) {   // no slab holds a pad row: zeros
        for (int o = threadIdx.x; o < (N - 1) * kD; o += 256) out[((int64_t)bh * N + 1) * kD + o] = (h16)0.f;
        return;
    }
    for (int n = 1; n < N; n++) {
        float s = 0.f;
        if ((live >> n) & 1u) {
            const float* src = ws + ((int64_t)bh * N + n) * kD + c;
            for (int k = par; k < S; k += 2)
                if ((flags[k * gy + y] >> n) & 1u) s += src[(int64_t)k * total];
        }
        __syncthreads();   // (part: the row before is folded)
        if (par) part[c] = s;
        __syncthreads();
        if (!par) out[((int64_t)bh * N + n) * kD + c] = (h16)(s + part[c]);
    }
}

// ------------------------------------------------------------------------------------------------ fused decode glue
// Replaces the PyTorch glue between the two SpMVs in the reference hook (llama_mustafar_kernel.py:270-317):
// window append (:270), q.K_window^T (:278), concat (:279), / sqrt(d) (:284), fp32 softmax -> fp16 (:304),
// p.V_window (:316) and the final sum (:317).  Both kernels run one workgroup per (batch, q-head) row and are
// latency-bound, so they are written for memory-level parallelism: 16-byte loads, everything a thread needs in
// flight at once, the score row kept in registers between the softmax sweeps.
constexpr int kGlueThreads = 512;
constexpr int kGlueWaves   = kGlueThreads / 64;
constexpr int kMaxRowVecs  = 8;     // 16-byte vectors of the score row a thread keeps in registers (T <= 32768)
constexpr int kMaxWindow   = 1024;  // window tokens whose scores are staged in LDS

template <int NW>
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* sh)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float u = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, u) : v + u;
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[wave] = v;
    __syncthreads();
    float r = sh[0];
#pragma unroll
    for (int w = 1; w < NW; w++) r = is_max ? fmaxf(r, sh[w]) : r + sh[w];
    return r;
}

// x = fp16(score / sqrt(d)) like the reference (an fp16 tensor divided by a Python float, :284), then fp32 (:304).
// The quotient is formed as a product with 1/sqrt(d): at most 1 fp32 ulp away before the fp16 rounding.
__device__ __forceinline__ float scaled(h16 a, float inv_sqrt_d) { return (float)(h16)((float)a * inv_sqrt_d); }

// Additive attention mask of the reference hook (llama_mustafar_kernel.py:293-301): `attn_weights + attention_mask` is
// an fp16 addition (both operands fp16; the fp32 sum of two fp16 values followed by one rounding is that addition), then
// `torch.max(., finfo(fp16).min)` lifts an overflow to -inf back to -65504.  `m` is the mask value of the column.
__device__ __forceinline__ float masked(float x, h16 m)
{
    return fmaxf((float)(h16)(x + (float)m), -65504.f);
}
// fp16 [rows, >= kv_len] mask, `stride` halfs between rows; score row bh uses mask row bh / heads (heads = q heads per
// batch entry: the reference mask is [bsz, 1, 1, kv_len], broadcast over heads).  ptr == nullptr: no mask.
struct MaskArg {
    const h16* ptr;
    int64_t stride;
    int heads;
};
struct __attribute__((packed, aligned(2))) Half8U {   // 8 halfs at 2-byte alignment (kv_len is arbitrary): one global_load_dwordx4
    h16 h[8];
};

template <bool MASK>
__global__ __launch_bounds__(kGlueThreads) void window_softmax_kernel(
    const h16* __restrict__ q, h16* __restrict__ k_win, const h16* __restrict__ k_new, h16* __restrict__ scores,
    int T, int ld, int w_len, int w_cap, int groups, float inv_sqrt_d, const int* __restrict__ w_extra, MaskArg mask)
{
    __shared__ float sh[kGlueWaves];
    __shared__ h16 wsc[kMaxWindow];
    if (w_extra) w_len = min(w_len + *w_extra, w_cap);   // device-side step counter (graph replay): tokens appended so far
    const int bh = blockIdx.x, kvh = bh / groups;
    const int tid = threadIdx.x;
    h16* row = scores + (int64_t)bh * ld;

    // (1) score row of the compressed part -> registers (issued first: the longest latency)
    const int nvec = T / 8;
    Vec8 x[kMaxRowVecs];
#pragma unroll
    for (int i = 0; i < kMaxRowVecs; i++) {
        const int v = tid + i * kGlueThreads;
        x[i].u = (v < nvec) ? reinterpret_cast<const uint4*>(row)[v] : make_uint4(0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u);  // -inf
    }

    // (2) window scores: already in the row (computed by the window workgroups of the key SpMV launch) ...
    if (k_win == nullptr) {
        for (int w = tid; w < w_len; w += kGlueThreads) wsc[w] = row[T + w];
        w_len = min(w_len, kMaxWindow);
    }
    // ... or computed here: 16 lanes per token (8 channels each), 32 tokens per sweep; fp32 accumulate, fp16 result (:278)
    const int sub = tid & 15, grp = tid >> 4;
    Vec8 qv;
    qv.u = reinterpret_cast<const uint4*>(q + (int64_t)bh * kD)[sub];
    const h16* knew = k_new ? k_new + (int64_t)kvh * kD : nullptr;
    if (k_win && knew && bh % groups == 0 && tid < 16)   // the group's first head stores the new key row (:270)
        reinterpret_cast<uint4*>(k_win + ((int64_t)kvh * w_cap + (w_len - 1)) * kD)[tid] = reinterpret_cast<const uint4*>(knew)[tid];
    // 4 sweeps (128 tokens) per iteration with all four row loads issued before any is used: the loop is a chain of
    // L2 round trips otherwise
    constexpr int kSweep = kGlueThreads / 16;
    for (int w0 = 0; k_win != nullptr && w0 < w_len; w0 += 4 * kSweep) {
        Vec8 kv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int w = w0 + u * kSweep + grp;
            const h16* kr = (knew && w == w_len - 1) ? knew : k_win + ((int64_t)kvh * w_cap + min(w, w_len - 1)) * kD;
            kv[u].u = reinterpret_cast<const uint4*>(kr)[sub];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int w = w0 + u * kSweep + grp;
            float sdot = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) sdot = __builtin_fmaf((float)qv.h[j], (float)kv[u].h[j], sdot);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sdot += __shfl_xor(sdot, o);
            if (sub == 0 && w < w_len && w < kMaxWindow) wsc[w] = (h16)sdot;
        }
    }
    __syncthreads();

    // (3) fp32 softmax over x_i = fp16(score_i / sqrt(d)) (+ mask)  (:284, :293-304).  xs() is the softmax input of
    // element j of register slot i; with a mask the row registers are rewritten once with the masked values
    const h16* mrow = MASK ? mask.ptr + (int64_t)(bh / mask.heads) * mask.stride : nullptr;
    if constexpr (MASK) {
#pragma unroll
        for (int i = 0; i < kMaxRowVecs; i++) {
            const int v = tid + i * kGlueThreads;
            if (v < nvec) {
                const Half8U mv = *reinterpret_cast<const Half8U*>(mrow + 8 * (int64_t)v);
#pragma unroll
                for (int j = 0; j < 8; j++) x[i].h[j] = (h16)masked(scaled(x[i].h[j], inv_sqrt_d), mv.h[j]);
            }
        }
        for (int w = tid; w < w_len; w += kGlueThreads) wsc[w] = (h16)masked(scaled(wsc[w], inv_sqrt_d), mrow[T + w]);
    }
    auto xs = [&](h16 a) -> float { return MASK ? (float)a : scaled(a, inv_sqrt_d); };
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < kMaxRowVecs; i++)
        if (i * kGlueThreads < nvec) {   // workgroup-uniform: skip register slots beyond the row
#pragma unroll
            for (int j = 0; j < 8; j++) m = fmaxf(m, xs(x[i].h[j]));
        }
    for (int w = tid; w < w_len; w += kGlueThreads) m = fmaxf(m, xs(wsc[w]));
    m = block_reduce<kGlueWaves>(m, true, sh);
    float l = 0.f;
    float e[kMaxRowVecs][8];
#pragma unroll
    for (int i = 0; i < kMaxRowVecs; i++)
        if (i * kGlueThreads < nvec) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                e[i][j] = __expf(xs(x[i].h[j]) - m);   // exp(-inf) = 0 for the padding lanes
                l += e[i][j];
            }
        }
    for (int w = tid; w < w_len; w += kGlueThreads) l += __expf(xs(wsc[w]) - m);
    l = block_reduce<kGlueWaves>(l, false, sh);
    const float inv = 1.f / l;
#pragma unroll
    for (int i = 0; i < kMaxRowVecs; i++) {
        const int v = tid + i * kGlueThreads;
        if (v < nvec) {
            Vec8 o;
#pragma unroll
            for (int j = 0; j < 8; j++) o.h[j] = (h16)(e[i][j] * inv);
            reinterpret_cast<uint4*>(row)[v] = o.u;
        }
    }
    for (int w = tid; w < w_len; w += kGlueThreads) row[T + w] = (h16)(__expf(xs(wsc[w]) - m) * inv);
}

// Rows longer than the register form holds (T > 32768): the same softmax as three passes over the row in global memory
// (max, denominator, write); the window scores must already be in the row (T > 0: the key launch's window workgroups).
template <bool MASK>
__global__ __launch_bounds__(kGlueThreads) void long_softmax_kernel(h16* __restrict__ scores, int T, int ld, int w_len, int w_cap,
                                                                    float inv_sqrt_d, const int* __restrict__ w_extra, MaskArg mask)
{
    __shared__ float sh[kGlueWaves];
    if (w_extra) w_len = min(w_len + *w_extra, w_cap);
    const int tid = threadIdx.x;
    h16* row = scores + (int64_t)blockIdx.x * ld;
    const h16* mrow = MASK ? mask.ptr + (int64_t)((int)blockIdx.x / mask.heads) * mask.stride : nullptr;
    const int nvec = T / 8, n = T + w_len;
    // softmax input of column c holding raw score a (:284, :293-301)
    auto xs = [&](h16 a, h16 mk) -> float { return MASK ? masked(scaled(a, inv_sqrt_d), mk) : scaled(a, inv_sqrt_d); };
    auto mvec = [&](int v) -> Half8U { return MASK ? *reinterpret_cast<const Half8U*>(mrow + 8 * (int64_t)v) : Half8U{}; };
    float m = -INFINITY;
    for (int v = tid; v < nvec; v += kGlueThreads) {
        Vec8 x;
        x.u = reinterpret_cast<const uint4*>(row)[v];
        const Half8U mv = mvec(v);
#pragma unroll
        for (int j = 0; j < 8; j++) m = fmaxf(m, xs(x.h[j], mv.h[j]));
    }
    for (int i = T + tid; i < n; i += kGlueThreads) m = fmaxf(m, xs(row[i], MASK ? mrow[i] : (h16)0));
    m = block_reduce<kGlueWaves>(m, true, sh);
    float l = 0.f;
    for (int v = tid; v < nvec; v += kGlueThreads) {
        Vec8 x;
        x.u = reinterpret_cast<const uint4*>(row)[v];
        const Half8U mv = mvec(v);
#pragma unroll
        for (int j = 0; j < 8; j++) l += __expf(xs(x.h[j], mv.h[j]) - m);
    }
    for (int i = T + tid; i < n; i += kGlueThreads) l += __expf(xs(row[i], MASK ? mrow[i] : (h16)0) - m);
    l = block_reduce<kGlueWaves>(l, false, sh);
    const float inv = 1.f / l;
    for (int v = tid; v < nvec; v += kGlueThreads) {
        Vec8 x, o;
        x.u = reinterpret_cast<const uint4*>(row)[v];
        const Half8U mv = mvec(v);
#pragma unroll
        for (int j = 0; j < 8; j++) o.h[j] = (h16)(__expf(xs(x.h[j], mv.h[j]) - m) * inv);
        reinterpret_cast<uint4*>(row)[v] = o.u;
    }
    for (int i = T + tid; i < n; i += kGlueThreads) row[i] = (h16)(__expf(xs(row[i], MASK ? mrow[i] : (h16)0) - m) * inv);
}

__global__ void counter_add_kernel(int* ctr, int delta) { if (threadIdx.x == 0 && blockIdx.x == 0) *ctr += delta; }

// out[bh, c] = fp16( sum_s ws[s, bh, c] + sum_w p[bh, T + w] * V_window[kvh, w, c] )   (:315-317), and window append (:309).
// 256 threads: 16 lanes per window row (8 channels each) x 16 rows per sweep; slab sums on (channel, parity).
__global__ __launch_bounds__(256) void value_finish_kernel(
    const float* __restrict__ ws, int S, const h16* __restrict__ probs, int ld, int T, h16* __restrict__ v_win,
    const h16* __restrict__ v_new, int w_len, int w_cap, h16* __restrict__ out, int BH, int groups,
    const int* __restrict__ w_extra)
{
    __shared__ float red[16][kD + 4];
    __shared__ float part[kD];
    if (w_extra) w_len = min(w_len + *w_extra, w_cap);
    const int bh = blockIdx.x, kvh = bh / groups;
    const int tid = threadIdx.x;
    // window p.V
    const int sub = tid & 15, grp = tid >> 4;
    const h16* vnew = v_new ? v_new + (int64_t)kvh * kD : nullptr;
    if (v_win && vnew && bh % groups == 0 && tid < 16)
        reinterpret_cast<uint4*>(v_win + ((int64_t)kvh * w_cap + (w_len - 1)) * kD)[tid] = reinterpret_cast<const uint4*>(vnew)[tid];
    if (v_win == nullptr) w_len = 0;   // the window p.V arrives as extra slabs (window workgroups of the value SpMV launch)
    const h16* pr = probs + (int64_t)bh * ld + T;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0.f;
#pragma unroll 4
    for (int w = grp; w < w_len; w += 16) {
        const h16* vr = (vnew && w == w_len - 1) ? vnew : v_win + ((int64_t)kvh * w_cap + w) * kD;
        Vec8 vv;
        vv.u = reinterpret_cast<const uint4*>(vr)[sub];
        const float pw = (float)pr[w];
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = __builtin_fmaf(pw, (float)vv.h[j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) red[grp][sub * 8 + j] = acc[j];
    // slab sums: thread = (channel, slab parity), independent loads
    const int c = tid & (kD - 1), par = tid >> 7;
    const int64_t total = (int64_t)BH * kD;
    const float* src = ws + (int64_t)bh * kD + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = par;
    for (; k + 6 < S; k += 8) {
        s0 += src[(int64_t)k * total];
        s1 += src[(int64_t)(k + 2) * total];
        s2 += src[(int64_t)(k + 4) * total];
        s3 += src[(int64_t)(k + 6) * total];
    }
    for (; k < S; k += 2) s0 += src[(int64_t)k * total];
    float s = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (par) {
        part[c] = s;
    } else {
#pragma unroll
        for (int g = 0; g < 16; g++) s += red[g][c];
    }
    __syncthreads();
    if (!par) out[(int64_t)bh * kD + c] = (h16)(s + part[c]);
}


// ------------------------------------------------------------------------------------------------ one-pass decode
// The whole compressed part of a layer's decode attention in ONE launch (+ a row kernel that merges the partial results):
// a wave takes a 64-token block of one kv-head group and runs, back to back,
//   key phase     scores of its 64 tokens for the G heads (key_tokblk, lane = token)
//   softmax step  x = fp16(fp16(score) / sqrt(d)) (+ mask) as the hook does (llama_mustafar_kernel.py:284-301); running
//                 maximum of the wave, e = exp(x - max) rounded to fp16, running sum; earlier partial outputs rescaled
//   value phase   partial output += sum_t e_t * V_t over the same 64 tokens (value_tokblks, lane = channel); the e row
//                 segment travels to the coefficient loads through the score scratch (128 bytes per head and block,
//                 written and read back by the SAME wave: a row stride of 64 bytes keeps every segment on scalar-cache
//                 lines of its own, and the wave waits for its stores before the scalar loads are issued)
// and the workgroup leaves one slab (max, sum, unnormalised output) per head; the dense window is a few more workgroups
// of the same launch with slabs of the same form; onepass_finish_kernel merges the slabs of a row (flash-decoding).
// Against the two-launch form (key SpMV -> softmax rows -> value SpMV -> sum): no softmax launch, one launch boundary
// and one ramp / tail less, and the value phase of a block starts behind its own key phase instead of behind ALL key
// blocks.  Numerics: the probabilities are normalised in fp32 at the very end instead of being rounded to fp16 after
// normalisation (:304); e carries the same 11 bits as the hook's fp16 probabilities.
// Wave-wide maximum / sum, the same value in every lane, by DPP: two quad permutes, two rotations inside the rows of 16,
// then the last lane of a row handed to the next row(s) (row_bcast:15 / :31) -- six VALU instructions whose operand comes
// through the data-parallel path, and one v_readlane of lane 63.  (__shfl_xor compiles to ds_bpermute_b32 here: six LDS round
// trips in a dependent chain per reduction, eight reductions per 64-token block in the softmax step.)
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float old, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_max(float v)
{
    // (written out: the compiler folds the DPP operand into v_add_f32 but emits v_mov_b32_dpp + v_max_f32 for the maximum)
    asm("s_nop 1\n\t"   // (a DPP operand written by the previous VALU instruction needs two wait states)
        "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"       // every lane of a row holds the row's maximum
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"    // into rows 1 and 3
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"          // into rows 2 and 3: lane 63 holds the wave's
        : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// A wave-uniform float kept in a scalar register (a running maximum, a rescale factor): VGPRs are what limits the
// one-pass kernel's occupancy, and a spilled SGPR costs a lane of one shared VGPR.
__device__ __forceinline__ float uniform_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_f<0xB1>(0.f, v);
    v += dpp_f<0x4E>(0.f, v);
    v += dpp_f<0x124>(0.f, v);
    v += dpp_f<0x128>(0.f, v);
    v += dpp_f<0x142, 0xa>(0.f, v);
    v += dpp_f<0x143, 0xc>(0.f, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

constexpr int kOneWinChunk = 64;   // window tokens per window workgroup of the one-pass launch (one slab each)

struct OneArgs {   // operands of the one-pass launch beyond the two caches (by value: one kernarg block)
    const h16* q;          // [BH, 128]
    h16* e_rows;           // [BH, ld] scratch: exp(x - running max) of the compressed part, fp16
    float* ws_o;           // [slabs, BH, 128] unnormalised partial outputs
    float* ws_ml;          // [slabs, BH, 2]   (max, sum) of every slab
    h16* k_win;            // [B', w_cap, 128] dense windows, appended in place
    h16* v_win;
    const h16* k_new;      // [B', 128] newest rows (or nullptr: already stored)
    const h16* v_new;
    const int* w_extra;    // device step counter added to w_len (graph replay), or nullptr
    MaskArg mask;
    int T, groups, BH, tb_per_wg, ld, w_len, w_cap, nchunks, win_rows;
    float inv_sqrt_d;
    int pair_slabs = 0;    // pair form: every PAIR of waves leaves a slab of its own (2 per workgroup) instead of merging through LDS first
    // appended extents (pair form): blocks [0, nb0) live in the views the launch carries, block nb0 + 4 i + j in entry i of these
    // DEVICE tables (a 256-token cache of its own each: mustafar_decode_attention_extents); nullptr: one extent
    const mustafar_cache_view* k_ext = nullptr;
    const mustafar_cache_view* v_ext = nullptr;
    int nb0 = 0;
    // tokens in use as a DEVICE quantity (extents only): T above is then the capacity the launch was sized for -- grid, slabs, score
    // scratch, mask columns -- and a captured graph of the launch stays valid while the cache grows up to it
    const int* t_dev = nullptr;
    // round 5 (super-block pair form): workgroups with a linear id >= this one issue at a raised priority for their whole life -- the few
    // workgroups of a launch's LAST, nearly empty round (a cache a trigger or two past one resident round of workgroups: T = 8448 at c3's
    // geometry is 2112 workgroups on 2048 slots) start when the first slots free up and would otherwise crawl along at an eighth of a SIMD
    int hi_prio_from = 0x7fffffff;
    // round 6 (experiment, mustafar_tune(12, bytes)): expected bytes of key stream per 64-token block; > 0: a wave asks for the lines around
    // the PREDICTED position of its first key chunk next to its bounds load (see decode_onepass_sb_kernel)
    int spec_k_bytes = 0;
};

// Window workgroup: 64 window tokens of one head batch -> scores, softmax partial, p.V partial -> slab (S + chunk).
// (the OneArgs fields arrive through a by-value copy of the few that are needed: a reference to the kernel argument pins the
// whole struct to a stack slot, and a kernel with a private segment -- even one it never touches -- is launched with scratch)
struct WinOneArgs {
    const h16* q; float* ws_o; float* ws_ml; h16* k_win; h16* v_win; const h16* k_new; const h16* v_new; const h16* mask_ptr;
    int64_t mask_stride;
    int mask_heads, T, groups, BH, w_len, w_cap, nchunks;
    float inv_sqrt_d;
};
__device__ __forceinline__ WinOneArgs win_args(const OneArgs& a, int T_used = -1)   // T_used: the tokens in use when a.T is a capacity
{
    return WinOneArgs{a.q, a.ws_o, a.ws_ml, a.k_win, a.v_win, a.k_new, a.v_new, a.mask.ptr, a.mask.stride, a.mask.heads, T_used >= 0 ? T_used : a.T, a.groups, a.BH,
                      window_len(a.w_extra, a.w_len, a.w_cap), a.w_cap, a.nchunks, a.inv_sqrt_d};
}
template <int G>
__device__ __forceinline__ void onepass_window_wg(unsigned char* smem, const WinOneArgs a, int task, int S)
{
    constexpr int kRedLd = kD + 4;
    float* xs = reinterpret_cast<float*>(smem);                    // [G][64] scores, then e
    float* ml = xs + G * 64;                                        // [G][2]
    float* red = ml + 2 * G;                                        // [16][kRedLd] fold buffer
    h16* qs = reinterpret_cast<h16*>(red + 16 * kRedLd);            // [G][128]
    static_assert((G * 64 + 2 * G + 16 * kRedLd) * 4 + G * kD * 2 <= kWaves * kStageBytes, "window scratch must fit in the stage area");
    const int hb_per_kv = a.groups / G;
    const int hb = task / a.nchunks, chunk = task % a.nchunks;
    const int kvh = hb / hb_per_kv, bh0 = kvh * a.groups + (hb % hb_per_kv) * G;
    const int w_len = a.w_len;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w0 = chunk * kOneWinChunk;
    float* slab_o = a.ws_o + ((int64_t)(S + chunk) * a.BH + bh0) * kD;
    float* slab_ml = a.ws_ml + ((int64_t)(S + chunk) * a.BH + bh0) * 2;
    if (w0 >= w_len) {   // empty chunk (beyond the current window): a slab of weight zero
        for (int o = tid; o < G * kD; o += kThreads) slab_o[o] = 0.f;
        if (tid < G) { slab_ml[2 * tid] = -INFINITY; slab_ml[2 * tid + 1] = 0.f; }
        return;
    }
    const bool first_hb = hb % hb_per_kv == 0;
    // ---- scores: 4 threads per token (32 channels each), q rows of the G heads in LDS (llama_mustafar_kernel.py:270, :278)
    if (tid < G * 16) reinterpret_cast<uint4*>(qs)[tid] = reinterpret_cast<const uint4*>(a.q + (int64_t)bh0 * kD)[tid];
    const int row = tid >> 2, part = tid & 3;
    const int w = w0 + row;
    const bool valid = w < w_len;
    const int wr = valid ? w : w_len - 1;
    const h16* kfresh = a.k_new ? a.k_new + (int64_t)kvh * kD : nullptr;
    h16* kwin = a.k_win + (int64_t)kvh * a.w_cap * kD;
    const h16* kr = ((kfresh && wr == w_len - 1) ? kfresh : kwin + (int64_t)wr * kD) + part * 32;
    Vec8 kv[4];
#pragma unroll
    for (int c = 0; c < 4; c++) kv[c].u = reinterpret_cast<const uint4*>(kr)[c];
    if (kfresh && valid && w == w_len - 1 && first_hb) {   // store the new key row (:270)
#pragma unroll
        for (int c = 0; c < 4; c++) reinterpret_cast<uint4*>(kwin + (int64_t)w * kD + part * 32)[c] = kv[c].u;
    }
    __syncthreads();
    const h16* mrow = a.mask_ptr ? a.mask_ptr + (int64_t)(bh0 / a.mask_heads) * a.mask_stride + a.T : nullptr;
#pragma unroll
    for (int h = 0; h < G; h++) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            Vec8 qv;
            qv.u = reinterpret_cast<const uint4*>(qs + h * kD + part * 32)[c];
#pragma unroll
            for (int j = 0; j < 8; j++) acc = __builtin_fmaf((float)kv[c].h[j], (float)qv.h[j], acc);
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (part == 0) {
            float x = -INFINITY;
            if (valid) {
                x = scaled((h16)acc, a.inv_sqrt_d);          // fp16 score (:278), / sqrt(d) in fp16 (:284)
                if (mrow) x = masked(x, mrow[w]);
            }
            xs[h * 64 + row] = x;
        }
    }
    __syncthreads();
    if (wave < G) {   // softmax partial of head `wave` over the chunk's tokens (lane = token)
        const float x = xs[wave * 64 + lane];
        const float m = wave_max(x);
        const float e = (float)(h16)__expf(x - m);           // (exp(-inf - m) = 0 for the lanes beyond the window)
        xs[wave * 64 + lane] = e;
        const float l = wave_sum(e);
        if (lane == 0) { ml[2 * wave] = m; ml[2 * wave + 1] = l; }
    }
    __syncthreads();
    // ---- p.V over the chunk's tokens (:309, :316): 16 lanes x 16 bytes per row, 16 rows per sweep
    const int sub = tid & 15, grp = tid >> 4;
    const h16* vfresh = a.v_new ? a.v_new + (int64_t)kvh * kD : nullptr;
    h16* vwin = a.v_win + (int64_t)kvh * a.w_cap * kD;
    const int w1 = min(w0 + kOneWinChunk, w_len);
    Vec8 vv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int ww = w0 + grp + u * 16;
        const int wr2 = ww < w1 ? ww : w0;
        const h16* vr = (vfresh && wr2 == w_len - 1) ? vfresh : vwin + (int64_t)wr2 * kD;
        vv[u].u = reinterpret_cast<const uint4*>(vr)[sub];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int ww = w0 + grp + u * 16;
        if (ww < w1 && vfresh && ww == w_len - 1 && first_hb)   // store the new value row (:309)
            reinterpret_cast<uint4*>(vwin + (int64_t)ww * kD)[sub] = vv[u].u;
    }
    // one head at a time (8 accumulators live, not 8 G: this path must not set the register allocation of the kernel)
#pragma unroll 1
    for (int h = 0; h < G; h++) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = 0.f;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int ww = w0 + grp + u * 16;
            if (ww < w1) {
                const float pw = xs[h * 64 + (ww - w0)];
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j] = __builtin_fmaf(pw, (float)vv[u].h[j], acc[j]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; j++) red[grp * kRedLd + sub * 8 + j] = acc[j];
        __syncthreads();
        if (tid < kD) {
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < 16; g++) sum += red[g * kRedLd + tid];
            slab_o[h * kD + tid] = sum;
        }
    }
    if (tid < 2 * G) slab_ml[tid] = ml[tid];
}

// grid: x = token chunks (tb_per_wg blocks each), y = kv-heads * (groups / G) (+ win_rows leading rows of window
// workgroups).  4 waves; PAIR = false: wave w takes blocks tb0 + w, tb0 + w + 4, ... whole; PAIR = true: two waves share a
// block -- 64 channels each in the key phase (partial scores folded through LDS), one 64-channel half of the output each in
// the value phase -- and the workgroup walks two blocks at a time with two barriers per step.  The finer grain is what the
// VALU engine wants (its two-launch forms of the same grain: key 24.6 vs 25.8 us, value 29.5 vs 34.9 us at c3); the
// matrix-pipe engine is faster with whole blocks.
template <int G, bool MF, bool PAIR>
__global__ __launch_bounds__(kThreads, MF ? 5 : 1) void decode_onepass_kernel(   // (matrix-pipe form: 97 registers unbounded, one over the 5-wave step)
    const uint64_t* __restrict__ k_bmp, const unsigned char* __restrict__ k_nz, const uint32_t* __restrict__ k_idx,
    const uint32_t* __restrict__ k_nz_off, const uint64_t* __restrict__ v_bmp, const unsigned char* __restrict__ v_nz,
    const uint32_t* __restrict__ v_idx, const uint32_t* __restrict__ v_nz_off, OneArgs a, int64_t k_bmp_stride,
    int64_t k_idx_stride, uint32_t k_nz_stride, int64_t v_bmp_stride, int64_t v_idx_stride, uint32_t v_nz_stride)
{
    constexpr int kTabBytes = (MF && G == 4) ? 4 * kKeyTabStride + kWaves * 4 * kValTabStride : 0;
    __shared__ __attribute__((aligned(16))) unsigned char smem[kWaves * kStageBytes + kTabBytes];
    static_assert(kWaves * kStageBytes >= (kWaves * 2 * 4 * 64 + 2 * kWaves * 4) * 4, "combine buffers must fit in the stage area");
    MUSTAFAR_PTRACE_BEGIN();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = gridDim.x;
    if ((int)blockIdx.y < a.win_rows) {   // dense window
        const int task = blockIdx.y * gridDim.x + blockIdx.x;
        if (task < (int)(gridDim.y - a.win_rows) * a.nchunks) onepass_window_wg<G>(smem, win_args(a), task, S);
        MUSTAFAR_PTRACE_END(5);
        return;
    }
    const int by = blockIdx.y - a.win_rows;
    const int hb_per_kv = a.groups / G;
    const int kvh = by / hb_per_kv;
    const int bh0 = kvh * a.groups + (by % hb_per_kv) * G;
    const int ntb = a.T >> 6;
    const int tb0 = blockIdx.x * a.tb_per_wg;
    const int tb_end = min(ntb, tb0 + a.tb_per_wg);
    const int64_t tiles = (int64_t)ntb * kTilesPerTb;
    const uint64_t* kb = k_bmp + (int64_t)kvh * (k_bmp_stride ? k_bmp_stride : tiles);
    const uint32_t* ki = k_idx + (int64_t)kvh * (k_idx_stride ? k_idx_stride : tiles + 1);
    const unsigned char* kn = k_nz + 16ull * (k_nz_stride ? (uint64_t)kvh * k_nz_stride : (uint64_t)k_nz_off[kvh]);
    const uint64_t* vb = v_bmp + (int64_t)kvh * (v_bmp_stride ? v_bmp_stride : tiles);
    const uint32_t* vi = v_idx + (int64_t)kvh * (v_idx_stride ? v_idx_stride : tiles + 1);
    const unsigned char* vn = v_nz + 16ull * (v_nz_stride ? (uint64_t)kvh * v_nz_stride : (uint64_t)v_nz_off[kvh]);
    const h16x2* qw = reinterpret_cast<const h16x2*>(a.q + (int64_t)bh0 * kD);
    h16* erow = a.e_rows + (int64_t)bh0 * a.ld;
    const h16x2* pw = reinterpret_cast<const h16x2*>(erow);
    const h16* mrow = a.mask.ptr ? a.mask.ptr + (int64_t)(bh0 / a.mask.heads) * a.mask.stride : nullptr;

    uint32_t ctab_lane = 0;
    unsigned char* ptab = nullptr;
    if constexpr (MF && G == 4) {   // key-side coefficient table: the q rows of the 4 heads (as key_spmv_kernel)
        unsigned char* tab = smem + kWaves * kStageBytes;
        if (threadIdx.x < 64)
            *reinterpret_cast<uint4*>(tab + (threadIdx.x >> 4) * kKeyTabStride + (threadIdx.x & 15) * 16) =
                *reinterpret_cast<const uint4*>(a.q + (int64_t)(bh0 + (threadIdx.x >> 4)) * kD + (threadIdx.x & 15) * 8);
        __syncthreads();
        ctab_lane = (uint32_t)reinterpret_cast<uintptr_t>(tab) + (lane & 3) * kKeyTabStride;
        ptab = tab + 4 * kKeyTabStride + wave * (4 * kValTabStride);
    }

    static_assert(!(MF && PAIR), "the matrix-pipe engine runs whole blocks per wave");
    float m_run[G], l_lane[G], acc0[G], acc1[G];   // (m_run, l_lane: wave-uniform, in scalar registers)
#pragma unroll
    for (int h = 0; h < G; h++) { m_run[h] = -INFINITY; l_lane[h] = 0.f; acc0[h] = 0.f; acc1[h] = 0.f; }
    // softmax step of one block on the lanes' scores s[] (lane = token): running maximum, e -> the score scratch, running
    // sum; returns the factors the earlier partial outputs are rescaled by
    auto softmax_step = [&](int tb, float (&s)[G], float (&alpha)[G]) {
        h16 mk = (h16)0.f;
        if (mrow) mk = mrow[tb * 64 + lane];
#pragma unroll
        for (int h = 0; h < G; h++) {
            float x = scaled((h16)s[h], a.inv_sqrt_d);        // fp16 score (SpMM_Kernel.cuh:418), / sqrt(d) in fp16 (model :284)
            if (mrow) x = masked(x, mk);
            const float m_new = uniform_f(fmaxf(m_run[h], wave_max(x)));
            alpha[h] = uniform_f(__expf(m_run[h] - m_new));   // 0 for the first block (m_run = -inf)
            const h16 e = (h16)__expf(x - m_new);
            if constexpr (MF && G == 4) {
                // matrix-pipe engine: e goes straight into the wave's LDS coefficient table [head][token] -- no memory round trip
                *reinterpret_cast<h16*>(ptab + h * kValTabStride + lane * 2) = e;
            } else {
                erow[(int64_t)h * a.ld + tb * 64 + lane] = e;
            }
            l_lane[h] = uniform_f(l_lane[h] * alpha[h] + wave_sum((float)e));   // (the running sum of the wave, uniform)
            m_run[h] = m_new;
        }
        // VALU engine: the value phase reads the e segments back as coefficients through scalar loads -- after the stores
        // have reached L2
        if constexpr (!(MF && G == 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if constexpr (!PAIR) {
        for (int tb = tb0 + wave; tb < tb_end; tb += kWaves) {
            float s[G], alpha[G];
#pragma unroll
            for (int h = 0; h < G; h++) s[h] = 0.f;
            key_tokblk<G, MF, 0, 4>(smem, wave * kStageBytes, kb + (int64_t)tb * kTilesPerTb, ki + (int64_t)tb * kTilesPerTb, kn, qw, kD / 2,
                                    lane, s, ctab_lane);
            MUSTAFAR_PTRACE_STAMP(2);
            softmax_step(tb, s, alpha);
            MUSTAFAR_PTRACE_STAMP(3);
#pragma unroll
            for (int h = 0; h < G; h++) { acc0[h] *= alpha[h]; acc1[h] *= alpha[h]; }
            value_tokblks<G, MF, 0, 4, 1, MF && G == 4>(smem, wave * kStageBytes, vb, vi, vn, pw, (uint32_t)a.ld / 2u, tb, tb + 1, lane, acc0, acc1, ptab);
            MUSTAFAR_PTRACE_STAMP(5);
        }
    } else {
        const int pair = wave >> 1, odd = wave & 1;
        // exchange area of the pair: the ODD wave's stage window (dead between the phases): G x 64 partial scores, then G factors
        float* xch = reinterpret_cast<float*>(smem + (2 * pair + 1) * kStageBytes);
        for (int t = tb0; t < tb_end; t += kWaves / 2) {   // workgroup-uniform: the barriers below are reached by every wave
            const int tb = t + pair;
            const bool active = tb < tb_end;
            float s[G], alpha[G];
#pragma unroll
            for (int h = 0; h < G; h++) { s[h] = 0.f; alpha[h] = 1.f; }
            if (active) {
                const uint64_t* kbt = kb + (int64_t)tb * kTilesPerTb;
                const uint32_t* kit = ki + (int64_t)tb * kTilesPerTb;
                if (odd) key_tokblk<G, MF, 2, 2>(smem, wave * kStageBytes, kbt, kit, kn, qw, kD / 2, lane, s, ctab_lane);
                else     key_tokblk<G, MF, 0, 2>(smem, wave * kStageBytes, kbt, kit, kn, qw, kD / 2, lane, s, ctab_lane);
                if (odd) {
#pragma unroll
                    for (int h = 0; h < G; h++) xch[h * 64 + lane] = s[h];
                }
            }
            MUSTAFAR_PTRACE_STAMP(2);
            __syncthreads();
            if (active && !odd) {
#pragma unroll
                for (int h = 0; h < G; h++) s[h] += xch[h * 64 + lane];
                softmax_step(tb, s, alpha);
                if (lane < G) {
                    float mine = alpha[0];
#pragma unroll
                    for (int h = 1; h < G; h++) mine = (lane == h) ? alpha[h] : mine;
                    xch[G * 64 + lane] = mine;
                }
            }
            __syncthreads();
            MUSTAFAR_PTRACE_STAMP(3);
            if (active) {
                if (odd) {
#pragma unroll
                    for (int h = 0; h < G; h++) alpha[h] = xch[G * 64 + h];
                }
#pragma unroll
                for (int h = 0; h < G; h++) { acc0[h] *= alpha[h]; acc1[h] *= alpha[h]; }
                // (the odd wave's LDS reads above are issued before its value phase rewrites the window: one wave, in order)
                if (odd) value_tokblks<G, MF, 2, 2, 1>(smem, wave * kStageBytes, vb, vi, vn, pw, (uint32_t)a.ld / 2u, tb, tb + 1, lane, acc0, acc1, ptab);
                else     value_tokblks<G, MF, 0, 2, 1>(smem, wave * kStageBytes, vb, vi, vn, pw, (uint32_t)a.ld / 2u, tb, tb + 1, lane, acc0, acc1, ptab);
            }
            MUSTAFAR_PTRACE_STAMP(5);
        }
        if (odd) {   // maximum and sum live in the even wave; the odd wave contributes its output half only
#pragma unroll
            for (int h = 0; h < G; h++) l_lane[h] = 0.f;
        }
    }
    // ---- merge the waves: common maximum, rescaled sums and outputs -> one slab per head.  (PAIR: an odd wave carries
    // the second channel half of its pair's blocks and must be scaled by the PAIR's maximum: it borrows the even wave's.)
    float* red = reinterpret_cast<float*>(smem);                 // [kWaves][2G][64]
    float* s_m = red + kWaves * 2 * G * 64;                      // [kWaves][G]
    float* s_l = s_m + kWaves * G;                               // [kWaves][G]
    __syncthreads();   // every wave is done with its stage window
    if (lane < G && !(PAIR && (wave & 1))) {
        float mine = m_run[0];
#pragma unroll
        for (int h = 1; h < G; h++) mine = (lane == h) ? m_run[h] : mine;
        s_m[wave * G + lane] = mine;
        if (PAIR) s_m[(wave + 1) * G + lane] = mine;
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < G; h++) {
        const float mw = s_m[wave * G + h];                      // (= m_run[h], or the partner's for an odd wave of a pair)
        float M = s_m[h];
#pragma unroll
        for (int w = 1; w < kWaves; w++) M = fmaxf(M, s_m[w * G + h]);
        const float scale = (mw == -INFINITY) ? 0.f : __expf(mw - M);   // a wave without blocks weighs nothing
        const float l = l_lane[h] * scale;
        if (lane == 0) s_l[wave * G + h] = l;
        red[(wave * 2 * G + h) * 64 + lane]     = acc0[h] * scale;
        red[(wave * 2 * G + G + h) * 64 + lane] = acc1[h] * scale;
    }
    __syncthreads();
    float* slab_o = a.ws_o + ((int64_t)blockIdx.x * a.BH + bh0) * kD;
    for (int o = threadIdx.x; o < 2 * G * 64; o += kThreads) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; w++) sum += red[w * 2 * G * 64 + o];
        const int hh = o >> 6, l = o & 63;   // hh = half * G + h
        slab_o[(hh % G) * kD + (hh / G) * 64 + l] = sum;
    }
    if (threadIdx.x < G) {
        const int h = threadIdx.x;
        float M = s_m[h], L = s_l[h];
#pragma unroll
        for (int w = 1; w < kWaves; w++) { M = fmaxf(M, s_m[w * G + h]); L += s_l[w * G + h]; }
        float* slab_ml = a.ws_ml + ((int64_t)blockIdx.x * a.BH + bh0 + h) * 2;
        slab_ml[0] = M;
        slab_ml[1] = L;
    }
    MUSTAFAR_PTRACE_END(MF ? 6 : 4);
}

// out[bh, c] = fp16( sum_s w_s * o_s[c] / sum_s w_s * l_s ),  w_s = exp(m_s - max_s m_s)   (the softmax of :304 and the
// sums of :315-317, merged over the slabs of the row).  One workgroup per row, 256 threads: thread = (channel, parity).
constexpr int kMaxSlabs = 512;
// Every wave works out the row's maximum, the slab weights and the denominator for ITSELF (lane = slab, DPP reductions, its own
// LDS copy of the weights): no barrier before the weighted sum, one behind it to fold the two slab parities.  (Round 2 reduced
// across the workgroup: five barriers in a kernel that is all latency; 5.4 -> 4.x us per layer at c3.)
template <int KPER = kMaxSlabs / 64>   // slabs per lane of the weight pass: 64 * KPER >= S (round 5: c3's 36 slabs run the KPER = 1 text)
__global__ __launch_bounds__(256) void onepass_finish_kernel(const float* __restrict__ ws_o, const float* __restrict__ ws_ml,
                                                             int S, h16* __restrict__ out, int BH)
{
    __shared__ float wgt_all[4][64 * KPER];
    __shared__ float part[kD];
    const int bh = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int c = tid & (kD - 1), par = tid >> 7;
    float* wgt = wgt_all[tid >> 6];
    const int64_t total = (int64_t)BH * kD;
    const float* src = ws_o + (int64_t)bh * kD + c;
    // The first 2 * MUSTAFAR_FINISH_EARLY slabs' outputs are requested BEFORE the weights are known: the loads fly while the maxima / sums are
    // loaded and reduced (the kernel is two dependent memory round trips otherwise; 36-68 slabs at c3).
#ifndef MUSTAFAR_FINISH_EARLY
#define MUSTAFAR_FINISH_EARLY 20   // (round 5: 32 -> 20; c3's rows have 36 slabs: tokens/s + 1-2 % at c3, c4 / c5 unchanged -- same-box A/B, profiles/r05_probes.txt)
#endif
    constexpr int kEarly = MUSTAFAR_FINISH_EARLY;   // per thread: slabs par, par + 2, ..., par + 2 * kEarly - 2 (128 slabs: a slab per pair at c3)
    float v[kEarly];
#pragma unroll
    for (int i = 0; i < kEarly; i++) {
        const int k = par + 2 * i;
        v[i] = (k < S) ? src[(int64_t)k * total] : 0.f;
    }
    constexpr int kPer = KPER;   // slabs lane, lane + 64, ... of the row, in every wave
    float m[kPer], l[kPer];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < kPer; i++) {
        const int k = lane + 64 * i;
        m[i] = -INFINITY;
        l[i] = 0.f;
        if (k < S) {
            const float2 ml = *reinterpret_cast<const float2*>(ws_ml + ((int64_t)k * BH + bh) * 2);
            m[i] = ml.x;
            l[i] = ml.y;
        }
        mx = fmaxf(mx, m[i]);
    }
    const float M = wave_max(mx);
    float dsum = 0.f;
#pragma unroll
    for (int i = 0; i < kPer; i++) {
        const int k = lane + 64 * i;
        const float w = (l[i] > 0.f) ? __expf(m[i] - M) : 0.f;
        if (k < S) wgt[k] = w;
        dsum += w * l[i];
    }
    const float denom = wave_sum(dsum);
    __builtin_amdgcn_wave_barrier();   // (the wave's own LDS writes, read back below: ordered within the wave)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if constexpr (kPer == 1) {
        // one slab per lane: the weight of slab k is lane k's register, and k = par + 2 i is the same in every lane of the wave (par = wave / 2):
        // a v_readlane per slab instead of an LDS round trip (the slabs beyond S were read as zero and weigh zero: lanes >= S hold w = 0)
        const float w0 = (l[0] > 0.f) ? __expf(m[0] - M) : 0.f;
        const int par_u = __builtin_amdgcn_readfirstlane(par);
#pragma unroll
        for (int i = 0; i < kEarly; i += 4) {
            static_assert(2 * kEarly + 1 <= 64, "the early slabs' weights are lanes of one register");
            const int k = par_u + 2 * i;
            s0 += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w0), k)) * v[i];
            s1 += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w0), k + 2)) * v[i + 1];
            s2 += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w0), k + 4)) * v[i + 2];
            s3 += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w0), k + 6)) * v[i + 3];
        }
    } else {
#pragma unroll
        for (int i = 0; i < kEarly; i += 4) {   // (slabs beyond S were read as zero; their weights are never read)
            const int k = par + 2 * i;
            if (k < S)     s0 += wgt[k] * v[i];
            if (k + 2 < S) s1 += wgt[k + 2] * v[i + 1];
            if (k + 4 < S) s2 += wgt[k + 4] * v[i + 2];
            if (k + 6 < S) s3 += wgt[k + 6] * v[i + 3];
        }
    }
    int k = par + 2 * kEarly;
    for (; k + 6 < S; k += 8) {
        s0 += wgt[k] * src[(int64_t)k * total];
        s1 += wgt[k + 2] * src[(int64_t)(k + 2) * total];
        s2 += wgt[k + 4] * src[(int64_t)(k + 4) * total];
        s3 += wgt[k + 6] * src[(int64_t)(k + 6) * total];
    }
    for (; k < S; k += 2) s0 += wgt[k] * src[(int64_t)k * total];
    const float s = (s0 + s1) + (s2 + s3);
    if (par) part[c] = s;
    __syncthreads();
    if (!par) out[(int64_t)bh * kD + c] = (h16)((s + part[c]) / denom);
}

// The row kernel for rows of at most 64 slabs (round 5; every BASELINE shape up to 8k x batch 8): ONE thread per channel walks all the slabs of
// the row -- no parity halves, no LDS, no barrier -- with the first kEarly1 slabs' outputs requested before the weights are known and each
// weight taken from the lane that computed it (v_readlane; a slab per lane).  Two waves per row, each working the weights out for itself.
#ifndef MUSTAFAR_FINISH1_EARLY
#define MUSTAFAR_FINISH1_EARLY 40
#endif
__global__ __launch_bounds__(128) void onepass_finish1_kernel(const float* __restrict__ ws_o, const float* __restrict__ ws_ml, int S,
                                                              h16* __restrict__ out, int BH)
{
    constexpr int kEarly1 = MUSTAFAR_FINISH1_EARLY;
    static_assert(kEarly1 <= 64 && kEarly1 % 4 == 0, "a slab per lane");
    const int bh = blockIdx.x, c = threadIdx.x, lane = threadIdx.x & 63;
    const int64_t total = (int64_t)BH * kD;
    const float* src = ws_o + (int64_t)bh * kD + c;
    float v[kEarly1];
#pragma unroll
    for (int i = 0; i < kEarly1; i++) v[i] = (i < S) ? src[(int64_t)i * total] : 0.f;
    float m = -INFINITY, l = 0.f;
    if (lane < S) {
        const float2 ml = *reinterpret_cast<const float2*>(ws_ml + ((int64_t)lane * BH + bh) * 2);
        m = ml.x;
        l = ml.y;
    }
    const float M = wave_max(m);
    const float w = (l > 0.f) ? __expf(m - M) : 0.f;   // (lanes >= S: zero)
    const float denom = wave_sum(w * l);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    auto wk = [&](int k) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w), k)); };
#pragma unroll
    for (int i = 0; i < kEarly1; i += 4) {
        s0 += wk(i) * v[i];
        s1 += wk(i + 1) * v[i + 1];
        s2 += wk(i + 2) * v[i + 2];
        s3 += wk(i + 3) * v[i + 3];
    }
    for (int k = kEarly1; k < S; k++) s0 += wk(k) * src[(int64_t)k * total];   // (S <= 64; wave-uniform k)
    out[(int64_t)bh * kD + c] = (h16)(((s0 + s1) + (s2 + s3)) / denom);
}

// ------------------------------------------------------------------------------------------------ one-pass decode, lean form (G = 4)
// Same algorithm and slabs as decode_onepass_kernel, rebuilt around what the round-2 counters showed: at c3 a workgroup of the
// pair form runs its block loop ONCE, and 2.3 of its 11.3 vector instructions per tile were the start-up and merge code around
// that single iteration (sixteen hoisted coefficient pointers and a dozen chunk pointers spilled to VGPR lanes, the pair
// exchange, two barriers).  Here
//   * a wave owns whole 64-token blocks (no exchange, no barrier before the final merge);
//   * every address inside a block is ONE base pointer + an immediate: the chunk's bitmaps / offsets (tile offset TOFF), the
//     coefficients of the four heads (rows HS bytes apart: the q rows are contiguous, and the e segments of a block are laid out
//     [head][64 tokens] in the score scratch, which is this kernel's to arrange), so nothing per chunk or per head is
//     computed, hoisted or spilled;
//   * FMA phase (ENG): 0 = four v_fma_mix_f32 per tile under EXEC = bitmap (as chunk32), or
//     2 = v_dot2_f32_f16 on PAIRS of tiles: the gathers run under EXEC = bitmap into zeroed registers (even tile -> low half,
//     odd tile -> high half through ds_read_u16_d16_hi, packed by one full-rate v_or_b32), then ONE v_dot2 per head and pair:
//     per tile 3 + 2 half-rate and 1.5 full-rate vector instructions instead of 3 + 4.  A v_dot2_f32_f16 product with an fp16
//     SUBNORMAL input can lose up to its whole value (tools/ubench/dot2_asm_numerics.hip: the instruction as issued here, under
//     1 % of such products); the softmax weights therefore travel scaled by 2^15 (e in (2^-29, 1] stays normal; the factor
//     cancels between the output and the denominator and is removed exactly when the slab is written), and a non-zero
//     K / V / q element below 2^-14 contributes with an absolute error of at most 6.1e-5 x |coefficient|, far inside the fp16
//     rounding of the scores and outputs themselves.
template <int TOFF>
__device__ __forceinline__ void metab_issue_at(MetaB& m, const uint64_t* __restrict__ bmp, const uint32_t* __restrict__ idx)
{
#ifdef MUSTAFAR_PROBE_HOTMETA
    asm volatile("s_load_dwordx16 %0, %2, %4\n\ts_load_dwordx8 %1, %3, %5"
                 : "=&s"(m.bm), "=&s"(m.ix)
                 : "s"(g_hot_bmp), "s"(g_hot_idx), "i"((TOFF % 32) * 8), "i"((TOFF % 32) * 4));
#else
    asm volatile("s_load_dwordx16 %0, %2, %4\n\ts_load_dwordx8 %1, %3, %5"
                 : "=&s"(m.bm), "=&s"(m.ix)
                 : "s"(bmp), "s"(idx), "i"(TOFF * 8), "i"(TOFF * 4));
#endif
}
// The same with a TOUCH of the metadata two steps ahead (TT = its tile offset, < 0: none): one dword of the bitmap line and one of the
// offset line that the request AFTER this one will want -- they ride in this request's wait (same issue time, same L2 latency) and pull
// the 64-byte lines into the scalar cache, so that from a chunk's second step on the 96-byte requests are scalar-cache hits instead of one
// L2 round trip per step (probe build with every wave on the same hot lines: c3 40.7 -> 36.8 us, c5 128.5 -> 117.5: the bound of this).
// MEASURED SLOWER (round 5, same box, kernel us, without / with: c3 42.0 / 43.2, c4 67.6 / 69.8, c5 128.4 / 133.4, c2 15.4 / 15.4): the
// touches are scalar-cache misses themselves, every wait is a full drain, and 64 waves share a 16 KB scalar cache.  Off
// (MUSTAFAR_META_TOUCH = 0); kept as an experiment knob.
template <int TOFF, int TT>
__device__ __forceinline__ void metab_issue_touch_at(MetaB& m, const uint64_t* __restrict__ bmp, const uint32_t* __restrict__ idx)
{
    if constexpr (TT < 0 || !MUSTAFAR_META_TOUCH) {
        metab_issue_at<TOFF>(m, bmp, idx);
    } else {
#if defined(MUSTAFAR_PROBE_HOTMETA) || !MUSTAFAR_META_TOUCH
        metab_issue_at<TOFF>(m, bmp, idx);
#else
        asm volatile("s_load_dwordx16 %0, %4, %6\n\ts_load_dwordx8 %1, %5, %7\n\ts_load_dword %2, %4, %8\n\ts_load_dword %3, %5, %9"
                     : "=&s"(m.bm), "=&s"(m.ix), "=&s"(m.t0), "=&s"(m.t1)
                     : "s"(bmp), "s"(idx), "i"(TOFF * 8), "i"(TOFF * 4), "i"(TT * 8), "i"(TT * 4));
#endif
    }
}
// coefficients of one step for the four heads: 16 bytes each at base + OFF + h * HS
template <int OFF, int HS>
__device__ __forceinline__ void coef4_issue_at(u32x4 (&c)[4], const void* __restrict__ base)
{
    asm volatile("s_load_dwordx4 %0, %4, %5\n\ts_load_dwordx4 %1, %4, %6\n\t"
                 "s_load_dwordx4 %2, %4, %7\n\ts_load_dwordx4 %3, %4, %8"
                 : "=&s"(c[0]), "=&s"(c[1]), "=&s"(c[2]), "=&s"(c[3])
                 : "s"(base), "i"(OFF), "i"(OFF + HS), "i"(OFF + 2 * HS), "i"(OFF + 3 * HS));
}

template <int G, int OFF, int HS>
__device__ __forceinline__ void coef_issue_at(u32x4 (&c)[G], const void* __restrict__ base)
{
    if constexpr (G == 4) {
        coef4_issue_at<OFF, HS>(c, base);
    } else if constexpr (G == 2) {
        asm volatile("s_load_dwordx4 %0, %2, %3\n\ts_load_dwordx4 %1, %2, %4" : "=&s"(c[0]), "=&s"(c[1]) : "s"(base), "i"(OFF), "i"(OFF + HS));
    } else {
        asm volatile("s_load_dwordx4 %0, %1, %2" : "=&s"(c[0]) : "s"(base), "i"(OFF));
    }
}

// Coefficient rows that are NOT a fixed distance apart (the value entry point's probabilities: rows N * ldb halfs apart, a runtime
// quantity): one base pointer per head, the offset inside the block still an immediate.
template <int G>
struct CoefPtrs {
    const void* p[G];
};
template <int G, int OFF, int HS>
__device__ __forceinline__ void coef_issue_at(u32x4 (&c)[G], const CoefPtrs<G>& cb)
{
    if constexpr (G == 4) {
        asm volatile("s_load_dwordx4 %0, %4, %8\n\ts_load_dwordx4 %1, %5, %8\n\t"
                     "s_load_dwordx4 %2, %6, %8\n\ts_load_dwordx4 %3, %7, %8"
                     : "=&s"(c[0]), "=&s"(c[1]), "=&s"(c[2]), "=&s"(c[3])
                     : "s"(cb.p[0]), "s"(cb.p[1]), "s"(cb.p[2]), "s"(cb.p[3]), "i"(OFF));
    } else if constexpr (G == 2) {
        asm volatile("s_load_dwordx4 %0, %2, %4\n\ts_load_dwordx4 %1, %3, %4" : "=&s"(c[0]), "=&s"(c[1]) : "s"(cb.p[0]), "s"(cb.p[1]), "i"(OFF));
    } else {
        asm volatile("s_load_dwordx4 %0, %1, %2" : "=&s"(c[0]) : "s"(cb.p[0]), "i"(OFF));
    }
}

// The same rows through ONE base pointer and a scalar byte offset per further head (round 5): gfx950's scalar loads take a scalar
// offset AND an immediate (s_load_dwordx4 sdst, sbase, soffset offset:imm), so rows a runtime distance apart cost three scalar
// registers next to the base instead of three more pointers, and nothing but the base moves from block to block.
template <int G>
struct CoefStride {
    const void* base;
    uint32_t off[G > 1 ? G - 1 : 1];   // byte offsets of heads 1 .. G-1 from head 0
};
template <int G, int OFF, int HS>
__device__ __forceinline__ void coef_issue_at(u32x4 (&c)[G], const CoefStride<G>& cb)
{
    if constexpr (G == 4) {
        asm volatile("s_load_dwordx4 %0, %4, %8\n\ts_load_dwordx4 %1, %4, %5 offset:%8\n\t"
                     "s_load_dwordx4 %2, %4, %6 offset:%8\n\ts_load_dwordx4 %3, %4, %7 offset:%8"
                     : "=&s"(c[0]), "=&s"(c[1]), "=&s"(c[2]), "=&s"(c[3])
                     : "s"(cb.base), "s"(cb.off[0]), "s"(cb.off[1]), "s"(cb.off[2]), "i"(OFF));
    } else if constexpr (G == 2) {
        asm volatile("s_load_dwordx4 %0, %2, %4\n\ts_load_dwordx4 %1, %2, %3 offset:%4" : "=&s"(c[0]), "=&s"(c[1]) : "s"(cb.base), "s"(cb.off[0]), "i"(OFF));
    } else {
        asm volatile("s_load_dwordx4 %0, %1, %2" : "=&s"(c[0]) : "s"(cb.base), "i"(OFF));
    }
}

struct Gathered2 {
    uint32_t t[8];   // gathered halfs, EXACT zero where the tile has no element in the lane: even tiles bits 15:0, odd tiles bits 31:16
};
// Zeroing the gather registers (the masked gathers leave the lanes without an element untouched) is one full-rate v_mov_b32 per tile
// -- 1 of the 6.5 vector instructions per tile of the dot2 loop, 1 of the 4.75 of the matrix-pipe loop.  MUSTAFAR_ZFILL = 1 lets the
// LDS pipe do it instead: an UNMASKED ds_read_u16 from an address beyond the workgroup's LDS allocation (the tile's own address
// register + 65534: out-of-range LDS reads return zero; LDS operations of a wave return in order, so the masked gather behind it lands
// on top of the zeros).  Correct (the GPU suite passes with it) and 1 vector instruction per tile cheaper, but SLOWER: the gathers
// already keep the CU's one LDS pipe busy a third of the launch, and a second LDS instruction per tile costs more than the v_mov it
// saves (round 4, c3, kernel us: dot2 42.8 -> 43.4, matrix pipe 32.9 -> 36.0; c4 57.4 -> 63.5, c5 113.7 -> 121.6:
// profiles/r04_probes.txt).  Off.
#ifndef MUSTAFAR_ZFILL
#define MUSTAFAR_ZFILL 0
#endif
#if MUSTAFAR_ZFILL
#define MUSTAFAR_D2_ZERO(j, k) "ds_read_u16 %[t" #j "], %[x" #k "] offset:65534\n\t"
#else
#define MUSTAFAR_D2_ZERO(j, k) "v_mov_b32 %[t" #j "], 0\n\t"
#endif
#define MUSTAFAR_D2_RANK(j, k)                                              \
    "s_lshl2_add_u32 %[u" #k "], %[o" #j "], %[adj]\n\t"                     \
    "v_mbcnt_lo_u32_b32 %[x" #k "], %[l" #j "], 0\n\t"                        \
    "v_mbcnt_hi_u32_b32 %[x" #k "], %[h" #j "], %[x" #k "]\n\t"               \
    "v_lshl_add_u32 %[x" #k "], %[x" #k "], 1, %[u" #k "]\n\t"                \
    MUSTAFAR_D2_ZERO(j, k)
#define MUSTAFAR_D2_LOAD(j, k) "s_mov_b64 exec, %[m" #j "]\n\tds_read_u16 %[t" #j "], %[x" #k "]\n\t"
#define MUSTAFAR_D2_LOAD_HI(j, k) "s_mov_b64 exec, %[m" #j "]\n\tds_read_u16_d16_hi %[t" #j "], %[x" #k "]\n\t"
// (EXEC contract as fma8 / gather8_clean: full wave at entry, restored before the statement ends)
__device__ __forceinline__ void gather8_d2(const MetaB& m, uint32_t adj, Gathered2& g)
{
    const uint64_t m0 = __builtin_bitreverse64(m.bm[0] | ((uint64_t)m.bm[1] << 32));
    const uint64_t m1 = __builtin_bitreverse64(m.bm[2] | ((uint64_t)m.bm[3] << 32));
    const uint64_t m2 = __builtin_bitreverse64(m.bm[4] | ((uint64_t)m.bm[5] << 32));
    const uint64_t m3 = __builtin_bitreverse64(m.bm[6] | ((uint64_t)m.bm[7] << 32));
    const uint64_t m4 = __builtin_bitreverse64(m.bm[8] | ((uint64_t)m.bm[9] << 32));
    const uint64_t m5 = __builtin_bitreverse64(m.bm[10] | ((uint64_t)m.bm[11] << 32));
    const uint64_t m6 = __builtin_bitreverse64(m.bm[12] | ((uint64_t)m.bm[13] << 32));
    const uint64_t m7 = __builtin_bitreverse64(m.bm[14] | ((uint64_t)m.bm[15] << 32));
    uint32_t x0, x1, x2, x3, u0, u1, u2, u3;
    asm volatile(MUSTAFAR_D2_RANK(0, 0) MUSTAFAR_D2_RANK(1, 1) MUSTAFAR_D2_RANK(2, 2) MUSTAFAR_D2_RANK(3, 3)
                 MUSTAFAR_D2_LOAD(0, 0) MUSTAFAR_D2_LOAD_HI(1, 1) MUSTAFAR_D2_LOAD(2, 2) MUSTAFAR_D2_LOAD_HI(3, 3)
                 "s_mov_b64 exec, -1\n\t"
                 MUSTAFAR_D2_RANK(4, 0) MUSTAFAR_D2_RANK(5, 1) MUSTAFAR_D2_RANK(6, 2) MUSTAFAR_D2_RANK(7, 3)
                 MUSTAFAR_D2_LOAD(4, 0) MUSTAFAR_D2_LOAD_HI(5, 1) MUSTAFAR_D2_LOAD(6, 2) MUSTAFAR_D2_LOAD_HI(7, 3)
                 "s_mov_b64 exec, -1"
                 : [t0] "=&v"(g.t[0]), [t1] "=&v"(g.t[1]), [t2] "=&v"(g.t[2]), [t3] "=&v"(g.t[3]), [t4] "=&v"(g.t[4]),
                   [t5] "=&v"(g.t[5]), [t6] "=&v"(g.t[6]), [t7] "=&v"(g.t[7]), [x0] "=&v"(x0), [x1] "=&v"(x1), [x2] "=&v"(x2),
                   [x3] "=&v"(x3), [u0] "=&s"(u0), [u1] "=&s"(u1), [u2] "=&s"(u2), [u3] "=&s"(u3)
                 : MUSTAFAR_MOPS(0), MUSTAFAR_MOPS(1), MUSTAFAR_MOPS(2), MUSTAFAR_MOPS(3), MUSTAFAR_MOPS(4), MUSTAFAR_MOPS(5),
                   MUSTAFAR_MOPS(6), MUSTAFAR_MOPS(7), [o0] "s"(m.ix[0]), [o1] "s"(m.ix[1]), [o2] "s"(m.ix[2]), [o3] "s"(m.ix[3]),
                   [o4] "s"(m.ix[4]), [o5] "s"(m.ix[5]), [o6] "s"(m.ix[6]), [o7] "s"(m.ix[7]), [adj] "s"(adj)
                 : "scc");
}
__device__ __forceinline__ void gather2_wait(Gathered2& g, u32x4 (&c)[4])   // drains the gathers and the coefficient loads
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(g.t[0]), "+v"(g.t[1]), "+v"(g.t[2]), "+v"(g.t[3]), "+v"(g.t[4]), "+v"(g.t[5]), "+v"(g.t[6]), "+v"(g.t[7]),
                   "+s"(c[0]), "+s"(c[1]), "+s"(c[2]), "+s"(c[3]));
}
#define MUSTAFAR_DOT4(p, w)                                                                                          \
    "v_dot2_f32_f16 %[a0], %[t" #p "], %[c0" #w "], %[a0]\n\tv_dot2_f32_f16 %[a1], %[t" #p "], %[c1" #w "], %[a1]\n\t"   \
    "v_dot2_f32_f16 %[a2], %[t" #p "], %[c2" #w "], %[a2]\n\tv_dot2_f32_f16 %[a3], %[t" #p "], %[c3" #w "], %[a3]\n\t"
// DOT hazard of gfx90a / gfx940 / gfx950 (LLVM GCNHazardRecognizer::checkMAIVALUHazards: DotWriteDifferentVALURead = 3,
// DotWriteDifferentVALUWrite = 4): the result of a v_dot2 may be READ by a different vector instruction only 3 wait states later and
// its register be WRITTEN by one only 4 later (the same dot opcode accumulating into it back to back is fine).  The compiler inserts
// those wait states for its own instructions; it cannot see a v_dot2 inside an asm statement, so the statement itself ends with them.
// Round 6 found this the hard way: with the online-softmax text removed the register allocator placed `v_mov_b32 v29, v9` one wait
// state behind the key phase's last `v_dot2_f32_f16 v9, ...` and half of the block pairs got a stale partial score for head 3 (the
// LAST accumulator written) -- bisected on the ISA with tools/isa_patch.sh (profiles/r06_probes.txt item 1); 16 wait states in front of
// that move fixed it, this is the principled form.  tools/check_smem_hazards.py check (6) holds every asm statement to it.
#ifndef MUSTAFAR_DOT_GUARD
#define MUSTAFAR_DOT_GUARD "s_nop 3\n\t"
#endif
// acc[h] += tile(2w) * coef(2w) + tile(2w + 1) * coef(2w + 1): the coefficient dword w of head h holds exactly that pair
__device__ __forceinline__ void fma8_d2(const u32x4 (&c)[4], Gathered2& g, float (&acc)[4])
{
    asm volatile("v_or_b32 %[t0], %[t0], %[t1]\n\tv_or_b32 %[t2], %[t2], %[t3]\n\t"
                 "v_or_b32 %[t4], %[t4], %[t5]\n\tv_or_b32 %[t6], %[t6], %[t7]\n\t"
                 MUSTAFAR_DOT4(0, 0) MUSTAFAR_DOT4(2, 1) MUSTAFAR_DOT4(4, 2) MUSTAFAR_DOT4(6, 3) MUSTAFAR_DOT_GUARD
                 : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), [t0] "+v"(g.t[0]), [t2] "+v"(g.t[2]),
                   [t4] "+v"(g.t[4]), [t6] "+v"(g.t[6])
                 : [t1] "v"(g.t[1]), [t3] "v"(g.t[3]), [t5] "v"(g.t[5]), [t7] "v"(g.t[7]), MUSTAFAR_COPS(0), MUSTAFAR_COPS(1),
                   MUSTAFAR_COPS(2), MUSTAFAR_COPS(3));
}

// Round 5, dot2 engine: the pair register built WITHOUT zeroing and WITHOUT switching EXEC per tile.  The gathers run unmasked (a lane
// without an element reads a neighbouring half of the same stream: a valid LDS address, a value nobody uses), then per pair
//   v_cndmask_b32      t_even, 0, t_even, mask_even                                   -> bits 15:0 = the even tile's element or 0, bits 31:16 = 0
//   v_cndmask_b32_sdwa t_even, zero, t_odd, vcc (= mask_odd)  dst_sel:WORD_1 PRESERVE  -> bits 31:16 = the odd tile's element or 0
// : two vector instructions per pair instead of three (two v_mov_b32 + v_or_b32), 6.0 per tile in the loop instead of 6.5, and one scalar
// move per pair (vcc) instead of two (exec).  `zero`: a register holding 0 (SDWA takes no inline constant on gfx9).  A partial (dst_sel)
// write needs one wait state before a vector instruction reads the register (gfx940+ forwarding hazard): the next pair's two selects
// stand between a pair's SDWA write and its v_dot2 -- and an s_nop behind the last pair's.
// MEASURED SLOWER (same box, kernel us, v_mov + v_or form / this form): c3 39.3-40.2 / 39.9-41.1, c4 66.5 / 68.6, c5 126.7 / 130.3 -- the v_mov_b32 and
// v_or_b32 it saves are the two FULL-rate instructions of the loop (tools/ubench/issue_rates.hip: twice the rate of anything with a
// scalar operand or a VOP3 / SDWA encoding), the two selects are not, and the masks now live to the FMA phase (68 scalar spills
// instead of 44).  Off (MUSTAFAR_D2_SDWA = 0); kept as an experiment knob.
#ifndef MUSTAFAR_D2_SDWA
#define MUSTAFAR_D2_SDWA 0
#endif
#define MUSTAFAR_D2U_RANK(j, k)                                              \
    "s_lshl2_add_u32 %[u" #k "], %[o" #j "], %[adj]\n\t"                     \
    "v_mbcnt_lo_u32_b32 %[x" #k "], %[l" #j "], 0\n\t"                        \
    "v_mbcnt_hi_u32_b32 %[x" #k "], %[h" #j "], %[x" #k "]\n\t"               \
    "v_lshl_add_u32 %[x" #k "], %[x" #k "], 1, %[u" #k "]\n\t"
#define MUSTAFAR_D2U_LOAD(j, k) "ds_read_u16 %[t" #j "], %[x" #k "]\n\t"
struct Gathered2u {
    uint32_t t[8];   // gathered halfs in bits 15:0 (garbage where the tile has no element in the lane: selected away by fma8_d2s)
    uint64_t m[8];   // bit i <=> element i of the tile non-zero
};
__device__ __forceinline__ void gather8_d2u(const MetaB& m, uint32_t adj, Gathered2u& g)
{
#pragma unroll
    for (int j = 0; j < 8; j++) g.m[j] = __builtin_bitreverse64(m.bm[2 * j] | ((uint64_t)m.bm[2 * j + 1] << 32));
    const uint64_t m0 = g.m[0], m1 = g.m[1], m2 = g.m[2], m3 = g.m[3], m4 = g.m[4], m5 = g.m[5], m6 = g.m[6], m7 = g.m[7];
    uint32_t x0, x1, x2, x3, u0, u1, u2, u3;
    asm volatile(MUSTAFAR_D2U_RANK(0, 0) MUSTAFAR_D2U_RANK(1, 1) MUSTAFAR_D2U_RANK(2, 2) MUSTAFAR_D2U_RANK(3, 3)
                 MUSTAFAR_D2U_LOAD(0, 0) MUSTAFAR_D2U_LOAD(1, 1) MUSTAFAR_D2U_LOAD(2, 2) MUSTAFAR_D2U_LOAD(3, 3)
                 MUSTAFAR_D2U_RANK(4, 0) MUSTAFAR_D2U_RANK(5, 1) MUSTAFAR_D2U_RANK(6, 2) MUSTAFAR_D2U_RANK(7, 3)
                 MUSTAFAR_D2U_LOAD(4, 0) MUSTAFAR_D2U_LOAD(5, 1) MUSTAFAR_D2U_LOAD(6, 2) MUSTAFAR_D2U_LOAD(7, 3)
                 : [t0] "=&v"(g.t[0]), [t1] "=&v"(g.t[1]), [t2] "=&v"(g.t[2]), [t3] "=&v"(g.t[3]), [t4] "=&v"(g.t[4]),
                   [t5] "=&v"(g.t[5]), [t6] "=&v"(g.t[6]), [t7] "=&v"(g.t[7]), [x0] "=&v"(x0), [x1] "=&v"(x1), [x2] "=&v"(x2),
                   [x3] "=&v"(x3), [u0] "=&s"(u0), [u1] "=&s"(u1), [u2] "=&s"(u2), [u3] "=&s"(u3)
                 : MUSTAFAR_MOPS(0), MUSTAFAR_MOPS(1), MUSTAFAR_MOPS(2), MUSTAFAR_MOPS(3), MUSTAFAR_MOPS(4), MUSTAFAR_MOPS(5),
                   MUSTAFAR_MOPS(6), MUSTAFAR_MOPS(7), [o0] "s"(m.ix[0]), [o1] "s"(m.ix[1]), [o2] "s"(m.ix[2]), [o3] "s"(m.ix[3]),
                   [o4] "s"(m.ix[4]), [o5] "s"(m.ix[5]), [o6] "s"(m.ix[6]), [o7] "s"(m.ix[7]), [adj] "s"(adj)
                 : "scc");
}
__device__ __forceinline__ void gather2u_wait(Gathered2u& g, u32x4 (&c)[4])   // drains the gathers and the coefficient loads
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(g.t[0]), "+v"(g.t[1]), "+v"(g.t[2]), "+v"(g.t[3]), "+v"(g.t[4]), "+v"(g.t[5]), "+v"(g.t[6]), "+v"(g.t[7]),
                   "+s"(c[0]), "+s"(c[1]), "+s"(c[2]), "+s"(c[3]));
}
#define MUSTAFAR_D2S_PAIR(e, o)                                                                                         \
    "v_cndmask_b32_e64 %[t" #e "], 0, %[t" #e "], %[m" #e "]\n\t"                                                        \
    "s_mov_b64 vcc, %[m" #o "]\n\t"                                                                                     \
    "v_cndmask_b32_sdwa %[t" #e "], %[z], %[t" #o "], vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:WORD_0\n\t"
__device__ __forceinline__ void fma8_d2s(const u32x4 (&c)[4], Gathered2u& g, uint32_t zero, float (&acc)[4])
{
    asm volatile(MUSTAFAR_D2S_PAIR(0, 1) MUSTAFAR_D2S_PAIR(2, 3) MUSTAFAR_DOT4(0, 0) MUSTAFAR_D2S_PAIR(4, 5) MUSTAFAR_DOT4(2, 1)
                 MUSTAFAR_D2S_PAIR(6, 7) MUSTAFAR_DOT4(4, 2) "s_nop 0\n\t" MUSTAFAR_DOT4(6, 3) MUSTAFAR_DOT_GUARD
                 : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), [t0] "+v"(g.t[0]), [t2] "+v"(g.t[2]),
                   [t4] "+v"(g.t[4]), [t6] "+v"(g.t[6])
                 : [t1] "v"(g.t[1]), [t3] "v"(g.t[3]), [t5] "v"(g.t[5]), [t7] "v"(g.t[7]), [z] "v"(zero),
                   [m0] "s"(g.m[0]), [m1] "s"(g.m[1]), [m2] "s"(g.m[2]), [m3] "s"(g.m[3]), [m4] "s"(g.m[4]), [m5] "s"(g.m[5]),
                   [m6] "s"(g.m[6]), [m7] "s"(g.m[7]), MUSTAFAR_COPS(0), MUSTAFAR_COPS(1), MUSTAFAR_COPS(2), MUSTAFAR_COPS(3)
                 : "vcc");
}

// The matrix-pipe engine on the lean addressing: the dot2 form's gather (tile pairs packed in one register, exact zeros where a lane
// has no element), then per FOUR tiles one v_mfma_f32_4x4x4_16B_f16 in place of eight v_dot2 -- the lane's A fragment is the four
// coefficient halfs of head (lane % 4), read with one ds_read_b64 from a [4 heads][coefficients] table in LDS at ctab_lane =
// table + (lane % 4) * row stride (layout and roles: the note at GatheredClean).  CBASE = byte offset of the chunk's first
// coefficient in a table row.  Per tile: 2 v_mbcnt + v_lshl_add + v_mov (zero) + 1/2 v_or + 1/4 v_mfma.
template <int OFF>
__device__ __forceinline__ void coefm_issue_at(uint32_t ctab_lane, uint64_t& a0, uint64_t& a1)
{
    asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4" : "=&v"(a0), "=&v"(a1) : "v"(ctab_lane), "i"(OFF), "i"(OFF + 8));
}
__device__ __forceinline__ void gatherm_wait(Gathered2& g, uint64_t& a0, uint64_t& a1)
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(g.t[0]), "+v"(g.t[1]), "+v"(g.t[2]), "+v"(g.t[3]), "+v"(g.t[4]), "+v"(g.t[5]), "+v"(g.t[6]), "+v"(g.t[7]),
                   "+v"(a0), "+v"(a1));
}
__device__ __forceinline__ void fma8_m2(const Gathered2& g, uint64_t a0, uint64_t a1, f32x4& acc)
{
    acc = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(h16x4, a0), pack4(g.t[0], g.t[1], g.t[2], g.t[3]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(h16x4, a1), pack4(g.t[4], g.t[5], g.t[6], g.t[7]), acc, 0, 0, 0);
}
template <int TOFF, int CBASE>
__device__ __forceinline__ void chunk32_mfma_at(uint32_t adj, const uint64_t* __restrict__ bmp_t, const uint32_t* __restrict__ idx_t,
                                                uint32_t ctab_lane, f32x4& acc)
{
    MetaB cur, nxt;
    metab_issue_at<TOFF>(cur, bmp_t, idx_t);
    metab_wait(cur);
#define MUSTAFAR_STEP(S)                                    \
    {                                                       \
        Gathered2 g;                                        \
        uint64_t a0, a1;                                    \
        if constexpr (MUSTAFAR_META_EARLY) metab_issue_at<TOFF + 8 * (S + 1)>(nxt, bmp_t, idx_t); \
        gather8_d2(cur, adj, g);                            \
        coefm_issue_at<CBASE + 16 * S>(ctab_lane, a0, a1);  \
        gatherm_wait(g, a0, a1);                            \
        if constexpr (!MUSTAFAR_META_EARLY) metab_issue_at<TOFF + 8 * (S + 1)>(nxt, bmp_t, idx_t); \
        fma8_m2(g, a0, a1, acc);                            \
        if constexpr (MUSTAFAR_META_EARLY) metab_ready(nxt); else metab_wait(nxt); \
        cur = nxt;                                          \
    }
    MUSTAFAR_STEP(0) MUSTAFAR_STEP(1) MUSTAFAR_STEP(2)
#undef MUSTAFAR_STEP
    {
        Gathered2 g;
        uint64_t a0, a1;
        gather8_d2(cur, adj, g);
        coefm_issue_at<CBASE + 48>(ctab_lane, a0, a1);
        gatherm_wait(g, a0, a1);
        fma8_m2(g, a0, a1, acc);
    }
}

// prefetch_meta without a divergent region (every lane loads; the lanes beyond the last sector repeat it): the lean kernels call it
// inside their block loops, right in front of asm statements that own EXEC.  Tiles [T0, T0 + NT) of the block (a wave of the pair
// form asks for its own half only), one 4-byte load per 64-byte line: NT / 8 lines of bitmaps, NT / 16 + 2 of offsets (the row may
// start anywhere).  (One load per 32-byte sector measured the same: L2 fills whole lines.  MUSTAFAR_PF_SECTOR: experiment knob.)
#ifndef MUSTAFAR_PF_SECTOR
#define MUSTAFAR_PF_SECTOR 64
#endif
template <int T0, int NT>
__device__ __forceinline__ uint32_t prefetch_meta_all(const uint64_t* __restrict__ bmp_t, const uint32_t* __restrict__ idx_t, int lane)
{
    constexpr int kSec = MUSTAFAR_PF_SECTOR;
    constexpr int kB = NT * 8 / kSec, kI = (NT * 4 + kSec - 1) / kSec + 1;
    static_assert(kB + kI <= 64, "one load per sector and lane");
    const int k = lane < kB + kI - 1 ? lane : kB + kI - 1;
    const unsigned char* a = reinterpret_cast<const unsigned char*>(bmp_t + T0) + k * kSec;
    const unsigned char* b = reinterpret_cast<const unsigned char*>(idx_t + T0) + (k - kB) * kSec;
    return *reinterpret_cast<const uint32_t*>(k < kB ? a : b);
}

// One staged chunk (32 tiles) of the lean kernel; the step schedule is chunk32's.
//   bmp_t / idx_t : the BLOCK's bitmaps / offsets (the chunk starts TOFF tiles in);  cbase + COFF + h * HS : the chunk's first
//   coefficient of head h
template <int ENG, int TOFF, int COFF, int HS, int G = 4, class CB = const void*, int NTOFF = -2>   // CB: const void* (rows HS bytes apart) or CoefPtrs<G>;
                                           // NTOFF: tile offset of the chunk the wave works on NEXT (-1: none; -2: no touches at all, the round-4 callers)
__device__ __forceinline__ void chunk32_at(uint32_t adj, const uint64_t* __restrict__ bmp_t, const uint32_t* __restrict__ idx_t,
                                           const CB& cbase, float (&acc)[G])
{
    // tile offset of the metadata the request of step S + 1 touches -- that of step S + 2, the next chunk's first step behind this chunk's last
    // (metab_issue_touch_at) -- or -1
#define MUSTAFAR_TT(S) (NTOFF == -2 ? -1 : ((S) + 2 < 4 ? TOFF + 8 * ((S) + 2) : (NTOFF >= 0 ? NTOFF + 8 * ((S) + 2 - 4) : -1)))
    static_assert(G == 4 || ENG == 0, "dot2 pairs four heads' coefficients; G < 4 runs v_fma_mix");
    // (with the coefficients in scalar registers as well, an early request leaves the loop 140+ registers short of the 78 a wave of
    // this launch has, and the compiler then spills registers that loads are still writing: tools/check_smem_hazards.py)
    constexpr bool kEarly = MUSTAFAR_META_EARLY > 1;
    MetaB cur, nxt;
    u32x4 c[G];
    uint32_t zero = 0;   // (ENG == 2: the SDWA select's zero operand, held in a vector register)
    if constexpr (ENG == 2 && MUSTAFAR_D2_SDWA) asm volatile("" : "+v"(zero));
    metab_issue_touch_at<TOFF, (NTOFF == -2 ? -1 : TOFF + 8)>(cur, bmp_t, idx_t);
    coef_issue_at<G, COFF, HS>(c, cbase);
    metab_wait(cur);
#define MUSTAFAR_STEP(S)                                        \
    if constexpr (kEarly) metab_issue_touch_at<TOFF + 8 * (S + 1), MUSTAFAR_TT(S)>(nxt, bmp_t, idx_t); \
    if constexpr (ENG == 2) {                                   \
        if constexpr (G == 4) {                                 \
            if constexpr (MUSTAFAR_D2_SDWA) {                   \
                Gathered2u g;                                   \
                gather8_d2u(cur, adj, g);                       \
                gather2u_wait(g, c);                            \
                if constexpr (!kEarly) metab_issue_touch_at<TOFF + 8 * (S + 1), MUSTAFAR_TT(S)>(nxt, bmp_t, idx_t);  \
                fma8_d2s(c, g, zero, acc);                      \
            } else {                                            \
                Gathered2 g;                                    \
                gather8_d2(cur, adj, g);                        \
                gather2_wait(g, c);                             \
                if constexpr (!kEarly) metab_issue_touch_at<TOFF + 8 * (S + 1), MUSTAFAR_TT(S)>(nxt, bmp_t, idx_t);  \
                fma8_d2(c, g, acc);                             \
            }                                                   \
        }                                                       \
    } else {                                                    \
        Gathered g;                                             \
        gather8(cur, adj, g);                                   \
        gather_wait<G>(g, c);                                   \
        if constexpr (!kEarly) metab_issue_touch_at<TOFF + 8 * (S + 1), MUSTAFAR_TT(S)>(nxt, bmp_t, idx_t);  \
        fma8<G>(c, g, acc);                                     \
    }                                                           \
    if constexpr (kEarly) metab_ready(nxt); else metab_wait(nxt); \
    coef_issue_at<G, COFF + 16 * (S + 1), HS>(c, cbase);        \
    cur = nxt;
    MUSTAFAR_STEP(0) MUSTAFAR_STEP(1) MUSTAFAR_STEP(2)
#undef MUSTAFAR_STEP
    if constexpr (ENG == 2) {
        if constexpr (G == 4) {
            if constexpr (MUSTAFAR_D2_SDWA) {
                Gathered2u g;
                gather8_d2u(cur, adj, g);
                gather2u_wait(g, c);
                fma8_d2s(c, g, zero, acc);
            } else {
                Gathered2 g;
                gather8_d2(cur, adj, g);
                gather2_wait(g, c);
                fma8_d2(c, g, acc);
            }
        }
    } else {
        Gathered g;
        gather8(cur, adj, g);
        gather_wait<G>(g, c);
        fma8<G>(c, g, acc);
    }
#undef MUSTAFAR_TT
}

// The 128 tiles of one 64-token block against the coefficients at cbase (four rows HS bytes apart).
//   VAL = false (key):   all four chunks -> accA (lane = token);  coefficient of tile d: halfs d of the rows
//   VAL = true  (value): chunks 0, 1 -> accA (channels 0..63), chunks 2, 3 -> accB (channels 64..127), lane = channel;
//                        coefficient of a tile: the row's half (token % 64)
//   bnd: the block's five chunk bounds (bnd_load), fetched by the caller ahead of time
struct NoMid {
    __device__ __forceinline__ void operator()() const {}
};
template <int ENG, int HS, bool VAL, int CB, int CN, int G = 4, class CBT = const void*, int BOFF = 0, class MID = NoMid>   // chunks [CB, CB + CN) of the block; G heads (G < 4: ENG 0);
                                                                                                      // BOFF: lane of `bnd` that holds the block's first bound;
                                                                                                      // MID: called in front of the phase's LAST chunk (the caller's requests for what follows the phase)
__device__ __forceinline__ void lean_block_phase(unsigned char* lds, uint32_t lds_addr, const uint64_t* __restrict__ bmp_t,
                                                 const uint32_t* __restrict__ idx_t, const unsigned char* __restrict__ nz_h,
                                                 const CBT& cbase, uint32_t bnd, int lane, float (&accA)[G],
                                                 float (&accB)[G]
#ifdef MUSTAFAR_WAVE_TRACE
                                                 , PhaseTrace& phase_trace_
#endif
                                                 , uint32_t ctab_lane = 0   // ENG == 1: the LDS coefficient table (cbase unused)
                                                 , const MID& mid = MID()
                                                 )
{
    f32x4 mA = {0.f, 0.f, 0.f, 0.f}, mB = {0.f, 0.f, 0.f, 0.f};   // (ENG == 1 only)
    if constexpr (ENG == 1) {
        static_assert(ENG != 1 || G == 4, "the matrix-pipe engine multiplies four heads at a time");
#pragma unroll
        for (int h = 0; h < G; h++) { mA[h] = accA[h]; mB[h] = accB[h]; }
    }
    uint32_t i0 = bnd_get(bnd, BOFF + CB);
    const uint32_t len0 = 4u * (bnd_get(bnd, BOFF + CB + 1) - i0);
    Stage st = stage_issue<ENG == 1>(nz_h + 4ull * i0, len0, lane);
    stage_commit<ENG == 1>(lds, st, lane, len0);
#ifdef MUSTAFAR_WAVE_TRACE
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    MUSTAFAR_PTRACE_STAMP(VAL ? 4 : 1);
#pragma unroll
    for (int c = CB; c < CB + CN; c++) {
        uint32_t n0 = 0, nlen = 0;
        if (c < CB + CN - 1) {
            n0 = bnd_get(bnd, BOFF + c + 1);
            nlen = 4u * (bnd_get(bnd, BOFF + c + 2) - n0);
            st = stage_issue<ENG == 1>(nz_h + 4ull * n0, nlen, lane);
        }
        if (c == CB + CN - 1) mid();
        __builtin_amdgcn_wave_barrier();
#ifdef MUSTAFAR_PROBE_HOTMETA
        const uint32_t adj = __builtin_amdgcn_readfirstlane(lds_addr);   // the fixed offsets of g_hot_idx stay inside the window
#else
        const uint32_t adj = __builtin_amdgcn_readfirstlane(lds_addr - 4u * i0);
#endif
        if constexpr (ENG == 1) {
            if (c == 0)      chunk32_mfma_at<0, 0>(adj, bmp_t, idx_t, ctab_lane, mA);
            else if (c == 1) chunk32_mfma_at<32, 64>(adj, bmp_t, idx_t, ctab_lane, mA);
            else if (c == 2) chunk32_mfma_at<64, VAL ? 0 : 128>(adj, bmp_t, idx_t, ctab_lane, VAL ? mB : mA);
            else             chunk32_mfma_at<96, VAL ? 64 : 192>(adj, bmp_t, idx_t, ctab_lane, VAL ? mB : mA);
        } else {
            if (c == 0)      chunk32_at<ENG, 0, 0, HS, G, CBT>(adj, bmp_t, idx_t, cbase, accA);
            else if (c == 1) chunk32_at<ENG, 32, 64, HS, G, CBT>(adj, bmp_t, idx_t, cbase, accA);
            else if (c == 2) chunk32_at<ENG, 64, VAL ? 0 : 128, HS, G, CBT>(adj, bmp_t, idx_t, cbase, VAL ? accB : accA);
            else             chunk32_at<ENG, 96, VAL ? 64 : 192, HS, G, CBT>(adj, bmp_t, idx_t, cbase, VAL ? accB : accA);
        }
        __builtin_amdgcn_wave_barrier();
        if (c < CB + CN - 1) {
            stage_commit<ENG == 1>(lds, st, lane, nlen);
            i0 = n0;
        }
    }
    if constexpr (ENG == 1) {   // (the pair form passes the same array for accA and accB: only the one its chunks went to is written)
        if (CB < 2 || !VAL) {
#pragma unroll
            for (int h = 0; h < G; h++) accA[h] = mA[h];
        }
        if (VAL && CB + CN > 2) {
#pragma unroll
            for (int h = 0; h < G; h++) accB[h] = mB[h];
        }
    }
}

// TWO consecutive blocks (A and B = A + 1) of a wave's half -- its 64 tiles of each -- as ONE pipeline of four chunks (A0, A1, B0, B1): the
// first chunk of B is in flight while the last chunk of A is worked on, so that a pair of phases starts with ONE exposed stream latency
// instead of two (round 5, decode_onepass_sb_kernel).  bmp_t / idx_t / cbase: block A's (block B's tiles lie 128 tiles behind; its e
// segments -- VAL -- G x 128 bytes behind, its q rows are the same); bnd: A's three bounds in lanes 0..2, B's in lanes 8..10.
//   mid1: called in front of chunk A1, mid3 in front of chunk B1 (the caller's requests for what comes next).
template <int ENG, int HS, bool VAL, int G, class MID1, class MID3, class CBT = const void*, int EB = (VAL ? G * 64 * 2 : 0)>   // EB: coefficient offset of block B (bytes): its e rows lie G segments of 64 halfs behind block A's
__device__ __forceinline__ void lean_pair_phase(unsigned char* lds, uint32_t lds_addr, const uint64_t* __restrict__ bmp_t,
                                                const uint32_t* __restrict__ idx_t, const unsigned char* __restrict__ nz_h,
                                                const CBT& cbase, uint32_t bnd, int lane, float (&accA)[G], float (&accB)[G]
#ifdef MUSTAFAR_WAVE_TRACE
                                                , PhaseTrace& phase_trace_
#endif
                                                , uint32_t ctab_lane, const MID1& mid1, const MID3& mid3)
{
    constexpr int kE = EB;   // coefficient offset of block B (bytes): the next block's e segments (one-pass launch); the same q rows (key phase)
    f32x4 mA = {0.f, 0.f, 0.f, 0.f}, mB = {0.f, 0.f, 0.f, 0.f};   // (ENG == 1 only)
    if constexpr (ENG == 1) {
        static_assert(ENG != 1 || G == 4, "the matrix-pipe engine multiplies four heads at a time");
#pragma unroll
        for (int h = 0; h < G; h++) { mA[h] = accA[h]; mB[h] = VAL ? 0.f : accB[h]; }   // (VAL: both blocks add into accA's registers)
    }
    uint32_t i0 = bnd_get(bnd, 0);
    const uint32_t len0 = 4u * (bnd_get(bnd, 1) - i0);
    Stage st = stage_issue<ENG == 1>(nz_h + 4ull * i0, len0, lane);
    stage_commit<ENG == 1>(lds, st, lane, len0);
#ifdef MUSTAFAR_WAVE_TRACE
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    MUSTAFAR_PTRACE_STAMP(VAL ? 4 : 1);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t n0 = 0, nlen = 0;
        if (k < 3) {
            const int nl = ((k + 1) >> 1) * 8 + ((k + 1) & 1);   // lane of the next chunk's lower bound: 1, 8, 9
            n0 = bnd_get(bnd, nl);
            nlen = 4u * (bnd_get(bnd, nl + 1) - n0);
            st = stage_issue<ENG == 1>(nz_h + 4ull * n0, nlen, lane);
        }
        if (k == 1) mid1();
        if (k == 3) mid3();
        __builtin_amdgcn_wave_barrier();
        const uint32_t adj = __builtin_amdgcn_readfirstlane(lds_addr - 4u * i0);
        if constexpr (ENG == 1) {
            if (k == 0)      chunk32_mfma_at<0, 0>(adj, bmp_t, idx_t, ctab_lane, mA);
            else if (k == 1) chunk32_mfma_at<32, 64>(adj, bmp_t, idx_t, ctab_lane, mA);
            else if (k == 2) chunk32_mfma_at<128, 0>(adj, bmp_t, idx_t, ctab_lane + (VAL ? 4 * kValTabStride : 0), VAL ? mA : mB);
            else             chunk32_mfma_at<160, 64>(adj, bmp_t, idx_t, ctab_lane + (VAL ? 4 * kValTabStride : 0), VAL ? mA : mB);
        } else {
            if (k == 0)      chunk32_at<ENG, 0, 0, HS, G, CBT, 32>(adj, bmp_t, idx_t, cbase, accA);
            else if (k == 1) chunk32_at<ENG, 32, 64, HS, G, CBT, 128>(adj, bmp_t, idx_t, cbase, accA);
            else if (k == 2) chunk32_at<ENG, 128, kE, HS, G, CBT, 160>(adj, bmp_t, idx_t, cbase, accB);
            else             chunk32_at<ENG, 160, kE + 64, HS, G, CBT, -1>(adj, bmp_t, idx_t, cbase, accB);
        }
        __builtin_amdgcn_wave_barrier();
        if (k < 3) {
            stage_commit<ENG == 1>(lds, st, lane, nlen);
            i0 = n0;
        }
    }
    if constexpr (ENG == 1) {
#pragma unroll
        for (int h = 0; h < G; h++) {
            accA[h] = mA[h];
            if (!VAL) accB[h] = mB[h];
        }
    }
}

// grid: x = ceil(T / 64 / (4 * tb_per_wg)) (a.tb_per_wg = consecutive 64-token blocks per WAVE here), y = kv-heads * groups / 4
// (+ win_rows leading rows of window workgroups, as decode_onepass_kernel).  e scratch: a.e_rows is used as
// [y][T / 64][4 heads][64 tokens] halfs (4 T halfs per y: fits the [BH, ld >= T] score buffer), every block's four segments
// 512 contiguous bytes on scalar-cache lines of their own.
// INVARIANT of the e round trip (vector stores, then scalar loads of the same bytes by the SAME wave): the wave waits vmcnt(0)
// -- the stores are acknowledged by L2 -- before it issues the scalar loads; no wave scalar-reads those lines earlier in the
// launch (a block's segments belong to one wave and are 64-byte aligned, ld_scores % 32 == 0 is checked by the host), so the
// only stale copy the scalar cache could hold is one from an EARLIER launch reusing the scratch, and every dispatch (graph
// nodes included) starts with an acquire that invalidates the scalar cache.  tests/test_gpu_benchshape.py replays a captured
// step with alternating queries to hold this.
template <int ENG>
__global__ __launch_bounds__(kThreads) void decode_onepass_lean_kernel(
    const uint64_t* __restrict__ k_bmp, const unsigned char* __restrict__ k_nz, const uint32_t* __restrict__ k_idx,
    const uint32_t* __restrict__ k_nz_off, const uint64_t* __restrict__ v_bmp, const unsigned char* __restrict__ v_nz,
    const uint32_t* __restrict__ v_idx, const uint32_t* __restrict__ v_nz_off, OneArgs a, int64_t k_bmp_stride,
    int64_t k_idx_stride, uint32_t k_nz_stride, int64_t v_bmp_stride, int64_t v_idx_stride, uint32_t v_nz_stride)
{
    constexpr int G = 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[kWaves * kStageBytes];
    static_assert(kWaves * kStageBytes >= (kWaves * 2 * 4 * 64 + 2 * kWaves * 4) * 4, "combine buffers must fit in the stage area");
    MUSTAFAR_PTRACE_BEGIN();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wrows = a.win_rows < 0 ? -a.win_rows : a.win_rows;   // window rows lead (win_rows > 0) or trail (< 0) the grid
    const int wy = a.win_rows < 0 ? (int)blockIdx.y - ((int)gridDim.y - wrows) : (int)blockIdx.y;
    if (wy >= 0 && wy < wrows) {   // dense window
        const int task = wy * gridDim.x + blockIdx.x;
        if (task < (int)(gridDim.y - wrows) * a.nchunks) onepass_window_wg<G>(smem, win_args(a), task, gridDim.x);
        MUSTAFAR_PTRACE_END(5);
        return;
    }
    const int by = blockIdx.y - (a.win_rows > 0 ? a.win_rows : 0);
    const int hb_per_kv = a.groups >> 2;
    const int kvh = hb_per_kv == 1 ? by : by / hb_per_kv;
    const int bh0 = kvh * a.groups + (by - kvh * hb_per_kv) * G;
    const int ntb = a.T >> 6;
    int tb = (blockIdx.x * kWaves + wave) * a.tb_per_wg;
    const int tb_end = min(ntb, tb + a.tb_per_wg);
    const int64_t tiles = (int64_t)ntb * kTilesPerTb;
    const uint64_t* kb = k_bmp + (int64_t)kvh * (k_bmp_stride ? k_bmp_stride : tiles);
    const uint32_t* ki = k_idx + (int64_t)kvh * (k_idx_stride ? k_idx_stride : tiles + 1);
    const unsigned char* kn = k_nz + 16ull * (k_nz_stride ? (uint64_t)kvh * k_nz_stride : (uint64_t)k_nz_off[kvh]);
    const uint64_t* vb = v_bmp + (int64_t)kvh * (v_bmp_stride ? v_bmp_stride : tiles);
    const uint32_t* vi = v_idx + (int64_t)kvh * (v_idx_stride ? v_idx_stride : tiles + 1);
    const unsigned char* vn = v_nz + 16ull * (v_nz_stride ? (uint64_t)kvh * v_nz_stride : (uint64_t)v_nz_off[kvh]);
    const h16* qb = a.q + (int64_t)bh0 * kD;                     // the four q rows, 256 bytes apart
    h16* eb = a.e_rows + (int64_t)by * ntb * (G * 64);            // this head batch's e segments, [block][4][64]
    const h16* mrow = a.mask.ptr ? a.mask.ptr + (int64_t)(bh0 / a.mask.heads) * a.mask.stride : nullptr;
    unsigned char* lds = smem + wave * kStageBytes;
    const uint32_t lds_addr = (uint32_t)reinterpret_cast<uintptr_t>(lds);
    // e = exp(x - max) * kEScale as fp16 (ENG 2: out of the subnormal range, see above); kEScaleLog2 is added to the exponent
    constexpr float kEScaleLog2 = ENG == 2 ? 15.f : 0.f;

    float m_run[G], l_run[G], acc0[G], acc1[G];   // (m_run, l_run: wave-uniform)
#pragma unroll
    for (int h = 0; h < G; h++) { m_run[h] = -INFINITY; l_run[h] = 0.f; acc0[h] = 0.f; acc1[h] = 0.f; }
#pragma unroll 1
    for (; tb < tb_end; tb++) {
        const uint64_t* kbt = kb + (int64_t)tb * kTilesPerTb;
        const uint32_t* kit = ki + (int64_t)tb * kTilesPerTb;
        const uint64_t* vbt = vb + (int64_t)tb * kTilesPerTb;
        const uint32_t* vit = vi + (int64_t)tb * kTilesPerTb;
        // everything the block needs from memory before its streams is requested up front: metadata lines into L2, the chunk
        // bounds of both sides, the mask column of the lane's token
        const uint32_t pfk = prefetch_meta_all<0, 128>(kbt, kit, lane);
        const uint32_t bnd_k = bnd_load(kit, lane);
        const uint32_t pfv = prefetch_meta_all<0, 128>(vbt, vit, lane);
        const uint32_t bnd_v = bnd_load(vit, lane);
        const h16 mk = mrow ? mrow[tb * 64 + lane] : (h16)0.f;   // (mrow is wave-uniform: a scalar branch)
        float s[G];
#pragma unroll
        for (int h = 0; h < G; h++) s[h] = 0.f;
        lean_block_phase<ENG, kD * 2, false, 0, 4>(lds, lds_addr, kbt, kit, kn, qb, bnd_k, lane, s, s MUSTAFAR_PTRACE_ARG);
        prefetch_done(pfk);
        MUSTAFAR_PTRACE_STAMP(2);
        // ---- softmax step (as decode_onepass_kernel): running maximum, e -> the block's segments, running sum, rescale
        h16* eblk = eb + (int64_t)tb * (G * 64);
#pragma unroll
        for (int h = 0; h < G; h++) {
            float x = scaled((h16)s[h], a.inv_sqrt_d);        // fp16 score (SpMM_Kernel.cuh:418), / sqrt(d) in fp16 (model :284)
            if (mrow) x = masked(x, mk);
            const float m_new = uniform_f(fmaxf(m_run[h], wave_max(x)));
            const float alpha = uniform_f(__expf(m_run[h] - m_new));   // 0 for the first block (m_run = -inf)
            const h16 e = (h16)__builtin_amdgcn_exp2f((x - m_new) * 1.44269504f + kEScaleLog2);
            eblk[h * 64 + lane] = e;
            l_run[h] = uniform_f(l_run[h] * alpha + wave_sum((float)e));
            m_run[h] = m_new;
            acc0[h] *= alpha;
            acc1[h] *= alpha;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the e stores have reached L2 (see the invariant above)
        MUSTAFAR_PTRACE_STAMP(3);
        lean_block_phase<ENG, 64 * 2, true, 0, 4>(lds, lds_addr, vbt, vit, vn, eblk, bnd_v, lane, acc0, acc1 MUSTAFAR_PTRACE_ARG);
        prefetch_done(pfv);
        MUSTAFAR_PTRACE_STAMP(5);
    }
    // ---- merge the four waves: common maximum, rescaled sums and outputs -> one slab per head (as decode_onepass_kernel)
    float* red = reinterpret_cast<float*>(smem);                 // [kWaves][2G][64]
    float* s_m = red + kWaves * 2 * G * 64;                      // [kWaves][G]
    float* s_l = s_m + kWaves * G;                               // [kWaves][G]
    __syncthreads();   // every wave is done with its stage window
    if (lane < G) {
        float mine = m_run[0];
#pragma unroll
        for (int h = 1; h < G; h++) mine = (lane == h) ? m_run[h] : mine;
        s_m[wave * G + lane] = mine;
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < G; h++) {
        float M = s_m[h];
#pragma unroll
        for (int w = 1; w < kWaves; w++) M = fmaxf(M, s_m[w * G + h]);
        // a wave without blocks weighs nothing; the e scale leaves here (a power of two: exact)
        const float scale = (m_run[h] == -INFINITY) ? 0.f : __expf(m_run[h] - M) * (ENG == 2 ? 0x1p-15f : 1.f);
        if (lane == 0) s_l[wave * G + h] = l_run[h] * scale;
        red[(wave * 2 * G + h) * 64 + lane]     = acc0[h] * scale;
        red[(wave * 2 * G + G + h) * 64 + lane] = acc1[h] * scale;
    }
    __syncthreads();
    float* slab_o = a.ws_o + ((int64_t)blockIdx.x * a.BH + bh0) * kD;
    for (int o = threadIdx.x; o < 2 * G * 64; o += kThreads) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; w++) sum += red[w * 2 * G * 64 + o];
        const int hh = o >> 6, l = o & 63;   // hh = half * G + h
        slab_o[(hh % G) * kD + (hh / G) * 64 + l] = sum;
    }
    if (threadIdx.x < G) {
        const int h = threadIdx.x;
        float M = s_m[h], L = s_l[h];
#pragma unroll
        for (int w = 1; w < kWaves; w++) { M = fmaxf(M, s_m[w * G + h]); L += s_l[w * G + h]; }
        float* slab_ml = a.ws_ml + ((int64_t)blockIdx.x * a.BH + bh0 + h) * 2;
        slab_ml[0] = M;
        slab_ml[1] = L;
    }
    MUSTAFAR_PTRACE_END(3);
}

// The lean form at the PAIR grain: two waves share a 64-token block -- 64 channels each in the key phase (partial scores folded
// through LDS), one 64-channel half of the output each in the value phase -- and a workgroup (two pairs) takes two blocks at a
// time.  The whole-block form above has the fewest instructions per tile but its waves live 40 us and wait on memory with nobody
// to cover for them (probe builds: -12 us without the stream loads, -8 us with hot metadata; the pair grain: -2 / -2): half-size
// waves, twice as many, is the grain at which the launch is bound by instruction issue, so this is the form that runs by default.
//   grid: x = ceil(T / 64 / a.tb_per_wg) (a.tb_per_wg = blocks per WORKGROUP, even), y as decode_onepass_lean_kernel.
//   e round trip (vector engines): each wave stores the e segments of the two heads it finishes (the softmax step is split between
//   the waves of a pair) and waits vmcnt(0) before the barrier; both waves then scalar-load all four segments (the invariant stated
//   at decode_onepass_lean_kernel, with "the same wave" read as "the same pair").  Matrix-pipe engine: e goes into the pair's LDS table.
// Compiled for 8 waves per SIMD (<= 64 vector registers; the scalar file then spills ~45 values to lanes of a vector register):
// with the stream loads non-temporal the launch is bound by how many waves are there to cover for each other (c3, dot2 form:
// 7 waves 45.6 us, 8 waves 44.3 us).  MUSTAFAR_LP_WAVES: experiment knob (tools/build_variant.sh).
#ifndef MUSTAFAR_LP_WAVES
#define MUSTAFAR_LP_WAVES 8
#endif
#define MUSTAFAR_LP_BOUNDS __launch_bounds__(kThreads, MUSTAFAR_LP_WAVES)
// Issue priority by PROGRESS (pair form): s_setprio 1 while a wave is on its FIRST block, 0 afterwards.  The arbiter serves equal
// priorities oldest wave first, so the workgroups dispatched first ran ahead and left (the first ones at 60 % of the launch's span)
// while the youngest dragged on at falling occupancy; with this the laggards overtake whoever has reached a second block.
// Measured (c3, kernel us, same box): dot2 45.3 -> 44.3, matrix pipe 35.5 -> 33.2; c4 76.1 -> 73.2 / 59.9 -> 57.3; c5 unchanged.  A level
// per PHASE (3, 2, 1, 0) was slower on the vector engines (c3 46.9, c5 148.6 vs 139.5: strict least-progress-first lines the waves
// up on memory) -- profiles/r04_probes.txt.  MUSTAFAR_PRIO=0: off (experiment builds).
#ifndef MUSTAFAR_PRIO
#define MUSTAFAR_PRIO 1
#endif
template <int ENG, bool EXT = false, int G = 4>   // G: q-heads per kv-head served by one pass (4, 2 or 1; G < 4 runs the v_fma_mix engine);
                                                // EXT: the cache grew by extents (a.k_ext / a.v_ext / a.nb0); an instantiation of its own, so that
                                       // the plain launch does not carry the extra arguments (matrix-pipe form at c3: 37.6 vs 38.7 us)
__global__ MUSTAFAR_LP_BOUNDS void decode_onepass_leanpair_kernel(
    const uint64_t* __restrict__ k_bmp, const unsigned char* __restrict__ k_nz, const uint32_t* __restrict__ k_idx,
    const uint32_t* __restrict__ k_nz_off, const uint64_t* __restrict__ v_bmp, const unsigned char* __restrict__ v_nz,
    const uint32_t* __restrict__ v_idx, const uint32_t* __restrict__ v_nz_off, OneArgs a, int64_t k_bmp_stride,
    int64_t k_idx_stride, uint32_t k_nz_stride, int64_t v_bmp_stride, int64_t v_idx_stride, uint32_t v_nz_stride)
{
    static_assert(G == 4 || ENG == 0, "dot2 and the matrix pipe work on four heads; G < 4 runs v_fma_mix");
    constexpr int HW = G >= 2 ? G / 2 : 1;   // heads a wave of the pair finishes in the softmax step (G = 1: the even wave its one head, the odd wave none)
    // matrix-pipe engine: behind the stage windows, the q rows of the four heads ([4][kKeyTabStride]) and one e table per pair ([4][kValTabStride])
    constexpr int kTabBytes = ENG == 1 ? 4 * kKeyTabStride + 2 * 4 * kValTabStride : 0;
    __shared__ __attribute__((aligned(16))) unsigned char smem[kWaves * kStageBytes + kTabBytes + 2 * G * 4];   // (+ the pairs' rescale factors)
    MUSTAFAR_PTRACE_BEGIN();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wrows = a.win_rows < 0 ? -a.win_rows : a.win_rows;   // window rows lead (win_rows > 0) or trail (< 0) the grid
    const int wy = a.win_rows < 0 ? (int)blockIdx.y - ((int)gridDim.y - wrows) : (int)blockIdx.y;
    if (wy >= 0 && wy < wrows) {   // dense window
#ifdef MUSTAFAR_WIN_PRIO   // experiment: the window workgroups (short latency chains dispatched behind the SpMV rows) at a high issue priority
        __builtin_amdgcn_s_setprio(MUSTAFAR_WIN_PRIO);
#endif
        const int task = wy * gridDim.x + blockIdx.x;
        if (task < (int)(gridDim.y - wrows) * a.nchunks) {
            int T_used = -1;
            if constexpr (EXT) { if (a.t_dev) T_used = __builtin_amdgcn_readfirstlane(*a.t_dev); }
            onepass_window_wg<G>(smem, win_args(a, T_used), task, a.pair_slabs ? 2 * gridDim.x : gridDim.x);
        }
        MUSTAFAR_PTRACE_END(5);
        return;
    }
    const int by = blockIdx.y - (a.win_rows > 0 ? a.win_rows : 0);
    const int hb_per_kv = a.groups / G;
    const int kvh = hb_per_kv == 1 ? by : by / hb_per_kv;
    const int bh0 = kvh * a.groups + (by - kvh * hb_per_kv) * G;
    const int ntb_cap = a.T >> 6;   // what the launch was sized for
    int ntb = ntb_cap;              // blocks in use
    if constexpr (EXT) { if (a.t_dev) ntb = min(ntb_cap, __builtin_amdgcn_readfirstlane(*a.t_dev) >> 6); }
    const int tb0 = blockIdx.x * a.tb_per_wg;
    const int tb_end = min(ntb, tb0 + a.tb_per_wg);
    if constexpr (EXT) {
        if (tb0 >= ntb) {   // a workgroup beyond the tokens in use (the cache has not grown into its blocks yet): slabs of weight zero
            const int nsl = a.pair_slabs ? 2 : 1;
            for (int sl = 0; sl < nsl; sl++) {
                const int64_t slab = (int64_t)blockIdx.x * nsl + sl;
                // (every thread stores, some the same bytes: no divergent region around the block loop's EXEC-owning asm statements)
                float* so = a.ws_o + (slab * a.BH + bh0) * kD;
#pragma unroll
                for (int o = 0; o < G * kD; o += kThreads) so[o + (threadIdx.x & (G * kD - 1) & (kThreads - 1))] = 0.f;   // (G = 1: 128 floats, stored twice)
                *reinterpret_cast<float2*>(a.ws_ml + (slab * a.BH + bh0 + (threadIdx.x & (G - 1))) * 2) = make_float2(-INFINITY, 0.f);
            }
            MUSTAFAR_PTRACE_END(7);
            return;
        }
    }
    const int pair = wave >> 1;
    const bool odd = wave & 1;
    const int64_t tiles = (int64_t)(EXT ? a.nb0 : ntb) * kTilesPerTb;
    const uint64_t* kb;
    const uint32_t* ki;
    const unsigned char* kn;
    const uint64_t* vb;
    const uint32_t* vi;
    const unsigned char* vn;
    if (EXT && tb0 >= a.nb0) {
        // a workgroup of an appended extent (its blocks never straddle two: extents are four blocks, workgroups two or four):
        // the extent's arrays, biased so that the block loop's `+ tb * 128` lands inside them.  One more scalar round trip than
        // the base extent's workgroups, whose pointers arrive with the launch.
        const int e = (tb0 - a.nb0) >> 2;
        const mustafar_cache_view ek = a.k_ext[e], ev = a.v_ext[e];
        const int64_t t0 = (int64_t)(a.nb0 + 4 * e) * kTilesPerTb;
        // (uniform_ptr: the entries are the same in every lane, and the block loop hands these pointers to scalar loads -- whatever
        // kind of load the compiler chose for the table)
        kb = uniform_ptr(ek.bmp + (int64_t)kvh * ek.bmp_head_stride - t0);
        ki = uniform_ptr(ek.idx + (int64_t)kvh * ek.idx_head_stride - t0);
        kn = uniform_ptr(static_cast<const unsigned char*>(ek.nz) + 16ull * (uint64_t)kvh * (uint64_t)ek.nz_head_stride);
        vb = uniform_ptr(ev.bmp + (int64_t)kvh * ev.bmp_head_stride - t0);
        vi = uniform_ptr(ev.idx + (int64_t)kvh * ev.idx_head_stride - t0);
        vn = uniform_ptr(static_cast<const unsigned char*>(ev.nz) + 16ull * (uint64_t)kvh * (uint64_t)ev.nz_head_stride);
    } else {
        kb = k_bmp + (int64_t)kvh * (k_bmp_stride ? k_bmp_stride : tiles);
        ki = k_idx + (int64_t)kvh * (k_idx_stride ? k_idx_stride : tiles + 1);
        kn = k_nz + 16ull * (k_nz_stride ? (uint64_t)kvh * k_nz_stride : (uint64_t)k_nz_off[kvh]);
        vb = v_bmp + (int64_t)kvh * (v_bmp_stride ? v_bmp_stride : tiles);
        vi = v_idx + (int64_t)kvh * (v_idx_stride ? v_idx_stride : tiles + 1);
        vn = v_nz + 16ull * (v_nz_stride ? (uint64_t)kvh * v_nz_stride : (uint64_t)v_nz_off[kvh]);
    }
    const h16* qb = a.q + (int64_t)bh0 * kD;
    h16* eb = a.e_rows + (int64_t)by * ntb_cap * (G * 64);
    const h16* mrow = a.mask.ptr ? a.mask.ptr + (int64_t)(bh0 / a.mask.heads) * a.mask.stride : nullptr;
    unsigned char* lds = smem + wave * kStageBytes;
    const uint32_t lds_addr = (uint32_t)reinterpret_cast<uintptr_t>(lds);
    // The softmax step is SPLIT between the two waves of a pair: the even wave finishes heads 0 and 1, the odd wave heads 2 and 3
    // (round 3a: the even wave did all four while the odd wave waited at the barrier).  Each wave leaves the partial scores of the
    // partner's two heads in its OWN stage window (dead from the end of its key phase to the start of its value phase) and reads
    // the partner's after the barrier; the four rescale factors of the pair travel through `alf` (behind windows and tables).
    float* xch_out = reinterpret_cast<float*>(lds);                                           // [2][64] partial scores for the partner
    const float* xch_in = reinterpret_cast<const float*>(smem + (wave ^ 1) * kStageBytes);    // the partner's, for my two heads
    float* alf = reinterpret_cast<float*>(smem + kWaves * kStageBytes + kTabBytes) + pair * G;
    const int h0 = (odd && G >= 2) ? HW : 0;   // my heads: h0 .. h0 + HW - 1 (G = 4: two, G = 2: one; G = 1: the even wave's head 0, the odd wave has none)
    const bool has_heads = G >= 2 || !odd;
    constexpr float kEScaleLog2 = ENG == 2 ? 15.f : 0.f;

    uint32_t ctab_q = 0, ctab_e = 0;
    unsigned char* ptab = nullptr;   // the pair's e table: e stays in LDS, no round trip through the score scratch
    if constexpr (ENG == 1) {
        unsigned char* tab = smem + kWaves * kStageBytes;
        if (threadIdx.x < 64)
            *reinterpret_cast<uint4*>(tab + (threadIdx.x >> 4) * kKeyTabStride + (threadIdx.x & 15) * 16) =
                *reinterpret_cast<const uint4*>(qb + (int64_t)(threadIdx.x >> 4) * kD + (threadIdx.x & 15) * 8);
        __syncthreads();
        ptab = tab + 4 * kKeyTabStride + pair * (4 * kValTabStride);
        ctab_q = (uint32_t)reinterpret_cast<uintptr_t>(tab) + (lane & 3) * kKeyTabStride;
        ctab_e = (uint32_t)reinterpret_cast<uintptr_t>(ptab) + (lane & 3) * kValTabStride;
    }
    float m_run[HW], l_run[HW], acc[G];   // acc: the wave's output half (even: channels 0..63, odd: 64..127), four heads; m_run, l_run: my two heads
#pragma unroll
    for (int h = 0; h < G; h++) acc[h] = 0.f;
#pragma unroll
    for (int j = 0; j < HW; j++) { m_run[j] = -INFINITY; l_run[j] = 0.f; }
    if (MUSTAFAR_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
    for (int t = tb0; t < tb_end; t += kWaves / 2) {   // workgroup-uniform: every wave reaches the barriers below
        const int tb = t + pair;
        const bool active = tb < tb_end;                // (wave-uniform)
        const int tbc = active ? tb : t;                // (an idle pair addresses the other pair's block and computes nothing)
        const uint64_t* kbt = kb + (int64_t)tbc * kTilesPerTb;
        const uint32_t* kit = ki + (int64_t)tbc * kTilesPerTb;
        const uint64_t* vbt = vb + (int64_t)tbc * kTilesPerTb;
        const uint32_t* vit = vi + (int64_t)tbc * kTilesPerTb;
        h16* eblk = eb + (int64_t)tbc * (G * 64);
        float s[G], alpha[G], own[HW];   // own: the partial scores of my heads
#pragma unroll
        for (int j = 0; j < HW; j++) own[j] = 0.f;
#pragma unroll
        for (int h = 0; h < G; h++) { s[h] = 0.f; alpha[h] = 1.f; }
        uint32_t bnd_v = 0, pfv = 0;
        h16 mk = (h16)0.f;
        if (active) {
            const uint32_t bnd_k = bnd_load(kit, lane);
            const uint32_t pfk = odd ? prefetch_meta_all<64, 64>(kbt, kit, lane) : prefetch_meta_all<0, 64>(kbt, kit, lane);   // (the wave's own half)
            if (mrow) mk = mrow[tb * 64 + lane];
            if (odd) lean_block_phase<ENG, kD * 2, false, 2, 2, G>(lds, lds_addr, kbt, kit, kn, qb, bnd_k, lane, s, s MUSTAFAR_PTRACE_ARG, ctab_q);
            else     lean_block_phase<ENG, kD * 2, false, 0, 2, G>(lds, lds_addr, kbt, kit, kn, qb, bnd_k, lane, s, s MUSTAFAR_PTRACE_ARG, ctab_q);
            prefetch_done(pfk);
            // the value side's chunk bounds and metadata lines are requested HERE, a barrier and a softmax step (~2 us) in front of
            // their use: requested at the block's start (~10 us ahead) the lines were often gone from L2 again by the time the
            // scalar loads came for them (c3: 44.4 -> 43.7 us; without any prefetch the launch takes 62 us)
            bnd_v = bnd_load(vit, lane);
            pfv = odd ? prefetch_meta_all<64, 64>(vbt, vit, lane) : prefetch_meta_all<0, 64>(vbt, vit, lane);
            // (static indices under a wave-uniform branch: `s[odd ? 0 : 2]` would move the array to scratch memory)
            if constexpr (G == 4) {
                if (odd) { xch_out[lane] = s[0]; xch_out[64 + lane] = s[1]; own[0] = s[2]; own[1] = s[3]; }
                else     { xch_out[lane] = s[2]; xch_out[64 + lane] = s[3]; own[0] = s[0]; own[1] = s[1]; }
            } else if constexpr (G == 2) {
                if (odd) { xch_out[lane] = s[0]; own[0] = s[1]; }
                else     { xch_out[lane] = s[1]; own[0] = s[0]; }
            } else {
                if (odd) xch_out[lane] = s[0];       // (G = 1: the even wave finishes the head)
                else     own[0] = s[0];
            }
        }
        MUSTAFAR_PTRACE_STAMP(2);
        __syncthreads();
        if (active && has_heads) {
            float al[HW];
#pragma unroll
            for (int j = 0; j < HW; j++) {
                float x = scaled((h16)(own[j] + xch_in[j * 64 + lane]), a.inv_sqrt_d);   // (even + odd in either wave: the same sum)   // fp16 score (SpMM_Kernel.cuh:418), / sqrt(d) in fp16 (model :284)
                if (mrow) x = masked(x, mk);
                const float m_new = uniform_f(fmaxf(m_run[j], wave_max(x)));
                al[j] = uniform_f(__expf(m_run[j] - m_new));   // 0 for the first block (m_run = -inf)
                const h16 e = (h16)__builtin_amdgcn_exp2f((x - m_new) * 1.44269504f + kEScaleLog2);
                if constexpr (ENG == 1) *reinterpret_cast<h16*>(ptab + (h0 + j) * kValTabStride + lane * 2) = e;
                else                    eblk[(h0 + j) * 64 + lane] = e;
                l_run[j] = uniform_f(l_run[j] * al[j] + wave_sum((float)e));
                m_run[j] = m_new;
            }
            if constexpr (HW == 2) { if (lane < 2) alf[h0 + lane] = lane ? al[1] : al[0]; }
            else                   { if (lane < 1) alf[h0] = al[0]; }
            if constexpr (ENG != 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my e stores have reached L2 before the pair's scalar loads
        }
        __syncthreads();
        MUSTAFAR_PTRACE_STAMP(3);
        if (active) {
#pragma unroll
            for (int h = 0; h < G; h++) alpha[h] = alf[h];
#pragma unroll
            for (int h = 0; h < G; h++) acc[h] *= alpha[h];
            // (the partner read my outgoing partial scores before the barrier above; my value phase now rewrites the window)
            if (odd) lean_block_phase<ENG, 64 * 2, true, 2, 2, G>(lds, lds_addr, vbt, vit, vn, eblk, bnd_v, lane, acc, acc MUSTAFAR_PTRACE_ARG, ctab_e);
            else     lean_block_phase<ENG, 64 * 2, true, 0, 2, G>(lds, lds_addr, vbt, vit, vn, eblk, bnd_v, lane, acc, acc MUSTAFAR_PTRACE_ARG, ctab_e);
            prefetch_done(pfv);
        }
        MUSTAFAR_PTRACE_STAMP(5);
        if (MUSTAFAR_PRIO) __builtin_amdgcn_s_setprio(0);   // (the first block is done)
    }
    if (a.pair_slabs) {
        // ---- a slab per pair: each wave stores its output half as it holds it and the (maximum, sum) of its two heads -- no exchange, no
        // barrier; the row kernel folds twice as many slabs (c3: 126).  A pair without a block leaves a slab of weight zero.
        constexpr float kOut = ENG == 2 ? 0x1p-15f : 1.f;   // the e scale leaves here (a power of two: exact)
        const int64_t slab = (int64_t)blockIdx.x * 2 + pair;
        float* so = a.ws_o + (slab * a.BH + bh0) * kD + (odd ? 64 : 0) + lane;
#pragma unroll
        for (int h = 0; h < G; h++) so[h * kD] = acc[h] * kOut;
        if (has_heads && lane < HW)   // (maximum, sum) of my heads
            *reinterpret_cast<float2*>(a.ws_ml + (slab * a.BH + bh0 + h0 + lane) * 2) =
                make_float2(lane ? m_run[HW - 1] : m_run[0], (lane ? l_run[HW - 1] : l_run[0]) * kOut);
        MUSTAFAR_PTRACE_END(7);
        return;
    }
    // ---- merge the two pairs: common maximum, rescaled sums and output halves -> one slab per head
    float* red = reinterpret_cast<float*>(smem);                 // [kWaves][G][64]
    float* s_m = red + kWaves * G * 64;                          // [2 pairs][G]
    float* s_l = s_m + 2 * G;                                    // [2 pairs][G]
    __syncthreads();   // every wave is done with its stage window
    if (has_heads && lane < HW) s_m[pair * G + h0 + lane] = lane ? m_run[HW - 1] : m_run[0];   // (each wave: the maxima of its heads)
    __syncthreads();
#pragma unroll
    for (int h = 0; h < G; h++) {
        const float mw = s_m[pair * G + h];                      // the PAIR's maximum of head h (kept by one of its two waves)
        const float M = fmaxf(s_m[h], s_m[G + h]);
        // a pair without blocks weighs nothing; the e scale leaves here (a power of two: exact)
        const float scale = (mw == -INFINITY) ? 0.f : __expf(mw - M) * (ENG == 2 ? 0x1p-15f : 1.f);
        if (lane == 0 && has_heads && h / HW == (G >= 2 ? (int)odd : 0)) s_l[pair * G + h] = l_run[h % HW] * scale;   // (the wave that finished head h)
        red[(wave * G + h) * 64 + lane] = acc[h] * scale;
    }
    __syncthreads();
    float* slab_o = a.ws_o + ((int64_t)blockIdx.x * a.BH + bh0) * kD;
    for (int o = threadIdx.x; o < 2 * G * 64; o += kThreads) {
        const int hh = o >> 6, l = o & 63;   // hh = half * G + h; waves `half` and `half + 2` hold that half
        const int half = hh / G, h = hh % G;
        slab_o[h * kD + half * 64 + l] = red[(half * G + h) * 64 + l] + red[((half + 2) * G + h) * 64 + l];
    }
    if (threadIdx.x < G) {
        const int h = threadIdx.x;
        float* slab_ml = a.ws_ml + ((int64_t)blockIdx.x * a.BH + bh0 + h) * 2;
        slab_ml[0] = fmaxf(s_m[h], s_m[G + h]);
        slab_ml[1] = s_l[h] + s_l[G + h];
    }
    MUSTAFAR_PTRACE_END(7);
}

// ------------------------------------------------------------------------------------------------ one-pass decode, SUPER-BLOCK pair form (round 5)
// decode_onepass_leanpair_kernel with the code AROUND the two phases cut down (profiles/r05_isa_breakdown.txt: 2.07 of its 8.78 vector
// instructions per tile sat there -- 85 v_readlane / v_writelane of spilled scalars, a softmax step per block and head with twelve DPP
// reductions, 23 v_cndmask selecting the even / odd wave's registers, the address arithmetic of the metadata prefetch):
//   * a pair of waves takes TWO consecutive blocks as one 128-token super-block: key phase A, key phase B, ONE softmax step over the
//     128 tokens (one maximum per head: one DPP reduction instead of two), value phase A, value phase B.  At four blocks per workgroup
//     (every BASELINE shape) a pair runs its loop once: no running maximum, no rescale factors, no exchange of them, two barriers per
//     workgroup instead of four (longer loops run decode_onepass_leanpair_kernel);
//   * the softmax denominator stays per LANE and is summed over the lanes once behind the loop (no DPP reduction in the step);
//   * one body for both waves of a pair: the odd wave's pointers are biased by its 64 tiles / 64 channels once, in front of the loop, so that
//     the phase code (lean_block_phase<.., 0, 2>) and every register choice are the same for both -- the phases are in the kernel's text
//     twice instead of four times and nothing is selected per lane;
//   * the partial scores cross through LDS whole ([block][lane][G] floats in the wave's own stage window, dead between its key and value
//     phases) and every wave reads back the two heads it finishes -- its own contribution and the partner's -- at an address that carries
//     the head offset: no register selection;
//   * bounds + offset-line prefetch are ONE saddr-form load (lane k < 3: bound k; lanes 3, 4: the lines in between), the bitmap-line
//     prefetch another: no vector address arithmetic;
//   * MASK is a template parameter (the unmasked launch carries no mask arithmetic).
// Same grid, slabs and window workgroups as decode_onepass_leanpair_kernel: a drop-in for it (`mustafar_tune(8, 0)` selects the old kernel).
#ifndef MUSTAFAR_SB_PRIO
#define MUSTAFAR_SB_PRIO 3   // s_setprio 1 from the start of a trip to: 1 = the end of its key phases, 2 = the end of its softmax step, 3 = the last chunk of
                             // its value phases (c3, kernel us: 42.6 / 42.3 / 42.0 / 40.8 for 0 / 1 / 2 / 3); 0 = never raised (experiment knob: tools/build_variant.sh)
#endif
template <int W>
struct FVec;
template <> struct FVec<1> { typedef float type; };
template <> struct FVec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct FVec<4> { typedef float type __attribute__((ext_vector_type(4))); };

// m = max over the wave of max(x, floor) (floor: a wave-uniform value in a scalar register), in a scalar register
__device__ __forceinline__ float wave_max_from(float x, float floor_s)
{
    asm("v_max_f32 %0, %1, %0\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(x)
        : "s"(floor_s));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
// a dword at (uniform base) + (per-lane byte offset): global_load_dword v, v_off, s[base] -- no address arithmetic
__device__ __forceinline__ uint32_t ld_at(const void* __restrict__ sbase, uint32_t voff)
{
    return *reinterpret_cast<const uint32_t*>(static_cast<const unsigned char*>(sbase) + voff);
}
#ifndef MUSTAFAR_SB_TRIPLOOP
#define MUSTAFAR_SB_TRIPLOOP 1
#endif
#ifndef MUSTAFAR_SPEC
#define MUSTAFAR_SPEC 0           // (experiment, measured slower: the speculative first-chunk request of DESIGN 4.1 item 25 is compiled in only with -DMUSTAFAR_SPEC=1)
#endif
#ifndef MUSTAFAR_SPEC_LINES
#define MUSTAFAR_SPEC_LINES 16    // (experiment: 128-byte lines of the speculative first-chunk request, a power of two <= 64; MUSTAFAR_SPEC_BACK bytes in front of the prediction)
#define MUSTAFAR_SPEC_BACK 512
#endif
template <int ENG, bool EXT = false, int G = 4, bool MASK = false>   // (a pair walks ONE super-block: launches of more than four blocks per workgroup run
                                                                     // decode_onepass_leanpair_kernel; round 5 carried an uninstantiated online form here -- removed in round 6)
__global__ MUSTAFAR_LP_BOUNDS void decode_onepass_sb_kernel(
    const uint64_t* __restrict__ k_bmp, const unsigned char* __restrict__ k_nz, const uint32_t* __restrict__ k_idx,
    const uint32_t* __restrict__ k_nz_off, const uint64_t* __restrict__ v_bmp, const unsigned char* __restrict__ v_nz,
    const uint32_t* __restrict__ v_idx, const uint32_t* __restrict__ v_nz_off, OneArgs a, int64_t k_bmp_stride,
    int64_t k_idx_stride, uint32_t k_nz_stride, int64_t v_bmp_stride, int64_t v_idx_stride, uint32_t v_nz_stride)
{
    static_assert(G == 4 || ENG == 0, "dot2 and the matrix pipe work on four heads; G < 4 runs v_fma_mix");
    constexpr int HW = G >= 2 ? G / 2 : 1;   // heads a wave of the pair finishes (G = 1: the even wave its one head, the odd wave none)
    constexpr int kTabBytes = ENG == 1 ? 4 * kKeyTabStride + 2 * 2 * 4 * kValTabStride : 0;   // q rows; per pair TWO e tables (blocks A, B)
    __shared__ __attribute__((aligned(16))) unsigned char smem[kWaves * kStageBytes + kTabBytes];
    typedef typename FVec<G>::type fvG;
    typedef typename FVec<HW>::type fvH;
    MUSTAFAR_PTRACE_BEGIN();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wrows = a.win_rows < 0 ? -a.win_rows : a.win_rows;   // window rows lead (win_rows > 0) or trail (< 0) the grid
    const int wy = a.win_rows < 0 ? (int)blockIdx.y - ((int)gridDim.y - wrows) : (int)blockIdx.y;
    if (wy >= 0 && wy < wrows) {   // dense window
        const int task = wy * gridDim.x + blockIdx.x;
        if (task < (int)(gridDim.y - wrows) * a.nchunks) {
            int T_used = -1;
            if constexpr (EXT) { if (a.t_dev) T_used = __builtin_amdgcn_readfirstlane(*a.t_dev); }
            onepass_window_wg<G>(smem, win_args(a, T_used), task, gridDim.x);
        }
        MUSTAFAR_PTRACE_END(5);
        return;
    }
    const int by = blockIdx.y - (a.win_rows > 0 ? a.win_rows : 0);
    const int hb_per_kv = a.groups / G;
    const int kvh = hb_per_kv == 1 ? by : by / hb_per_kv;
    const int bh0 = kvh * a.groups + (by - kvh * hb_per_kv) * G;
    const int ntb_cap = a.T >> 6;   // what the launch was sized for
    int ntb = ntb_cap;              // blocks in use
    if constexpr (EXT) { if (a.t_dev) ntb = min(ntb_cap, __builtin_amdgcn_readfirstlane(*a.t_dev) >> 6); }
    const int tb0 = blockIdx.x * a.tb_per_wg;
    const int tb_end = min(ntb, tb0 + a.tb_per_wg);
    if constexpr (EXT) {
        if (tb0 >= ntb) {   // a workgroup beyond the tokens in use (the cache has not grown into its blocks yet): a slab of weight zero
            float* so = a.ws_o + ((int64_t)blockIdx.x * a.BH + bh0) * kD;
#pragma unroll
            for (int o = 0; o < G * kD; o += kThreads) so[o + (threadIdx.x & (G * kD - 1) & (kThreads - 1))] = 0.f;   // (G = 1: 128 floats, stored twice)
            *reinterpret_cast<float2*>(a.ws_ml + ((int64_t)blockIdx.x * a.BH + bh0 + (threadIdx.x & (G - 1))) * 2) = make_float2(-INFINITY, 0.f);
            MUSTAFAR_PTRACE_END(7);
            return;
        }
    }
    const int pair = wave >> 1;
    const int odd = wave & 1;
    // the pair's blocks: the first pair takes the first half of the workgroup's blocks (rounded up), the second pair the rest: at most two
    // each (the host launches this kernel for tb_per_wg <= 4 only), walked as one super-block
    const int nblk = tb_end - tb0;
    const int nfirst = (nblk + 1) >> 1;
    const int pb0 = tb0 + (pair ? nfirst : 0);
    const int pb_end = pair ? tb_end : tb0 + nfirst;
    const int64_t tiles = (int64_t)(EXT ? a.nb0 : ntb) * kTilesPerTb;
    const uint64_t* kb;
    const uint32_t* ki;
    const unsigned char* kn;
    const uint64_t* vb;
    const uint32_t* vi;
    const unsigned char* vn;
    if (EXT && tb0 >= a.nb0) {
        // a workgroup of an appended extent (its blocks never straddle two: extents are four blocks, workgroups two or four): the extent's
        // arrays, biased so that `+ tb * 128` lands inside them
        const int e = (tb0 - a.nb0) >> 2;
        const mustafar_cache_view ek = a.k_ext[e], ev = a.v_ext[e];
        const int64_t t0 = (int64_t)(a.nb0 + 4 * e) * kTilesPerTb;
        kb = uniform_ptr(ek.bmp + (int64_t)kvh * ek.bmp_head_stride - t0);
        ki = uniform_ptr(ek.idx + (int64_t)kvh * ek.idx_head_stride - t0);
        kn = uniform_ptr(static_cast<const unsigned char*>(ek.nz) + 16ull * (uint64_t)kvh * (uint64_t)ek.nz_head_stride);
        vb = uniform_ptr(ev.bmp + (int64_t)kvh * ev.bmp_head_stride - t0);
        vi = uniform_ptr(ev.idx + (int64_t)kvh * ev.idx_head_stride - t0);
        vn = uniform_ptr(static_cast<const unsigned char*>(ev.nz) + 16ull * (uint64_t)kvh * (uint64_t)ev.nz_head_stride);
    } else {
        kb = k_bmp + (int64_t)kvh * (k_bmp_stride ? k_bmp_stride : tiles);
        ki = k_idx + (int64_t)kvh * (k_idx_stride ? k_idx_stride : tiles + 1);
        kn = k_nz + 16ull * (k_nz_stride ? (uint64_t)kvh * k_nz_stride : (uint64_t)k_nz_off[kvh]);
        vb = v_bmp + (int64_t)kvh * (v_bmp_stride ? v_bmp_stride : tiles);
        vi = v_idx + (int64_t)kvh * (v_idx_stride ? v_idx_stride : tiles + 1);
        vn = v_nz + 16ull * (v_nz_stride ? (uint64_t)kvh * v_nz_stride : (uint64_t)v_nz_off[kvh]);
    }
    // ---- one body for both waves of a pair: the odd wave's half (tiles 64..127 of every block; channels 64..127 of the q rows) is a bias
    kb += odd * 64;
    ki += odd * 64;
    vb += odd * 64;
    vi += odd * 64;
    const h16* qb = a.q + (int64_t)bh0 * kD + odd * 64;
    h16* eb = a.e_rows + (int64_t)by * ntb_cap * (G * 64);
    const h16* mrow = nullptr;
    if constexpr (MASK) mrow = a.mask.ptr + (int64_t)(bh0 / a.mask.heads) * a.mask.stride;
    unsigned char* lds = smem + wave * kStageBytes;
    const uint32_t lds_addr = (uint32_t)reinterpret_cast<uintptr_t>(lds);
    const int h0 = (odd && G >= 2) ? HW : 0;   // my heads: h0 .. h0 + HW - 1
    const bool has_heads = G >= 2 || !odd;
    // partial scores, whole: [block A | block B][lane][G] floats in my window; I read back my heads' [HW] floats of mine and the partner's
    fvG* xch_out = reinterpret_cast<fvG*>(lds) + lane;                                                       // (+ 64 for block B)
    const float* xch_mine = reinterpret_cast<const float*>(lds) + lane * G + h0;                               // (+ 64 * G for block B)
    const float* xch_part = reinterpret_cast<const float*>(smem + (wave ^ 1) * kStageBytes) + lane * G + h0;
    constexpr float kEScaleLog2 = ENG == 2 ? 15.f : 0.f;
    // per-lane byte offsets of the two metadata loads of a phase (no address arithmetic at the call sites):
    //   bounds + offset lines: lane k < 3 reads idx[32 k] (the wave's three chunk bounds), lanes 3 / 4 the 64-byte lines in between
    //   bitmap lines: 64 tiles x 8 bytes = eight 64-byte lines
    // (lanes 0..7 serve block A, lanes 8..15 -- the same offsets -- block B of the super-block: one register holds both blocks' bounds)
    const uint32_t off_bnd = (lane & 7) < 3 ? (lane & 7) * 128u : ((lane & 7) == 3 ? 64u : 192u);
    const uint32_t off_bmp = (lane & 7) * 64u;
    const bool lanesB = lane >= 8 && lane < 16;

    uint32_t ctab_q = 0, ctab_e = 0;
    unsigned char* ptab = nullptr;   // matrix-pipe engine: the pair's e tables (blocks A, B): e stays in LDS
    if constexpr (ENG == 1) {
        unsigned char* tab = smem + kWaves * kStageBytes;
        if (threadIdx.x < 64)
            *reinterpret_cast<uint4*>(tab + (threadIdx.x >> 4) * kKeyTabStride + (threadIdx.x & 15) * 16) =
                *reinterpret_cast<const uint4*>(a.q + (int64_t)bh0 * kD + (int64_t)(threadIdx.x >> 4) * kD + (threadIdx.x & 15) * 8);
        __syncthreads();
        ptab = tab + 4 * kKeyTabStride + pair * (2 * 4 * kValTabStride);
        ctab_q = (uint32_t)reinterpret_cast<uintptr_t>(tab) + (lane & 3) * kKeyTabStride + odd * 128;
        ctab_e = (uint32_t)reinterpret_cast<uintptr_t>(ptab) + (lane & 3) * kValTabStride;
    }
    float m_run[HW], l_lane[HW], acc[G];   // acc: the wave's output half (even: channels 0..63, odd: 64..127); m_run (uniform), l_lane (per lane): my heads
#pragma unroll
    for (int h = 0; h < G; h++) acc[h] = 0.f;
#pragma unroll
    for (int j = 0; j < HW; j++) { m_run[j] = -INFINITY; l_lane[j] = 0.f; }
    const bool late_round = (int)(blockIdx.y * gridDim.x + blockIdx.x) >= a.hi_prio_from;   // (workgroup-uniform)
    if (late_round) __builtin_amdgcn_s_setprio(3);
    else if (MUSTAFAR_SB_PRIO) __builtin_amdgcn_s_setprio(1);
    asm volatile("; sb_trips_begin");   // (markers for tools/isa_breakdown.py --markers: a comment in the ISA, no instruction)
#if MUSTAFAR_SB_TRIPLOOP
    // ONE trip, written as a loop the compiler may not unroll: a scheduling region boundary around the trip.  It computes the same thing as the bare block;
    // what it changes is where the register allocator reloads its 30-odd spilled scalars (bare block: 14 more v_readlane, several of them inside the steps of
    // the phases: ~1 us at c3, profiles/r06_probes.txt item 6).  Correctness does not hang on it (round 6, item 1: the DOT guard does that job).
#pragma unroll 1
    for (int trip = 0; trip < 1; trip++)
#endif
    {
        const int tA = pb0;
        const bool actA = tA < pb_end, actB = tA + 1 < pb_end;   // (wave-uniform)
        const int tAc = actA ? tA : tb0;                          // (an idle pair addresses the workgroup's first block and computes nothing)
        const int tBc = actB ? tA + 1 : tAc;
        h16* eA = eb + (int64_t)tAc * (G * 64);
        h16* eB = eb + (int64_t)tBc * (G * 64);
        float sA[G], sB[G];
#pragma unroll
        for (int h = 0; h < G; h++) { sA[h] = 0.f; sB[h] = 0.f; }
        uint32_t bndV = 0;   // bounds of the value side: block A in lanes 0..2, block B in lanes 8..10
        uint32_t pfVA = 0;   // (the value of a metadata-line prefetch is never used; it is held to the end of the phase it was made for)
        h16 mkA = (h16)0.f, mkB = (h16)0.f;
        const uint64_t* vbA = vb + (int64_t)tAc * kTilesPerTb;
        const uint32_t* viA = vi + (int64_t)tAc * kTilesPerTb;
        // What a phase needs from memory before its stream -- its three chunk bounds and its metadata lines in L2 -- is requested in
        // front of the LAST chunk of the phase before it (~2-3 us ahead: sooner and the lines are gone from L2 again by the time the
        // scalar loads come for them, later and the phase starts with two dependent round trips)
        if (actA) {
            const uint64_t* kbA = kb + (int64_t)tAc * kTilesPerTb;
            const uint32_t* kiA = ki + (int64_t)tAc * kTilesPerTb;
            uint32_t bndK = ld_at(kiA, off_bnd);
            const uint32_t pfKA = ld_at(kbA, off_bmp);
            uint32_t pfS = 0;
            if (MUSTAFAR_SPEC && a.spec_k_bytes > 0 && k_nz_stride && !(EXT && tb0 >= a.nb0)) {
                // speculative request for the first key chunk (behind the bounds load in program order: the bounds' wait does not cover it): 32 lines of
                // 128 bytes from 1 KiB in front of where block tAc's half would start if every block had the average length, clamped to the head's region
                const int64_t pred = (int64_t)tAc * a.spec_k_bytes + (odd ? a.spec_k_bytes / 2 : 0) - MUSTAFAR_SPEC_BACK + (lane & (MUSTAFAR_SPEC_LINES - 1)) * 128;
                const int64_t last = (int64_t)k_nz_stride * 16 - 4;
                pfS = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(kn + ((uint32_t)(pred < 0 ? 0 : pred > last ? last : pred) & ~3u)));
            }
            if constexpr (MASK) {
                mkA = mrow[tAc * 64 + lane];
                if (actB) mkB = mrow[tBc * 64 + lane];
            }
            auto reqVA = [&]() {   // the value side's bounds (both blocks: they are consumed as one pipeline) and block A's metadata lines
                bndV = ld_at(viA, off_bnd);
                if (actB && lanesB) bndV = ld_at(viA + kTilesPerTb, off_bnd);
                pfVA = ld_at(vbA, off_bmp);
            };
            if (actB) {
                // both blocks as one pipeline of four chunks: B's bounds are needed while A's last chunk is worked on -- they come with A's
                // (four bytes per lane); B's metadata lines are requested in front of A's last chunk
                if (lanesB) bndK = ld_at(kiA + kTilesPerTb, off_bnd);
                uint32_t pfKB = 0;
                auto reqKB = [&]() { pfKB = ld_at(kbA + kTilesPerTb, off_bmp); };
                lean_pair_phase<ENG, kD * 2, false, G>(lds, lds_addr, kbA, kiA, kn, qb, bndK, lane, sA, sB MUSTAFAR_PTRACE_ARG, ctab_q, reqKB, reqVA);
                prefetch_done(pfKA);
                prefetch_done(pfKB);
                if (MUSTAFAR_SPEC) prefetch_done(pfS);
            } else {
                lean_block_phase<ENG, kD * 2, false, 0, 2, G, const void*, 0>(lds, lds_addr, kbA, kiA, kn, qb, bndK, lane, sA, sA MUSTAFAR_PTRACE_ARG, ctab_q, reqVA);
                prefetch_done(pfKA);
                if (MUSTAFAR_SPEC) prefetch_done(pfS);
            }
            fvG oA, oB;
            if constexpr (G == 1) { oA = sA[0]; oB = sB[0]; }
            else {
#pragma unroll
                for (int h = 0; h < G; h++) { oA[h] = sA[h]; oB[h] = sB[h]; }
            }
            xch_out[0] = oA;
            xch_out[64] = oB;
        }
        if (MUSTAFAR_SB_PRIO == 1 && !late_round) __builtin_amdgcn_s_setprio(0);
        MUSTAFAR_PTRACE_STAMP(2);
        __syncthreads();
        if (actA && has_heads) {
            const fvH mineA = *reinterpret_cast<const fvH*>(xch_mine), partA = *reinterpret_cast<const fvH*>(xch_part);
            const fvH mineB = *reinterpret_cast<const fvH*>(xch_mine + 64 * G), partB = *reinterpret_cast<const fvH*>(xch_part + 64 * G);
#pragma unroll
            for (int j = 0; j < HW; j++) {
                float pa, pb;
                if constexpr (HW == 1) { pa = mineA + partA; pb = mineB + partB; }
                else                   { pa = mineA[j] + partA[j]; pb = mineB[j] + partB[j]; }
                float xa = scaled((h16)pa, a.inv_sqrt_d);   // fp16 score (SpMM_Kernel.cuh:418), / sqrt(d) in fp16 (model :284)
                float xb = scaled((h16)pb, a.inv_sqrt_d);
                if constexpr (MASK) { xa = masked(xa, mkA); xb = masked(xb, mkB); }
                if (!actB) xb = -INFINITY;
                const float m_new = wave_max_from(fmaxf(xa, xb), m_run[j]);
                const h16 ea = (h16)__builtin_amdgcn_exp2f((xa - m_new) * 1.44269504f + kEScaleLog2);
                const h16 eb2 = (h16)__builtin_amdgcn_exp2f((xb - m_new) * 1.44269504f + kEScaleLog2);
                if constexpr (ENG == 1) {
                    *reinterpret_cast<h16*>(ptab + (h0 + j) * kValTabStride + lane * 2) = ea;
                    *reinterpret_cast<h16*>(ptab + 4 * kValTabStride + (h0 + j) * kValTabStride + lane * 2) = eb2;
                } else {
                    eA[(h0 + j) * 64 + lane] = ea;
                    if (actB) eB[(h0 + j) * 64 + lane] = eb2;
                }
                l_lane[j] += (float)ea + (float)eb2;
                m_run[j] = m_new;
            }
            if constexpr (ENG != 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my e stores have reached L2 before the pair's scalar loads
        }
        if (MUSTAFAR_SB_PRIO == 2 && !late_round) __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        MUSTAFAR_PTRACE_STAMP(3);
        if (actA) {
            // (the partner read my outgoing partial scores before the barrier above; my value phase now rewrites the window)
            if (actB) {
                uint32_t pfVB = 0;
                auto reqVB = [&]() { pfVB = ld_at(vbA + kTilesPerTb, off_bmp); };
                auto dropPrio = [&]() { if (MUSTAFAR_SB_PRIO == 3 && !late_round) __builtin_amdgcn_s_setprio(0); };   // (in front of the last chunk: the trip is nearly done)
                lean_pair_phase<ENG, 64 * 2, true, G>(lds, lds_addr, vbA, viA, vn, eA, bndV, lane, acc, acc MUSTAFAR_PTRACE_ARG, ctab_e, reqVB, dropPrio);
                prefetch_done(pfVA);
                prefetch_done(pfVB);
            } else {
                lean_block_phase<ENG, 64 * 2, true, 0, 2, G, const void*, 0>(lds, lds_addr, vbA, viA, vn, eA, bndV, lane, acc, acc MUSTAFAR_PTRACE_ARG, ctab_e);
                prefetch_done(pfVA);
            }
        }
        MUSTAFAR_PTRACE_STAMP(5);
    }
    asm volatile("; sb_trips_end");
    // ---- the softmax denominators: summed over the lanes once
    float l_run[HW];
#pragma unroll
    for (int j = 0; j < HW; j++) l_run[j] = wave_sum(l_lane[j]);
    // ---- merge the two pairs: common maximum, rescaled sums and output halves -> one slab per head (as decode_onepass_leanpair_kernel)
    float* red = reinterpret_cast<float*>(smem);                 // [kWaves][G][64]
    float* s_m = red + kWaves * G * 64;                          // [2 pairs][G]
    float* s_l = s_m + 2 * G;                                    // [2 pairs][G]
    __syncthreads();   // every wave is done with its stage window
    if (has_heads && lane < HW) s_m[pair * G + h0 + lane] = lane ? m_run[HW - 1] : m_run[0];   // (each wave: the maxima of its heads)
    __syncthreads();
    // lane h < G works out head h's weight (the four heads side by side: one exp for the wave, not one per head)
    const int hl = lane & (G - 1);
    const float mw = s_m[pair * G + hl];                         // the PAIR's maximum of head hl (kept by one of its two waves)
    const float M = fmaxf(s_m[hl], s_m[G + hl]);
    // a pair without blocks weighs nothing; the e scale leaves here (a power of two: exact)
    const float scale_l = (mw == -INFINITY) ? 0.f : __expf(mw - M) * (ENG == 2 ? 0x1p-15f : 1.f);
    if (has_heads && lane >= h0 && lane < h0 + HW) s_l[pair * G + lane] = (lane == h0 ? l_run[0] : l_run[HW - 1]) * scale_l;   // (the wave that finished the head)
#pragma unroll
    for (int h = 0; h < G; h++)
        red[(wave * G + h) * 64 + lane] = acc[h] * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, scale_l), h));
    __syncthreads();
    float* slab_o = a.ws_o + ((int64_t)blockIdx.x * a.BH + bh0) * kD;
    for (int o = threadIdx.x; o < 2 * G * 64; o += kThreads) {
        const int hh = o >> 6, l = o & 63;   // hh = half * G + h; waves `half` and `half + 2` hold that half
        const int half = hh / G, h = hh % G;
        slab_o[h * kD + half * 64 + l] = red[(half * G + h) * 64 + l] + red[((half + 2) * G + h) * 64 + l];
    }
    if (threadIdx.x < G) {
        const int h = threadIdx.x;
        float* slab_ml = a.ws_ml + ((int64_t)blockIdx.x * a.BH + bh0 + h) * 2;
        slab_ml[0] = fmaxf(s_m[h], s_m[G + h]);
        slab_ml[1] = s_l[h] + s_l[G + h];
    }
    MUSTAFAR_PTRACE_END(7);
}

// ------------------------------------------------------------------------------------------------ one-pass decode, SMALL launches (round 6)
// decode_onepass_sb_kernel for launches of two blocks per workgroup (under 768 workgroups: Llama-3-8B 8k x batch 1, c2) -- the same grid, slabs,
// window workgroups and row kernel.  Such a launch has ~2 waves per SIMD and nothing to hide a latency behind: a wave's life is a chain of
// dependent round trips (wave timeline at batch 1, us: 2.3 to the first key chunk, 2.7 key steps, 0.7 softmax, 0.7 to the first value chunk,
// 3.4 value steps, 0.9 merge; profiles/r06_wave_trace_b1.txt), every step of a phase waiting once for its scalar loads (metadata of the next
// step, coefficients) and once for its gathers.  Here the chain is cut to TWO trips to memory per wave, and no step waits for memory at all:
//   trip 1 (everything whose address does not depend on data, all at once): the wave's 64 key and 64 value bitmaps and offsets as VECTOR
//           loads (lane = tile: 8 + 4 bytes per lane and side), the offsets that close the two halves, the q rows (2 dwords per lane), the mask;
//   trip 2: the four stream chunks of the wave's block -- key 0 / 1 AND value 0 / 1 -- into 64 registers (the launch is compiled for four waves
//           per SIMD: 128 registers), as soon as the offsets have landed;
//   steps : a step's bitmaps, offsets and coefficients come out of those registers with v_readlane (24 + 4 G per step: vector issue is idle
//           in a launch like this), so the only wait of a step is the one for its own LDS gathers;
//   e     : crosses the pair through an LDS table ([G][64] halfs per pair) instead of a store / vmcnt(0) / scalar-load round trip through L2.
// One block per pair (a.tb_per_wg == 2), G = 4 on dot2 or v_fma_mix, G = 2 / 1 on v_fma_mix.  mustafar_tune(11, 0) selects the super-block kernel.
template <int L>
__device__ __forceinline__ uint32_t rl_at(uint32_t v)   // v_readlane with a constant lane, pinned where it is written (asm volatile keeps the order
{                                                       // of these statements and of the asm helpers that consume the scalars)
    uint32_t r;
    asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(r) : "v"(v), "i"(L));
    return r;
}
// bitmaps / offsets of the 8 tiles [T0, T0 + 8) of the wave's 64 (lane = tile) -> the scalar image the gather helpers take
template <int T0>
__device__ __forceinline__ void metab_from_lanes(MetaB& m, uint32_t bm_lo, uint32_t bm_hi, uint32_t ix)
{
#define MUSTAFAR_ML(j)                          \
    m.bm[2 * j] = rl_at<T0 + j>(bm_lo);         \
    m.bm[2 * j + 1] = rl_at<T0 + j>(bm_hi);     \
    m.ix[j] = rl_at<T0 + j>(ix);
    MUSTAFAR_ML(0) MUSTAFAR_ML(1) MUSTAFAR_ML(2) MUSTAFAR_ML(3) MUSTAFAR_ML(4) MUSTAFAR_ML(5) MUSTAFAR_ML(6) MUSTAFAR_ML(7)
#undef MUSTAFAR_ML
}
// coefficients of step D0 / 4 (dwords [D0, D0 + 4) of every head's 32): cv[r] lane (h & 1) * 32 + d holds dword d of head 2 r + (h & 1)
template <int G, int D0>
__device__ __forceinline__ void coef_from_lanes(u32x4 (&c)[G], const uint32_t (&cv)[(G + 1) / 2])
{
#define MUSTAFAR_CL(h, r, base)          \
    c[h][0] = rl_at<base + D0>(cv[r]);   \
    c[h][1] = rl_at<base + D0 + 1>(cv[r]); \
    c[h][2] = rl_at<base + D0 + 2>(cv[r]); \
    c[h][3] = rl_at<base + D0 + 3>(cv[r]);
    MUSTAFAR_CL(0, 0, 0)
    if constexpr (G >= 2) { MUSTAFAR_CL(1, 0, 32) }
    if constexpr (G == 4) { MUSTAFAR_CL(2, 1, 0) MUSTAFAR_CL(3, 1, 32) }
#undef MUSTAFAR_CL
    // a vector instruction may read a scalar register a vector instruction wrote only 2 wait states later (gfx940+); the consumers sit in asm
    // statements the compiler does not look into
    asm volatile("s_nop 1");
}
template <int ENG, int G, int T0>   // the 8 tiles [T0, T0 + 8) of the wave's 64 against coefficient dwords [T0 / 2, T0 / 2 + 4) of every head
__device__ __forceinline__ void small_step(uint32_t adj, uint32_t bm_lo, uint32_t bm_hi, uint32_t ix, const uint32_t (&cv)[(G + 1) / 2], float (&acc)[G])
{
    MetaB m;
    u32x4 c[G];
    metab_from_lanes<T0>(m, bm_lo, bm_hi, ix);
    coef_from_lanes<G, T0 / 2>(c, cv);
    if constexpr (ENG == 2) {
        Gathered2 g;
        gather8_d2(m, adj, g);
        gather2_wait(g, c);
        fma8_d2(c, g, acc);
    } else {
        Gathered g;
        gather8(m, adj, g);
        gather_wait<G>(g, c);
        fma8<G>(c, g, acc);
    }
}
// the wave's 64 tiles of one side of its block: chunk 0 (tiles 0..31, stream offsets [i0, i1)) and chunk 1 (tiles 32..63, [i1, ...)), both
// already in registers
template <int ENG, int G>
__device__ __forceinline__ void small_phase(unsigned char* lds, uint32_t lds_addr, int lane, const Stage& s0, const Stage& s1, uint32_t i0, uint32_t i1,
                                            uint32_t bm_lo, uint32_t bm_hi, uint32_t ix, const uint32_t (&cv)[(G + 1) / 2], float (&acc)[G])
{
    stage_commit(lds, s0, lane, 4096u);
    __builtin_amdgcn_wave_barrier();
    const uint32_t adj0 = __builtin_amdgcn_readfirstlane(lds_addr - 4u * i0);
    small_step<ENG, G, 0>(adj0, bm_lo, bm_hi, ix, cv, acc);
    small_step<ENG, G, 8>(adj0, bm_lo, bm_hi, ix, cv, acc);
    small_step<ENG, G, 16>(adj0, bm_lo, bm_hi, ix, cv, acc);
    small_step<ENG, G, 24>(adj0, bm_lo, bm_hi, ix, cv, acc);
    __builtin_amdgcn_wave_barrier();
    stage_commit(lds, s1, lane, 4096u);
    __builtin_amdgcn_wave_barrier();
    const uint32_t adj1 = __builtin_amdgcn_readfirstlane(lds_addr - 4u * i1);
    small_step<ENG, G, 32>(adj1, bm_lo, bm_hi, ix, cv, acc);
    small_step<ENG, G, 40>(adj1, bm_lo, bm_hi, ix, cv, acc);
    small_step<ENG, G, 48>(adj1, bm_lo, bm_hi, ix, cv, acc);
    small_step<ENG, G, 56>(adj1, bm_lo, bm_hi, ix, cv, acc);
    __builtin_amdgcn_wave_barrier();
}
template <int ENG, bool EXT = false, int G = 4, bool MASK = false>
__global__ __launch_bounds__(kThreads, 4) void decode_onepass_small_kernel(
    const uint64_t* __restrict__ k_bmp, const unsigned char* __restrict__ k_nz, const uint32_t* __restrict__ k_idx,
    const uint32_t* __restrict__ k_nz_off, const uint64_t* __restrict__ v_bmp, const unsigned char* __restrict__ v_nz,
    const uint32_t* __restrict__ v_idx, const uint32_t* __restrict__ v_nz_off, OneArgs a, int64_t k_bmp_stride,
    int64_t k_idx_stride, uint32_t k_nz_stride, int64_t v_bmp_stride, int64_t v_idx_stride, uint32_t v_nz_stride)
{
    static_assert(ENG == 0 || (ENG == 2 && G == 4), "v_fma_mix for every group count, dot2 for four heads (the matrix-pipe engine keeps the super-block kernel)");
    constexpr int HW = G >= 2 ? G / 2 : 1;   // heads a wave of the pair finishes (G = 1: the even wave its one head, the odd wave none)
    constexpr int NR = (G + 1) / 2;          // coefficient registers: two heads of 32 dwords each per register
    constexpr int kETab = G * 64 * 2;        // bytes of a pair's e table
    __shared__ __attribute__((aligned(16))) unsigned char smem[kWaves * kStageBytes + 2 * kETab];
    typedef typename FVec<G>::type fvG;
    typedef typename FVec<HW>::type fvH;
    MUSTAFAR_PTRACE_BEGIN();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wrows = a.win_rows < 0 ? -a.win_rows : a.win_rows;
    const int wy = a.win_rows < 0 ? (int)blockIdx.y - ((int)gridDim.y - wrows) : (int)blockIdx.y;
    if (wy >= 0 && wy < wrows) {   // dense window (as decode_onepass_sb_kernel)
        const int task = wy * gridDim.x + blockIdx.x;
        if (task < (int)(gridDim.y - wrows) * a.nchunks) {
            int T_used = -1;
            if constexpr (EXT) { if (a.t_dev) T_used = __builtin_amdgcn_readfirstlane(*a.t_dev); }
            onepass_window_wg<G>(smem, win_args(a, T_used), task, gridDim.x);
        }
        MUSTAFAR_PTRACE_END(5);
        return;
    }
    const int by = blockIdx.y - (a.win_rows > 0 ? a.win_rows : 0);
    const int hb_per_kv = a.groups / G;
    const int kvh = hb_per_kv == 1 ? by : by / hb_per_kv;
    const int bh0 = kvh * a.groups + (by - kvh * hb_per_kv) * G;
    const int ntb_cap = a.T >> 6;
    int ntb = ntb_cap;
    if constexpr (EXT) { if (a.t_dev) ntb = min(ntb_cap, __builtin_amdgcn_readfirstlane(*a.t_dev) >> 6); }
    const int tb0 = blockIdx.x * 2;   // (the host launches this kernel for two blocks per workgroup only)
    if constexpr (EXT) {
        if (tb0 >= ntb) {   // a workgroup beyond the tokens in use: a slab of weight zero (as decode_onepass_sb_kernel)
            float* so = a.ws_o + ((int64_t)blockIdx.x * a.BH + bh0) * kD;
#pragma unroll
            for (int o = 0; o < G * kD; o += kThreads) so[o + (threadIdx.x & (G * kD - 1) & (kThreads - 1))] = 0.f;
            *reinterpret_cast<float2*>(a.ws_ml + ((int64_t)blockIdx.x * a.BH + bh0 + (threadIdx.x & (G - 1))) * 2) = make_float2(-INFINITY, 0.f);
            return;
        }
    }
    const int pair = wave >> 1;
    const int odd = wave & 1;
    const int t = tb0 + pair;            // the pair's block
    const bool act = t < ntb;            // (wave-uniform; an idle pair addresses the workgroup's first block and computes nothing)
    const int tc = act ? t : tb0;
    const int64_t tiles = (int64_t)(EXT ? a.nb0 : ntb) * kTilesPerTb;
    const uint64_t* kb;
    const uint32_t* ki;
    const unsigned char* kn;
    const uint64_t* vb;
    const uint32_t* vi;
    const unsigned char* vn;
    if (EXT && tb0 >= a.nb0) {
        const int e = (tb0 - a.nb0) >> 2;
        const mustafar_cache_view ek = a.k_ext[e], ev = a.v_ext[e];
        const int64_t t0 = (int64_t)(a.nb0 + 4 * e) * kTilesPerTb;
        kb = uniform_ptr(ek.bmp + (int64_t)kvh * ek.bmp_head_stride - t0);
        ki = uniform_ptr(ek.idx + (int64_t)kvh * ek.idx_head_stride - t0);
        kn = uniform_ptr(static_cast<const unsigned char*>(ek.nz) + 16ull * (uint64_t)kvh * (uint64_t)ek.nz_head_stride);
        vb = uniform_ptr(ev.bmp + (int64_t)kvh * ev.bmp_head_stride - t0);
        vi = uniform_ptr(ev.idx + (int64_t)kvh * ev.idx_head_stride - t0);
        vn = uniform_ptr(static_cast<const unsigned char*>(ev.nz) + 16ull * (uint64_t)kvh * (uint64_t)ev.nz_head_stride);
    } else {
        kb = k_bmp + (int64_t)kvh * (k_bmp_stride ? k_bmp_stride : tiles);
        ki = k_idx + (int64_t)kvh * (k_idx_stride ? k_idx_stride : tiles + 1);
        kn = k_nz + 16ull * (k_nz_stride ? (uint64_t)kvh * k_nz_stride : (uint64_t)k_nz_off[kvh]);
        vb = v_bmp + (int64_t)kvh * (v_bmp_stride ? v_bmp_stride : tiles);
        vi = v_idx + (int64_t)kvh * (v_idx_stride ? v_idx_stride : tiles + 1);
        vn = v_nz + 16ull * (v_nz_stride ? (uint64_t)kvh * v_nz_stride : (uint64_t)v_nz_off[kvh]);
    }
    // the wave's 64 tiles of either side: tiles odd * 64 ... + 63 of the block
    const uint64_t* kbT = kb + (int64_t)tc * kTilesPerTb + odd * 64;
    const uint32_t* kiT = ki + (int64_t)tc * kTilesPerTb + odd * 64;
    const uint64_t* vbT = vb + (int64_t)tc * kTilesPerTb + odd * 64;
    const uint32_t* viT = vi + (int64_t)tc * kTilesPerTb + odd * 64;
    unsigned char* lds = smem + wave * kStageBytes;
    const uint32_t lds_addr = (uint32_t)reinterpret_cast<uintptr_t>(lds);
    h16* etab = reinterpret_cast<h16*>(smem + kWaves * kStageBytes + pair * kETab);   // [G][64] halfs
    const int h0 = (odd && G >= 2) ? HW : 0;
    const bool has_heads = G >= 2 || !odd;
    fvG* xch_out = reinterpret_cast<fvG*>(lds) + lane;
    const float* xch_mine = reinterpret_cast<const float*>(lds) + lane * G + h0;
    const float* xch_part = reinterpret_cast<const float*>(smem + (wave ^ 1) * kStageBytes) + lane * G + h0;
    constexpr float kEScaleLog2 = ENG == 2 ? 15.f : 0.f;

    // ---- trip 1: offsets first (the streams hang on them), then bitmaps, q rows, mask
    const uint32_t kix = kiT[lane], kie = kiT[64], vix = viT[lane], vie = viT[64];
    const uint64_t kbm = kbT[lane], vbm = vbT[lane];
    uint32_t qv[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const int h = 2 * r + (lane >> 5);
        qv[r] = 0u;
        if (h < G) qv[r] = reinterpret_cast<const uint32_t*>(a.q + (int64_t)(bh0 + h) * kD + odd * 64)[lane & 31];
    }
    h16 mk = (h16)0.f;
    if constexpr (MASK) mk = (a.mask.ptr + (int64_t)(bh0 / a.mask.heads) * a.mask.stride)[tc * 64 + lane];
    // ---- trip 2: the four chunks
    const uint32_t k0 = rl_at<0>(kix), k1 = rl_at<32>(kix), k2 = __builtin_amdgcn_readfirstlane(kie);
    const Stage sk0 = stage_issue(kn + 4ull * k0, 4u * (k1 - k0), lane);
    const Stage sk1 = stage_issue(kn + 4ull * k1, 4u * (k2 - k1), lane);
    const uint32_t v0 = rl_at<0>(vix), v1 = rl_at<32>(vix), v2 = __builtin_amdgcn_readfirstlane(vie);
    const Stage sv0 = stage_issue(vn + 4ull * v0, 4u * (v1 - v0), lane);
    const Stage sv1 = stage_issue(vn + 4ull * v1, 4u * (v2 - v1), lane);
    // Everything above stays above: the phases below are straight-line code in THIS basic block (an idle pair walks the workgroup's first block
    // and weighs nothing at the merge) -- with a branch around a phase the compiler sinks that phase's loads into it, next to their first use,
    // and the wave is back to one round trip per phase.
    __builtin_amdgcn_sched_barrier(0);

    MUSTAFAR_PTRACE_STAMP(1);   // (trace builds: requests issued; the phases' own stamps follow)
    float sA[G], acc[G];
#pragma unroll
    for (int h = 0; h < G; h++) { sA[h] = 0.f; acc[h] = 0.f; }
    float m_run[HW], l_lane[HW];
#pragma unroll
    for (int j = 0; j < HW; j++) { m_run[j] = -INFINITY; l_lane[j] = 0.f; }
    // ---- key phase: partial scores of the wave's 64 channels, lane = token
    {
        small_phase<ENG, G>(lds, lds_addr, lane, sk0, sk1, k0, k1, (uint32_t)kbm, (uint32_t)(kbm >> 32), kix, qv, sA);
        fvG o;
        if constexpr (G == 1) o = sA[0];
        else {
#pragma unroll
            for (int h = 0; h < G; h++) o[h] = sA[h];
        }
        xch_out[0] = o;
    }
    MUSTAFAR_PTRACE_STAMP(2);
    __syncthreads();
    // ---- softmax step over the block's 64 tokens: each wave its heads; e -> the pair's table
    if (has_heads) {
        const fvH mine = *reinterpret_cast<const fvH*>(xch_mine), part = *reinterpret_cast<const fvH*>(xch_part);
#pragma unroll
        for (int j = 0; j < HW; j++) {
            float p;
            if constexpr (HW == 1) p = mine + part;
            else                   p = mine[j] + part[j];
            float x = scaled((h16)p, a.inv_sqrt_d);   // fp16 score (SpMM_Kernel.cuh:418), / sqrt(d) in fp16 (model :284)
            if constexpr (MASK) x = masked(x, mk);
            const float m_new = wave_max_from(x, m_run[j]);
            const h16 e = (h16)__builtin_amdgcn_exp2f((x - m_new) * 1.44269504f + kEScaleLog2);
            etab[(h0 + j) * 64 + lane] = e;
            l_lane[j] += (float)e;
            m_run[j] = m_new;
        }
    }
    __syncthreads();   // (also: the partner has read my outgoing partial scores; my value phase now rewrites the window)
    MUSTAFAR_PTRACE_STAMP(3);
    // ---- value phase: lane = channel of the wave's half
    {
        uint32_t ev[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int h = 2 * r + (lane >> 5);
            ev[r] = reinterpret_cast<const uint32_t*>(etab)[(h < G ? h : 0) * 32 + (lane & 31)];
        }
        small_phase<ENG, G>(lds, lds_addr, lane, sv0, sv1, v0, v1, (uint32_t)vbm, (uint32_t)(vbm >> 32), vix, ev, acc);
    }
    MUSTAFAR_PTRACE_STAMP(5);
    // ---- the softmax denominators, then the merge of the two pairs into one slab per head (as decode_onepass_sb_kernel)
    float l_run[HW];
#pragma unroll
    for (int j = 0; j < HW; j++) {
        l_run[j] = wave_sum(l_lane[j]);
        if (!act) m_run[j] = -INFINITY;   // an idle pair (a workgroup's second block beyond the cache) weighs nothing
    }
    float* red = reinterpret_cast<float*>(smem);                 // [kWaves][G][64]
    float* s_m = red + kWaves * G * 64;                          // [2 pairs][G]
    float* s_l = s_m + 2 * G;                                    // [2 pairs][G]
    __syncthreads();   // every wave is done with its stage window
    if (has_heads && lane < HW) s_m[pair * G + h0 + lane] = lane ? m_run[HW - 1] : m_run[0];
    __syncthreads();
    const int hl = lane & (G - 1);
    const float mw = s_m[pair * G + hl];
    const float M = fmaxf(s_m[hl], s_m[G + hl]);
    const float scale_l = (mw == -INFINITY) ? 0.f : __expf(mw - M) * (ENG == 2 ? 0x1p-15f : 1.f);
    if (has_heads && lane >= h0 && lane < h0 + HW) s_l[pair * G + lane] = (lane == h0 ? l_run[0] : l_run[HW - 1]) * scale_l;
#pragma unroll
    for (int h = 0; h < G; h++)
        red[(wave * G + h) * 64 + lane] = acc[h] * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, scale_l), h));
    __syncthreads();
    float* slab_o = a.ws_o + ((int64_t)blockIdx.x * a.BH + bh0) * kD;
    for (int o = threadIdx.x; o < 2 * G * 64; o += kThreads) {
        const int hh = o >> 6, l = o & 63;
        const int half = hh / G, h = hh % G;
        slab_o[h * kD + half * 64 + l] = red[(half * G + h) * 64 + l] + red[((half + 2) * G + h) * 64 + l];
    }
    if (threadIdx.x < G) {
        const int h = threadIdx.x;
        float* slab_ml = a.ws_ml + ((int64_t)blockIdx.x * a.BH + bh0 + h) * 2;
        slab_ml[0] = fmaxf(s_m[h], s_m[G + h]);
        slab_ml[1] = s_l[h] + s_l[G + h];
    }
    MUSTAFAR_PTRACE_END(7);
}

// ------------------------------------------------------------------------------------------------ key SpMV, lean pair form (round 4)
// The reference entry point Key_SplitK_API (kernel/csrc/SpMM_API.cu:86-139 -> Key_Kernel, SpMM_Kernel.cuh:156-419) on the machinery
// of the one-pass launch's key phase: two waves share a 64-token block (64 channels each, partial scores folded through LDS), every
// address inside the block is one base pointer + an immediate (lean_block_phase), the stream loads are non-temporal, the launch is
// compiled for 8 waves per SIMD.  Exact products only (v_fma_mix, or the matrix pipe for four heads): this is what an unchanged hook
// calls.  N = rows per head of the dense operand: 1, or the hook's 8 (llama_mustafar_kernel.py:273: rows 1..7 are zero padding and
// are written as exact zeros unless a row holds a non-zero, in which case it is computed like row 0).
template <int G, int ENG, int N, bool WIN = false>   // WIN: the launch carries window workgroups (fused two-launch decode); as value_lean_kernel
__global__ MUSTAFAR_LP_BOUNDS void key_lean_kernel(
    const uint64_t* __restrict__ bmp, const unsigned char* __restrict__ nz, const uint32_t* __restrict__ idx,
    const uint32_t* __restrict__ nz_off, const h16* __restrict__ q, h16* __restrict__ out, int T, int groups, int ldc, WinArgs wa,
    int64_t bmp_stride, int64_t idx_stride, uint32_t nz_stride)
{
    static_assert(ENG == 0 || (ENG == 1 && G == 4), "exact products: v_fma_mix, or the matrix pipe for four heads");
    constexpr int kTabBytes = ENG == 1 ? 4 * kKeyTabStride : 0;
    __shared__ __attribute__((aligned(16))) unsigned char smem[kWaves * kStageBytes + kTabBytes];
    MUSTAFAR_TRACE_BEGIN(1);
    MUSTAFAR_PTRACE_BEGIN();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wrows = wa.rows < 0 ? -wa.rows : wa.rows;            // window rows lead (rows > 0) or trail (rows < 0) the grid
    const int wy = wa.rows < 0 ? (int)blockIdx.y - ((int)gridDim.y - wrows) : (int)blockIdx.y;
    if constexpr (WIN) {
        if (wa.rows != 0 && wy >= 0 && wy < wrows) {   // fused decode only (N == 1): window scores
            const int task = wy * gridDim.x + blockIdx.x;
            if (task < (int)(gridDim.y - wrows) * wa.nchunks)
                key_window_wg<G>(smem, q, wa.win, wa.fresh, window_len(wa.w_extra, wa.w_len, wa.w_cap), wa.w_cap, wa.nchunks, out, T, ldc,
                                 groups, task);
            MUSTAFAR_TRACE_END();
            return;
        }
    }
    const int by = blockIdx.y - (WIN && wa.rows > 0 ? wa.rows : 0);
    const int hb_per_kv = groups / G;
    const int kvh = by / hb_per_kv;
    const int bh0 = kvh * groups + (by % hb_per_kv) * G;
    const int ntb = T >> 6;
    const int pair = wave >> 1;
    const bool odd = wave & 1;
    const int tb = blockIdx.x * 2 + pair;
    const bool active = tb < ntb;                                   // (wave-uniform)
    const int tbc = active ? tb : ntb - 1;
    const int64_t tiles = (int64_t)ntb * kTilesPerTb;
    const uint64_t* kbt = bmp + (int64_t)kvh * (bmp_stride ? bmp_stride : tiles) + (int64_t)tbc * kTilesPerTb;
    const uint32_t* kit = idx + (int64_t)kvh * (idx_stride ? idx_stride : tiles + 1) + (int64_t)tbc * kTilesPerTb;
    const unsigned char* kn = nz + 16ull * (nz_stride ? (uint64_t)kvh * nz_stride : (uint64_t)nz_off[kvh]);
    unsigned char* lds = smem + wave * kStageBytes;
    const uint32_t lds_addr = (uint32_t)reinterpret_cast<uintptr_t>(lds);
    float* xch_out = reinterpret_cast<float*>(lds);                                        // odd wave: [G][64] partial scores for the even wave
    const float* xch_in = reinterpret_cast<const float*>(smem + (wave | 1) * kStageBytes);

    uint32_t rows = 1u;   // bit n: row n has to be computed
    if constexpr (N > 1) rows |= pad_row_mask<G>(q, kD, bh0, N, 0, kD, reinterpret_cast<uint32_t*>(smem));
    const int tok0 = blockIdx.x * 128;
    const int ntok = min(128, T - tok0);
    uint32_t bnd = 0;
    if (active) bnd = bnd_load(kit, lane);                          // (the same five chunk bounds for every row)
#pragma unroll 1
    for (int n = 0; n < N; n++) {
        if ((rows >> n) & 1u) {
            const h16* qb = q + ((int64_t)bh0 * N + n) * kD;        // the G rows, N * 256 bytes apart
            uint32_t ctab_q = 0;
            if constexpr (ENG == 1) {   // coefficient table: row n of the 4 heads, 16 bytes per thread
                unsigned char* tab = smem + kWaves * kStageBytes;
                if (n > 0) __syncthreads();
                if (threadIdx.x < 64)
                    *reinterpret_cast<uint4*>(tab + (threadIdx.x >> 4) * kKeyTabStride + (threadIdx.x & 15) * 16) =
                        *reinterpret_cast<const uint4*>(qb + (int64_t)(threadIdx.x >> 4) * N * kD + (threadIdx.x & 15) * 8);
                __syncthreads();
                ctab_q = (uint32_t)reinterpret_cast<uintptr_t>(tab) + (lane & 3) * kKeyTabStride;
            }
            float s[G];
#pragma unroll
            for (int h = 0; h < G; h++) s[h] = 0.f;
            if (active) {
                const uint32_t pf = odd ? prefetch_meta_all<64, 64>(kbt, kit, lane) : prefetch_meta_all<0, 64>(kbt, kit, lane);
                if (odd) lean_block_phase<ENG, N * kD * 2, false, 2, 2, G>(lds, lds_addr, kbt, kit, kn, qb, bnd, lane, s, s MUSTAFAR_PTRACE_ARG, ctab_q);
                else     lean_block_phase<ENG, N * kD * 2, false, 0, 2, G>(lds, lds_addr, kbt, kit, kn, qb, bnd, lane, s, s MUSTAFAR_PTRACE_ARG, ctab_q);
                prefetch_done(pf);
                if (odd) {
#pragma unroll
                    for (int h = 0; h < G; h++) xch_out[h * 64 + lane] = s[h];
                }
            }
            __syncthreads();
            if (active && !odd) {
#pragma unroll
                for (int h = 0; h < G; h++)
                    out[((int64_t)(bh0 + h) * N + n) * ldc + (int64_t)tb * 64 + lane] = (h16)(s[h] + xch_in[h * 64 + lane]);
            }
            if constexpr (N > 1) __syncthreads();   // the exchange areas are stage windows again in the next row
        } else {   // exact zeros, 16 bytes per lane
            const int per_row = ntok / 8;
            const uint4 z = {0u, 0u, 0u, 0u};
            for (int u = threadIdx.x; u < G * per_row; u += kThreads) {
                const int h = u / per_row, k = u % per_row;
                *reinterpret_cast<uint4*>(out + ((int64_t)(bh0 + h) * N + n) * ldc + tok0 + k * 8) = z;
            }
        }
    }
    MUSTAFAR_TRACE_END();
}

// ------------------------------------------------------------------------------------------------ value SpMV, lean pair form (round 4)
// The reference entry point Value_SplitK_API (kernel/csrc/SpMM_API.cu:193-254 -> Value_Kernel, SpMM_Kernel.cuh:421-676) on the
// machinery of the one-pass launch's value phase: two waves share a 64-token block (one 64-channel half of the output each), a
// workgroup (two pairs) walks its token chunk two blocks at a time, every address inside a block is a base pointer + an immediate
// (one pointer per head for the probabilities: their rows are N * ldb halfs apart, a runtime quantity), non-temporal stream loads,
// 8 waves per SIMD, issue priority by progress.  v_fma_mix: exact products.  Slabs, flags and the combine pass as value_spmv_kernel.
#ifndef MUSTAFAR_VL_STRIDE
#define MUSTAFAR_VL_STRIDE 1   // the probabilities' rows through one base pointer + a scalar offset per head (0: a pointer per head, round 4 / 5a)
#endif
#ifndef MUSTAFAR_VL_PFCOEF
#define MUSTAFAR_VL_PFCOEF 1   // the probabilities of a pair's blocks prefetched into L2 at the top of the trip (round 6; 0: not)
#endif
#if MUSTAFAR_VL_STRIDE
#define MUSTAFAR_VL_COEF CoefStride
#else
#define MUSTAFAR_VL_COEF CoefPtrs
#endif
// WIN: the launch carries window workgroups (fused two-launch decode, N == 1).  An instantiation of its own: within this kernel's 64
// vector registers the window path spills to scratch, and a kernel with a private segment -- even one its SpMV workgroups never touch --
// is launched with scratch; the reference entry point (no window) must not pay for that.
template <int G, int N, bool WIN = false>
__global__ MUSTAFAR_LP_BOUNDS void value_lean_kernel(
    const uint64_t* __restrict__ bmp, const unsigned char* __restrict__ nz, const uint32_t* __restrict__ idx,
    const uint32_t* __restrict__ nz_off, const h16* __restrict__ p, h16* __restrict__ out, float* __restrict__ ws,
    uint32_t* __restrict__ flags, int T, int groups, int BH, int tb_per_wg, int direct, int ldb, WinArgs wa,
    int64_t bmp_stride, int64_t idx_stride, uint32_t nz_stride)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[kWaves * kStageBytes];
    static_assert(kWaves * kStageBytes >= kWaves * G * 64 * 4, "reduce buffer must fit in the stage area");
    MUSTAFAR_TRACE_BEGIN(2);
    MUSTAFAR_PTRACE_BEGIN();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wrows = wa.rows < 0 ? -wa.rows : wa.rows;            // window rows lead (rows > 0) or trail (rows < 0) the grid
    const int wy = wa.rows < 0 ? (int)blockIdx.y - ((int)gridDim.y - wrows) : (int)blockIdx.y;
    if constexpr (WIN) {
        if (wa.rows != 0 && wy >= 0 && wy < wrows) {   // fused decode only (N == 1): window p.V -> slabs gridDim.x ..
            const int task = wy * gridDim.x + blockIdx.x;
            if (task < (int)(gridDim.y - wrows) * wa.nchunks)
                value_window_wg<G, kWaves>(smem, p, wa.win, wa.fresh, window_len(wa.w_extra, wa.w_len, wa.w_cap), wa.w_cap, wa.nchunks, ws,
                                           (int64_t)BH * kD, gridDim.x, T, ldb, groups, task);
            MUSTAFAR_TRACE_END();
            return;
        }
    }
    // N > 1 (round 6): the pad rows have workgroups of their OWN.  Grid rows [0, gy) -- dispatched first -- compute row 0 and never look at the
    // pad rows: they are the N = 1 launch.  The rows behind them are pad workgroups: one per head group and kPadGroup consecutive token chunks;
    // it reads its slice of the G x (N - 1) pad rows (2 KiB runs, every load of a thread in flight at once), publishes the row masks of its
    // chunks' slabs and computes, chunk by chunk, the rows that hold a non-zero.  With the hook's zero pads (model :313) it reads ~14 vectors per
    // thread and leaves -- in the tail of the launch, where the chip has wave slots to spare.  (Rounds 4-5: every workgroup read its pad slice in
    // FRONT of row 0 -- three barriers and a trip to memory before its first stream request: c3, N = 8, 48.4 us per call against 26 for N = 1.)
    constexpr int kPadGroup = MUSTAFAR_PAD_GROUP;
    const int hb_per_kv = groups / G;
    const int gy_main = N > 1 ? (BH / groups) * hb_per_kv : (int)gridDim.y;
    const bool pad_wg = N > 1 && (int)blockIdx.y >= gy_main;
    int by = (int)blockIdx.y - (WIN && wa.rows > 0 ? wa.rows : 0), slab0 = (int)blockIdx.x, nck = 1;
    if constexpr (N > 1) {
        if (pad_wg) {
            const int S = (int)gridDim.x, S4 = (S + kPadGroup - 1) / kPadGroup;
            const int j = ((int)blockIdx.y - gy_main) * S + (int)blockIdx.x;
            if (j >= gy_main * S4) { MUSTAFAR_TRACE_END(); return; }   // (the pad rows of the grid are rounded up to whole rows)
            by = j / S4;
            slab0 = (j - by * S4) * kPadGroup;
            nck = min(kPadGroup, S - slab0);
        }
    }
    const int kvh = by / hb_per_kv;
    const int bh0 = kvh * groups + (by % hb_per_kv) * G;
    const int ntb = T >> 6;
    const int64_t tiles = (int64_t)ntb * kTilesPerTb;
    const int pair = wave >> 1;
    const bool odd = wave & 1;
    const uint64_t* vb = bmp + (int64_t)kvh * (bmp_stride ? bmp_stride : tiles);
    const uint32_t* vi = idx + (int64_t)kvh * (idx_stride ? idx_stride : tiles + 1);
    const unsigned char* vn = nz + 16ull * (nz_stride ? (uint64_t)kvh * nz_stride : (uint64_t)nz_off[kvh]);
    unsigned char* lds = smem + wave * kStageBytes;
    const uint32_t lds_addr = (uint32_t)reinterpret_cast<uintptr_t>(lds);
    float* red = reinterpret_cast<float*>(smem);   // [kWaves][G][64], overlays the stage windows

    // rows [n_lo, n_hi) of the token chunk `slab` (rows: which of them hold a non-zero; the others are written as zeros in a direct launch, skipped otherwise)
    auto run_chunk = [&](const int slab, const uint32_t rows, const int n_lo, const int n_hi) {
    const int tb0 = slab * tb_per_wg;
    const int tb_end = min(ntb, tb0 + tb_per_wg);
    float* ws_slab = ws + (int64_t)slab * BH * N * kD;
#pragma unroll 1
    for (int n = n_lo; n < n_hi; n++) {
        const bool live = (rows >> n) & 1u;
        if (!live && !direct) continue;   // the combine pass skips this row of this slab
        float acc[G];
#pragma unroll
        for (int h = 0; h < G; h++) acc[h] = 0.f;
        if (live) {
            if (MUSTAFAR_PRIO) __builtin_amdgcn_s_setprio(1);
            // round 5: a pair takes a contiguous half of the workgroup's blocks and walks it TWO blocks at a time as one pipeline of four
            // chunks (lean_pair_phase: the next block's first chunk is in flight while this block's last one is worked on -- what the
            // round-1 kernel's cross-block prefetch bought it over round 4's lean form); one body for both waves (biased pointers)
            const int nblk = tb_end - tb0, nfirst = (nblk + 1) >> 1;
            const int pb0 = tb0 + (pair ? nfirst : 0), pb_end = pair ? tb_end : tb0 + nfirst;
            const uint64_t* vbo = vb + (odd ? 64 : 0);
            const uint32_t* vio = vi + (odd ? 64 : 0);
            const uint32_t off_bnd = (lane & 7) < 3 ? (lane & 7) * 128u : ((lane & 7) == 3 ? 64u : 192u);
            const uint32_t off_bmp = (lane & 7) * 64u;
            const bool lanesB = lane >= 8 && lane < 16;
            const uint32_t head_bytes = (uint32_t)N * (uint32_t)ldb * 2u;   // (rows of consecutive heads; < 2^32: N * ldb < 2^31 halfs is checked by the launcher's T limit)
            (void)head_bytes;
#pragma unroll 1
            for (int tb = pb0; tb < pb_end; tb += 2) {   // (wave-uniform; no barrier inside the loop: the pairs run freely)
                const uint64_t* vbt = vbo + (int64_t)tb * kTilesPerTb;
                const uint32_t* vit = vio + (int64_t)tb * kTilesPerTb;
                const bool two = tb + 1 < pb_end;
                uint32_t bnd = ld_at(vit, off_bnd);
                if (two && lanesB) bnd = ld_at(vit + kTilesPerTb, off_bnd);
                const uint32_t pfA = ld_at(vbt, off_bmp);
                MUSTAFAR_VL_COEF<G> cb;
#if MUSTAFAR_VL_STRIDE
                cb.base = p + ((int64_t)bh0 * N + n) * ldb + (int64_t)tb * 64;
#pragma unroll
                for (int h = 1; h < G; h++) cb.off[h - 1] = (uint32_t)h * head_bytes;
#else
#pragma unroll
                for (int h = 0; h < G; h++) cb.p[h] = p + ((int64_t)(bh0 + h) * N + n) * ldb + (int64_t)tb * 64;
#endif
#if MUSTAFAR_VL_STRIDE && MUSTAFAR_VL_PFCOEF
                // round 6: the probabilities of these (<= 2) blocks, 128 bytes per block and head, asked into L2 by a vector load NOW -- the steps read them with
                // scalar loads inside their one wait, and a row the softmax kernel wrote a moment ago (or one 127 KiB from its neighbour: the hook's 8-row
                // operand) is not in L2 when they do: us per call at c3, same box, without / with: 8 rows 34.0-34.3 / 28.0-28.2, one row 26.1-26.9 / 24.9-25.0; c4 51.0 / 49.6 and 45.0 / 41.4-42.9 (profiles/r06_probes.txt item 9).  Lanes 0..2G-1: (head, block); behind the
                // bounds and the bitmap lines in program order, so no wait of theirs covers it.
                const uint32_t pfP = ld_at(cb.base, (uint32_t)(((lane & (2 * G - 1)) >> 1) * head_bytes + ((lane & 1) && two ? 128u : 0u)));
#endif
                if (two) {
                    uint32_t pfB = 0;
                    auto reqB = [&]() { pfB = ld_at(vbt + kTilesPerTb, off_bmp); };
                    lean_pair_phase<0, 0, true, G, decltype(reqB), NoMid, MUSTAFAR_VL_COEF<G>, 64 * 2>(lds, lds_addr, vbt, vit, vn, cb, bnd, lane, acc, acc MUSTAFAR_PTRACE_ARG, 0u,
                                                                                                           reqB, NoMid());
                    prefetch_done(pfB);
                } else {
                    lean_block_phase<0, 0, true, 0, 2, G, MUSTAFAR_VL_COEF<G>>(lds, lds_addr, vbt, vit, vn, cb, bnd, lane, acc, acc MUSTAFAR_PTRACE_ARG);
                }
                prefetch_done(pfA);
#if MUSTAFAR_VL_STRIDE && MUSTAFAR_VL_PFCOEF
                prefetch_done(pfP);
#endif
                if (MUSTAFAR_PRIO) __builtin_amdgcn_s_setprio(0);   // (the first blocks are done)
            }
        }
        __syncthreads();   // every wave is done with its stage window (and with the previous row's sums)
#pragma unroll
        for (int h = 0; h < G; h++) red[(wave * G + h) * 64 + lane] = acc[h];
        __syncthreads();
        for (int o = threadIdx.x; o < 2 * G * 64; o += kThreads) {
            const int hh = o >> 6, l = o & 63;   // hh = half * G + h; waves `half` and `half + 2` hold that half
            const int half = hh / G, h = hh % G;
            const float sum = red[(half * G + h) * 64 + l] + red[((half + 2) * G + h) * 64 + l];
            const int64_t row = (int64_t)(bh0 + h) * N + n;
            if (direct) out[row * kD + half * 64 + l] = (h16)sum;
            else        ws_slab[row * kD + half * 64 + l] = sum;
        }
        if constexpr (N > 1) __syncthreads();
    }
    };

    if constexpr (N > 1) {
        // ONE inlined copy of run_chunk for both kinds of workgroup (two copies: 188 scalar-spill reloads against 37 in the N = 1 instantiation, 7 us at c3)
        uint32_t all = 1u;   // row-0 workgroup: chunk 0 of its "group" = its own slab, row 0
        int n_lo = 0, n_hi = 1;
        if (pad_wg) {
            const int cw = tb_per_wg * 64;                              // tokens (columns of the dense operand) per chunk
            const int col0 = slab0 * cw, ncols = min(T, col0 + nck * cw) - col0;
#ifdef MUSTAFAR_PROBE_NOPADREAD   // (timing probe: the pad workgroups are dispatched and publish "nothing live" without reading anything -- wrong for non-zero pads)
            all = 0u;
            (void)col0; (void)ncols;
#else
            all = (uint32_t)__builtin_amdgcn_readfirstlane(
                (int)pad_group_mask<G, N>(p, ldb, bh0, col0, ncols, cw, reinterpret_cast<uint32_t*>(smem)));   // bits 8 c + n: pad row n over chunk c
#endif
            if (!direct && (int)threadIdx.x < nck)
                flags[(slab0 + (int)threadIdx.x) * gy_main + by] = ((all >> (8 * threadIdx.x)) & 0xfeu) | 1u;   // (row 0: every slab, always; no window rows when N > 1)
            if (all == 0u && !direct) { MUSTAFAR_TRACE_END(); return; }
            n_lo = 1;
            n_hi = N;
        }
#pragma unroll 1
        for (int c = 0; c < nck; c++) {
            const uint32_t rows = (all >> (8 * c)) & 0xffu;
            if (rows != 0u || direct) run_chunk(slab0 + c, rows, n_lo, n_hi);
        }
    } else {
        run_chunk(slab0, 1u, 0, 1);
    }
    MUSTAFAR_TRACE_END();
}

inline int pick_g(int groups) { return (groups % 4 == 0) ? 4 : (groups % 2 == 0) ? 2 : 1; }

// FMA engine of the G = 4 kernels: 2 = v_dot2_f32_f16 on pairs of tiles (the DEFAULT: one-pass launch only; the two reference
// entry points and every G < 4 kernel then run v_fma_mix), 0 = v_fma_mix_f32 everywhere (exact fp16 products, subnormals
// included; MUSTAFAR_FMA_ENGINE=valu), 1 = matrix pipe as a 4-wide FMA unit (v_mfma_f32_4x4x4_16B_f16; opt-in, MFMA left off by
// default as the north_star asks: MUSTAFAR_FMA_ENGINE=mfma or mustafar_set_fma_engine(1)).
inline int fma_engine();
int g_key_lean = -1;    // the two reference entry points on the lean pair machinery (round 4): MUSTAFAR_KEY_LEAN / MUSTAFAR_VALUE_LEAN = 0 | 1
inline bool key_lean()
{
    if (g_key_lean < 0) {
        const char* e = getenv("MUSTAFAR_KEY_LEAN");
        g_key_lean = e ? atoi(e) != 0 : 1;
    }
    return g_key_lean != 0;
}
int g_key_split = -1;   // 0 = automatic; MUSTAFAR_KEY_SPLIT=1|2 forces
inline int key_split(int ntb, int gy)
{
    if (g_key_split < 0) {
        const char* e = getenv("MUSTAFAR_KEY_SPLIT");
        g_key_split = e ? atoi(e) : 0;
    }
    if (g_key_split == 1 || g_key_split == 2) return g_key_split;
    if (fma_engine() == 1) return 1;   // matrix-pipe engine: one wave per block at every size (tools/sweep_forms.sh: c3 19.1 vs 20.2 us, c5 57 vs 65)
    // One wave per token block unless that grid is small (<= 2048 workgroups; the chip holds 256 CUs x 6 of them, see
    // tools/wave_trace.py): then a wave's latency chain, not throughput, sets the time, and two waves per block halve it.
    return ((int64_t)((ntb + kWaves - 1) / kWaves) * gy <= 2048) ? 2 : 1;
}
// MUSTAFAR_WINDOW=rows keeps the dense-window work of the fused path inside the softmax / finish row kernels (the form
// used when T == 0) instead of window workgroups in the SpMV launches; =key / =value lets only that side ride.
int g_window_mode = -1;   // bit 0: key side rides, bit 1: value side rides
inline int window_ride_mask()
{
    if (g_window_mode < 0) {
        const char* e = getenv("MUSTAFAR_WINDOW");
        g_window_mode = !e ? 3 : e[0] == 'r' ? 0 : e[0] == 'k' ? 1 : e[0] == 'v' ? 2 : 3;
    }
    return g_window_mode;
}
// Where the window workgroups sit in the grid: in front of the SpMV rows in the key launch (the softmax behind it waits
// for their scores anyway, and they are done in the first microseconds), behind them in the value launch (they are
// short and fill the tail while the last SpMV workgroups drain; measured +0.3 % over leading rows).
// MUSTAFAR_WINDOW_POS=first|last|klast|vlast overrides.
int g_window_last = -1;   // bit 0: key launch, bit 1: value launch
inline bool window_rows_last(int side)
{
    if (g_window_last < 0) {
        const char* e = getenv("MUSTAFAR_WINDOW_POS");
        g_window_last = !e ? 2 : e[0] == 'l' ? 3 : e[0] == 'k' ? 1 : e[0] == 'v' ? 2 : 0;
    }
    return (g_window_last >> side) & 1;
}
// Structure of the fused decode entry point: 1 = one-pass launch (decode_onepass_kernel + onepass_finish_kernel),
// 0 = key SpMV -> softmax rows -> value SpMV -> sum, 2 = by size (default).  MUSTAFAR_ONEPASS=0|1|auto, mustafar_set_onepass().
// Measured (round 2, fused + graph, tokens/s one-pass vs two-launch).  VALU engine: c2 1566 vs 1282, c3 3970 vs 3890,
// c4 1194 vs 1285, c5 2561 vs 2819 -- the one-pass launch saves the softmax launch, a boundary and a ramp, which is what
// counts while a launch is tens of microseconds, and loses once the launches are long (its pair form pays two barriers per
// block).  Matrix-pipe engine (e stays in LDS, no barriers): c3 5200 vs 4590, c4 1676 vs 1543, c5 3672 vs 3391 -- one-pass
// at every size.  `by size` = that rule.
// Process DEFAULTS live in the g_* variables (environment, mustafar_set_fma_engine / mustafar_set_onepass); a fused call may carry
// its own choice in its `flags` argument, in force for that call only (t_engine / t_onepass: set and cleared by decode_attention
// on the calling thread).
thread_local int t_engine = -1, t_onepass = -1;
thread_local int t_last_choice = -1;   // what the last fused call on this thread launched: engine | structure << 4 | one-pass form << 8
int g_onepass = -1;
inline int onepass_mode()
{
    if (t_onepass >= 0) return t_onepass;
    if (g_onepass < 0) {
        const char* e = getenv("MUSTAFAR_ONEPASS");
        g_onepass = !e ? 2 : e[0] == '0' ? 0 : e[0] == '1' ? 1 : 2;
    }
    return g_onepass;
}
inline bool onepass_enabled(int64_t kv_heads, int T);
int g_onepass_wgs = -1;   // MUSTAFAR_ONEPASS_WGS=n overrides the workgroup target of the one-pass launch
inline int onepass_target_wgs(bool pair)
{
    if (g_onepass_wgs < 0) {
        const char* e = getenv("MUSTAFAR_ONEPASS_WGS");
        g_onepass_wgs = e ? atoi(e) : 0;
    }
    return g_onepass_wgs > 0 ? g_onepass_wgs : (pair ? 4096 : 0);   // (VALU pair form: flat from 4096 workgroups up at c3; 0: fixed blocks per workgroup)
}
// GQA-4 one-pass launches on the vector engines: MUSTAFAR_ONEPASS_LEAN=2 (default) the lean kernel at the pair grain, 1 the lean
// kernel with whole blocks per wave, 0 the round-2 pair form; MUSTAFAR_LEAN_TBW=n: blocks per wave (1) / block pairs per
// workgroup (2) instead of the automatic choice (raised when the slabs would not fit).
int g_finish1 = [] { const char* e = getenv("MUSTAFAR_FINISH1"); return e ? atoi(e) != 0 : 1; }();   // round 5: the one-thread-per-channel row kernel for rows of <= 64 slabs (mustafar_tune(10, v))
int g_sb = [] { const char* e = getenv("MUSTAFAR_SB"); return e ? atoi(e) != 0 : 1; }();   // round 5: the super-block pair form (mustafar_tune(8, 0): round 4's pair kernel)
int g_small = [] { const char* e = getenv("MUSTAFAR_SMALL"); return e ? atoi(e) : 1; }();   // round 6: decode_onepass_small_kernel: 1 = for launches of at most small_waves() waves (default), 2 = for every launch of two
                                                                                          // blocks per workgroup, 0 = never (mustafar_tune(11, v))
inline int small_waves()   // one wave per SIMD: 4 x the CU count (asked of the runtime at the first fused call, not when the library is loaded)
{
    static const int n = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return (cus > 0 ? cus : 256) * 4;
    }();
    return n;
}
int g_spec_k_bytes = 0; // mustafar_tune(12, bytes): experiment, see OneArgs::spec_k_bytes
int g_late_prio = 1;    // mustafar_tune(9, 0): no raised priority for a small last round of workgroups (experiments)
int g_pair_slabs = 0;   // pair form, mustafar_tune(4, 1): a slab per pair instead of one per workgroup (kernel 1.1 us shorter at c3, row kernel 1.5 us longer)
// g_lean_win_last: the pair form's window workgroups sit BEHIND the SpMV rows of the grid (mustafar_tune(3, 0): in front, round 3a).
// In front they hold 434 of the chip's 2048 workgroup slots for their ~10 us while the SpMV rows wait; behind, they fill the tail
// (matrix pipe c3 37.1 -> 36.2 us, c4 61.4 -> 60.0, c5 119.0 -> 115.3; dot2 +-0 at c3 / c4, 139.2 -> 137.3 at c5).
int g_lean = -1, g_lean_tbw = -1, g_lean_win_last = 1;
inline int onepass_lean()
{
    if (g_lean < 0) {
        const char* e = getenv("MUSTAFAR_ONEPASS_LEAN");
        g_lean = !e ? 2 : (e[0] == '0') ? 0 : (e[0] == '1') ? 1 : 2;
    }
    return g_lean;
}
inline int onepass_lean_tbw()
{
    if (g_lean_tbw < 0) {
        const char* e = getenv("MUSTAFAR_LEAN_TBW");
        g_lean_tbw = e ? atoi(e) : 0;
    }
    return g_lean_tbw;
}
int g_engine = -1;
inline int fma_engine()
{
    if (t_engine >= 0) return t_engine;
    if (g_engine < 0) {
        const char* e = getenv("MUSTAFAR_FMA_ENGINE");
        g_engine = !e ? 2 : (e[0] == 'm' || e[0] == 'M' || e[0] == '1') ? 1 : (e[0] == 'v' || e[0] == 'V' || e[0] == '0') ? 0 : 2;
    }
    return g_engine;
}

inline bool onepass_enabled(int64_t kv_heads, int T)
{
    const int mode = onepass_mode();
    if (mode != 2) return mode == 1;
    return true;   // (decode_attention narrows this for the round-2 pair form, the only one that loses to two launches at c4 / c5)
}

// Optional live timing of the two SpMV kernels inside mustafar_decode_attention (bench.py's roofline leg): HIP
// events that receive each kernel's own start / stop timestamps (hipExtLaunchKernel), i.e. what rocprofv3 reports.  Off by default; not thread-safe by design
// (the hook is single-threaded, mustafar_wrapper.cu holds the GIL throughout as well).
struct Profile {
    bool on = false;
    int cap = 0, n = 0;
    hipEvent_t* ev = nullptr;   // 4 per record: key begin/end, value begin/end (one-pass launch: the first pair only)
    int onepass = 0;            // records taken on the one-pass launch
    int finish = 0;             // ... whose second event pair holds the row kernel (onepass_finish_kernel) behind it
} g_prof;


// One place that picks the key kernel instantiation: G heads per pass, FMA engine, waves per token block.
void launch_key(hipStream_t st, const uint64_t* bmp, const unsigned char* nz, const uint32_t* idx, const uint32_t* nz_off,
                const h16* q, h16* out, int T, int N, int groups, int Batch_Size, int ldc, WinArgs wa = WinArgs{},
                hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr,   // ev0/ev1: the kernel's own start / stop timestamps
                int64_t bmp_stride = 0, int64_t idx_stride = 0, uint32_t nz_stride = 0)
{
    const int G = pick_g(groups);
    const int gy = (Batch_Size / groups) * (groups / G);
    const int ntb = T / 64;
    if (key_lean() && (N == 1 || N == 8)) {
        // round 4: the lean pair form (key_lean_kernel); MUSTAFAR_KEY_LEAN=0 / mustafar_tune(6, 0) selects the round-1 kernel below
        dim3 grid((ntb + 1) / 2, gy);
        if (wa.win) {
            wa.nchunks = (wa.w_cap + kKeyWinChunk - 1) / kKeyWinChunk;
            wa.rows = (gy * wa.nchunks + (int)grid.x - 1) / (int)grid.x;
            grid.y += wa.rows;
            if (window_rows_last(0)) wa.rows = -wa.rows;
        }
        const bool mf = G == 4 && fma_engine() == 1;
#define MUSTAFAR_LKL(GG, EE)                                                                                                     \
    do {                                                                                                                         \
        if (N == 1 && wa.rows != 0)                                                                                              \
            hipExtLaunchKernelGGL((key_lean_kernel<GG, EE, 1, true>), grid, dim3(kThreads), 0, st, ev0, ev1, 0, bmp, nz, idx, nz_off, q, out,     \
                                  T, groups, ldc, wa, bmp_stride, idx_stride, nz_stride);                                        \
        else if (N == 1) hipExtLaunchKernelGGL((key_lean_kernel<GG, EE, 1>), grid, dim3(kThreads), 0, st, ev0, ev1, 0, bmp, nz, idx, nz_off, q, out,   \
                                          T, groups, ldc, wa, bmp_stride, idx_stride, nz_stride);                                \
        else        hipExtLaunchKernelGGL((key_lean_kernel<GG, EE, 8>), grid, dim3(kThreads), 0, st, ev0, ev1, 0, bmp, nz, idx, nz_off, q, out,   \
                                          T, groups, ldc, wa, bmp_stride, idx_stride, nz_stride);                                \
    } while (0)
        if (G == 4) { if (mf) MUSTAFAR_LKL(4, 1); else MUSTAFAR_LKL(4, 0); }
        else if (G == 2) MUSTAFAR_LKL(2, 0);
        else MUSTAFAR_LKL(1, 0);
#undef MUSTAFAR_LKL
        return;
    }
    const int split = key_split(ntb, gy);
    const int per_wg = kWaves / split;
    dim3 grid((ntb + per_wg - 1) / per_wg, gy);
    if (wa.win) {   // window workgroups first: whole grid rows in front of the SpMV rows
        wa.nchunks = (wa.w_cap + kKeyWinChunk - 1) / kKeyWinChunk;
        wa.rows = (gy * wa.nchunks + (int)grid.x - 1) / (int)grid.x;
        grid.y += wa.rows;
        if (window_rows_last(0)) wa.rows = -wa.rows;
    }
#define MUSTAFAR_LK(GG, MFF)                                                                                                   \
    do {                                                                                                                       \
        if (split == 2) hipExtLaunchKernelGGL((key_spmv_kernel<GG, MFF, 2>), grid, dim3(kThreads), 0, st, ev0, ev1, 0,         \
                                              bmp, nz, idx, nz_off, q, out, T, N, groups, ldc, wa, bmp_stride, idx_stride, nz_stride);   \
        else            hipExtLaunchKernelGGL((key_spmv_kernel<GG, MFF, 1>), grid, dim3(kThreads), 0, st, ev0, ev1, 0,         \
                                              bmp, nz, idx, nz_off, q, out, T, N, groups, ldc, wa, bmp_stride, idx_stride, nz_stride);   \
    } while (0)
    switch (G) {
        case 4:
            if (fma_engine() == 1) MUSTAFAR_LK(4, true);
            else              MUSTAFAR_LK(4, false);
            break;
        case 2: MUSTAFAR_LK(2, false); break;
        default: MUSTAFAR_LK(1, false); break;
    }
#undef MUSTAFAR_LK
}

// The value kernel's form: 8 waves per workgroup, two per token block (one per 64-channel half) by default;
// MUSTAFAR_VALUE_SPLIT=1 selects the older 4-wave / one-wave-per-block form.
constexpr int kValueWaves = 8;
int g_value_split = -1;
inline int value_split()
{
    if (g_value_split < 0) {
        const char* e = getenv("MUSTAFAR_VALUE_SPLIT");
        g_value_split = e ? (atoi(e) == 1 ? 1 : 2) : 0;
    }
    if (g_value_split) return g_value_split;
    return fma_engine() == 1 ? 1 : 2;   // the MFMA form needs > 80 VGPRs: 8-wave workgroups would drop to 4 waves per SIMD
}
// round 4: the lean pair form of the value entry point (value_lean_kernel, v_fma_mix engine) was opt-in -- MUSTAFAR_VALUE_LEAN=1 /
// mustafar_tune(7, 1).  Measured against value_spmv_kernel (kernel + combine, us, N = 1): c3 28.0-29.9 vs 29.4-30.1, c4 48.8 vs 46.7,
// c5 85.6 vs 76.8; N = 8 (the hook's padded rows): c3 55.7 vs 41.2 -- the round-1 kernel keeps its next block's bounds, metadata lines
// and first chunk in flight while it works on the current one, which a value-only launch (no softmax step between the blocks) can do
// and the lean block phase does not; lean addressing alone does not make up for it (profiles/r04_probes.txt).
// Round 5: the lean form walks a pair's blocks two at a time as one pipeline of four chunks (lean_pair_phase) and is the default for N = 1
// (the operands a caller of the C ABI / the hook's api="native" passes) up to ~24 k workgroup-blocks per launch: c2 22.5 -> 17.4 us, c3 30.8 ->
// 28.0, c4 47.4 -> 45.1; c5 (32 k) 78.6 -> 80.6 and the hook's 8 padded rows (c3 41.6 -> 48.4) stay with round 1's kernel.  (Its GQA-4
// instantiation spilled 83 scalar + 12 vector registers: the window path it carried and four coefficient pointers -- both gone since round 5b:
// template flag WIN, CoefStride; 21 scalar spills, no private segment.  The call stays 26-27 us at c3: ~22 us of kernel + the combine launch.)
// MUSTAFAR_VALUE_LEAN = 0 | 1 / mustafar_tune(7, .) force one form; unset (2): by size.
int g_value_lean = -1;
int g_value_lean8 = 1;   // round 6: the lean form also for the hook's 8 padded rows (mustafar_tune(13, 0): round 1's kernel for N = 8, as rounds 1-5)
inline int value_lean_mode()
{
    if (g_value_lean < 0) {
        const char* e = getenv("MUSTAFAR_VALUE_LEAN");
        g_value_lean = e ? (atoi(e) != 0 ? 1 : 0) : 2;
    }
    return fma_engine() == 1 ? 0 : g_value_lean;
}
// Round 6: with the probabilities prefetched (MUSTAFAR_VL_PFCOEF) the lean form wins at every size -- c5 (32 k workgroup-blocks), us per call, lean / round 1's kernel:
// one row 83.4 / 92.5, eight rows 83.6 / 95.9 (round 5, without the prefetch: 80.6 / 78.6, and the size limit of 24 k workgroup-blocks that came from it) -- so "by size"
// (mode 2, the default) now means: lean whenever the vector engines run, for N = 1 and -- mustafar_tune(13, .) -- the hook's 8 padded rows.
inline bool value_lean() { return value_lean_mode() != 0; }   // (the workgroup shape follows it, value_tb_stride)
inline bool value_lean_for(int N, int64_t /*wg_blocks*/) { const int m = value_lean_mode(); return m == 1 || (m == 2 && (N == 1 || g_value_lean8)); }
inline int value_tb_stride() { return value_lean() ? kWaves / 2 : value_split() == 2 ? kValueWaves / 2 : kWaves; }   // token blocks in flight per workgroup

// One place that picks the value kernel instantiation.
void launch_value(hipStream_t st, dim3 grid, const uint64_t* bmp, const unsigned char* nz, const uint32_t* idx,
                  const uint32_t* nz_off, const h16* p, h16* out, float* ws, uint32_t* flags, int T, int N, int groups,
                  int Batch_Size, int tb_per_wg, int direct, int ldb, WinArgs wa = WinArgs{}, hipEvent_t ev0 = nullptr,
                  hipEvent_t ev1 = nullptr, int64_t bmp_stride = 0, int64_t idx_stride = 0, uint32_t nz_stride = 0)
{
    const int G = pick_g(groups);
    if (wa.win) {   // window workgroups first; their partial slabs follow the grid.x token-chunk slabs
        wa.nchunks = (wa.w_cap + kValueWinChunk - 1) / kValueWinChunk;
        wa.rows = ((int)grid.y * wa.nchunks + (int)grid.x - 1) / (int)grid.x;
        grid.y += wa.rows;
        if (window_rows_last(1)) wa.rows = -wa.rows;
    }
    if ((N == 1 || N == 8) && value_lean_for(N, (int64_t)(grid.y - (wa.rows < 0 ? -wa.rows : wa.rows)) * (T / 64))) {
#ifndef MUSTAFAR_PROBE_NOPADWG   // (timing probe: no pad workgroups at all -- the slabs' row masks stay unwritten)
        if (N > 1) grid.y += (grid.y * ((grid.x + MUSTAFAR_PAD_GROUP - 1) / MUSTAFAR_PAD_GROUP) + grid.x - 1) / grid.x;
#endif   // (the pad workgroups behind the row-0 workgroups, one per head group and four chunks: value_lean_kernel)
#define MUSTAFAR_LVL(GG)                                                                                                         \
    do {                                                                                                                         \
        if (N == 1 && wa.rows != 0)                                                                                                              \
            hipExtLaunchKernelGGL((value_lean_kernel<GG, 1, true>), grid, dim3(kThreads), 0, st, ev0, ev1, 0, bmp, nz, idx, nz_off, p, out, ws,    \
                                  flags, T, groups, Batch_Size, tb_per_wg, direct, ldb, wa, bmp_stride, idx_stride, nz_stride);                    \
        else if (N == 1) hipExtLaunchKernelGGL((value_lean_kernel<GG, 1>), grid, dim3(kThreads), 0, st, ev0, ev1, 0, bmp, nz, idx, nz_off, p, out, ws,  \
                                          flags, T, groups, Batch_Size, tb_per_wg, direct, ldb, wa, bmp_stride, idx_stride, nz_stride);            \
        else if (MUSTAFAR_PROBE_N1AS8) hipExtLaunchKernelGGL((value_lean_kernel<GG, 1>), grid, dim3(kThreads), 0, st, ev0, ev1, 0, bmp, nz, idx, nz_off, p, out, ws,  \
                                          flags, T, groups, Batch_Size, tb_per_wg, direct, 8 * ldb, wa, bmp_stride, idx_stride, nz_stride);        \
        else        hipExtLaunchKernelGGL((value_lean_kernel<GG, 8>), grid, dim3(kThreads), 0, st, ev0, ev1, 0, bmp, nz, idx, nz_off, p, out, ws,  \
                                          flags, T, groups, Batch_Size, tb_per_wg, direct, ldb, wa, bmp_stride, idx_stride, nz_stride);            \
    } while (0)
        if (G == 4) MUSTAFAR_LVL(4);
        else if (G == 2) MUSTAFAR_LVL(2);
        else MUSTAFAR_LVL(1);
#undef MUSTAFAR_LVL
        return;
    }
#define MUSTAFAR_LV(GG, MFF)                                                                                                   \
    do {                                                                                                                       \
        if (value_split() == 2 && wa.rows != 0)                                                                                \
            hipExtLaunchKernelGGL((value_spmv_kernel<GG, MFF, kValueWaves, 2, true>), grid, dim3(kValueWaves * 64), 0, st, ev0, ev1, 0, \
                                  bmp, nz, idx, nz_off, p, out, ws, flags, T, N, groups, Batch_Size, tb_per_wg, direct, ldb, wa, bmp_stride, idx_stride, nz_stride); \
        else if (value_split() == 2)                                                                                           \
            hipExtLaunchKernelGGL((value_spmv_kernel<GG, MFF, kValueWaves, 2, false>), grid, dim3(kValueWaves * 64), 0, st, ev0, ev1, 0, \
                                  bmp, nz, idx, nz_off, p, out, ws, flags, T, N, groups, Batch_Size, tb_per_wg, direct, ldb, wa, bmp_stride, idx_stride, nz_stride); \
        else if (wa.rows != 0)                                                                                                 \
            hipExtLaunchKernelGGL((value_spmv_kernel<GG, MFF, kWaves, 1, true>), grid, dim3(kThreads), 0, st, ev0, ev1, 0,     \
                                  bmp, nz, idx, nz_off, p, out, ws, flags, T, N, groups, Batch_Size, tb_per_wg, direct, ldb, wa, bmp_stride, idx_stride, nz_stride); \
        else                                                                                                                   \
            hipExtLaunchKernelGGL((value_spmv_kernel<GG, MFF, kWaves, 1, false>), grid, dim3(kThreads), 0, st, ev0, ev1, 0,    \
                                  bmp, nz, idx, nz_off, p, out, ws, flags, T, N, groups, Batch_Size, tb_per_wg, direct, ldb, wa, bmp_stride, idx_stride, nz_stride); \
    } while (0)
    switch (G) {
        case 4:
            if (fma_engine() == 1) MUSTAFAR_LV(4, true);
            else              MUSTAFAR_LV(4, false);
            break;
        case 2: MUSTAFAR_LV(2, false); break;
        default: MUSTAFAR_LV(1, false); break;
    }
#undef MUSTAFAR_LV
}

}  // namespace

extern "C" {

int mustafar_abi_version(void) { return 106; }   // 106 (round 6): mustafar_compress_get_form; the compression form and the test hook are per host thread; 105 (round 5): mustafar_profile_end2, mustafar_convert_*, mustafar_cache_consolidate_extents, mustafar_compress_set_form

int Key_SplitK_API(void* stream, const void* /*A*/, const uint64_t* bmp, const void* NZ, const uint32_t* idx,
                   const uint32_t* NZ_offset, const void* B, void* C, int M_Global, int N_Global, int K_Global,
                   void* /*Reduction_Workspace*/, int Split_K, int Batch_Size, int num_key_value_groups)
{
    const int T = M_Global, N = N_Global, groups = num_key_value_groups;
    if (K_Global != kD || T <= 0 || (T & 63) || (N != 1 && N != 8) || Split_K != 1 || groups < 1 ||
        Batch_Size < 1 || Batch_Size % groups)
        return MUSTAFAR_EINVAL;
    if (!bmp || !NZ || !idx || !NZ_offset || !B || !C) return MUSTAFAR_EINVAL;
    launch_key(static_cast<hipStream_t>(stream), bmp, static_cast<const unsigned char*>(NZ), idx, NZ_offset,
               static_cast<const h16*>(B), static_cast<h16*>(C), T, N, groups, Batch_Size, T);
    return (int)hipGetLastError();
}

int mustafar_value_pick_split_k(int M_Global, int N_Global, int K_Global, int Batch_Size, int num_key_value_groups)
{
    (void)N_Global;
    if (M_Global != kD || K_Global <= 0 || (K_Global & 63) || num_key_value_groups < 1 || Batch_Size < num_key_value_groups ||
        Batch_Size % num_key_value_groups)
        return 1;   // (arguments Value_SplitK_API rejects: no split, no division by an empty grid)
    const int ntb = K_Global / 64;
    const int G = pick_g(num_key_value_groups);
    const int gy = (Batch_Size / num_key_value_groups) * (num_key_value_groups / G);
    // ~2048 workgroups (2-3 rounds of what the chip holds: the start-up latency chains of one round hide behind the
    // steady state of another -- a single exact round measured 10-25 % slower, tools/sweep_split.py), every workgroup
    // a multiple of the token blocks it keeps in flight so that its waves finish together.
    const int stride = value_tb_stride();
    const int want = (2048 + gy - 1) / gy;
    int tb = (ntb + want - 1) / want;
    tb = (tb + stride - 1) / stride * stride;
    return (ntb + tb - 1) / tb;   // normalised: every chunk is non-empty
}

int64_t mustafar_value_workspace_bytes(int M_Global, int N_Global, int K_Global, int Batch_Size,
                                       int num_key_value_groups, int Split_K)
{
    (void)K_Global;
    if (Split_K <= 1 || num_key_value_groups < 1 || Batch_Size < num_key_value_groups || Batch_Size % num_key_value_groups ||
        N_Global < 1 || M_Global < 1)
        return 0;
    const int G = pick_g(num_key_value_groups);
    const int64_t gy = (int64_t)(Batch_Size / num_key_value_groups) * (num_key_value_groups / G);
    const int64_t slabs = (int64_t)Split_K * Batch_Size * N_Global * M_Global * (int64_t)sizeof(float);
    return slabs + ((Split_K * gy * (int64_t)sizeof(uint32_t) + 255) / 256) * 256;
}

int Value_SplitK_API(void* stream, const void* /*A*/, const uint64_t* bmp, const void* NZ, const uint32_t* idx,
                     const uint32_t* NZ_offset, const void* B, void* C, int M_Global, int N_Global, int K_Global,
                     void* Reduction_Workspace, int Split_K, int Batch_Size, int num_key_value_groups)
{
    const int T = K_Global, N = N_Global, groups = num_key_value_groups;
    if (M_Global != kD || T <= 0 || (T & 63) || (N != 1 && N != 8) || Split_K < 1 || groups < 1 ||
        Batch_Size < 1 || Batch_Size % groups)
        return MUSTAFAR_EINVAL;
    if (!bmp || !NZ || !idx || !NZ_offset || !B || !C) return MUSTAFAR_EINVAL;
    if (Split_K > 1 && !Reduction_Workspace) return MUSTAFAR_EINVAL;
    const int ntb = T / 64;
    const int G = pick_g(groups);
    const int gy = (Batch_Size / groups) * (groups / G);
    const int tb_per_wg = (ntb + Split_K - 1) / Split_K;
    const int S = (ntb + tb_per_wg - 1) / tb_per_wg;   // non-empty chunks (<= Split_K)
    const int direct = (S == 1);
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto nz = static_cast<const unsigned char*>(NZ);
    auto p  = static_cast<const h16*>(B);
    auto o  = static_cast<h16*>(C);
    float* ws = static_cast<float*>(Reduction_Workspace);
    uint32_t* flags = nullptr;
    if (!direct) {
        const int64_t slabs = (int64_t)Split_K * Batch_Size * N * kD * (int64_t)sizeof(float);
        flags = reinterpret_cast<uint32_t*>(static_cast<unsigned char*>(Reduction_Workspace) + slabs);
    }
    launch_value(st, dim3(S, gy), bmp, nz, idx, NZ_offset, p, o, ws, flags, T, N, groups, Batch_Size, tb_per_wg, direct, T);
    int err = (int)hipGetLastError();
    if (err || direct) return err;
    value_combine_kernel<<<(unsigned)Batch_Size, 256, 0, st>>>(ws, flags, o, Batch_Size, N, S, groups, G);
    return (int)hipGetLastError();
}


int64_t mustafar_decode_workspace_bytes(int T, int Batch_Size, int num_key_value_groups, int Split_K)
{
    (void)num_key_value_groups;   // token-chunk slabs + the window workgroups' slabs (one-pass forms: 64-token window chunks and a
    // (max, sum) pair per slab and row); sized for a slab per 64-token block (the forms in use leave at most one per two)
    const int ntb = T > 0 ? T / 64 : 1;
    const int sk = Split_K < 1 ? 1 : Split_K;
    const int per_block = ntb + 1 < kMaxSlabs ? ntb + 1 : kMaxSlabs;   // (pair form: two slabs per workgroup of two blocks, the last one may hold one)
    const int slabs = sk > per_block ? sk : per_block;
    return (int64_t)(slabs + kMaxWindow / kOneWinChunk) * Batch_Size * (kD + 2) * (int64_t)sizeof(float);
}

}  // extern "C"

namespace {
int decode_attention(void* stream, const mustafar_cache_view& kc, const mustafar_cache_view& vc, const void* q, void* k_window,
                     void* v_window, const void* k_new, const void* v_new, int window_len, int window_capacity, void* scores,
                     int ld_scores, void* out, void* workspace, int Split_K, int T, int Batch_Size, int num_key_value_groups,
                     float sqrt_d, const int32_t* window_len_extra, const void* attention_mask, int64_t mask_row_stride,
                     int heads_per_mask_row, uint32_t flags, const mustafar_cache_view* k_ext = nullptr,
                     const mustafar_cache_view* v_ext = nullptr, int T_base = 0, const int32_t* T_device = nullptr)
{
    const int groups = num_key_value_groups;
    const bool extents = k_ext != nullptr;   // (validated by mustafar_decode_attention_extents)
    // the call's own engine / structure (mustafar_hip.h: MUSTAFAR_FLAG_*), in force until this function returns
    const uint32_t f_eng = flags & 7u, f_str = (flags >> 4) & 3u;
    if (f_eng > 3u || f_str > 2u || (flags & ~0x37u)) return MUSTAFAR_EINVAL;
    struct Override {
        Override(int e, int o) { t_engine = e; t_onepass = o; }
        ~Override() { t_engine = -1; t_onepass = -1; }
    } override_(f_eng == 1 ? 0 : f_eng == 2 ? 1 : f_eng == 3 ? 2 : -1, f_str == 1 ? 0 : f_str == 2 ? 1 : -1);
    if (attention_mask && (heads_per_mask_row < 1 || Batch_Size % heads_per_mask_row || mask_row_stride < 0)) return MUSTAFAR_EINVAL;
    const MaskArg mask{static_cast<const h16*>(attention_mask), mask_row_stride, heads_per_mask_row > 0 ? heads_per_mask_row : 1};
    if (T < 0 || (T & 63) || groups < 1 || Batch_Size < 1 || Batch_Size % groups || window_len < 1 ||
        window_len > window_capacity || window_capacity > kMaxWindow ||
        ld_scores < T + (window_len_extra ? window_capacity : window_len) || (ld_scores & 7) || Split_K < 1 || !(sqrt_d > 0.f))
        return MUSTAFAR_EINVAL;
    if (!q || !k_window || !v_window || !scores || !out || !workspace) return MUSTAFAR_EINVAL;
    if (T > 0 && (!kc.bmp || !kc.nz || !kc.idx || !kc.nz_offset || !vc.bmp || !vc.nz || !vc.idx || !vc.nz_offset)) return MUSTAFAR_EINVAL;
    const int64_t tiles = (int64_t)(extents ? T_base : T) * 2;   // (the base views hold T_base of the T tokens when the cache grew by extents)
    if (T > 0 && ((kc.bmp_head_stride && kc.bmp_head_stride < tiles) || (kc.idx_head_stride && kc.idx_head_stride < tiles + 1) ||
                  (vc.bmp_head_stride && vc.bmp_head_stride < tiles) || (vc.idx_head_stride && vc.idx_head_stride < tiles + 1)))
        return MUSTAFAR_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto qh = static_cast<const h16*>(q);
    auto sc = static_cast<h16*>(scores);
    const int G = pick_g(groups);
    const int gy = (Batch_Size / groups) * (groups / G);
    int S = 0, nwin_slabs = 0;
    const bool prof = g_prof.on && g_prof.n < g_prof.cap && T > 0;
    auto kwin = static_cast<h16*>(k_window);
    auto vwin = static_cast<h16*>(v_window);
    auto knew = static_cast<const h16*>(k_new);
    auto vnew = static_cast<const h16*>(v_new);
    const float inv_sqrt_d0 = (float)(1.0 / (double)sqrt_d);
    // Structure by size (mode 2): the lean kernels (GQA-4, vector engines) and the matrix-pipe form run one-pass at every size
    // (round 3, tokens/s one-pass vs two launches -- dot2: c3 5110 vs 4180, c4 1540 vs 1350, c5 3470 vs 3130; fma_mix: c4 1416 vs
    // 1353, c5 3111 vs 3131); the round-2 pair form (G < 4) while kv-heads x T is small (c2: 1650 vs 1310)
    // (the matrix-pipe engine has the pair form only: lean == 1 keeps its round-2 whole-block kernel, as lean == 0 does)
    // (G < 4 -- MHA, GQA-2 -- has the pair form only, on the v_fma_mix engine: round 4; before, its fused calls ran round 2's
    // decode_onepass_kernel, which reads neither extents nor a device-side T)
    const bool lean_form = G == 4 ? onepass_lean() != 0 && !(fma_engine() == 1 && onepass_lean() == 1) : onepass_lean() == 2;
    const int eng = G == 4 ? fma_engine() : 0;
    const bool small = (int64_t)(Batch_Size / groups) * T <= 768000;
    // appended extents are read by the pair form of the one-pass launch only
    if (extents && !(T > 0 && lean_form && onepass_lean() == 2 && onepass_enabled(Batch_Size / groups, T) && (ld_scores & 31) == 0))
        return MUSTAFAR_EINVAL;
    if (T > 0 && onepass_enabled(Batch_Size / groups, T) && (onepass_mode() == 1 || lean_form || fma_engine() == 1 || small) &&
        (ld_scores & 31) == 0) {
        // ---- one-pass form: every wave runs key phase -> softmax step -> value phase on its token blocks; slabs merged per row
        const int ntb = T / 64;
        if (lean_form) {
            // lean forms (GQA-4, vector engines).  2 (default): pair grain -- two waves per block, two blocks per workgroup at a
            // time; 1: a wave owns `tbw` consecutive whole blocks.  Four waves merge into one slab either way.
            const bool lp = onepass_lean() == 2;
            const int nchunks = (window_capacity + kOneWinChunk - 1) / kOneWinChunk;
            int per_wg;   // 64-token blocks per workgroup
            if (lp) {
                // two block pairs per workgroup (each pair of waves runs its block loop twice) unless that leaves fewer than 1024
                // workgroups: start-up and merge code are paid once per two blocks and the row kernel folds half the slabs.  With the
                // window workgroups BEHIND the SpMV rows (below) this is the better shape at c3 too -- round 3a, window rows first:
                // 3968 workgroups of one pair each (tokens/s, four vs two blocks per workgroup, windows last: matrix pipe c3 6204 vs
                // 6050, c4 1877 vs 1719, c5 4070 vs 3900; dot2 c3 5122 vs 4992).  MUSTAFAR_ONEPASS_WGS / MUSTAFAR_LEAN_TBW override
                // (round 5, super-block form: from 768 workgroups -- three per CU -- on; Llama-3-8B 4k x batch 8 is 960 workgroups of four blocks:
                // 23.5 vs 24.6 us; at 480 (c2) and 248 (8k x batch 1) two blocks per workgroup win, 13.3 vs 15.4 and 11.1 vs 14.6 us)
                per_wg = (int64_t)((ntb + 3) / 4) * gy >= (g_sb ? 768 : 1024) ? 4 : 2;
                (void)onepass_target_wgs(true);   // (reads MUSTAFAR_ONEPASS_WGS once)
                if (g_onepass_wgs > 0) {
                    const int want = (g_onepass_wgs + gy - 1) / gy;
                    per_wg = ((ntb + want - 1) / want + 1) / 2 * 2;
                }
                if (onepass_lean_tbw() > 0) per_wg = 2 * onepass_lean_tbw();
            } else {
                per_wg = kWaves * (onepass_lean_tbw() > 0 ? onepass_lean_tbw() : 1);
            }
            const int step = lp ? 2 : kWaves;
            const int spw = lp && g_pair_slabs && !(g_sb && per_wg <= 4) ? 2 : 1;   // slabs per workgroup (the super-block form always merges its pairs)
            while (spw * ((ntb + per_wg - 1) / per_wg) + nchunks > kMaxSlabs) per_wg += step;
            const int S1 = (ntb + per_wg - 1) / per_wg;   // workgroups per head group
            const int NS = spw * S1;                      // their slabs
            float* ws_o = static_cast<float*>(workspace);
            float* ws_ml = ws_o + (int64_t)(NS + nchunks) * Batch_Size * kD;
            const int win_rows = (gy * nchunks + S1 - 1) / S1;
            if (extents && (!lp || (per_wg != 2 && per_wg != 4))) return MUSTAFAR_EINVAL;   // a workgroup's blocks stay inside one extent
            OneArgs a{qh, sc, ws_o, ws_ml, kwin, vwin, knew, vnew, window_len_extra, mask, T, groups, Batch_Size, lp ? per_wg : per_wg / kWaves,
                      ld_scores, window_len, window_capacity, nchunks, g_lean_win_last ? -win_rows : win_rows, inv_sqrt_d0, spw == 2};
            a.spec_k_bytes = g_spec_k_bytes;
            if (extents) {
                a.k_ext = k_ext;
                a.v_ext = v_ext;
                a.nb0 = T_base / 64;
                a.t_dev = T_device;
            }
            const dim3 grid(S1, gy + win_rows);
            {   // the last round of SpMV workgroups, when it is a small fraction of a resident round (8 workgroups per CU): raised priority
                static const int slots = [] {
                    int dev = 0, cus = 256;
                    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
                    return (cus > 0 ? cus : 256) * 8;
                }();
                const int64_t W = (int64_t)S1 * gy, full = W / slots * slots;
                if (g_late_prio && full > 0 && W - full > 0 && (W - full) * 4 <= slots && !(a.win_rows > 0)) a.hi_prio_from = (int)full;
            }
            hipEvent_t e0 = prof ? g_prof.ev[4 * g_prof.n] : nullptr, e1 = prof ? g_prof.ev[4 * g_prof.n + 1] : nullptr;
            auto kz = static_cast<const unsigned char*>(kc.nz), vz = static_cast<const unsigned char*>(vc.nz);
            bool small_form = false;
#define MUSTAFAR_LL(KERNEL)                                                                                                           \
    hipExtLaunchKernelGGL(KERNEL, grid, dim3(kThreads), 0, st, e0, e1, 0, kc.bmp, kz, kc.idx, kc.nz_offset,                            \
                          vc.bmp, vz, vc.idx, vc.nz_offset, a, kc.bmp_head_stride, kc.idx_head_stride, (uint32_t)kc.nz_head_stride,    \
                          vc.bmp_head_stride, vc.idx_head_stride, (uint32_t)vc.nz_head_stride)
            if (lp) {
#define MUSTAFAR_LP(EXTV)                                                                                   \
    do {                                                                                                    \
        if (G == 2)        MUSTAFAR_LL((decode_onepass_leanpair_kernel<0, EXTV, 2>));                       \
        else if (G == 1)   MUSTAFAR_LL((decode_onepass_leanpair_kernel<0, EXTV, 1>));                       \
        else if (eng == 2) MUSTAFAR_LL((decode_onepass_leanpair_kernel<2, EXTV>));                          \
        else if (eng == 1) MUSTAFAR_LL((decode_onepass_leanpair_kernel<1, EXTV>));                          \
        else               MUSTAFAR_LL((decode_onepass_leanpair_kernel<0, EXTV>));                          \
    } while (0)
                // (G = 1 without extents: the plain instantiation comes out of the register allocator with a 68-byte private segment it
                // never touches, and a kernel with a private segment is launched with scratch (+1 % measured on the GQA-4 form); the
                // extents instantiation has none and serves the same launch with every block in the base views)
                if (!extents && G == 1) a.nb0 = ntb;
                if (g_sb && per_wg == 2 && eng != 1 && onepass_lean_tbw() <= 0 && (g_small == 2 || (g_small == 1 && (int64_t)S1 * gy * kWaves <= small_waves()))) {
                    // round 6: the form without a memory wait inside the phases, for launches that do not even put one wave on every SIMD (c1;
                    // Llama-3-8B 4k x batch 1): there its shorter chain wins 10 %; from two waves per SIMD on (8k x batch 1, c2) its extra
                    // v_readlane per step cost more than the waits they replace (profiles/r06_probes.txt item 3).  mustafar_tune(11, 2): always
#define MUSTAFAR_SM(EXTV, MASKV)                                                                              \
    do {                                                                                                      \
        if (G == 2)        MUSTAFAR_LL((decode_onepass_small_kernel<0, EXTV, 2, MASKV>));                      \
        else if (G == 1)   MUSTAFAR_LL((decode_onepass_small_kernel<0, EXTV, 1, MASKV>));                      \
        else if (eng == 2) MUSTAFAR_LL((decode_onepass_small_kernel<2, EXTV, 4, MASKV>));                      \
        else               MUSTAFAR_LL((decode_onepass_small_kernel<0, EXTV, 4, MASKV>));                      \
    } while (0)
                    if (extents) { if (mask.ptr) MUSTAFAR_SM(true, true); else MUSTAFAR_SM(true, false); }
                    else         { if (mask.ptr) MUSTAFAR_SM(false, true); else MUSTAFAR_SM(false, false); }
#undef MUSTAFAR_SM
                    small_form = true;
                } else if (g_sb && per_wg <= 4) {   // round 5: the super-block pair form (same grid, slabs and window workgroups; a pair walks its <= 2 blocks once)
#define MUSTAFAR_SB(EXTV, MASKV)                                                                            \
    do {                                                                                                    \
        if (G == 2)        MUSTAFAR_LL((decode_onepass_sb_kernel<0, EXTV, 2, MASKV>));                       \
        else if (G == 1)   MUSTAFAR_LL((decode_onepass_sb_kernel<0, EXTV, 1, MASKV>));                       \
        else if (eng == 2) MUSTAFAR_LL((decode_onepass_sb_kernel<2, EXTV, 4, MASKV>));                       \
        else if (eng == 1) MUSTAFAR_LL((decode_onepass_sb_kernel<1, EXTV, 4, MASKV>));                       \
        else               MUSTAFAR_LL((decode_onepass_sb_kernel<0, EXTV, 4, MASKV>));                       \
    } while (0)
                    if (extents || G == 1) { if (mask.ptr) MUSTAFAR_SB(true, true); else MUSTAFAR_SB(true, false); }
                    else                   { if (mask.ptr) MUSTAFAR_SB(false, true); else MUSTAFAR_SB(false, false); }
#undef MUSTAFAR_SB
                } else if (extents || G == 1) MUSTAFAR_LP(true);
                else                          MUSTAFAR_LP(false);
#undef MUSTAFAR_LP
            } else {
                if (fma_engine() == 2) MUSTAFAR_LL((decode_onepass_lean_kernel<2>));
                else                   MUSTAFAR_LL((decode_onepass_lean_kernel<0>));
            }
#undef MUSTAFAR_LL
            // (the row kernel's own start / stop timestamps go into the record's second event pair: mustafar_profile_end2)
            if (NS + nchunks <= 64 && g_finish1)
                hipExtLaunchKernelGGL(onepass_finish1_kernel, dim3(Batch_Size), dim3(128), 0, st, prof ? g_prof.ev[4 * g_prof.n + 2] : nullptr,
                                      prof ? g_prof.ev[4 * g_prof.n + 3] : nullptr, 0, ws_o, ws_ml, NS + nchunks, static_cast<h16*>(out), Batch_Size);
            else if (NS + nchunks <= 64)
                hipExtLaunchKernelGGL(onepass_finish_kernel<1>, dim3(Batch_Size), dim3(256), 0, st, prof ? g_prof.ev[4 * g_prof.n + 2] : nullptr,
                                      prof ? g_prof.ev[4 * g_prof.n + 3] : nullptr, 0, ws_o, ws_ml, NS + nchunks, static_cast<h16*>(out), Batch_Size);
            else if (NS + nchunks <= 128)
                hipExtLaunchKernelGGL(onepass_finish_kernel<2>, dim3(Batch_Size), dim3(256), 0, st, prof ? g_prof.ev[4 * g_prof.n + 2] : nullptr,
                                      prof ? g_prof.ev[4 * g_prof.n + 3] : nullptr, 0, ws_o, ws_ml, NS + nchunks, static_cast<h16*>(out), Batch_Size);
            else
                hipExtLaunchKernelGGL(onepass_finish_kernel<>, dim3(Batch_Size), dim3(256), 0, st, prof ? g_prof.ev[4 * g_prof.n + 2] : nullptr,
                                      prof ? g_prof.ev[4 * g_prof.n + 3] : nullptr, 0, ws_o, ws_ml, NS + nchunks, static_cast<h16*>(out), Batch_Size);
            if (prof) { g_prof.onepass++; g_prof.finish++; g_prof.n++; }
            t_last_choice = eng | (1 << 4) | ((lp ? (small_form ? 4 : g_sb && per_wg <= 4 ? 3 : 2) : 1) << 8);
            return (int)hipGetLastError();
        }
        const bool pair = fma_engine() != 1 || G != 4;                      // two waves per block unless the matrix-pipe engine runs
        const int round = pair ? kWaves / 2 : kWaves;                     // token blocks a workgroup has in flight
        // workgroups of the SpMV part: ~onepass_target_wgs(), every workgroup whole rounds of its waves (Split_K only sizes the
        // workspace here: the slab count below never exceeds it by more than the rounding)
        (void)Split_K;
        int tb_per_wg;
        if (onepass_target_wgs(pair) > 0) {
            const int want = (onepass_target_wgs(pair) + gy - 1) / gy;
            tb_per_wg = (ntb + want - 1) / want;
        } else {
            // matrix-pipe form: two whole blocks per wave, or one where that fills the last round of waves much better (the chip
            // holds 6 of these waves per SIMD: c4 with two blocks per wave runs 1.33 rounds, with one 2.65: 67 -> 63 us; c3 and c5
            // stay at two: 0.65 and 2.67 rounds)
            static const int slots = [] {
                int dev = 0, cus = 256;
                if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
                return (cus > 0 ? cus : 256) * 4 * 6;
            }();
            auto fill = [&](int tb) {
                const double rounds = (double)gy * ((ntb + tb - 1) / tb) * kWaves / slots;
                return rounds / ceil(rounds);
            };
            tb_per_wg = fill(4) > fill(8) + 0.1 ? 4 : 8;
        }
        tb_per_wg = (tb_per_wg + round - 1) / round * round;
        const int S1 = (ntb + tb_per_wg - 1) / tb_per_wg;
        const int nchunks = (window_capacity + kOneWinChunk - 1) / kOneWinChunk;
        if (S1 + nchunks <= kMaxSlabs) {
            float* ws_o = static_cast<float*>(workspace);
            float* ws_ml = ws_o + (int64_t)(S1 + nchunks) * Batch_Size * kD;
            const int win_rows = (gy * nchunks + S1 - 1) / S1;
            const OneArgs a{qh, sc, ws_o, ws_ml, kwin, vwin, knew, vnew, window_len_extra, mask, T, groups, Batch_Size, tb_per_wg, ld_scores,
                            window_len, window_capacity, nchunks, win_rows, inv_sqrt_d0};
            const dim3 grid(S1, gy + win_rows);
            hipEvent_t e0 = prof ? g_prof.ev[4 * g_prof.n] : nullptr, e1 = prof ? g_prof.ev[4 * g_prof.n + 1] : nullptr;
            auto kz = static_cast<const unsigned char*>(kc.nz), vz = static_cast<const unsigned char*>(vc.nz);
#define MUSTAFAR_L1(GG, MFF)                                                                                                     \
    hipExtLaunchKernelGGL((decode_onepass_kernel<GG, MFF, !MFF>), grid, dim3(kThreads), 0, st, e0, e1, 0, kc.bmp, kz, kc.idx, kc.nz_offset,  \
                          vc.bmp, vz, vc.idx, vc.nz_offset, a, kc.bmp_head_stride, kc.idx_head_stride, (uint32_t)kc.nz_head_stride,    \
                          vc.bmp_head_stride, vc.idx_head_stride, (uint32_t)vc.nz_head_stride)
            switch (G) {
                case 4:
                    if (fma_engine() == 1) MUSTAFAR_L1(4, true);
                    else              MUSTAFAR_L1(4, false);
                    break;
                case 2: MUSTAFAR_L1(2, false); break;
                default: MUSTAFAR_L1(1, false); break;
            }
#undef MUSTAFAR_L1
            if (prof) { g_prof.onepass++; g_prof.n++; }
            onepass_finish_kernel<><<<Batch_Size, 256, 0, st>>>(ws_o, ws_ml, S1 + nchunks, static_cast<h16*>(out), Batch_Size);
            t_last_choice = (fma_engine() == 1 && G == 4 ? 1 : 0) | (1 << 4);
            return (int)hipGetLastError();
        }
    }
    // ---- two-launch form.  With a compressed part the dense-window work rides in the two SpMV launches (window workgroups);
    // without one (T == 0) the two row kernels do it themselves.
    const bool long_rows = T > kMaxRowVecs * kGlueThreads * 8;   // beyond the register form of the softmax kernel
    const bool ride_k = T > 0 && ((window_ride_mask() & 1) || long_rows), ride_v = T > 0 && (window_ride_mask() & 2);
    if (T > 0) {
        const WinArgs kw = ride_k ? WinArgs{kwin, knew, window_len_extra, window_len, window_capacity, 0, 0} : WinArgs{};
        launch_key(st, kc.bmp, static_cast<const unsigned char*>(kc.nz), kc.idx, kc.nz_offset, qh, sc, T, 1, groups, Batch_Size, ld_scores, kw,
                   prof ? g_prof.ev[4 * g_prof.n] : nullptr, prof ? g_prof.ev[4 * g_prof.n + 1] : nullptr, kc.bmp_head_stride,
                   kc.idx_head_stride, (uint32_t)kc.nz_head_stride);
    }
    const float inv_sqrt_d = (float)(1.0 / (double)sqrt_d);
    if (long_rows) {
        if (mask.ptr) long_softmax_kernel<true><<<Batch_Size, kGlueThreads, 0, st>>>(sc, T, ld_scores, window_len, window_capacity, inv_sqrt_d, window_len_extra, mask);
        else          long_softmax_kernel<false><<<Batch_Size, kGlueThreads, 0, st>>>(sc, T, ld_scores, window_len, window_capacity, inv_sqrt_d, window_len_extra, mask);
    } else {
        if (mask.ptr) window_softmax_kernel<true><<<Batch_Size, kGlueThreads, 0, st>>>(qh, ride_k ? nullptr : kwin, ride_k ? nullptr : knew, sc, T, ld_scores, window_len,
                                                                                     window_capacity, groups, inv_sqrt_d, window_len_extra, mask);
        else          window_softmax_kernel<false><<<Batch_Size, kGlueThreads, 0, st>>>(qh, ride_k ? nullptr : kwin, ride_k ? nullptr : knew, sc, T, ld_scores, window_len,
                                                                                      window_capacity, groups, inv_sqrt_d, window_len_extra, mask);
    }
    float* ws = static_cast<float*>(workspace);
    if (T > 0) {
        const int ntb = T / 64;
        const int tb_per_wg = (ntb + Split_K - 1) / Split_K;
        S = (ntb + tb_per_wg - 1) / tb_per_wg;
        const dim3 gv(S, gy);
        auto nz = static_cast<const unsigned char*>(vc.nz);
        h16* no_out = nullptr;
        uint32_t* no_flags = nullptr;
        const WinArgs vw = ride_v ? WinArgs{vwin, vnew, window_len_extra, window_len, window_capacity, 0, 0} : WinArgs{};
        if (ride_v) nwin_slabs = (window_capacity + kValueWinChunk - 1) / kValueWinChunk;
        launch_value(st, gv, vc.bmp, nz, vc.idx, vc.nz_offset, sc, no_out, ws, no_flags, T, 1, groups, Batch_Size, tb_per_wg, 0, ld_scores, vw,
                     prof ? g_prof.ev[4 * g_prof.n + 2] : nullptr, prof ? g_prof.ev[4 * g_prof.n + 3] : nullptr, vc.bmp_head_stride,
                     vc.idx_head_stride, (uint32_t)vc.nz_head_stride);
        if (prof) g_prof.n++;
    }
    value_finish_kernel<<<Batch_Size, 256, 0, st>>>(ws, S + nwin_slabs, sc, ld_scores, T, ride_v ? nullptr : vwin, ride_v ? nullptr : vnew,
                                                    window_len, window_capacity, static_cast<h16*>(out), Batch_Size, groups,
                                                    window_len_extra);
    t_last_choice = (fma_engine() == 1 && G == 4 ? 1 : 0);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" {

int mustafar_decode_attention(void* stream, const uint64_t* k_bmp, const void* k_nz, const uint32_t* k_idx,
                              const uint32_t* k_nz_offset, const uint64_t* v_bmp, const void* v_nz, const uint32_t* v_idx,
                              const uint32_t* v_nz_offset, const void* q, void* k_window, void* v_window, const void* k_new,
                              const void* v_new, int window_len, int window_capacity, void* scores, int ld_scores, void* out,
                              void* workspace, int Split_K, int T, int Batch_Size, int num_key_value_groups, float sqrt_d,
                              const int32_t* window_len_extra, const void* attention_mask, int64_t mask_row_stride,
                              int heads_per_mask_row, uint32_t flags)
{
    const mustafar_cache_view kc{const_cast<uint64_t*>(k_bmp), const_cast<void*>(k_nz), const_cast<uint32_t*>(k_idx),
                                 const_cast<uint32_t*>(k_nz_offset), 0, 0, 0};
    const mustafar_cache_view vc{const_cast<uint64_t*>(v_bmp), const_cast<void*>(v_nz), const_cast<uint32_t*>(v_idx),
                                 const_cast<uint32_t*>(v_nz_offset), 0, 0, 0};
    return decode_attention(stream, kc, vc, q, k_window, v_window, k_new, v_new, window_len, window_capacity, scores, ld_scores, out,
                            workspace, Split_K, T, Batch_Size, num_key_value_groups, sqrt_d, window_len_extra, attention_mask,
                            mask_row_stride, heads_per_mask_row, flags);
}

int mustafar_decode_attention_view(void* stream, const mustafar_cache_view* k_cache, const mustafar_cache_view* v_cache,
                                   const void* q, void* k_window, void* v_window, const void* k_new, const void* v_new,
                                   int window_len, int window_capacity, void* scores, int ld_scores, void* out, void* workspace,
                                   int Split_K, int T, int Batch_Size, int num_key_value_groups, float sqrt_d,
                                   const int32_t* window_len_extra, const void* attention_mask, int64_t mask_row_stride,
                                   int heads_per_mask_row, uint32_t flags)
{
    const mustafar_cache_view none{nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
    if (T > 0 && (!k_cache || !v_cache)) return MUSTAFAR_EINVAL;
    return decode_attention(stream, k_cache ? *k_cache : none, v_cache ? *v_cache : none, q, k_window, v_window, k_new, v_new,
                            window_len, window_capacity, scores, ld_scores, out, workspace, Split_K, T, Batch_Size,
                            num_key_value_groups, sqrt_d, window_len_extra, attention_mask, mask_row_stride, heads_per_mask_row, flags);
}

int mustafar_decode_attention_extents(void* stream, const mustafar_cache_view* k_base, const mustafar_cache_view* v_base, int T_base,
                                      const mustafar_cache_view* k_extents, const mustafar_cache_view* v_extents,
                                      const void* q, void* k_window, void* v_window, const void* k_new, const void* v_new,
                                      int window_len, int window_capacity, void* scores, int ld_scores, void* out, void* workspace,
                                      int Split_K, int T, int Batch_Size, int num_key_value_groups, float sqrt_d,
                                      const int32_t* window_len_extra, const void* attention_mask, int64_t mask_row_stride,
                                      int heads_per_mask_row, uint32_t flags, const int32_t* T_device)
{
    if (!k_base || !v_base || T_base <= 0 || (T_base & 255) || T < T_base || ((T - T_base) & 255)) return MUSTAFAR_EINVAL;
    if (T_device && T == T_base) return MUSTAFAR_EINVAL;   // (a capacity of exactly the base tokens: nothing to grow into)
    if (T == T_base)   // no appended extent: the plain call
        return decode_attention(stream, *k_base, *v_base, q, k_window, v_window, k_new, v_new, window_len, window_capacity, scores, ld_scores,
                                out, workspace, Split_K, T, Batch_Size, num_key_value_groups, sqrt_d, window_len_extra, attention_mask,
                                mask_row_stride, heads_per_mask_row, flags);
    if (!k_extents || !v_extents || k_base->nz_head_stride == 0 || v_base->nz_head_stride == 0) return MUSTAFAR_EINVAL;
    return decode_attention(stream, *k_base, *v_base, q, k_window, v_window, k_new, v_new, window_len, window_capacity, scores, ld_scores,
                            out, workspace, Split_K, T, Batch_Size, num_key_value_groups, sqrt_d, window_len_extra, attention_mask,
                            mask_row_stride, heads_per_mask_row, flags, k_extents, v_extents, T_base, T_device);
}

int mustafar_decode_reads_extents(int num_key_value_groups, int ld_scores, uint32_t flags)
{
    const uint32_t f_eng = flags & 7u, f_str = (flags >> 4) & 3u;
    if (f_eng > 3u || f_str > 2u || (flags & ~0x37u)) return 0;
    if (f_str == 1u || (f_str == 0u && onepass_mode() == 0)) return 0;   // two launches asked for (by the call or by the process default)
    (void)onepass_target_wgs(true);   // (reads MUSTAFAR_ONEPASS_WGS: the shape rule below depends on it)
    // rows so long that four blocks per workgroup leave more slabs than the row kernel folds (T > ~127 k tokens; ld_scores >= T
    // stands in for T): decode_attention then gives a workgroup more blocks, which may straddle extents -> not served
    if (((ld_scores / 64 + 3) / 4) + kMaxWindow / kOneWinChunk > kMaxSlabs) return 0;
    return num_key_value_groups >= 1 && onepass_lean() == 2 && (ld_scores & 31) == 0 && g_onepass_wgs <= 0 &&
           (onepass_lean_tbw() == 0 || onepass_lean_tbw() == 1 || onepass_lean_tbw() == 2);
}

int mustafar_profile_begin(int max_records)
{
    if (g_prof.ev || max_records < 1) return MUSTAFAR_EINVAL;
    g_prof.ev = new hipEvent_t[4 * (size_t)max_records];
    for (int i = 0; i < 4 * max_records; i++)
        if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) {   // release what exists: a failed begin leaves no state behind
            const int err = (int)hipGetLastError();
            for (int j = 0; j < i; j++) (void)hipEventDestroy(g_prof.ev[j]);
            delete[] g_prof.ev;
            g_prof = Profile();
            return err ? err : MUSTAFAR_EINVAL;
        }
    g_prof.cap = max_records;
    g_prof.n = 0;
    g_prof.on = true;
    return 0;
}

int mustafar_profile_end(double* key_us_avg, double* value_us_avg, int* records)
{
    if (!g_prof.ev) return MUSTAFAR_EINVAL;
    g_prof.on = false;
    double k = 0, v = 0;
    const bool one = g_prof.onepass == g_prof.n && g_prof.n > 0;   // every record is a one-pass launch: no value launch exists
    for (int i = 0; i < g_prof.n; i++) {
        float ms = 0;
        (void)hipEventSynchronize(g_prof.ev[4 * i + (one ? 1 : 3)]);
        (void)hipEventElapsedTime(&ms, g_prof.ev[4 * i], g_prof.ev[4 * i + 1]);
        k += ms * 1e3;
        if (!one) {
            (void)hipEventElapsedTime(&ms, g_prof.ev[4 * i + 2], g_prof.ev[4 * i + 3]);
            v += ms * 1e3;
        }
    }
    if (records) *records = g_prof.n;
    if (key_us_avg) *key_us_avg = g_prof.n ? k / g_prof.n : 0;
    if (value_us_avg) *value_us_avg = g_prof.n ? v / g_prof.n : 0;
    for (int i = 0; i < 4 * g_prof.cap; i++) (void)hipEventDestroy(g_prof.ev[i]);
    delete[] g_prof.ev;
    g_prof = Profile();
    return 0;
}


int mustafar_profile_end2(double* key_us_avg, double* value_us_avg, double* finish_us_avg, int* records)
{
    if (!g_prof.ev) return MUSTAFAR_EINVAL;
    double f = 0;
    const bool have = g_prof.finish == g_prof.n && g_prof.n > 0;   // every record is a one-pass launch of a lean form: pair 2 = its row kernel
    if (have) {
        for (int i = 0; i < g_prof.n; i++) {
            float ms = 0;
            (void)hipEventSynchronize(g_prof.ev[4 * i + 3]);
            (void)hipEventElapsedTime(&ms, g_prof.ev[4 * i + 2], g_prof.ev[4 * i + 3]);
            f += ms * 1e3;
        }
    }
    if (finish_us_avg) *finish_us_avg = have ? f / g_prof.n : 0;
    return mustafar_profile_end(key_us_avg, value_us_avg, records);
}


int mustafar_counter_add(void* stream, int32_t* counter, int delta)
{
    if (!counter) return MUSTAFAR_EINVAL;
    counter_add_kernel<<<1, 64, 0, static_cast<hipStream_t>(stream)>>>(counter, delta);
    return (int)hipGetLastError();
}


int mustafar_set_fma_engine(int engine)
{
    if (engine < 0 || engine > 2) return MUSTAFAR_EINVAL;
    g_engine = engine;
    return 0;
}

int mustafar_get_fma_engine(void) { return fma_engine(); }

int mustafar_set_onepass(int mode)
{
    if (mode < 0 || mode > 2) return MUSTAFAR_EINVAL;
    g_onepass = mode;
    return 0;
}

int mustafar_get_onepass(void) { return onepass_mode(); }

int mustafar_last_decode_choice(void) { return t_last_choice; }

// Tuning knobs of the experiment scripts (tools/): 0 = lean one-pass form on / off, 1 = blocks per wave of the lean form
// (0 = automatic), 2 = workgroup target of the pair form (0 = automatic).  Not part of the operator interface.
int mustafar_tune(int knob, int value)
{
    if (value < 0) return MUSTAFAR_EINVAL;
    switch (knob) {
        case 0: g_lean = value > 2 ? 2 : value; return 0;
        case 1: g_lean_tbw = value; return 0;
        case 2: g_onepass_wgs = value; return 0;
        case 3: g_lean_win_last = value ? 1 : 0; return 0;
        case 4: g_pair_slabs = value ? 1 : 0; return 0;
        case 6: g_key_lean = value ? 1 : 0; return 0;
        case 7: g_value_lean = value > 1 ? 2 : value ? 1 : 0; return 0;   // (2: by size, the default)
        case 8: g_sb = value ? 1 : 0; return 0;
        case 9: g_late_prio = value ? 1 : 0; return 0;
        case 10: g_finish1 = value ? 1 : 0; return 0;
        case 12: g_spec_k_bytes = value < 0 ? 0 : value; return 0;
        case 13: g_value_lean8 = value ? 1 : 0; return 0;   // (round 6: 1 = the value entry point's 8-row calls on the lean form with pad workgroups behind the row-0 workgroups, 0 = round 1's kernel)
        case 11: g_small = value < 0 ? 0 : value > 2 ? 2 : value; return 0;   // (round 6: 0 = never the small-launch kernel, 1 = below one wave per SIMD, 2 = for every launch of two blocks per workgroup)
        default: return MUSTAFAR_EINVAL;
    }
}

#ifdef MUSTAFAR_WAVE_TRACE
// Tool-only (tools/wave_trace.py): records go to `buf` (4 x u64 each, `cap` slots; zero it first).
int mustafar_trace_set(void* buf, unsigned int cap)
{
    unsigned long long* b = static_cast<unsigned long long*>(buf);
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_trace_buf), &b, sizeof(b)) != hipSuccess) return (int)hipGetLastError();
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_trace_cap), &cap, sizeof(cap)) != hipSuccess) return (int)hipGetLastError();
    return 0;
}
#endif

}  // extern "C"
