// Round-2 probes on gfx950 (design inputs for the SpMV inner loop, second batch):
//   1. what ds_read_u16_d16 / _d16_hi leave in the other half of the destination register
//   2. where buffer_load_dwordx4 ... lds (LDS-DMA) puts its bytes: M0 base, instruction offset, out-of-range lanes
//   3. issue rates: v_mfma_f32_4x4x4_16B_f16 alone and beside slow VALU, v_or_b32 / v_lshl_or_b32 / v_mov const,
//      ds_read_b128 / ds_read_b64 with wave-uniform (broadcast) addresses, ds_read_u16 gathers beside them
// Build: hipcc --offload-arch=gfx950 -O2 -o probe2 probe2.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %d\n", (int)e_, __LINE__); return 1; } } while (0)

// ---- 1. d16 loads ---------------------------------------------------------------------------------------------
__global__ void k_d16(uint32_t* out)
{
    __shared__ uint16_t lds[128];
    lds[threadIdx.x] = 0x1100 + threadIdx.x;
    lds[threadIdx.x + 64] = 0x2200 + threadIdx.x;
    __syncthreads();
    uint32_t a = 0xAAAABBBBu, b = 0xAAAABBBBu, c = 0xAAAABBBBu;
    const uint32_t addr = (uint32_t)(uintptr_t)lds + threadIdx.x * 2;
    asm volatile("ds_read_u16_d16 %0, %3\n\tds_read_u16_d16_hi %1, %3\n\tds_read_u16 %2, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "+v"(a), "+v"(b), "+v"(c) : "v"(addr));
    // both halves into ONE register: low first, then high
    uint32_t d = 0xAAAABBBBu;
    asm volatile("ds_read_u16_d16 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tds_read_u16_d16_hi %0, %1 offset:128\n\ts_waitcnt lgkmcnt(0)"
                 : "+v"(d) : "v"(addr));
    out[threadIdx.x * 4 + 0] = a;
    out[threadIdx.x * 4 + 1] = b;
    out[threadIdx.x * 4 + 2] = c;
    out[threadIdx.x * 4 + 3] = d;
}

// ---- 2. LDS-DMA ----------------------------------------------------------------------------------------------
__global__ void k_dma(const unsigned char* src, uint32_t len, uint32_t* out)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[8192];
    for (int i = threadIdx.x; i < 2048; i += 64) reinterpret_cast<uint32_t*>(lds)[i] = 0xEEEEEEEEu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), (short)0, (int)len, 0x00020000);
    // A: base lds+0, voffset lane*16, no immediate      -> expect src[lane*16..] at lds[lane*16]
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, threadIdx.x * 16, 0, 0, 0);
    // B: base lds+2048, immediate offset 1024            -> where does src[1024 + lane*16] land: 2048 or 3072?
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + 2048), 16, threadIdx.x * 16, 0, 1024, 0);
    // C: base lds+5120, soffset 256 (scalar)             -> src[256 + lane*16] at 5120?
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + 5120), 16, threadIdx.x * 16, 256, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = reinterpret_cast<uint32_t*>(lds)[i];
}

// ---- 3. issue rates -----------------------------------------------------------------------------------------
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define REP8(x) x x x x x x x x

#define RATE_KERNEL(name, decl, body, sink)                                                   \
    __global__ __launch_bounds__(256) void name(float* out, int iters, uint64_t* clk)         \
    {                                                                                         \
        __shared__ __attribute__((aligned(16))) uint32_t lds[4096];                           \
        for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 2654435761u;               \
        __syncthreads();                                                                      \
        decl                                                                                  \
        uint64_t t0 = __builtin_amdgcn_s_memtime();                                           \
        for (int i = 0; i < iters; i++) { body }                                              \
        uint64_t t1 = __builtin_amdgcn_s_memtime();                                           \
        out[blockIdx.x * 256 + threadIdx.x] = sink;                                           \
        if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;                                      \
    }

#define DECL_MFMA                                                                                          \
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;                                                     \
    h16x4 a = {(_Float16)threadIdx.x, (_Float16)1, (_Float16)2, (_Float16)3}, b = a;                        \
    unsigned long long m = 0x123456789abcdefULL + blockIdx.x;                                              \
    uint32_t x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;                                                      \
    const uint32_t la = (uint32_t)(uintptr_t)lds + (threadIdx.x & 63) * 2, lu = (uint32_t)(uintptr_t)lds; \
    uint32_t g0 = 0, g1 = 0, g2 = 0, g3 = 0; uint4 w0 = {0, 0, 0, 0}; uint2 w1 = {0, 0};
#define SINK_MFMA (c0[0] + c1[1] + c2[2] + c3[3] + (float)(x0 + x1 + x2 + x3 + g0 + g1 + g2 + g3 + w0.x + w0.w + w1.x + w1.y))

#define MF4 asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %4, %5, %0\n v_mfma_f32_4x4x4_16b_f16 %1, %4, %5, %1\n"  \
                         "v_mfma_f32_4x4x4_16b_f16 %2, %4, %5, %2\n v_mfma_f32_4x4x4_16b_f16 %3, %4, %5, %3"    \
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
#define MF1 asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
#define MB4 asm volatile("v_mbcnt_lo_u32_b32 %0, %4, 0\n v_mbcnt_hi_u32_b32 %0, %5, %0\n v_mbcnt_lo_u32_b32 %1, %4, 0\n v_mbcnt_hi_u32_b32 %1, %5, %1" \
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"((unsigned)m), "s"((unsigned)(m >> 32)));
#define OR4 asm volatile("v_or_b32 %0, %4, %0\n v_or_b32 %1, %4, %1\n v_or_b32 %2, %4, %2\n v_or_b32 %3, %4, %3" \
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(g0));
#define LOR4 asm volatile("v_lshl_or_b32 %0, %4, 16, %0\n v_lshl_or_b32 %1, %4, 16, %1\n v_lshl_or_b32 %2, %4, 16, %2\n v_lshl_or_b32 %3, %4, 16, %3" \
                          : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(g0));
#define MOVC4 asm volatile("v_mov_b32 %0, 0\n v_mov_b32 %1, 0\n v_mov_b32 %2, 0\n v_mov_b32 %3, 0" : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3));
#define ADDVV4 asm volatile("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
#define G4 asm volatile("ds_read_u16 %0, %4\n ds_read_u16 %1, %4 offset:130\n ds_read_u16 %2, %4 offset:260\n ds_read_u16 %3, %4 offset:390\n s_waitcnt lgkmcnt(0)" \
                        : "=v"(g0), "=v"(g1), "=v"(g2), "=v"(g3) : "v"(la));
#define B128U asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(w0) : "v"(lu));
#define B64Q asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(w1) : "v"(lu + (threadIdx.x & 3) * 256));
#define B128U4 asm volatile("ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:16\n ds_read_b128 %0, %1 offset:32\n ds_read_b128 %0, %1 offset:48\n s_waitcnt lgkmcnt(0)" : "=v"(w0) : "v"(lu));
#define B64Q4 asm volatile("ds_read_b64 %0, %1\n ds_read_b64 %0, %1 offset:8\n ds_read_b64 %0, %1 offset:16\n ds_read_b64 %0, %1 offset:24\n s_waitcnt lgkmcnt(0)" : "=v"(w1) : "v"(lu + (threadIdx.x & 3) * 256));

RATE_KERNEL(r_mfma, DECL_MFMA, REP8(MF4 MF4), SINK_MFMA)                         // 64 mfma
RATE_KERNEL(r_mfma_chain, DECL_MFMA, REP8(MF1 MF1 MF1 MF1 MF1 MF1 MF1 MF1), SINK_MFMA)   // 64 dependent mfma
RATE_KERNEL(r_mb, DECL_MFMA, REP8(MB4 MB4), SINK_MFMA)                           // 64 mbcnt
RATE_KERNEL(r_mb_mfma_8_1, DECL_MFMA, REP8(MB4 MB4 MF1), SINK_MFMA)              // 64 mbcnt + 8 mfma
RATE_KERNEL(r_mb_mfma_4_1, DECL_MFMA, REP8(MB4 MF1 MB4 MF1), SINK_MFMA)          // 64 mbcnt + 16 mfma
RATE_KERNEL(r_or, DECL_MFMA, REP8(OR4 OR4), SINK_MFMA)
RATE_KERNEL(r_lor, DECL_MFMA, REP8(LOR4 LOR4), SINK_MFMA)
RATE_KERNEL(r_movc, DECL_MFMA, REP8(MOVC4 MOVC4), SINK_MFMA)
RATE_KERNEL(r_addvv, DECL_MFMA, REP8(ADDVV4 ADDVV4), SINK_MFMA)
RATE_KERNEL(r_gather, DECL_MFMA, REP8(G4 G4), SINK_MFMA)                         // 64 ds_read_u16
RATE_KERNEL(r_b128u, DECL_MFMA, REP8(B128U4 B128U4), SINK_MFMA)                  // 64 uniform ds_read_b128
RATE_KERNEL(r_b64q, DECL_MFMA, REP8(B64Q4 B64Q4), SINK_MFMA)                     // 64 ds_read_b64, 4 distinct addresses
RATE_KERNEL(r_gather_b128u, DECL_MFMA, REP8(G4 B128U G4 B128U), SINK_MFMA)       // 64 gathers + 16 uniform b128
RATE_KERNEL(r_mb_gather, DECL_MFMA, REP8(MB4 G4 MB4 G4), SINK_MFMA)              // 64 mbcnt + 64 gathers

typedef void (*kern_t)(float*, int, uint64_t*);

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);

    {   // 1
        uint32_t* d;
        CHECK(hipMalloc(&d, 64 * 4 * 4));
        k_d16<<<1, 64>>>(d);
        std::vector<uint32_t> h(256);
        CHECK(hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost));
        printf("d16 probe (register preset 0xAAAABBBB, LDS half = 0x1100+lane / 0x2200+lane):\n");
        for (int l : {0, 5, 63})
            printf("  lane %2d: ds_read_u16_d16 -> %08x   ds_read_u16_d16_hi -> %08x   ds_read_u16 -> %08x   d16 then d16_hi into one reg -> %08x\n",
                   l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
        CHECK(hipFree(d));
    }
    {   // 2
        std::vector<unsigned char> src(4096);
        for (int i = 0; i < 4096; i++) src[i] = (unsigned char)(i / 16);   // byte value = 16-byte granule index (mod 256)
        unsigned char* ds;
        uint32_t* dout;
        CHECK(hipMalloc(&ds, 4096));
        CHECK(hipMalloc(&dout, 8192));
        CHECK(hipMemcpy(ds, src.data(), 4096, hipMemcpyHostToDevice));
        const uint32_t len = 1024 + 40 * 16;   // granules >= 104 are out of range
        k_dma<<<1, 64>>>(ds, len, dout);
        std::vector<uint32_t> h(2048);
        CHECK(hipMemcpy(h.data(), dout, 8192, hipMemcpyDeviceToHost));
        printf("LDS-DMA probe (num_records = %u bytes; source byte = granule index; LDS prefilled 0xEE):\n", len);
        int run_start = 0;
        for (int g = 1; g <= 512; g++) {   // print runs of 16-byte granules with the same "kind"
            auto kind = [&](int gg) -> long { uint32_t v = h[gg * 4]; return v == 0xEEEEEEEEu ? -1 : (v == 0 ? -2 : (long)(v & 0xff) - gg); };
            if (g == 512 || kind(g) != kind(run_start)) {
                const uint32_t v = h[run_start * 4];
                printf("  lds granules [%3d, %3d): %s first word %08x\n", run_start, g,
                       v == 0xEEEEEEEEu ? "untouched" : v == 0 ? "zeros" : "data", v);
                run_start = g;
            }
        }
        CHECK(hipFree(ds));
        CHECK(hipFree(dout));
    }
    {   // 3
        float* out;
        uint64_t* clk;
        const int wgs = cus * 8;   // 32 waves per CU
        CHECK(hipMalloc(&out, (size_t)wgs * 256 * 4));
        CHECK(hipMalloc(&clk, (size_t)wgs * 8));
        struct { const char* name; kern_t k; int per_iter; } ks[] = {
            {"mfma_4x4x4 (4 accumulators)", r_mfma, 64}, {"mfma_4x4x4 (dependent chain)", r_mfma_chain, 64},
            {"mbcnt", r_mb, 64}, {"mbcnt + mfma 8:1 (72/iter)", r_mb_mfma_8_1, 72}, {"mbcnt + mfma 4:1 (80/iter)", r_mb_mfma_4_1, 80},
            {"v_or_b32 (vv)", r_or, 64}, {"v_lshl_or_b32", r_lor, 64}, {"v_mov_b32 v, 0", r_movc, 64}, {"v_add_u32 (vvv same)", r_addvv, 64},
            {"ds_read_u16 gather", r_gather, 64}, {"ds_read_b128 uniform address", r_b128u, 64}, {"ds_read_b64 4 addresses", r_b64q, 64},
            {"gather + uniform b128 4:1 (80/iter)", r_gather_b128u, 80}, {"mbcnt + gather 1:1 (128/iter)", r_mb_gather, 128},
        };
        const int iters = 2000;
        for (auto& e : ks) {
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            e.k<<<wgs, 256>>>(out, 10, clk);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            e.k<<<wgs, 256>>>(out, iters, clk);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double insts = (double)wgs * 4 * iters * e.per_iter;   // wave-instructions
            printf("waves/CU 32  %-38s %8.3f ms  %6.2f wave-inst/ns/CU\n", e.name, ms, insts / (ms * 1e6) / cus);
        }
        CHECK(hipFree(out));
        CHECK(hipFree(clk));
    }
    return 0;
}
