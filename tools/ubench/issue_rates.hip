// Micro-benchmark of instruction issue rates on gfx950 (design input for the SpMV inner loop).
// Build: hipcc --offload-arch=gfx950 -O2 -o issue_rates issue_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

#define KERNEL(name, body)                                                                          \
    __global__ __launch_bounds__(256) void name(float* out, int iters, uint64_t* clk)                \
    {                                                                                               \
        float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f; \
        unsigned s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;                                            \
        unsigned long long m = 0x123456789abcdefULL + blockIdx.x;                                    \
        uint64_t t0 = __builtin_amdgcn_s_memtime();                                                  \
        for (int i = 0; i < iters; i++) { body }                                                     \
        uint64_t t1 = __builtin_amdgcn_s_memtime();                                                  \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + s0 + s1 + s2 + s3 + (float)m; \
        if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;                                             \
    }

#define V4 asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
#define V4B asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#define S4 asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");
#define MIX4 asm volatile("v_fma_mix_f32 %0, %0, %4, %0 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %1, %1, %4, %1 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n v_fma_mix_f32 %2, %2, %4, %2 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %3, %3, %4, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s0));
#define MB4 asm volatile("v_mbcnt_lo_u32_b32 %0, %4, 0\n v_mbcnt_hi_u32_b32 %0, %5, %0\n v_mbcnt_lo_u32_b32 %1, %4, 0\n v_mbcnt_hi_u32_b32 %1, %5, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"((unsigned)m), "s"((unsigned)(m >> 32)));
#define CND4 asm volatile("v_cndmask_b32 %0, 0, %0, %4\n v_cndmask_b32 %1, 0, %1, %4\n v_cndmask_b32 %2, 0, %2, %4\n v_cndmask_b32 %3, 0, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(m));
#define DOT4 asm volatile("v_dot2_f32_f16 %0, %4, %5, %0\n v_dot2_f32_f16 %1, %4, %5, %1\n v_dot2_f32_f16 %2, %4, %5, %2\n v_dot2_f32_f16 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "s"(s0));
#define DOTC4 asm volatile("v_dot2c_f32_f16 %0, %4, %5\n v_dot2c_f32_f16 %1, %4, %5\n v_dot2c_f32_f16 %2, %4, %5\n v_dot2c_f32_f16 %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));
#define PK4 asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1" : "+v"(d0), "+v"(d1));
#define LSH4 asm volatile("v_lshl_add_u32 %0, %0, 1, %4\n v_lshl_add_u32 %1, %1, 1, %4\n v_lshl_add_u32 %2, %2, 1, %4\n v_lshl_add_u32 %3, %3, 1, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s0));
#define BREV4 asm volatile("s_brev_b64 %0, %0\n s_brev_b64 %0, %0\n s_brev_b64 %0, %0\n s_brev_b64 %0, %0" : "+s"(m));

#define CVT4 asm volatile("v_cvt_f32_f16 %0, %4\n v_cvt_f32_f16 %1, %4\n v_cvt_f32_f16 %2, %4\n v_cvt_f32_f16 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));
#define AND4 asm volatile("v_and_b32 %0, %4, %0\n v_and_b32 %1, %4, %1\n v_and_b32 %2, %4, %2\n v_and_b32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));
#define ADDU4 asm volatile("v_add_u32 %0, %4, %0\n v_add_u32 %1, %4, %1\n v_add_u32 %2, %4, %2\n v_add_u32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));
#define FMAS4 asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s0));
#define FMAC4 asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));
#define MOV4 asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));
#define PKF4 asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1" : "+v"(d0), "+v"(d1));
#define MADMIX4 asm volatile("v_fma_mix_f32 %0, %4, %5, %0 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %1, %4, %5, %1 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %2, %4, %5, %2 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %3, %4, %5, %3 op_sel_hi:[1,1,0]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));
#define EXEC4 asm volatile("s_mov_b64 exec, %0\n s_mov_b64 exec, -1\n s_mov_b64 exec, %0\n s_mov_b64 exec, -1" :: "s"(m|1ull));
#define MBV4 asm volatile("v_mbcnt_lo_u32_b32 %0, %4, %0\n v_mbcnt_hi_u32_b32 %0, %5, %0\n v_mbcnt_lo_u32_b32 %1, %4, %1\n v_mbcnt_hi_u32_b32 %1, %5, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));
#define LSHV4 asm volatile("v_lshl_add_u32 %0, %0, 1, %4\n v_lshl_add_u32 %1, %1, 1, %4\n v_lshl_add_u32 %2, %2, 1, %4\n v_lshl_add_u32 %3, %3, 1, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));
#define CNDV4 asm volatile("v_cndmask_b32 %0, %4, %0, vcc\n v_cndmask_b32 %1, %4, %1, vcc\n v_cndmask_b32 %2, %4, %2, vcc\n v_cndmask_b32 %3, %4, %3, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4) : "vcc");
#define ANDS4 asm volatile("v_and_b32 %0, %4, %0\n v_and_b32 %1, %4, %1\n v_and_b32 %2, %4, %2\n v_and_b32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s0));
#define BCNT4 asm volatile("v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %4, %1\n v_bcnt_u32_b32 %2, %4, %2\n v_bcnt_u32_b32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));
#define DOTV4 asm volatile("v_dot2_f32_f16 %0, %4, %5, %0\n v_dot2_f32_f16 %1, %4, %5, %1\n v_dot2_f32_f16 %2, %4, %5, %2\n v_dot2_f32_f16 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));
#define LSHR4 asm volatile("v_lshrrev_b32 %0, %4, %0\n v_lshrrev_b32 %1, %4, %1\n v_lshrrev_b32 %2, %4, %2\n v_lshrrev_b32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));
#define LSHR64 asm volatile("v_lshrrev_b64 %0, %2, %0\n v_lshrrev_b64 %1, %2, %1\n v_lshrrev_b64 %0, %2, %0\n v_lshrrev_b64 %1, %2, %1" : "+v"(dd0), "+v"(dd1) : "v"(a4));
#define MULF4 asm volatile("v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %4, %1\n v_mul_f32 %2, %4, %2\n v_mul_f32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s0));
#define PERM4 asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));
#define MAD24 asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));
#define CMP4 asm volatile("v_cmp_ne_u32 vcc, %0, %1\n v_cmp_ne_u32 vcc, %1, %2\n v_cmp_ne_u32 vcc, %2, %3\n v_cmp_ne_u32 vcc, %3, %0" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");
#define READL4 asm volatile("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %4, 5\n v_readlane_b32 %2, %4, 7\n v_readlane_b32 %3, %4, 9" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a4));
KERNEL(k_mbv, REP8(MBV4 MBV4))
KERNEL(k_lshv, REP8(LSHV4 LSHV4))
KERNEL(k_cndv, REP8(CNDV4 CNDV4))
KERNEL(k_ands, REP8(ANDS4 ANDS4))
KERNEL(k_bcnt, REP8(BCNT4 BCNT4))
KERNEL(k_dotv, REP8(DOTV4 DOTV4))
KERNEL(k_lshr, REP8(LSHR4 LSHR4))
KERNEL(k_mulfs, REP8(MULF4 MULF4))
KERNEL(k_perm, REP8(PERM4 PERM4))
KERNEL(k_mad24, REP8(MAD24 MAD24))
KERNEL(k_cmp, REP8(CMP4 CMP4))
KERNEL(k_readl, REP8(READL4 READL4))
__global__ __launch_bounds__(256) void k_lshr64(float* out, int iters, uint64_t* clk)
{
    unsigned long long dd0 = threadIdx.x * 0x9e3779b97f4a7c15ull, dd1 = ~dd0; float a4 = 1;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) { REP8(LSHR64 LSHR64) }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = (float)(dd0 + dd1);
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
#define MOV64x4 asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %0\n v_mov_b64 %0, %1\n v_mov_b64 %1, %0" : "+v"(dd0), "+v"(dd1));
__global__ __launch_bounds__(256) void k_mov64(float* out, int iters, uint64_t* clk)   // 64 v_mov_b64 / iter (two registers each)
{
    unsigned long long dd0 = threadIdx.x * 0x9e3779b97f4a7c15ull, dd1 = ~dd0;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) { REP8(MOV64x4 MOV64x4) }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = (float)(dd0 + dd1);
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
#define PKMOVx4 asm volatile("v_pk_mov_b32 %0, %1, %1\n v_pk_mov_b32 %1, %0, %0\n v_pk_mov_b32 %0, %1, %1\n v_pk_mov_b32 %1, %0, %0" : "+v"(dd0), "+v"(dd1));
__global__ __launch_bounds__(256) void k_pkmov(float* out, int iters, uint64_t* clk)   // 64 v_pk_mov_b32 / iter
{
    unsigned long long dd0 = threadIdx.x * 0x9e3779b97f4a7c15ull, dd1 = ~dd0;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) { REP8(PKMOVx4 PKMOVx4) }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = (float)(dd0 + dd1);
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
KERNEL(k_valu, REP8(V4 V4B))
KERNEL(k_cvt, REP8(CVT4 CVT4))
KERNEL(k_and, REP8(AND4 AND4))
KERNEL(k_addu, REP8(ADDU4 ADDU4))
KERNEL(k_fmas, REP8(FMAS4 FMAS4))
KERNEL(k_fmac, REP8(FMAC4 FMAC4))
KERNEL(k_mov, REP8(MOV4 MOV4))
KERNEL(k_mixvv, REP8(MADMIX4 MADMIX4))
KERNEL(k_slow_s11, REP8(MB4 S4 MB4 S4))
KERNEL(k_slow_s21, REP8(MB4 MB4 S4))
KERNEL(k_slow_fast, REP8(MB4 V4 MB4 V4B))
KERNEL(k_exec, REP8(EXEC4 EXEC4))
__global__ __launch_bounds__(256) void k_pkf(float* out, int iters, uint64_t* clk)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 d0 = {(float)threadIdx.x, 1.f}, d1 = {2.f, 3.f};
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) { REP8(PKF4 PKF4) }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = d0.x + d0.y + d1.x + d1.y;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}                       // 64 VALU / iter
KERNEL(k_salu, REP8(S4 S4))                        // 64 SALU / iter
KERNEL(k_mix11, REP8(V4 S4 V4B S4))                // 64 VALU + 64 SALU
KERNEL(k_mix21, REP8(V4 V4B S4))                   // 64 VALU + 32 SALU
KERNEL(k_fmamix, REP8(MIX4 MIX4))                  // 64 fma_mix
KERNEL(k_mbcnt, REP8(MB4 MB4))                     // 64 mbcnt
KERNEL(k_cnd, REP8(CND4 CND4))                     // 64 cndmask with SGPR mask
KERNEL(k_dot2, REP8(DOT4 DOT4))                    // 64 dot2
KERNEL(k_dot2c, REP8(DOTC4 DOTC4))                 // 64 dot2c
KERNEL(k_lshl, REP8(LSH4 LSH4))                    // 64 lshl_add
KERNEL(k_brev, REP8(BREV4 BREV4))                  // 64 s_brev_b64

typedef void (*kern_t)(float*, int, uint64_t*);

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
    struct K { const char* name; kern_t fn; int valu, salu; };
    std::vector<K> ks = {{"valu_fma", k_valu, 64, 0}, {"salu_add", k_salu, 0, 64}, {"mix 1:1", k_mix11, 64, 64},
                         {"mix 2:1", k_mix21, 64, 32}, {"fma_mix(sgpr)", k_fmamix, 64, 0}, {"mbcnt", k_mbcnt, 64, 0},
                         {"cndmask(smask)", k_cnd, 64, 0}, {"dot2_f32_f16", k_dot2, 64, 0}, {"dot2c_f32_f16", k_dot2c, 64, 0},
                         {"lshl_add", k_lshl, 64, 0}, {"s_brev_b64", k_brev, 0, 64},
                         {"cvt_f32_f16", k_cvt, 64, 0}, {"v_and", k_and, 64, 0}, {"v_add_u32", k_addu, 64, 0}, {"fma_f32(sgpr)", k_fmas, 64, 0},
                         {"fmac_f32", k_fmac, 64, 0}, {"v_mov", k_mov, 64, 0}, {"fma_mix(vv)", k_mixvv, 64, 0}, {"pk_fma_f32", k_pkf, 64, 0},
                         {"mbcnt+salu 1:1", k_slow_s11, 64, 64}, {"mbcnt+salu 2:1", k_slow_s21, 64, 32}, {"mbcnt+fma 1:1", k_slow_fast, 128, 0},
                         {"s_mov exec", k_exec, 0, 64},
     {"mbcnt(vgpr mask)", k_mbv, 64, 0}, {"lshl_add(vvv)", k_lshv, 64, 0}, {"cndmask(vcc)", k_cndv, 64, 0}, {"v_and(sgpr)", k_ands, 64, 0},
     {"v_bcnt", k_bcnt, 64, 0}, {"dot2(vvv)", k_dotv, 64, 0}, {"v_lshrrev_b32", k_lshr, 64, 0}, {"v_lshrrev_b64", k_lshr64, 64, 0},
     {"v_mul_f32(sgpr)", k_mulfs, 64, 0}, {"v_perm_b32", k_perm, 64, 0}, {"v_mad_u32_u24", k_mad24, 64, 0}, {"v_cmp->vcc", k_cmp, 64, 0},
     {"v_readlane", k_readl, 64, 0}, {"v_mov_b64", k_mov64, 64, 0}, {"v_pk_mov_b32", k_pkmov, 64, 0}};
    float* out; uint64_t* clk;
    const int iters = 2000;
    for (int wpc : {32}) {           // waves per CU (256-thread blocks -> 4 waves each)
        const int blocks = cus * wpc / 4;
        hipMalloc(&out, blocks * 256 * sizeof(float));
        hipMalloc(&clk, blocks * sizeof(uint64_t));
        for (auto& k : ks) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            k.fn<<<blocks, 256>>>(out, 10, clk);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            k.fn<<<blocks, 256>>>(out, iters, clk);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<uint64_t> h(blocks);
            hipMemcpy(h.data(), clk, blocks * sizeof(uint64_t), hipMemcpyDeviceToHost);
            double cyc = 0; for (auto c : h) cyc += c; cyc /= blocks;     // s_memtime ticks (100 MHz? or core clk)
            const double insts = (double)(k.valu + k.salu) * iters * wpc;   // wave-instructions per CU
            printf("waves/CU %2d  %-16s  %8.3f ms  %7.2f wave-inst/ns/CU  ticks/iter %8.1f  (valu %d salu %d per iter)\n", wpc, k.name, ms,
                   insts / (ms * 1e6), cyc / iters, k.valu, k.salu);
        }
        hipFree(out); hipFree(clk);
    }
    return 0;
}
