// Throughput of 2-byte gathers from an L1-resident window: ds_read_u16 (LDS) vs buffer_load_ushort idxen (TA / L1).
// Build: hipcc --offload-arch=gfx950 -O2 -o gather_rate gather_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_lds(const uint16_t* src, float* out, int iters)
{
    __shared__ uint16_t win[4][2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = lane; i < 2048; i += 64) win[wave][i] = src[blockIdx.x * 2048 + i];
    __syncthreads();
    uint32_t base = (uint32_t)reinterpret_cast<uintptr_t>(&win[wave][0]);
    uint32_t idx = (lane * 7) & 63, acc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
        uint32_t a = base + 2 * idx + ((it & 15) << 7);
        asm volatile("ds_read_u16 %0, %8\n ds_read_u16 %1, %8 offset:128\n ds_read_u16 %2, %8 offset:256\n ds_read_u16 %3, %8 offset:384\n"
                     "ds_read_u16 %4, %8 offset:512\n ds_read_u16 %5, %8 offset:640\n ds_read_u16 %6, %8 offset:768\n ds_read_u16 %7, %8 offset:896\n"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7) : "v"(a));
        acc += t0 + t1 + t2 + t3 + t4 + t5 + t6 + t7;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_buf(const uint16_t* src, float* out, int iters)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint16_t* win = src + ((size_t)blockIdx.x * 4 + wave) * 2048;   // 4 KiB window per wave, L1/L2 resident after first touch
    // stride 2 in the resource: idxen indexes halfs.  Words: base lo | base hi[15:0] + stride << 16 | num_records | flags
    const uint64_t b = reinterpret_cast<uint64_t>(win);
    u32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((uint32_t)b);
    rs.y = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xffffu) | (2u << 16);
    rs.z = 2048u;
    rs.w = 0x00020000u;
    uint32_t idx = (lane * 7) & 63, acc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
        uint32_t so = (it & 15) << 7;
        asm volatile("buffer_load_ushort %0, %8, %9, %10 idxen\n buffer_load_ushort %1, %8, %9, %10 idxen offset:128\n"
                     "buffer_load_ushort %2, %8, %9, %10 idxen offset:256\n buffer_load_ushort %3, %8, %9, %10 idxen offset:384\n"
                     "buffer_load_ushort %4, %8, %9, %10 idxen offset:512\n buffer_load_ushort %5, %8, %9, %10 idxen offset:640\n"
                     "buffer_load_ushort %6, %8, %9, %10 idxen offset:768\n buffer_load_ushort %7, %8, %9, %10 idxen offset:896\n"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                     : "v"(idx), "s"(rs), "s"(so));
        acc += t0 + t1 + t2 + t3 + t4 + t5 + t6 + t7;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    const int blocks = 256 * 6, iters = 4000;   // 6 workgroups of 4 waves per CU
    uint16_t* src; float* out;
    hipMalloc(&src, (size_t)blocks * 4 * 2048 * 2);
    hipMemset(src, 1, (size_t)blocks * 4 * 2048 * 2);
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; which++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (which == 0) k_lds<<<blocks, 256>>>(src, out, iters); else k_buf<<<blocks, 256>>>(src, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double inst = (double)blocks * 4 * iters * 8;   // gather wave-instructions
        printf("%s: %.3f ms, %.3f gather wave-inst/ns/CU (%.2f per cycle per CU at 2.4 GHz)\n", which ? "buffer_load_ushort idxen" : "ds_read_u16",
               ms, inst / (ms * 1e6) / 256, inst / (ms * 1e6) / 256 / 2.4);
    }
    return 0;
}
