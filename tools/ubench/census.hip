// census.hip -- how many workgroups of a given size does a CU of gfx950 hold at once?  (round 4: could a PAIR of waves be a
// workgroup of its own -- 16 workgroups of 128 threads per CU -- in the one-pass launch?)
// Every workgroup bumps a counter of its CU (XCC_ID / SE_ID / CU_ID of HW_ID), records the peak, holds its slot for ~40 us, leaves.
//   hipcc --offload-arch=gfx950 -O2 -o census census.hip && ./census
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int THREADS, int LDS_BYTES>
__global__ __launch_bounds__(THREADS, 8) void census(int* cur, int* peak, int hold_ticks)
{
    __shared__ unsigned char pad[LDS_BYTES];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    const unsigned cu = (xcc & 15) * 64 + ((hw >> 13) & 7) * 16 + ((hw >> 8) & 15);   // xcc, se_id, cu_id
    if (threadIdx.x == 0) {
        pad[0] = 1;
        const int now = atomicAdd(cur + cu, 1) + 1;
        atomicMax(peak + cu, now);
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)hold_ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) atomicSub(cur + cu, pad[0]);
}

template <int THREADS, int LDS_BYTES>
void run(const char* name)
{
    int *cur, *peak;
    (void)hipMalloc(&cur, 1024 * sizeof(int));
    (void)hipMalloc(&peak, 1024 * sizeof(int));
    (void)hipMemset(cur, 0, 1024 * sizeof(int));
    (void)hipMemset(peak, 0, 1024 * sizeof(int));
    census<THREADS, LDS_BYTES><<<256 * 40, THREADS>>>(cur, peak, 4000);   // 40 us at 100 MHz
    (void)hipDeviceSynchronize();
    std::vector<int> h(1024);
    (void)hipMemcpy(h.data(), peak, 1024 * sizeof(int), hipMemcpyDeviceToHost);
    int mx = 0, mn = 1 << 30, used = 0;
    for (int v : h) if (v) { used++; mx = v > mx ? v : mx; mn = v < mn ? v : mn; }
    printf("%-40s CUs seen %3d  workgroups per CU at once: min %d max %d\n", name, used, mn, mx);
    (void)hipFree(cur); (void)hipFree(peak);
}

int main()
{
    run<256, 16384>("256 threads, 16 KiB LDS");
    run<128, 8192>("128 threads,  8 KiB LDS");
    run<128, 4096>("128 threads,  4 KiB LDS");
    run<64, 4096>(" 64 threads,  4 KiB LDS");
    run<128, 64>("128 threads, 64 B LDS");
    return 0;
}
