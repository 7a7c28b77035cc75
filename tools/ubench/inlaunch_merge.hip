// What would folding the one-pass launch's row kernel INTO the launch cost?  (round-5 review, item 1b: "the slab merge inside the launch for launches of
// <= 512 workgroups ... per-row arrival counter, sc1 slab stores and loads as in the guide's valid forms".)  A stand-alone measurement of exactly that
// hand-off on this hardware, at the product's shapes, against what the product does today (a second, dependent launch):
//
//   producers : grid (P, R) workgroups of 256 threads = the one-pass launch of a small configuration (Llama-3-8B 8k x batch 1: R = 8 head groups of
//               4 heads, P = 62 workgroups each; c2: R = 32 rows of one head -- here run as 8 rows of 4 heads --, P = 30).  A workgroup "works" for a
//               given time (a spin on s_memrealtime: every workgroup the same, so that they all arrive together -- the hardest case for a counter and
//               the one the product's one-round launches are in), then writes its slab: [4 heads][128] floats + 4 (max, sum) pairs.
//   (a) two launches (the product): plain slab stores; a row kernel of R * 4 workgroups x 128 threads behind it folds the P slabs of a row
//       (weights exp(m - M), weighted sum, divide, fp16) -- onepass_finish1_kernel's arithmetic.
//   (b) one launch, acquire form: slab stores `sc1`, every wave s_waitcnt vmcnt(0), barrier, ONE lane's agent-scope atomic add on the row's counter;
//       the workgroup whose add returns P - 1 runs an agent-scope acquire (buffer_inv sc1), waits, barriers, reads the row's P slabs with plain
//       loads and folds them (all 256 threads: 4 heads x 128 channels / 2), resets the counter.
//   (c) one launch, sc1-load form: as (b) without the acquire, every slab load `sc1` (the guide's table row 1: measured there at one workgroup per CU).
// Reported: us per iteration of each form (32 iterations captured as one hipGraph -- the product's step is a graph of 32 layers -- replayed 25 times between
// hipEvents, after a warm-up), for several amounts of work, and the
// outputs of the three forms compared word by word (the one-launch fold adds the even and the odd slabs separately: equal within 2e-3 relative).
// Build + run (GPU box): hipcc --offload-arch=gfx950 -O2 -o inlaunch_merge tools/ubench/inlaunch_merge.hip && ./inlaunch_merge
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kD = 128, kH = 4, kThreads = 256;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void spin_us(float us)
{
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();           // 100 MHz
    const uint64_t ticks = (uint64_t)(us * 100.f);
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}

// a slab's values: anything deterministic that differs per workgroup, head and channel
__device__ __forceinline__ float slab_val(int row, int p, int h, int c) { return (float)((row * 131 + p * 17 + h * 7 + c) % 97) * 0.03125f - 1.f; }
__device__ __forceinline__ float slab_max(int row, int p, int h) { return (float)((row * 5 + p * 3 + h) % 11) * 0.25f; }
__device__ __forceinline__ float slab_sum(int row, int p, int h) { return 1.f + (float)((row + p * 7 + h * 3) % 13); }

// ws_o [P][R * 4][128] floats, ws_ml [P][R * 4][2] floats: the product's slab layout
template <int FORM>   // 0: plain stores (two launches), 1: sc1 stores + counter + acquire + plain loads, 2: sc1 stores + counter + sc1 loads
__global__ __launch_bounds__(kThreads) void producer(float* ws_o, float* ws_ml, uint32_t* counter, __half* out, int P, int R, float work_us)
{
    __shared__ uint32_t s_old;
    __shared__ float s_w[64 * kH];   // weights of up to 64 slabs x 4 heads
    __shared__ float s_den[kH];
    __shared__ __attribute__((aligned(16))) float s_fold[128 * 4];
    const int p = blockIdx.x, row = blockIdx.y, BH = R * kH;
    spin_us(work_us);
    // the slab: thread t < 128 writes head t / 32, channels (t % 32) * 4 .. + 3 as ONE 16-byte store (the guide: 4-byte sc1 stores cost ~6 x the 16-byte time per byte)
    if (threadIdx.x < 128) {
        const int h = threadIdx.x >> 5, c = (threadIdx.x & 31) * 4;
        float* dst = ws_o + ((int64_t)p * BH + row * kH + h) * kD + c;
        f32x4 v = {slab_val(row, p, h, c), slab_val(row, p, h, c + 1), slab_val(row, p, h, c + 2), slab_val(row, p, h, c + 3)};
        if (FORM == 0) *reinterpret_cast<f32x4*>(dst) = v;
        else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(v) : "memory");
    } else if (threadIdx.x < 128 + kH) {
        const int h = threadIdx.x - 128;
        float* ml = ws_ml + ((int64_t)p * BH + row * kH + h) * 2;
        f32x2 v = {slab_max(row, p, h), slab_sum(row, p, h)};
        if (FORM == 0) *reinterpret_cast<f32x2*>(ml) = v;
        else asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(ml), "v"(v) : "memory");
    }
    if (FORM == 0) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave
    __syncthreads();
    if (threadIdx.x == 0) s_old = __hip_atomic_fetch_add(counter + row, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_old != (uint32_t)(P - 1)) return;
    // ---- the last arriver of the row folds its P slabs
    if (FORM == 1) {
        if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // weights: thread (h, k) for k < P (P <= 64)
    {
        const int h = threadIdx.x >> 6, k = threadIdx.x & 63;
        f32x2 ml = {-INFINITY, 0.f};
        if (k < P) {
            const float* a = ws_ml + ((int64_t)k * BH + row * kH + h) * 2;
            if (FORM == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(ml) : "v"(a) : "memory");
            else ml = *reinterpret_cast<const f32x2*>(a);
        }
        const float m = ml.x, l = ml.y;
        float M = m;
        for (int o = 32; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o));
        const float w = l > 0.f ? __expf(m - M) : 0.f;
        float den = w * l;
        for (int o = 32; o > 0; o >>= 1) den += __shfl_xor(den, o);
        s_w[h * 64 + k] = w;
        if (k == 0) s_den[h] = den;
    }
    __syncthreads();
    {
        // thread: q = t % 128 -> head q / 32, channels (q % 32) * 4 .. + 3; the slabs of parity t / 128; sixteen 16-byte loads in flight (64 registers:
        // what a kernel compiled for eight waves per SIMD can hold)
        const int q = threadIdx.x & 127, par = threadIdx.x >> 7, h = q >> 5, c = (q & 31) * 4;
        const float* src = ws_o + ((int64_t)row * kH + h) * kD + c;
        const int64_t total = (int64_t)BH * kD;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = par; k0 < P; k0 += 32) {
            f32x4 x[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int k = k0 + 2 * i;
                x[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (k < P) {
                    if (FORM == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(x[i]) : "v"(src + k * total) : "memory");
                    else x[i] = *reinterpret_cast<const f32x4*>(src + k * total);
                }
            }
            if (FORM == 2)   // (the loaded registers are operands of the wait: nothing that reads them may be scheduled in front of it)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]),
                             "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) :: "memory");
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int k = k0 + 2 * i;
                const float w = k < P ? s_w[h * 64 + k] : 0.f;
                acc += w * x[i];
            }
        }
        f32x4* s_half = reinterpret_cast<f32x4*>(s_fold);
        if (par) s_half[q] = acc;
        __syncthreads();
        if (!par) {
            const f32x4 o = (acc + s_half[q]) / s_den[h];   // (even slabs + odd slabs: NOT the two-launch kernel's order of additions -- compared with a tolerance below)
            __half* dst = out + ((int64_t)row * kH + h) * kD + c;
            dst[0] = __float2half(o.x); dst[1] = __float2half(o.y); dst[2] = __float2half(o.z); dst[3] = __float2half(o.w);
        }
    }
    if (threadIdx.x == 0) __hip_atomic_store(counter + row, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
}

// the second launch of form (a): one workgroup of 128 threads per (row, head), a slab per lane (P <= 64) -- onepass_finish1_kernel's shape
__global__ __launch_bounds__(128) void row_kernel(const float* ws_o, const float* ws_ml, __half* out, int P, int R)
{
    const int bh = blockIdx.x, c = threadIdx.x, lane = threadIdx.x & 63, BH = R * kH;
    const float* src = ws_o + (int64_t)bh * kD + c;
    const int64_t total = (int64_t)BH * kD;
    float v[64];   // every slab's value requested before the weights are known (onepass_finish1_kernel asks for the first 40 that way)
#pragma unroll
    for (int k = 0; k < 64; k++) v[k] = k < P ? src[k * total] : 0.f;
    float m = -INFINITY, l = 0.f;
    if (lane < P) {
        const float* ml = ws_ml + ((int64_t)lane * BH + bh) * 2;
        m = ml[0];
        l = ml[1];
    }
    float M = m;
    for (int o = 32; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o));
    const float w = l > 0.f ? __expf(m - M) : 0.f;
    float den = w * l;
    for (int o = 32; o > 0; o >>= 1) den += __shfl_xor(den, o);
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 64; k++) a += __shfl(w, k) * v[k];
    out[(int64_t)bh * kD + c] = __float2half(a / den);
}

int main()
{
    struct Shape { const char* name; int P, R; };
    const Shape shapes[] = {{"8k x batch 1 (8 rows x 62 workgroups = 496)", 62, 8}, {"c2-like (8 rows x 30 = 240)", 30, 8}, {"4k x batch 8 (64 rows x 30 = 1920)", 30, 64}};
    const float works[] = {0.f, 4.f, 8.f};
    const int iters = 200, warm = 20;
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (const Shape& s : shapes) {
        const int BH = s.R * kH;
        float *ws_o, *ws_ml;
        uint32_t* counter;
        __half* out[3];
        CHECK(hipMalloc(&ws_o, sizeof(float) * (size_t)s.P * BH * kD));
        CHECK(hipMalloc(&ws_ml, sizeof(float) * (size_t)s.P * BH * 2));
        CHECK(hipMalloc(&counter, sizeof(uint32_t) * s.R));
        CHECK(hipMemset(counter, 0, sizeof(uint32_t) * s.R));
        for (auto& o : out) { CHECK(hipMalloc(&o, sizeof(__half) * (size_t)BH * kD)); CHECK(hipMemset(o, 0xff, sizeof(__half) * (size_t)BH * kD)); }
        printf("== %s\n", s.name);
        for (float w : works) {
            float us[3];
            for (int form = 0; form < 3; form++) {
                auto once = [&]() {
                    const dim3 grid(s.P, s.R);
                    if (form == 0) {
                        hipLaunchKernelGGL(producer<0>, grid, dim3(kThreads), 0, st, ws_o, ws_ml, counter, out[0], s.P, s.R, w);
                        hipLaunchKernelGGL(row_kernel, dim3(BH), dim3(128), 0, st, ws_o, ws_ml, out[0], s.P, s.R);
                    } else if (form == 1) {
                        hipLaunchKernelGGL(producer<1>, grid, dim3(kThreads), 0, st, ws_o, ws_ml, counter, out[1], s.P, s.R, w);
                    } else {
                        hipLaunchKernelGGL(producer<2>, grid, dim3(kThreads), 0, st, ws_o, ws_ml, counter, out[2], s.P, s.R, w);
                    }
                };
                // as the product runs its step: 32 iterations ("layers") captured as one graph, replayed
                hipGraph_t graph;
                hipGraphExec_t exec;
                CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
                for (int i = 0; i < 32; i++) once();
                CHECK(hipStreamEndCapture(st, &graph));
                CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                for (int i = 0; i < warm; i++) CHECK(hipGraphLaunch(exec, st));
                CHECK(hipStreamSynchronize(st));
                CHECK(hipEventRecord(e0, st));
                for (int i = 0; i < iters / 8; i++) CHECK(hipGraphLaunch(exec, st));
                CHECK(hipEventRecord(e1, st));
                CHECK(hipEventSynchronize(e1));
                float ms = 0.f;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                us[form] = ms * 1e3f / (32 * (iters / 8));
                CHECK(hipGraphExecDestroy(exec));
                CHECK(hipGraphDestroy(graph));
            }
            // the three outputs, word by word
            std::vector<uint16_t> h[3];
            int bad1 = 0, bad2 = 0;
            for (int f = 0; f < 3; f++) { h[f].resize((size_t)BH * kD); CHECK(hipMemcpy(h[f].data(), out[f], sizeof(uint16_t) * h[f].size(), hipMemcpyDeviceToHost)); }
            auto f = [](uint16_t u) { __half hh; memcpy(&hh, &u, 2); return __half2float(hh); };
            for (size_t i = 0; i < h[0].size(); i++) {   // (another order of additions: equal within 2 fp16 ulps, not bit for bit)
                const float r = f(h[0][i]), tol = fmaxf(fabsf(r), 1e-3f) * 0.002f;
                bad1 += !(fabsf(f(h[1][i]) - r) <= tol);
                bad2 += !(fabsf(f(h[2][i]) - r) <= tol);
            }
            printf("  work %4.1f us per workgroup:  two launches %6.2f us   one launch, acquire form %6.2f us   one launch, sc1-load form %6.2f us   (words off the two-launch result by more than 2e-3 relative: %d, %d)\n",
                   w, us[0], us[1], us[2], bad1, bad2);
        }
        CHECK(hipFree(ws_o)); CHECK(hipFree(ws_ml)); CHECK(hipFree(counter));
        for (auto& o : out) CHECK(hipFree(o));
    }
    return 0;
}
