// What does the v_dot2_f32_f16 INSTRUCTION (inline asm, as the kernels issue it: VGPR pair x SGPR pair) do with fp16 subnormal
// inputs, fp32-subnormal products and accumulators?  Compared against double on the host, class by class.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include <random>
#include <cstring>
typedef _Float16 h16;
typedef h16 h16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const uint32_t* a, const uint32_t* b, const float* c, float* d, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t av = a[i];
    uint32_t bs = __builtin_amdgcn_readfirstlane(b[blockIdx.x]);   // one coefficient pair per workgroup, in an SGPR
    float acc = c[i];
    asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "s"(bs));
    d[i] = acc;
}
static uint32_t pack(h16 x, h16 y) { uint16_t u = __builtin_bit_cast(uint16_t, x), v = __builtin_bit_cast(uint16_t, y); return u | ((uint32_t)v << 16); }
int main()
{
    const int wg = 256, nwg = 4096, n = wg * nwg;
    std::mt19937 g(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<uint32_t> a(n), b(nwg); std::vector<float> c(n); std::vector<h16x2> ah(n), bh(nwg);
    // classes by workgroup: 0 normal x normal, 1 subnormal a x normal b, 2 normal a x subnormal b, 3 subnormal a x large b, 4 tiny accumulators
    for (int w = 0; w < nwg; w++) {
        int cls = w % 5;
        float sb = (cls == 2) ? 3e-5f : (cls == 3) ? 2e4f : 1.f;
        bh[w] = {(h16)(nd(g) * sb), (h16)(nd(g) * sb)};
        b[w] = pack(bh[w].x, bh[w].y);
        for (int t = 0; t < wg; t++) {
            int i = w * wg + t;
            float sa = (cls == 1 || cls == 3) ? 3e-5f : 1.f;
            ah[i] = {(h16)(nd(g) * sa), (h16)(nd(g) * sa)};
            a[i] = pack(ah[i].x, ah[i].y);
            c[i] = (cls == 4) ? nd(g) * 1e-39f : ((t % 3 == 0) ? 0.f : nd(g));
        }
    }
    uint32_t *da, *db; float *dc, *dd;
    (void)hipMalloc(&da, n * 4); (void)hipMalloc(&db, nwg * 4); (void)hipMalloc(&dc, n * 4); (void)hipMalloc(&dd, n * 4);
    (void)hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), nwg * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    k<<<nwg, wg>>>(da, db, dc, dd, n);
    std::vector<float> r(n);
    (void)hipMemcpy(r.data(), dd, n * 4, hipMemcpyDeviceToHost);
    const char* names[5] = {"normal x normal", "subnormal a x normal b", "normal a x subnormal b", "subnormal a x large b", "fp32-subnormal accumulator"};
    for (int cls = 0; cls < 5; cls++) {
        double worst = 0; long cnt = 0, flushed = 0, exactly = 0, nonzero_expected = 0;
        for (int w = cls; w < nwg; w += 5)
            for (int t = 0; t < wg; t++) {
                int i = w * wg + t;
                double ex = (double)(float)ah[i].x * (double)(float)bh[w].x + (double)(float)ah[i].y * (double)(float)bh[w].y + (double)c[i];
                double exp_only = (double)(float)ah[i].x * (double)(float)bh[w].x + (double)(float)ah[i].y * (double)(float)bh[w].y;
                double mag = std::fabs((double)(float)ah[i].x * (double)(float)bh[w].x) + std::fabs((double)(float)ah[i].y * (double)(float)bh[w].y) + std::fabs((double)c[i]);
                double ulp = std::ldexp(1.0, std::ilogb(mag > 1e-300 ? mag : 1e-300) - 23);
                double e = std::fabs((double)r[i] - ex) / ulp;
                if (e > worst) worst = e;
                cnt++;
                exactly += (r[i] == (float)ex);
                if (exp_only != 0.0) { nonzero_expected++; flushed += ((double)r[i] == (double)c[i]); }
            }
        printf("%-28s n=%ld  worst err %.3f ulp(largest term)  correctly rounded %.2f%%  products ignored (result == c) %ld of %ld\n", names[cls], cnt, worst,
               100.0 * exactly / cnt, flushed, nonzero_expected);
    }
    return 0;
}
