// Does v_dot2_f32_f16 keep fp16 subnormals and exact products?  Compares against double on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include <random>
typedef _Float16 h16;
typedef h16 h16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const h16x2* a, const h16x2* b, const float* c, float* d_dot, float* d_mix, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    d_dot[i] = __builtin_amdgcn_fdot2(a[i], b[i], c[i], false);
    float t = __builtin_fmaf((float)a[i].x, (float)b[i].x, c[i]);
    d_mix[i] = __builtin_fmaf((float)a[i].y, (float)b[i].y, t);
}
static h16 rnd(std::mt19937& g, int mode)
{
    std::normal_distribution<float> nd(0.f, 1.f);
    float v = nd(g);
    if (mode == 1) v *= 3e-5f;       // subnormal range
    if (mode == 2) v *= 200.f;
    return (h16)v;
}
int main()
{
    const int n = 1 << 20;
    std::mt19937 g(7);
    std::vector<h16x2> a(n), b(n); std::vector<float> c(n);
    for (int i = 0; i < n; i++) {
        int ma = i % 3, mb = (i / 3) % 3;
        a[i] = {rnd(g, ma), rnd(g, ma)}; b[i] = {rnd(g, mb), rnd(g, mb)};
        c[i] = (i % 5 == 0) ? 0.f : std::normal_distribution<float>(0.f, (i % 7 == 0) ? 1e-6f : 3.f)(g);
    }
    h16x2 *da, *db; float *dc, *dd, *dm;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dd, n * 4); hipMalloc(&dm, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, db, dc, dd, dm, n);
    std::vector<float> rd(n), rm(n);
    hipMemcpy(rd.data(), dd, n * 4, hipMemcpyDeviceToHost); hipMemcpy(rm.data(), dm, n * 4, hipMemcpyDeviceToHost);
    double worst_dot = 0, worst_mix = 0; int exact_dot = 0, exact_mix = 0, bad_sub = 0, differ = 0, wi = 0; int hist[8] = {0};
    for (int i = 0; i < n; i++) {
        double ex = (double)(float)a[i].x * (double)(float)b[i].x + (double)(float)a[i].y * (double)(float)b[i].y + (double)c[i];
        float rn = (float)ex;   // correctly rounded single-rounding result
        double mag = std::fabs((double)(float)a[i].x * (double)(float)b[i].x) + std::fabs((double)(float)a[i].y * (double)(float)b[i].y) + std::fabs((double)c[i]);
        double ulp = std::ldexp(1.0, std::ilogb(mag > 1e-300 ? mag : 1e-300) - 23);   // ulp of the largest term
        differ += (rd[i] != rm[i]);
        double ed = std::fabs((double)rd[i] - ex) / ulp, em = std::fabs((double)rm[i] - ex) / ulp;
        if (ed > worst_dot) { worst_dot = ed; wi = i; }
        if (em > worst_mix) worst_mix = em;
        { int b_ = ed < 0.5001 ? 0 : ed < 1.01 ? 1 : ed < 2.01 ? 2 : ed < 8 ? 3 : ed < 64 ? 4 : 5; hist[b_]++; }
        exact_dot += (rd[i] == rn); exact_mix += (rm[i] == rn);
        if ((i % 3 == 1 || (i / 3) % 3 == 1) && c[i] == 0.f && ex != 0.0 && rd[i] == 0.f) bad_sub++;
    }
    printf("dot2: worst err %.3f ulp, correctly rounded %.2f%% | fma chain: worst %.3f ulp, correctly rounded %.2f%% | subnormal-input results flushed to 0: %d | dot2 != fma-chain in %d of %d\n",
           worst_dot, 100.0 * exact_dot / n, worst_mix, 100.0 * exact_mix / n, bad_sub, differ, n);
    printf("worst: a=(%g,%g) b=(%g,%g) c=%.9g got=%.9g exact=%.12g\n", (float)a[wi].x, (float)a[wi].y, (float)b[wi].x, (float)b[wi].y, c[wi], rd[wi],
           (double)(float)a[wi].x * (double)(float)b[wi].x + (double)(float)a[wi].y * (double)(float)b[wi].y + (double)c[wi]);
    printf("error histogram (ulp of largest term): <=0.5:%d <=1:%d <=2:%d <8:%d <64:%d more:%d\n", hist[0], hist[1], hist[2], hist[3], hist[4], hist[5]);
    return 0;
}
