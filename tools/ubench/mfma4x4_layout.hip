// Lane/register layout of v_mfma_f32_4x4x4_16B_f16 on gfx950, probed with exact small integers.
// Claim to verify: with a[lane] = A_blk[i = lane%4][k=0..3], b[lane] = B_blk[k=0..3][j = lane%4] (blk = lane/4),
// d[lane][r] = sum_k A_blk[r][k] * B_blk[k][lane%4].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out)
{
    const int lane = threadIdx.x, blk = lane / 4, r = lane % 4;
    h16x4 a, b;
    for (int kk = 0; kk < 4; kk++) {
        a[kk] = (_Float16)(float)(1 + r + 4 * kk + blk);          // A_blk[i=r][k=kk]
        b[kk] = (_Float16)(float)((kk + 1) * (r + 2) - blk);       // B_blk[k=kk][j=r]
    }
    f32x4 c = {0, 0, 0, 0};
    f32x4 d = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; i++) out[lane * 4 + i] = d[i];
}
int main()
{
    float* d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; lane++) {
        const int blk = lane / 4, j = lane % 4;
        for (int i = 0; i < 4; i++) {
            float want = 0;
            for (int kk = 0; kk < 4; kk++) want += (float)(1 + i + 4 * kk + blk) * (float)((kk + 1) * (j + 2) - blk);
            if (h[lane * 4 + i] != want) { if (bad < 5) printf("lane %d reg %d got %g want %g\n", lane, i, h[lane * 4 + i], want); bad++; }
        }
    }
    printf("mfma 4x4x4 layout %s (%d mismatches)\n", bad ? "DIFFERS" : "as claimed", bad);
    return 0;
}
