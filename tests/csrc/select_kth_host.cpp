// Host build of mustafar_amd/csrc/select_kth.h for tests/test_select_kth.py (g++ -O2 -shared -fPIC).
#include "../../mustafar_amd/csrc/select_kth.h"
extern "C" void kth_rows(const uint32_t* rows, int n_rows, int kth, uint32_t* out)
{
    for (int r = 0; r < n_rows; r++) {
        uint32_t raw[64];
        for (int j = 0; j < 64; j++) raw[j] = rows[(long)r * 64 + j];
        out[r] = kth_magnitude128(raw, kth);
    }
}
extern "C" void transpose_block(uint32_t* a32)
{
    uint32_t a[32];
    for (int j = 0; j < 32; j++) a[j] = a32[j];
    bit_transpose32(a);
    for (int j = 0; j < 32; j++) a32[j] = a[j];
}
