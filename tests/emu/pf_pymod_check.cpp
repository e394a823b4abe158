// TEST INFRASTRUCTURE: pf_pymod (prior-flow_amd/csrc/pf_elem.h) against the ATen statement of `%`
// (fmod, then + b when the signs differ) on exact multiples, their neighbours, tiny negatives, wide
// ranges and random bit patterns.  Prints the number of bitwise mismatches; run by tests/test_emu_kernels.py.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include "pf_elem.h"

static float aten_remainder(float a, float b) {
    float m = fmodf(a, b);
    if (m != 0.f && m < 0.f) m += b;
    return m;
}

int main() {
    std::mt19937_64 g(1);
    const float widths[] = {4, 5, 11, 16, 22, 27, 32, 45, 64, 120, 128, 160, 1024, 1280, 2048};
    long bad = 0, n = 0;
    for (float W : widths)
        for (int i = 0; i < 600000; ++i) {
            const uint64_t r = g();
            const double u = (double)(r >> 11) / (double)(1ull << 53);
            float a;
            switch (i % 6) {
                case 0: { uint32_t b32 = (uint32_t)r; memcpy(&a, &b32, 4); if (!std::isfinite(a) || fabsf(a) > 1e7f) a = (float)(u * 8 * W - 4 * W); break; }
                case 1: a = (float)((int)(r % 41) - 20) * W; break;
                case 2: a = nextafterf((float)((int)(r % 41) - 20) * W, (r & 64) ? 1e9f : -1e9f); break;
                case 3: a = -(float)u * 1e-6f; break;
                case 4: a = (float)(u * 6 * W - 3 * W); break;
                default: a = (float)((int)(r % 4001) - 2000) * 0.25f;
            }
            const float x = aten_remainder(a, W), y = pf_pymod(a, W);
            ++n;
            if (memcmp(&x, &y, 4) != 0 && !(std::isnan(x) && std::isnan(y))) {
                if (bad < 5) printf("W=%g a=%.9g aten=%.9g pf=%.9g\n", W, a, x, y);
                ++bad;
            }
        }
    printf("%ld mismatches of %ld\n", bad, n);
    return bad != 0;
}
