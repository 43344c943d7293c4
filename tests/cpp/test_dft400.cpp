// Host-side check of the 400-point split-radix algebra in speech-cloner_amd/csrc/fe_dft400.h
// against a naive double-precision DFT.  Built and run by tests/test_dft400_host.py (no GPU).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "fe_dft400.h"

int main() {
    const double PI = 3.14159265358979323846;
    std::vector<float> x(400);
    unsigned s = 12345u;
    double worst = 0.0, scale = 0.0, worst_inv = 0.0;
    for (int trial = 0; trial < 8; ++trial) {
        for (int n = 0; n < 400; ++n) {
            s = s * 1664525u + 1013904223u;
            float u = (float)((s >> 8) & 0xFFFFFF) / 16777216.0f - 0.5f;
            x[n] = (trial == 0) ? (n == 3 ? 1.0f : 0.0f) : u * (trial == 1 ? 1e-3f : 1.0f);
        }
        // step 1 + 2
        static float Ar[13][16], Ai[13][16];
        for (int n2 = 0; n2 < 16; ++n2) {
            float v[25], ar[13], ai[13];
            for (int n1 = 0; n1 < 25; ++n1) v[n1] = x[16 * n1 + n2];
            vcfe::rdft25_13(v, ar, ai);
            for (int k1 = 0; k1 < 13; ++k1) {
                const double a = -2.0 * PI * (double)(n2 * k1) / 400.0;
                float wr = (float)cos(a), wi = (float)sin(a);
                vcfe::cmul(ar[k1], ai[k1], wr, wi);
                Ar[k1][n2] = ar[k1]; Ai[k1][n2] = ai[k1];
            }
        }
        std::vector<double> P(201, -1.0);
        int written = 0;
        static float Br[13][16], Bi[13][16];
        for (int k1 = 0; k1 < 13; ++k1) {
            float yr[16], yi[16];
            vcfe::cdft16(Ar[k1], Ai[k1], yr, yi);
            {   // inverse direction: the 16 values this k1 holds are S[k1 + 25 k2] (src_bin says which
                // bin, conjugated above 200) -- check that claim against the naive DFT below, then
                // run the inverse 16-point stage and the conjugate twiddle.
                float zr[16], zi[16], br[16], bi[16];
                for (int k2 = 0; k2 < 16; ++k2) { zr[k2] = yr[k2]; zi[k2] = yi[k2]; }
                vcfe::cidft16(zr, zi, br, bi);
                for (int n2 = 0; n2 < 16; ++n2) {
                    const double a = 2.0 * PI * (double)(n2 * k1) / 400.0;
                    vcfe::cmul(br[n2], bi[n2], (float)cos(a), (float)sin(a));
                    Br[k1][n2] = br[n2]; Bi[k1][n2] = bi[n2];
                }
                for (int k2 = 0; k2 < 16; ++k2) {
                    bool cj; const int b = vcfe::src_bin(k1, k2, cj);
                    double re = 0, im = 0;
                    for (int n = 0; n < 400; ++n) {
                        double ang = -2.0 * PI * (double)((b * n) % 400) / 400.0;
                        re += x[n] * cos(ang); im += x[n] * sin(ang);
                    }
                    if (cj) im = -im;
                    const double e = fabs(re - yr[k2]) + fabs(im - yi[k2]);
                    if (e > 2e-3) { printf("src_bin mismatch k1=%d k2=%d err %.3e\n", k1, k2, e); return 3; }
                }
            }
            for (int k2 = 0; k2 < 16; ++k2) {
                int b = vcfe::bin_of(k1, k2);
                if (b < 0) continue;
                if (P[b] >= 0.0) { printf("bin %d written twice\n", b); return 1; }
                P[b] = (double)yr[k2] * yr[k2] + (double)yi[k2] * yi[k2];
                ++written;
            }
        }
        if (written != 201) { printf("written %d bins\n", written); return 1; }
        double pmax = 0.0;
        std::vector<double> Pref(201);
        for (int k = 0; k <= 200; ++k) {
            double re = 0, im = 0;
            for (int n = 0; n < 400; ++n) {
                double a = -2.0 * PI * (double)((k * n) % 400) / 400.0;
                re += x[n] * cos(a); im += x[n] * sin(a);
            }
            Pref[k] = re * re + im * im;
            if (Pref[k] > pmax) pmax = Pref[k];
        }
        for (int k = 0; k <= 200; ++k) {
            double e = fabs(P[k] - Pref[k]) / pmax;
            if (e > worst) worst = e;
        }
        scale = pmax;
        // inverse 25-point hermitian stage: must give back 400 * x
        double xmax = 0.0, xerr = 0.0;
        for (int n2 = 0; n2 < 16; ++n2) {
            float br[13], bi[13], xr[25];
            for (int k1 = 0; k1 < 13; ++k1) { br[k1] = Br[k1][n2]; bi[k1] = Bi[k1][n2]; }
            vcfe::hdft25_real(br, bi, xr);
            for (int n1 = 0; n1 < 25; ++n1) {
                const double ref = 400.0 * x[16 * n1 + n2];
                xerr = fmax(xerr, fabs(xr[n1] - ref));
                xmax = fmax(xmax, fabs(ref));
            }
        }
        if (xerr / xmax > worst_inv) worst_inv = xerr / xmax;
    }
    printf("inverse round trip max |x' - 400 x| / max = %.3e\n", worst_inv);
    if (worst_inv > 5e-6) return 4;
    printf("max |P - Pref| / max(Pref) = %.3e (last pmax %.3e)\n", worst, scale);
    return worst < 2e-6 ? 0 : 2;
}
