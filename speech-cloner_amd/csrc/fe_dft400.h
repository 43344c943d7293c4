// 400-point real DFT building blocks for the STFT kernel (vc_frontend.hip).
//
// The reference computes librosa.core.stft with n_fft = 400 (/root/reference/audio_lib.py:141-147,
// hp/ds_*_cfg_d.json: win 25 ms @ 16 kHz).  400 = 25 x 16, so the transform is split with
//   n = 16*n1 + n2   (n1 in [0,25), n2 in [0,16))        input index
//   k = k1 + 25*k2   (k1 in [0,25), k2 in [0,16))        output bin
//   W400^(n k) = W25^(n1 k1) * W400^(n2 k1) * W16^(n2 k2)
// step 1: 16 real-input 25-point DFTs over n1 (only k1 = 0..12 are needed: real input =>
//         A[25-k1] = conj A[k1], and bins > 200 are never needed),
// step 2: twiddle by W400^(n2 k1),
// step 3: 13 complex 16-point DFTs over n2.
// Bins k in [0,200] with (k mod 25) <= 12 come out directly; the others are |Y[400-k]|.
//
// Everything here is plain inline C++ usable from host code too (tests/ compile it with g++
// to check the index algebra against a naive double-precision DFT without a GPU).
#pragma once

#if defined(__HIPCC__)
#define VC_HD __host__ __device__ __forceinline__
#else
#define VC_HD inline
#endif

namespace vcfe {

constexpr float C5_1 = 0.30901699437494742f;    // cos(2pi/5)
constexpr float C5_2 = -0.80901699437494742f;   // cos(4pi/5)
constexpr float S5_1 = 0.95105651629515357f;    // sin(2pi/5)
constexpr float S5_2 = 0.58778525229247313f;    // sin(4pi/5)

// W25^m = exp(-2 pi i m / 25), m = 0..8 (needed: b*c with b<=4, c<=2)
constexpr float W25_RE[9] = {1.0f, 0.96858316112863108f, 0.87630668004386358f, 0.72896862742141155f,
                             0.53582679497899666f, 0.30901699437494742f, 0.06279051952931337f,
                             -0.18738131458572463f, -0.42577929156507272f};
constexpr float W25_IM[9] = {-0.0f, -0.24868988716485479f, -0.48175367410171532f, -0.68454710592868873f,
                             -0.84432792550201508f, -0.95105651629515357f, -0.99802672842827156f,
                             -0.98228725072868872f, -0.90482705246601958f};

constexpr float C16_1 = 0.92387953251128674f;   // cos(pi/8)
constexpr float S16_1 = 0.38268343236508977f;   // sin(pi/8)
constexpr float R2 = 0.70710678118654752f;      // sqrt(1/2)

// Real-input 5-point DFT, outputs X0 (real), X1, X2.
VC_HD void rdft5(float x0, float x1, float x2, float x3, float x4,
                 float& r0, float& r1, float& i1, float& r2, float& i2) {
    const float p1 = x1 + x4, p2 = x2 + x3, q1 = x1 - x4, q2 = x2 - x3;
    r0 = x0 + p1 + p2;
    r1 = x0 + C5_1 * p1 + C5_2 * p2;
    i1 = -(S5_1 * q1 + S5_2 * q2);
    r2 = x0 + C5_2 * p1 + C5_1 * p2;
    i2 = -(S5_2 * q1 - S5_1 * q2);
}

// Complex 5-point DFT, in place (zr/zi hold z0..z4 -> Z0..Z4).
VC_HD void cdft5(float* zr, float* zi) {
    const float p1r = zr[1] + zr[4], p1i = zi[1] + zi[4];
    const float p2r = zr[2] + zr[3], p2i = zi[2] + zi[3];
    const float q1r = zr[1] - zr[4], q1i = zi[1] - zi[4];
    const float q2r = zr[2] - zr[3], q2i = zi[2] - zi[3];
    const float u1r = zr[0] + C5_1 * p1r + C5_2 * p2r, u1i = zi[0] + C5_1 * p1i + C5_2 * p2i;
    const float u2r = zr[0] + C5_2 * p1r + C5_1 * p2r, u2i = zi[0] + C5_2 * p1i + C5_1 * p2i;
    const float w1r = S5_1 * q1r + S5_2 * q2r, w1i = S5_1 * q1i + S5_2 * q2i;
    const float w2r = S5_2 * q1r - S5_1 * q2r, w2i = S5_2 * q1i - S5_1 * q2i;
    zr[0] = zr[0] + p1r + p2r;
    zi[0] = zi[0] + p1i + p2i;
    // Z1 = u1 - i w1, Z4 = u1 + i w1 ; -i*(a+ib) = b - ia
    zr[1] = u1r + w1i; zi[1] = u1i - w1r;
    zr[4] = u1r - w1i; zi[4] = u1i + w1r;
    zr[2] = u2r + w2i; zi[2] = u2i - w2r;
    zr[3] = u2r - w2i; zi[3] = u2i + w2r;
}

// Real-input 25-point DFT: v[0..24] real -> A[k1], k1 = 0..12 (re/im).
VC_HD void rdft25_13(const float* v, float* ar, float* ai) {
    // stage A: for each b, DFT-5 over a of v[5a+b]; keep c = 0,1,2; twiddle by W25^(b c)
    float t0[5];                    // c = 0 (real)
    float t1r[5], t1i[5], t2r[5], t2i[5];
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        float r0, r1, i1, r2, i2;
        rdft5(v[b], v[5 + b], v[10 + b], v[15 + b], v[20 + b], r0, r1, i1, r2, i2);
        t0[b] = r0;
        const float w1r = W25_RE[b], w1i = W25_IM[b];
        const float w2r = W25_RE[2 * b], w2i = W25_IM[2 * b];
        t1r[b] = r1 * w1r - i1 * w1i; t1i[b] = r1 * w1i + i1 * w1r;
        t2r[b] = r2 * w2r - i2 * w2i; t2i[b] = r2 * w2i + i2 * w2r;
    }
    // stage B, c = 0: real DFT-5 over b -> k1 = 0, 5, 10
    {
        float r0, r1, i1, r2, i2;
        rdft5(t0[0], t0[1], t0[2], t0[3], t0[4], r0, r1, i1, r2, i2);
        ar[0] = r0; ai[0] = 0.0f;
        ar[5] = r1; ai[5] = i1;
        ar[10] = r2; ai[10] = i2;
    }
    // c = 1: k1 = 1, 6, 11, 16 -> conj -> 9, 21 -> conj -> 4
    cdft5(t1r, t1i);
    ar[1] = t1r[0]; ai[1] = t1i[0];
    ar[6] = t1r[1]; ai[6] = t1i[1];
    ar[11] = t1r[2]; ai[11] = t1i[2];
    ar[9] = t1r[3]; ai[9] = -t1i[3];
    ar[4] = t1r[4]; ai[4] = -t1i[4];
    // c = 2: k1 = 2, 7, 12, 17 -> conj -> 8, 22 -> conj -> 3
    cdft5(t2r, t2i);
    ar[2] = t2r[0]; ai[2] = t2i[0];
    ar[7] = t2r[1]; ai[7] = t2i[1];
    ar[12] = t2r[2]; ai[12] = t2i[2];
    ar[8] = t2r[3]; ai[8] = -t2i[3];
    ar[3] = t2r[4]; ai[3] = -t2i[4];
}

// radix-4 butterfly (forward, W4 = -i): in place on 4 complex values.
VC_HD void bfly4(float& r0, float& i0, float& r1, float& i1, float& r2, float& i2, float& r3, float& i3) {
    const float ar = r0 + r2, ai = i0 + i2, br = r0 - r2, bi = i0 - i2;
    const float cr = r1 + r3, ci = i1 + i3, dr = r1 - r3, di = i1 - i3;
    r0 = ar + cr; i0 = ai + ci;            // X0
    r2 = ar - cr; i2 = ai - ci;            // X2
    r1 = br + di; i1 = bi - dr;            // X1 = b - i d
    r3 = br - di; i3 = bi + dr;            // X3 = b + i d
}

VC_HD void cmul(float& r, float& i, float wr, float wi) {
    const float t = r * wr - i * wi;
    i = r * wi + i * wr;
    r = t;
}

// Complex 16-point DFT.  Input z[n2] (natural order), output Y[k2] in natural order written
// to yr/yi.  n2 = 4p + q, k2 = r + 4s.
VC_HD void cdft16(const float* zr, const float* zi, float* yr, float* yi) {
    float ur[4][4], ui[4][4];              // [q][r]
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float r0 = zr[q], i0 = zi[q], r1 = zr[4 + q], i1 = zi[4 + q];
        float r2 = zr[8 + q], i2 = zi[8 + q], r3 = zr[12 + q], i3 = zi[12 + q];
        bfly4(r0, i0, r1, i1, r2, i2, r3, i3);
        ur[q][0] = r0; ui[q][0] = i0; ur[q][1] = r1; ui[q][1] = i1;
        ur[q][2] = r2; ui[q][2] = i2; ur[q][3] = r3; ui[q][3] = i3;
    }
    // twiddles W16^(q r) = exp(-2 pi i q r / 16)
    // q=1: r=1: (c,-s)  r=2: (R2,-R2)  r=3: (s,-c)
    cmul(ur[1][1], ui[1][1], C16_1, -S16_1);
    cmul(ur[1][2], ui[1][2], R2, -R2);
    cmul(ur[1][3], ui[1][3], S16_1, -C16_1);
    // q=2: r=1: (R2,-R2)  r=2: (0,-1)  r=3: (-R2,-R2)
    cmul(ur[2][1], ui[2][1], R2, -R2);
    { const float t = ur[2][2]; ur[2][2] = ui[2][2]; ui[2][2] = -t; }
    cmul(ur[2][3], ui[2][3], -R2, -R2);
    // q=3: r=1: W^3 = (s,-c)  r=2: W^6 = (-R2,-R2)  r=3: W^9 = (-c, s)
    cmul(ur[3][1], ui[3][1], S16_1, -C16_1);
    cmul(ur[3][2], ui[3][2], -R2, -R2);
    cmul(ur[3][3], ui[3][3], -C16_1, S16_1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float r0 = ur[0][r], i0 = ui[0][r], r1 = ur[1][r], i1 = ui[1][r];
        float r2 = ur[2][r], i2 = ui[2][r], r3 = ur[3][r], i3 = ui[3][r];
        bfly4(r0, i0, r1, i1, r2, i2, r3, i3);
        yr[r] = r0; yi[r] = i0; yr[r + 4] = r1; yi[r + 4] = i1;
        yr[r + 8] = r2; yi[r + 8] = i2; yr[r + 12] = r3; yi[r + 12] = i3;
    }
}

// Which spectrum bin (0..200) does output (k1, k2) of step 3 feed?  -1 = none (duplicate).
VC_HD int bin_of(int k1, int k2) {
    const int k = k1 + 25 * k2;
    if (k <= 200) return k;
    if (k1 == 0) return -1;
    return 400 - k;
}

// ---- inverse direction (Griffin-Lim vocoder, vc_vocoder.hip) -------------------------------------
// x[16 n1 + n2] = (1/400) sum_k1 W25^(-n1 k1) B[k1, n2],
// B[k1, n2] = W400^(-n2 k1) sum_k2 W16^(-n2 k2) S[k1 + 25 k2]   (S hermitian: S[400-k] = conj S[k]).
// B[25-k1, n2] = conj B[k1, n2], so k1 = 0..12 suffice and the 25-point stage is a
// hermitian -> real transform.

constexpr float COS25[25] = {1.0f, 0.96858316112863108f, 0.87630668004386358f, 0.72896862742141155f, 0.53582679497899655f, 0.30901699437494745f, 0.062790519529313527f, -0.1873813145857246f, -0.42577929156507272f, -0.63742398974868975f, -0.80901699437494734f, -0.92977648588825135f, -0.99211470131447776f, -0.99211470131447788f, -0.92977648588825146f, -0.80901699437494778f, -0.63742398974868952f, -0.42577929156507216f, -0.18738131458572463f, 0.062790519529312833f, 0.30901699437494723f, 0.53582679497899677f, 0.72896862742141122f, 0.87630668004386314f, 0.96858316112863097f};
constexpr float SIN25[25] = {0.0f, 0.24868988716485479f, 0.48175367410171532f, 0.68454710592868862f, 0.84432792550201508f, 0.95105651629515353f, 0.99802672842827156f, 0.98228725072868872f, 0.90482705246601947f, 0.77051324277578925f, 0.58778525229247325f, 0.36812455268467814f, 0.12533323356430454f, -0.12533323356430429f, -0.36812455268467792f, -0.58778525229247269f, -0.77051324277578936f, -0.9048270524660198f, -0.98228725072868872f, -0.99802672842827156f, -0.95105651629515364f, -0.84432792550201496f, -0.68454710592868895f, -0.4817536741017161f, -0.24868988716485535f};

// Inverse complex 16-point DFT (no 1/16): idft(z) = conj(dft(conj z)).
VC_HD void cidft16(float* zr, float* zi, float* yr, float* yi) {
#pragma unroll
    for (int i = 0; i < 16; ++i) zi[i] = -zi[i];
    cdft16(zr, zi, yr, yi);
#pragma unroll
    for (int i = 0; i < 16; ++i) yi[i] = -yi[i];
}

// Hermitian 25-point inverse DFT (no 1/25): b[k1], k1 = 0..12 (Im b[0] ignored) -> x[0..24] real,
//   x[n1] = b0 + 2 sum_{k=1..12} (br[k] cos(2 pi n1 k/25) - bi[k] sin(2 pi n1 k/25)).
VC_HD void hdft25_real(const float* br, const float* bi, float* x) {
#pragma unroll
    for (int n1 = 0; n1 <= 12; ++n1) {
        float e = 0.5f * br[0], o = 0.0f;
#pragma unroll
        for (int k = 1; k <= 12; ++k) {
            e = fmaf(br[k], COS25[(n1 * k) % 25], e);
            o = fmaf(bi[k], SIN25[(n1 * k) % 25], o);
        }
        x[n1] = 2.0f * (e - o);
        if (n1 > 0) x[25 - n1] = 2.0f * (e + o);
    }
}

// Spectrum bin (0..200) whose value (conjugated when k > 200) is S[k1 + 25 k2].
VC_HD int src_bin(int k1, int k2, bool& conj) {
    const int k = k1 + 25 * k2;
    conj = k > 200;
    return conj ? 400 - k : k;
}

}  // namespace vcfe
