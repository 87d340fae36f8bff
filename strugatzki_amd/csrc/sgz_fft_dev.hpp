// Complex fp32 FFT of N = 4096, 8192 or 16384 points held in LDS, for the spectral screening of the sliding
// cross term (the sum over lags of impl/MathUtil.scala:177-196, as a circular correlation).
//
// Stockham auto-sort passes: three of radix 16 (strides 1, 16, 256), then one of radix N / 4096 (2 or 4) where
// N > 4096.  A pass works on N / radix butterflies; butterfly j of a pass with stride Ns reads the elements
// j + r * N / radix, multiplies element r by exp(DIR 2 pi i r k / (Ns radix)), k = j mod Ns, transforms, and
// writes element r of the result to (j - k) * radix + k + r * Ns: input and output are both in natural order.
// A workgroup of N / 16 threads owns one radix-16 butterfly per thread and pass; all threads read, meet at a
// barrier, write to the same buffer, meet again.
//
// Element i lives at fft_phys(i) = i ^ ((i >> 4) & 15): the four low bits are exchanged with those of the next
// four.  An 8-byte LDS store is served in groups of 16 consecutive lanes over 32 banks of 4 bytes, an 8-byte load in
// groups of 32 lanes over 64 banks: a group is free of conflicts when its elements' positions differ modulo 16 resp.
// 32.  Loads and the last pass's stores touch consecutive elements (an aligned run of 16 stays a permutation of
// itself); the strided stores of the radix-16 passes -- 16 j + r, 256 a + k + 16 r and 4096 a + k + 256 r over
// consecutive j resp. k -- have a constant low digit or constant higher digits, and the exchange turns either into
// 16 distinct positions.  (A pad of one element per 16 does the same for the stores but lets every aligned run
// of 32 consecutive elements straddle 33 positions: one two-way conflict per group of every load.)
//
// The pieces are __host__ __device__ so that a host loop over "threads" runs the same index arithmetic and the
// same fp32 operations (explicit fma, no contraction) as the kernels: sgz_fft_reference_host.
//
// Twiddle factors come from a table tab[j] = (cos, sin)(2 pi j / kFftTab) rounded from fp64; per butterfly the
// powers 1, 2, 4, 8 are read and the other eleven formed by at most three multiplications (error <= 10 ulp).
#pragma once
#include <cmath>

#include <hip/hip_runtime.h>

namespace sgz {

constexpr int kFftTab = 16384;  // entries of the twiddle table == the largest transform
// Behind the table: for the radix-16 passes with strides 16 and 256, per k < stride the four entries a butterfly
// reads -- tab[k s], tab[2 k s], tab[4 k s], tab[8 k s], s = kFftTab / (16 stride) -- side by side, so that a lane
// fetches its four in one 32-byte piece and a wave's lanes in a few cache lines (taken one by one out of the table
// they lie up to eight times as far apart as the lanes' k: 240 lines per wave and transform instead of 20).
constexpr int kFftPack16 = kFftTab, kFftPack256 = kFftTab + 4 * 16, kFftTabAll = kFftTab + 4 * 16 + 4 * 256;
// Fills the table: tab has kFftTabAll entries.
inline void fft_fill_table(float2 *tab)
{
    const double step = 2.0 * 3.14159265358979323846 / kFftTab;
    for (int j = 0; j < kFftTab; j++) tab[j] = make_float2((float)std::cos(step * j), (float)std::sin(step * j));
    for (int Ns : {16, 256}) {
        float2 *pk = tab + (Ns == 16 ? kFftPack16 : kFftPack256);
        const int stride = kFftTab / (16 * Ns);
        for (int k = 0; k < Ns; k++)
            for (int p = 0; p < 4; p++) pk[4 * k + p] = tab[(k << p) * stride];
    }
}

__host__ __device__ constexpr int fft_phys(int i) { return i ^ ((i >> 4) & 15); }
__host__ __device__ constexpr int fft_lds_elems(int N) { return N; }

__host__ __device__ inline float2 c_mul(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
__host__ __device__ inline float2 c_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__host__ __device__ inline float2 c_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// a * (DIR i)
template <int DIR>
__host__ __device__ inline float2 c_mul_i(float2 a)
{
    return DIR > 0 ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}
// table entry as exp(DIR i angle)
template <int DIR>
__host__ __device__ inline float2 c_dir(float2 w)
{
    return DIR > 0 ? w : make_float2(w.x, -w.y);
}

// 4-point transform in place: (a, b, c, d) = x0..x3 -> X0..X3, kernel exp(DIR 2 pi i n k / 4)
template <int DIR>
__host__ __device__ inline void dft4(float2 &a, float2 &b, float2 &c, float2 &d)
{
    const float2 s02 = c_add(a, c), d02 = c_sub(a, c), s13 = c_add(b, d), d13 = c_mul_i<DIR>(c_sub(b, d));
    a = c_add(s02, s13);
    b = c_add(d02, d13);
    c = c_sub(s02, s13);
    d = c_sub(d02, d13);
}

// 16-point transform of v[0..15] (natural order in).  Element k of the result is left in v[fft16_slot(k)].
__host__ __device__ constexpr int fft16_slot(int k) { return 4 * (k & 3) + (k >> 2); }

template <int DIR>
__host__ __device__ inline void dft16(float2 (&v)[16])
{
    // n = 4 n1 + n2, k = k1 + 4 k2: 4-point transforms over n1, factors W16^(n2 k1), 4-point transforms over n2
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) dft4<DIR>(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);  // -> v[4 k1 + n2]
    constexpr float c1 = 0.9238795325112867f, s1 = 0.3826834323650898f, h = 0.7071067811865476f;
    const float2 w1 = make_float2(c1, DIR * s1), w2 = make_float2(h, DIR * h), w3 = make_float2(s1, DIR * c1);
    const float2 w6 = make_float2(-h, DIR * h), w9 = make_float2(-c1, -DIR * s1);
    v[4 * 1 + 1] = c_mul(v[4 * 1 + 1], w1);
    v[4 * 2 + 1] = c_mul(v[4 * 2 + 1], w2);
    v[4 * 3 + 1] = c_mul(v[4 * 3 + 1], w3);
    v[4 * 1 + 2] = c_mul(v[4 * 1 + 2], w2);
    v[4 * 2 + 2] = c_mul_i<DIR>(v[4 * 2 + 2]);  // W16^4
    v[4 * 3 + 2] = c_mul(v[4 * 3 + 2], w6);
    v[4 * 1 + 3] = c_mul(v[4 * 1 + 3], w3);
    v[4 * 2 + 3] = c_mul(v[4 * 2 + 3], w6);
    v[4 * 3 + 3] = c_mul(v[4 * 3 + 3], w9);
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) dft4<DIR>(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);  // -> v[4 k1 + k2]
}

// ---- one radix-16 pass, butterfly j (0 <= j < N / 16) ----
template <int N>
__host__ __device__ inline void fft16_load(const float2 *buf, int j, float2 (&v)[16])
{
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = buf[fft_phys(j + r * (N / 16))];
}

template <int DIR, int Ns>
__host__ __device__ inline void fft16_math(float2 (&v)[16], int j, const float2 *__restrict__ tab)
{
    if constexpr (Ns > 1) {
        const int k = j & (Ns - 1);
        constexpr int stride = kFftTab / (16 * Ns);  // table steps per unit angle 2 pi / (16 Ns)
        float2 w[16];
        static_assert(Ns == 16 || Ns == 256, "packed factors exist for these strides");
        const float2 *pk2 = tab + (Ns == 16 ? kFftPack16 : kFftPack256) + 4 * k;  // tab[k s], [2 k s], [4 k s], [8 k s]
        (void)stride;
#if defined(__HIP_DEVICE_COMPILE__)
        const float4 *pk = reinterpret_cast<const float4 *>(pk2);  // (32-byte aligned on the device)
        const float4 p01 = pk[0], p23 = pk[1];
        w[1] = c_dir<DIR>(make_float2(p01.x, p01.y));
        w[2] = c_dir<DIR>(make_float2(p01.z, p01.w));
        w[4] = c_dir<DIR>(make_float2(p23.x, p23.y));
        w[8] = c_dir<DIR>(make_float2(p23.z, p23.w));
#else
        w[1] = c_dir<DIR>(pk2[0]);
        w[2] = c_dir<DIR>(pk2[1]);
        w[4] = c_dir<DIR>(pk2[2]);
        w[8] = c_dir<DIR>(pk2[3]);
#endif
        w[3] = c_mul(w[1], w[2]);
        w[5] = c_mul(w[1], w[4]);
        w[6] = c_mul(w[2], w[4]);
        w[9] = c_mul(w[1], w[8]);
        w[10] = c_mul(w[2], w[8]);
        w[12] = c_mul(w[4], w[8]);
        w[7] = c_mul(w[3], w[4]);
        w[11] = c_mul(w[3], w[8]);
        w[13] = c_mul(w[5], w[8]);
        w[14] = c_mul(w[6], w[8]);
        w[15] = c_mul(w[7], w[8]);
#pragma unroll
        for (int r = 1; r < 16; r++) v[r] = c_mul(v[r], w[r]);
    }
    dft16<DIR>(v);
}

template <int Ns>
__host__ __device__ inline void fft16_store(float2 *buf, int j, const float2 (&v)[16])
{
    const int k = j & (Ns - 1);
    const int j0 = (j - k) * 16 + k;
#pragma unroll
    for (int r = 0; r < 16; r++) buf[fft_phys(j0 + r * Ns)] = v[fft16_slot(r)];
}

// ---- the last pass (radix N / 4096, stride 4096), butterfly j (0 <= j < 4096), in place ----
template <int N, int DIR>
__host__ __device__ inline void fft_last(float2 *buf, int j, const float2 *__restrict__ tab)
{
    constexpr int R = N / 4096;
    static_assert(R == 2 || R == 4, "last pass is radix 2 or 4");
    constexpr int stride = kFftTab / N;
    if constexpr (R == 2) {
        const float2 a = buf[fft_phys(j)];
        const float2 b = c_mul(buf[fft_phys(j + 4096)], c_dir<DIR>(tab[j * stride]));
        buf[fft_phys(j)] = c_add(a, b);
        buf[fft_phys(j + 4096)] = c_sub(a, b);
    } else {
        float2 a = buf[fft_phys(j)];
        float2 b = c_mul(buf[fft_phys(j + 4096)], c_dir<DIR>(tab[j * stride]));
        float2 c = c_mul(buf[fft_phys(j + 8192)], c_dir<DIR>(tab[2 * j * stride]));
        float2 d = c_mul(buf[fft_phys(j + 12288)], c_dir<DIR>(tab[3 * j * stride]));
        dft4<DIR>(a, b, c, d);
        buf[fft_phys(j)] = a;
        buf[fft_phys(j + 4096)] = b;
        buf[fft_phys(j + 8192)] = c;
        buf[fft_phys(j + 12288)] = d;
    }
}

#if defined(__HIPCC__)
// The whole transform by a workgroup of N / 16 threads; buf holds the input in natural order (the caller has
// synchronised after filling it) and the result on return (synchronised).
template <int N, int DIR>
__device__ __forceinline__ void fft_lds(float2 *buf, const float2 *__restrict__ tab, int tid)
{
    constexpr int T = N / 16;
    float2 v[16];
    fft16_load<N>(buf, tid, v);
    fft16_math<DIR, 1>(v, tid, tab);
    __syncthreads();
    fft16_store<1>(buf, tid, v);
    __syncthreads();
    fft16_load<N>(buf, tid, v);
    fft16_math<DIR, 16>(v, tid, tab);
    __syncthreads();
    fft16_store<16>(buf, tid, v);
    __syncthreads();
    fft16_load<N>(buf, tid, v);
    fft16_math<DIR, 256>(v, tid, tab);
    __syncthreads();
    fft16_store<256>(buf, tid, v);
    __syncthreads();
    if constexpr (N > 4096) {
#pragma unroll
        for (int m = 0; m < 4096 / T; m++) fft_last<N, DIR>(buf, tid + m * T, tab);
        __syncthreads();
    }
}

// A transform of M = 256, 512 or 1024 points by ONE wave (lane = butterfly of the radix-16 passes; with M < 1024 the
// upper lanes idle), for the decimated sub-transforms of a longer one: two radix-16 passes (strides 1, 16) and one of
// radix M / 256 at stride 256.  `v` holds the input of the lane's first butterfly, element r = point lane + r M / 16
// (the caller forms it in registers); buf takes the exchanges and, on return (synchronised), the result in natural
// order.  Every wave of the workgroup must call it (the barriers are the workgroup's).
// The table factors are fetched by fft_wave_factors, which the caller places ahead of its own memory work: between
// the barriers of the transform every such load would be a round trip of its own.
template <int M>
struct FftWaveFactors {
    float2 p2[4];      // second pass: powers 1, 2, 4, 8 of the butterfly's factor
    float2 last[4][3]; // last pass: butterflies lane + 64 m, elements 1 .. M / 256 - 1
};

// (A lane's table entries lie far apart -- a gather of 64 cache lines per load of a wave --, so as few as possible are
// read: the last pass' factors exp(2 pi i r (lane + 64 m) / M) are the one of m = 0 times a power of exp(2 pi i 64 r / M),
// which is the same for all lanes; two multiplications at most, within the ten units of the last place the error
// account grants a derived factor.)
template <int M>
__device__ __forceinline__ void fft_wave_factors(FftWaveFactors<M> &f, const float2 *__restrict__ tab, int lane)
{
    const int k = lane & 15;
    constexpr int s2 = kFftTab / (16 * 16);
    f.p2[0] = tab[k * s2];
    f.p2[1] = tab[2 * k * s2];
    f.p2[2] = tab[4 * k * s2];
    f.p2[3] = tab[8 * k * s2];
    constexpr int R = M / 256, stride = kFftTab / M;
#pragma unroll
    for (int r = 1; r < R; r++) {
        const float2 base = tab[r * lane * stride], step = tab[r * 64 * stride], step2 = tab[r * 128 * stride];
        f.last[0][r - 1] = base;
        f.last[1][r - 1] = c_mul(base, step);
        f.last[2][r - 1] = c_mul(base, step2);
        f.last[3][r - 1] = c_mul(f.last[1][r - 1], step2);
    }
}

template <int M, int DIR>
__device__ __forceinline__ void fft_wave(float2 (&v)[16], float2 *buf, const FftWaveFactors<M> &f, int lane)
{
    static_assert(M == 256 || M == 512 || M == 1024, "sub-transform length");
    constexpr int B = M / 16;  // butterflies of a radix-16 pass
    const bool act = lane < B;
    if (act) {
        dft16<DIR>(v);  // first pass: stride 1, no factors
        fft16_store<1>(buf, lane, v);
    }
    __syncthreads();
    if (act) {
        fft16_load<M>(buf, lane, v);
        // second pass, stride 16: the factors' powers as fft16_math forms them
        float2 w[16];
        w[1] = c_dir<DIR>(f.p2[0]);
        w[2] = c_dir<DIR>(f.p2[1]);
        w[4] = c_dir<DIR>(f.p2[2]);
        w[8] = c_dir<DIR>(f.p2[3]);
        w[3] = c_mul(w[1], w[2]);
        w[5] = c_mul(w[1], w[4]);
        w[6] = c_mul(w[2], w[4]);
        w[9] = c_mul(w[1], w[8]);
        w[10] = c_mul(w[2], w[8]);
        w[12] = c_mul(w[4], w[8]);
        w[7] = c_mul(w[3], w[4]);
        w[11] = c_mul(w[3], w[8]);
        w[13] = c_mul(w[5], w[8]);
        w[14] = c_mul(w[6], w[8]);
        w[15] = c_mul(w[7], w[8]);
#pragma unroll
        for (int r = 1; r < 16; r++) v[r] = c_mul(v[r], w[r]);
        dft16<DIR>(v);
    }
    __syncthreads();
    if (act) fft16_store<16>(buf, lane, v);
    __syncthreads();
    constexpr int R = M / 256;
    if constexpr (R > 1) {
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int j = lane + 64 * m;  // butterfly j < 256: elements j + 256 r, factors exp(DIR 2 pi i r j / M)
            if constexpr (R == 2) {
                const float2 a = buf[fft_phys(j)];
                const float2 b = c_mul(buf[fft_phys(j + 256)], c_dir<DIR>(f.last[m][0]));
                buf[fft_phys(j)] = c_add(a, b);
                buf[fft_phys(j + 256)] = c_sub(a, b);
            } else {
                float2 a = buf[fft_phys(j)];
                float2 b = c_mul(buf[fft_phys(j + 256)], c_dir<DIR>(f.last[m][0]));
                float2 c = c_mul(buf[fft_phys(j + 512)], c_dir<DIR>(f.last[m][1]));
                float2 d = c_mul(buf[fft_phys(j + 768)], c_dir<DIR>(f.last[m][2]));
                dft4<DIR>(a, b, c, d);
                buf[fft_phys(j)] = a;
                buf[fft_phys(j + 256)] = b;
                buf[fft_phys(j + 512)] = c;
                buf[fft_phys(j + 768)] = d;
            }
        }
        __syncthreads();
    }
}
#endif

}  // namespace sgz
