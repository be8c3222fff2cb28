// Device kernels of the PTQ4ViT calibration engine for gfx950 (CDNA4, wave64).
//
// Pipeline of one search pass (reference quant_layers/linear.py:455-533 and siblings):
//   k_pack      fake-quantise an operand to int8 grid indices (or fp32 values), once per
//               candidate for the searched operand, once for the fixed one       [HBM-bound]
//   k_sweep     candidate-sweep GEMM on MFMA (i8->i32 or f32) with the similarity
//               metric fused into the epilogue -> per-column partial sums        [MFMA-bound]
//   k_finish    deterministic fixed-order reduction of the partials -> score[c][block]
//   k_select    argmax over candidates (first max, NaN counts as max) + gather of the interval
//
// No atomics on floating-point data anywhere: results are run-to-run and 1-vs-N-GPU identical.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace p4v {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------
// One kernel body, two entry points (round 6: grouped launches)
// ------------------------------------------------------------------------------------------
// Every kernel of the calibration path is a __device__ body k_x_body(params, blockIdx, gridDim) -- the parameters shadow HIP's
// built-ins, so the body reads exactly like a kernel -- with two __global__ entry points:
//   k_x(P p)              one launch = one module's grid                           (calibration_step2() of a single module)
//   k_x_g(GroupArgs<P> a) one launch = the CONCATENATED grids of several modules  (p4v_calibrate_group: the modules of a network are
//                         independent, reference utils/quant_calib.py:316-372, and search in lock step; a ViT-B calibration is
//                         ~3 800 launches of mostly 1-250 workgroups one module at a time, ~12-70 x fewer grouped)
// A workgroup of k_x_g finds its member m by a scalar scan of the block offsets, rebuilds the member's own (blockIdx, gridDim) from
// its flat index and runs the body on a.p[m] -- the same instructions on the same values as the single launch, so the results are
// bit-identical by construction.  Every member's block count is padded to a multiple of 8 (the XCD round-robin of the hardware
// dispatcher then sees each member's blocks as a launch of its own would have; the padding blocks return at once).  The
// argument block travels as a kernel argument: CAP members per launch (the kernel-argument segment, with the 256 bytes of
// implicit arguments the compiler appends, stays within 4 KB).
#ifndef P4V_GROUP_KERNARG
#define P4V_GROUP_KERNARG (4096 - 256)
#endif
template <typename P> struct GroupArgs {
    static constexpr int CAP_RAW = (P4V_GROUP_KERNARG - 16) / (int)(sizeof(P) + 16);
    static constexpr int CAP = CAP_RAW > 64 ? 64 : CAP_RAW;
    int n;
    unsigned off[CAP + 1];            // first flat block of member m; off[n] = total
    unsigned gx[CAP], gy[CAP], gz[CAP];
    P p[CAP];
};
#define P4V_BIDX uint3{blockIdx.x, blockIdx.y, blockIdx.z}
#define P4V_GDIM uint3{gridDim.x, gridDim.y, gridDim.z}
#define P4V_GROUP_ENTER(a)                                                                           \
    int m_ = 0;                                                                                      \
    while (m_ + 1 < a.n && blockIdx.x >= a.off[m_ + 1]) ++m_;                                        \
    m_ = __builtin_amdgcn_readfirstlane(m_);                                                         \
    const unsigned b_ = blockIdx.x - a.off[m_];                                                      \
    const uint3 vg_{a.gx[m_], a.gy[m_], a.gz[m_]};                                                   \
    if (b_ >= vg_.x * vg_.y * vg_.z) return;                                                         \
    const uint3 vb_{b_ % vg_.x, (b_ / vg_.x) % vg_.y, b_ / (vg_.x * vg_.y)}

// ------------------------------------------------------------------------------------------
// block abs-max (reference linear.py:385,395; matmul.py:435-436; conv.py:487,494)
// ------------------------------------------------------------------------------------------
// Monotonic float <-> uint encoding so that an integer atomicMax implements an exact
// (order-independent, hence deterministic) floating-point max.
__device__ __forceinline__ unsigned enc_ordered(float f) {
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_ordered(unsigned e) {
    unsigned b = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
    return __uint_as_float(b);
}

struct AbsMaxParams {
    const float* src;
    long s0, s1, s2, s3;      // element strides of the 4-D view [D0][D1][R][C]
    int D0, D1, R, C;
    int nV, nH, crb_r, crb_c; // row / column blocking of (R, C); D1 indexes the group (head)
    int row_tile;             // rows handled by one workgroup
    int signed_max;           // 1: plain max (post-GELU, linear.py:597), 0: abs max
    unsigned* out;            // [D1][nV][nH] ordered-encoded running max
};

__device__ __forceinline__ void k_absmax_body(const AbsMaxParams& p, const uint3 blockIdx, const uint3 gridDim) {
    // grid.x = ceil(crb_r / row_tile) * nV * nH, grid.y = D1, grid.z = D0
    int bx = blockIdx.x;
    const int h = bx % p.nH; bx /= p.nH;
    const int v = bx % p.nV; bx /= p.nV;
    const int rt = bx;
    const int r0 = v * p.crb_r + rt * p.row_tile;
    const int r1 = min(min(r0 + p.row_tile, (v + 1) * p.crb_r), p.R);
    const int c0 = h * p.crb_c;
    const int c1 = min(c0 + p.crb_c, p.C);
    const int w = c1 - c0;
    const float* base = p.src + (long)blockIdx.z * p.s0 + (long)blockIdx.y * p.s1;
    float m = -INFINITY;
    if (w > 0 && r1 > r0 && p.s3 == 1 && (w & 3) == 0 && (p.s2 & 3) == 0 && (c0 & 3) == 0 && ((((unsigned long long)base) & 15) == 0)) {
        // contiguous, 16-byte aligned rows (every Linear / MatMul operand as captured): dwordx4 loads, a fixed (row, column) split of
        // the 256 threads -- the element-wise path below pays an integer division per element and 4-byte loads (1.1-1.7 TB/s on
        // the activations of a ViT-B layer; this one is bound by the read)
        const int w4 = w >> 2;
        const int tpr = w4 >= 256 ? 256 : w4 >= 128 ? 128 : w4 >= 64 ? 64 : w4 >= 32 ? 32 : 16;
        const int tx = threadIdx.x & (tpr - 1), ty = threadIdx.x / tpr, rpar = 256 / tpr;
        for (int r = r0 + ty; r < r1; r += rpar) {
            const v4f* row = reinterpret_cast<const v4f*>(base + (long)r * p.s2 + c0);
            for (int c = tx; c < w4; c += tpr) {
                const v4f x = row[c];
                if (p.signed_max) m = fmaxf(fmaxf(m, fmaxf(x[0], x[1])), fmaxf(x[2], x[3]));
                else m = fmaxf(fmaxf(m, fmaxf(fabsf(x[0]), fabsf(x[1]))), fmaxf(fabsf(x[2]), fabsf(x[3])));
            }
        }
    } else if (w > 0 && r1 > r0) {
        const int n = (r1 - r0) * w;
        for (int i = threadIdx.x; i < n; i += 256) {
            const int r = r0 + i / w, c = c0 + i % w;
            float x = base[(long)r * p.s2 + (long)c * p.s3];
            m = fmaxf(m, p.signed_max ? x : fabsf(x));
        }
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (m > -INFINITY) atomicMax(p.out + ((long)blockIdx.y * p.nV + v) * p.nH + h, enc_ordered(m));
    }
}
__global__ __launch_bounds__(256) void k_absmax(AbsMaxParams p) { k_absmax_body(p, P4V_BIDX, P4V_GDIM); }
__global__ __launch_bounds__(256) void k_absmax_g(GroupArgs<AbsMaxParams> a) { P4V_GROUP_ENTER(a); k_absmax_body(a.p[m_], vb_, vg_); }

// interval[j] = max[j] / (qmax - 0.5)   (linear.py:385); `broadcast`: init_layerwise (linear.py:383)
struct IntervalParams { const unsigned* enc; int n; float denom; int broadcast; float* interval; };
__device__ __forceinline__ void k_interval_from_max_body(const IntervalParams& a, const uint3 blockIdx, const uint3 gridDim) {
    const auto& [enc, n, denom, broadcast, interval] = a;
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float m;
    if (broadcast) {
        m = -INFINITY;
        for (int i = 0; i < n; ++i) m = fmaxf(m, dec_ordered(enc[i]));
    } else {
        m = dec_ordered(enc[j]);
    }
    interval[j] = m / denom;
}
__global__ void k_interval_from_max(IntervalParams p) { k_interval_from_max_body(p, P4V_BIDX, P4V_GDIM); }
__global__ void k_interval_from_max_g(GroupArgs<IntervalParams> a) { P4V_GROUP_ENTER(a); k_interval_from_max_body(a.p[m_], vb_, vg_); }

// cands[c][j] = mult[c] * interval[j]  (fp32 multiply, linear.py:544-545)
struct CandsParams { const float* mult; const float* interval; int ncand, nblk; float* cands; };
__device__ __forceinline__ void k_make_cands_body(const CandsParams& a, const uint3 blockIdx, const uint3 gridDim) {
    const auto& [mult, interval, ncand, nblk, cands] = a;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncand * nblk) return;
    cands[i] = mult[i / nblk] * interval[i % nblk];
}
__global__ void k_make_cands(CandsParams p) { k_make_cands_body(p, P4V_BIDX, P4V_GDIM); }
__global__ void k_make_cands_g(GroupArgs<CandsParams> a) { P4V_GROUP_ENTER(a); k_make_cands_body(a.p[m_], vb_, vg_); }

// S[c][j] = X(c,j) * Y(c,j); each factor is a device array (with candidate / block strides) or a constant.
struct ScaleParams {
    const float* x; int x_cs, x_js; float x_const;
    const float* y; int y_cs, y_js; float y_const;
    int C, nblk;
    float* S;
};
__device__ __forceinline__ void k_scale_table_body(const ScaleParams& p, const uint3 blockIdx, const uint3 gridDim) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.C * p.nblk) return;
    const int c = i / p.nblk, j = i % p.nblk;
    const float x = p.x ? p.x[c * p.x_cs + j * p.x_js] : p.x_const;
    const float y = p.y ? p.y[c * p.y_cs + j * p.y_js] : p.y_const;
    p.S[i] = x * y;
}
__global__ void k_scale_table(ScaleParams p) { k_scale_table_body(p, P4V_BIDX, P4V_GDIM); }
__global__ void k_scale_table_g(GroupArgs<ScaleParams> a) { P4V_GROUP_ENTER(a); k_scale_table_body(a.p[m_], vb_, vg_); }

// ------------------------------------------------------------------------------------------
// k_pack: fake quantisation of one operand into a K-contiguous, zero-padded plane
// ------------------------------------------------------------------------------------------
enum PackMode {
    PACK_RAW = 0,     // copy (fp32 only)
    PACK_SYM = 1,     // clamp(rint(x/s), lo, hi)                      linear.py:167
    PACK_SOS_HI = 2,  // clamp(rint(clamp(x,split,1)*(q-1)), 0, q-1)   matmul.py:596
    PACK_SOS_LO = 3,  // clamp(rint(clamp(x,0,split)/(split/(q-1))), 0, q-1)   matmul.py:597
    PACK_SOS_SIM = 4, // fp32 only: hi/(q-1) + lo*(split/(q-1))        matmul.py:613-615
    PACK_TWIN_SIM = 5, // fp32 only: pos*s + neg*s_neg                  linear.py:605-607
    PACK_TWIN_I8 = 6   // int8 only: clamp(rint(x/s),0,hi) + clamp(rint(x/neg_scale),lo,0) -- the two grid indices of the post-GELU
                       // twin (linear.py:605-606) in ONE plane: their supports are disjoint (x > 0 clamps the negative range to 0,
                       // x < 0 the positive one), so the sum is the one that is not zero and fits the int8 range
};

struct PackParams {
    const float* src;
    long s_z, s_r, s_k;   // element strides of the logical [Z][R][K] view
    long s_z2; int zdiv;  // two-level batch: offset = (z / zdiv) * s_z2 + (z % zdiv) * s_z  (zdiv <= 0: single level)
    int Z, R, K;
    int Rp, Kp;           // padded plane: dst is [C][Z][Rp][Kp], [Z][Rp][C][Kp] when c_inner == 1,
                          // [Z][Rp][C/2][Kp/64][2][64] (candidate pairs interleaved per k-tile) when c_inner == 2,
                          // MFMA-fragment order of k_sweep6's register-stationary operand when c_inner == 3 (C == 1):
                          // 16-byte chunk ((((r / 64) * Kp/64 + k-tile) * 2 + (r / 32) % 2) * 2 + half) * 64 + g * 32 + r % 32
                          // for bytes [k-tile * 64 + half * 32 + g * 16, + 16) of row r -- one dwordx4 of a wave is 1 KB contiguous
    int c_inner;
    void* dst;
    int C;
    const float* scales;  // scales[c*sc_cs + blk]; for SOS modes: split candidates / the split
    int sc_cs;
    int blk_mode, blk_div, blk_div2; // 0: blk=0; 1: blk = min(r/blk_div, nblk_r-1)*nblk_k + k/blk_div2 (k part only if blk_div2); 2: blk = z % blk_div
    int nblk_r, nblk_k;
    int mode, lo, hi;
    float qm1;            // q-1 for SOS modes
    float neg_scale;      // twin: fixed negative-range interval
    // pruned passes: only the candidates in [crange[0], crange[1]) - c_base are packed, and of those only the ones whose `done`
    // flag (one per candidate) is still 0 (planes kept across the rounds of a call)
    const int* crange; int c_base; const unsigned char* done;
    // optional im2col gather (conv): logical r = (b, oy, ox), k = (ci, ki, kj)
    int conv, ic, H, W, kh, kw, sh, sw, ph, pw, dh, dw, fw, L;
    float qbias;          // != 0 (set by launch_pack for PACK_SYM on the full symmetric 8-bit grid): quant16_sat8 with this bias
};

__device__ __forceinline__ float pack_value(const PackParams& p, float x, float s) {
    switch (p.mode) {
        case PACK_SYM: return fminf(fmaxf(rintf(x / s), (float)p.lo), (float)p.hi);
        case PACK_SOS_HI: return fminf(fmaxf(rintf(fminf(fmaxf(x, s), 1.0f) * p.qm1), 0.0f), p.qm1);
        case PACK_SOS_LO: {
            const float a_int = s / p.qm1;
            return fminf(fmaxf(rintf(fminf(fmaxf(x, 0.0f), s) / a_int), 0.0f), p.qm1);
        }
        default: return x;
    }
}

__device__ __forceinline__ float pack_value_f32(const PackParams& p, float x, float s) {
    switch (p.mode) {
        case PACK_RAW: return x;
        case PACK_SYM: return fminf(fmaxf(rintf(x / s), (float)p.lo), (float)p.hi) * s;
        case PACK_SOS_SIM: {
            const float a_int = s / p.qm1;
            const float hi = fminf(fmaxf(rintf(fminf(fmaxf(x, s), 1.0f) * p.qm1), 0.0f), p.qm1) / p.qm1;
            const float lo = fminf(fmaxf(rintf(fminf(fmaxf(x, 0.0f), s) / a_int), 0.0f), p.qm1) * a_int;
            return hi + lo;
        }
        case PACK_TWIN_SIM: {
            const float pos = fminf(fmaxf(rintf(x / s), 0.0f), (float)p.hi) * s;
            const float neg = fminf(fmaxf(rintf(x / p.neg_scale), (float)p.lo), 0.0f) * p.neg_scale;
            return pos + neg;
        }
        default: return x;
    }
}

__device__ __forceinline__ float pack_load(const PackParams& p, const float* zbase, int r, int k) {
    if (r >= p.R || k >= p.K) return 0.0f;
    if (!p.conv) return zbase[(long)r * p.s_r + (long)k * p.s_k];
    const int b = r / p.L, l = r % p.L;   // zbase already points at image z (flat layout: Z = 1, b = r / L)
    const int oy = l / p.fw, ox = l % p.fw;
    const int kk = p.kh * p.kw;
    const int ci = k / kk, ki = (k % kk) / p.kw, kj = k % p.kw;
    const int y = oy * p.sh - p.ph + ki * p.dh, x = ox * p.sw - p.pw + kj * p.dw;
    if (y < 0 || y >= p.H || x < 0 || x >= p.W) return 0.0f;
    return zbase[(((long)b * p.ic + ci) * p.H + y) * p.W + x];
}

__device__ __forceinline__ int pack_blk(const PackParams& p, int z, int r, int k) {
    if (p.blk_mode == 1) {
        int b = min(r / p.blk_div, p.nblk_r - 1);
        if (p.blk_div2) b = b * p.nblk_k + min(k / p.blk_div2, p.nblk_k - 1);
        return b;
    }
    if (p.blk_mode == 2) return z % p.blk_div;
    return 0;
}

// Exact clamp(rint(x / s), lo, hi) without a division per element.  With r = fl(1/s) correctly rounded,
// t = fl(x * r) = (x/s)(1+e), |e| <= 2^-23, and q = fl(x/s) = (x/s)(1+e'), |e'| <= 2^-24: rint(t) and rint(q) can
// only differ if a half-integer lies within |x/s| * 2^-22 of t.  t is first clamped to [lo - 0.49, hi + 0.49]
// (beyond that both paths saturate to the same index, and |lo|, |hi| <= 128 bounds the distance by 3.2e-5), so an
// element needs the IEEE division only if |tc - rint(tc)| > 0.5 - 4e-5: about 1 in 12 000; the lane then redoes its
// run (5 % of the waves take that branch).  rint is done by adding 1.5 * 2^23 (round-to-nearest-even of the add):
// the low byte of the sum's bit pattern is the two's-complement grid index, which is what the plane stores.
static constexpr float PACK_MAGIC = 12582912.0f;
__device__ __forceinline__ unsigned quant_fast1(float x, float r, float lo49, float hi49, float magic, float& maxdev) {
    const float t = __builtin_amdgcn_fmed3f(x * r, lo49, hi49);
    const float biased = t + magic;
#ifndef P4V_PACK_DBG        // timing-only ablation: no exactness check
    maxdev = fmaxf(maxdev, fabsf(t - (biased - magic)));
#endif
    return __builtin_bit_cast(unsigned, biased);
}

// ---- exact 8-bit quantisation of 16 values at 4.5 VALU operations per element (round 5) --------------------------------------
// clamp(rint(x / s), -128, 127) needs, per element, a product, a rounding, a clamp and a byte insert -- and the proof that
// x * fl(1/s) rounds like the IEEE quotient the reference divides (linear.py:167).  quant_fast1 spends 6.75 operations on it
// (mul, med3, add magic, sub, sub, max-abs, 3/4 perm).  v_cvt_pk_u8_f32 converts, SATURATES to [0, 255] and inserts the byte in
// ONE operation, so with u = fma(x, r, bias) (one rounding of the exact product + bias; bias = 128.5 where the conversion
// truncates, 128 where it rounds to nearest -- probed once per process, k_probe_cvt) the grid index + 128 is two operations,
// and the proof is the same two operations again with the bias moved: lo = cvt(fma(x, r, bias - d)), hi = cvt(fma(x, r, bias + d)),
// d = 2^-14.  With q = fl(x / s): |x r - q| <= 3 * 2^-24 |x / s| <= 2.3e-5 inside the unsaturated range and the fma's own rounding
// is <= 2^-16 below 512, so u_lo < q + bias < u_hi strictly, by more than 2e-5 on either side; the conversion is monotone, hence
// lo <= cvt(q + bias) <= hi, and where q is an exact tie (q + bias lands on the conversion's own breakpoint) lo != hi.  So
// lo == hi  =>  that byte is clamp(rint(q), -128, 127) + 128; a dword whose four bytes do not all agree (1.2e-4 of the elements sit
// within d of a breakpoint: 0.05 % of the dwords) is redone with the IEEE division, as is everything when 1/s overflowed.  One
// xor turns the four biased bytes into two's complement.  Only for the full symmetric 8-bit grid (the saturation IS the clamp);
// other grids keep quant_fast1.
__global__ void k_probe_cvt(unsigned* out) {
    const float v[5] = {0.7f, 1.5f, 2.5f, -3.0f, 300.0f};
    if (threadIdx.x < 5) out[threadIdx.x] = __builtin_amdgcn_cvt_pk_u8_f32(v[threadIdx.x], 0u, 0u);
}
// The band, tightened for the conversion this part has (round to nearest: bias 128): |x r - q| <= 2 * 2^-24 * 128.5 = 1.53e-5
// (r = fl(1/s) and q = fl(x/s) are each within 2^-24 of the quotient) + the fma's own rounding, at most 2^-17 = 7.6e-6 below 256
// (above, both conversions saturate) = 2.3e-5; the two biases are the nearest floats that clear it: 128 - 5 * 2^-17 (3.8e-5) and
// 128 + 2 * 2^-16 (3.05e-5): 6.9e-5 of the elements are flagged.  Any other bias (a truncating conversion): 2^-14 on either side.
static constexpr float QUANT_D = 6.103515625e-05f;                 // 2^-14
__device__ __forceinline__ void quant16_sat8(const float (&x)[16], float s, float rcp, float qbias, v4i& out) {
    const bool rne = qbias == 128.0f;
    const float b1 = rne ? 127.99996185302734375f : qbias - QUANT_D, b2 = rne ? 128.000030517578125f : qbias + QUANT_D;
    unsigned lo[4], hi[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        lo[q] = 0u; hi[q] = 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lo[q] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(x[q * 4 + e], rcp, b1), (unsigned)e, lo[q]);
            hi[q] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(x[q * 4 + e], rcp, b2), (unsigned)e, hi[q]);
        }
    }
    const bool inf = !(rcp < 3.0e38f);
    const bool bad = inf || lo[0] != hi[0] || lo[1] != hi[1] || lo[2] != hi[2] || lo[3] != hi[3];
    if (__any(bad)) {
        float sd = s;
        asm volatile("" : "+v"(sd));       // the divisions depend on this: they cannot be hoisted out of the rare branch
        if (bad) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (inf || lo[q] != hi[q]) {
                    unsigned w = 0u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = fminf(fmaxf(rintf(x[q * 4 + e] / sd), -128.0f), 127.0f);
                        w |= (unsigned)(((int)v + 128) & 0xff) << (8 * e);
                    }
                    lo[q] = w;
                }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) out[q] = (int)(lo[q] ^ 0x80808080u);
}
// ... and the general grid (any clamp, <= 8 bit): quant_fast1 + its exactness check (k_pack's hot path)
__device__ __forceinline__ void quant16_any(const float (&x)[16], float s, float rcp, float flo, float fhi, bool wide, v4i& out) {
    unsigned qb[16];
    float maxdev = 0.0f, magic = PACK_MAGIC;
    asm volatile("" : "+v"(magic));
#pragma unroll
    for (int e = 0; e < 16; ++e) qb[e] = quant_fast1(x[e], rcp, flo - 0.49f, fhi + 0.49f, magic, maxdev);
    const bool bad = !(maxdev <= 0.49996f) || !(rcp < 3.0e38f) || wide;
    if (__any(bad)) {
        float sd = s;
        asm volatile("" : "+v"(sd));
        if (bad) {
#pragma unroll
            for (int e = 0; e < 16; ++e) qb[e] = __builtin_bit_cast(unsigned, fminf(fmaxf(rintf(x[e] / sd), flo), fhi) + PACK_MAGIC);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned lo16 = __builtin_amdgcn_perm(qb[q * 4 + 1], qb[q * 4], 0x0c0c0400u);
        const unsigned hi16 = __builtin_amdgcn_perm(qb[q * 4 + 3], qb[q * 4 + 2], 0x0c0c0400u);
        out[q] = (int)(lo16 | (hi16 << 16));
    }
}

// One thread owns 16 consecutive k of one (z, r) for a group of PACK_CG candidates (blockIdx.y): the source
// (L2 / Infinity-Cache resident: it is re-read once per candidate group) is loaded once per group, the scales
// are loaded up front, and every plane is written as one contiguous stream of full 16-byte (int8) /
// 64-byte (fp32) runs -- a pure streaming-write kernel.
static constexpr int PACK_CG = 10;

template <typename T>
__device__ __forceinline__ void k_pack_body(const PackParams& p, const uint3 blockIdx, const uint3 gridDim) {
    const unsigned kchunks = p.Kp / 16;
    const unsigned total = (unsigned)p.Z * p.Rp * kchunks;      // < 2^31: checked by the launcher
    const int cbeg = blockIdx.y * PACK_CG, cend = min(p.C, cbeg + PACK_CG);
    unsigned need = (1u << (cend - cbeg)) - 1u;            // the candidates of this group to pack (bit j: candidate cbeg + j)
    if (p.crange) {
        const int a = p.crange[0] - p.c_base, b = p.crange[1] - p.c_base;
        if (cend <= a || cbeg >= b) return;
        for (int j = 0; j < PACK_CG; ++j)
            if (cbeg + j < a || cbeg + j >= b || (p.done && cbeg + j < cend && p.done[cbeg + j])) need &= ~(1u << j);
        if (!need) return;
    }
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned row = i / kchunks;
        const int kc = (int)(i - row * kchunks);
        const int z = (int)(row / (unsigned)p.Rp);
        const int r = (int)(row - (unsigned)z * p.Rp);
        const float* zbase = p.zdiv > 0 ? p.src + (long)(z / p.zdiv) * p.s_z2 + (long)(z % p.zdiv) * p.s_z
                                        : p.src + (long)z * p.s_z;
        float x[16];
        int blk[16];
        const bool per_k_blk = (p.blk_mode == 1 && p.blk_div2);
        // a whole 16-element run of a K-contiguous, 16-byte aligned source: four dwordx4 loads (the scalar path below costs
        // 16 load instructions per thread, each touching 32 cache lines per wave)
        const bool vec = !p.conv && p.s_k == 1 && r < p.R && kc * 16 + 16 <= p.K && (p.s_r & 3) == 0 &&
                         ((((unsigned long long)zbase) & 15) == 0);
        if (vec) {
            const v4f* src4 = reinterpret_cast<const v4f*>(zbase + (long)r * p.s_r + kc * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const v4f t = src4[q];
                x[q * 4 + 0] = t[0]; x[q * 4 + 1] = t[1]; x[q * 4 + 2] = t[2]; x[q * 4 + 3] = t[3];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) blk[e] = per_k_blk ? pack_blk(p, z, r, kc * 16 + e) : 0;
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                x[e] = pack_load(p, zbase, r, kc * 16 + e);
                blk[e] = per_k_blk ? pack_blk(p, z, r, min(kc * 16 + e, p.K - 1)) : 0;
            }
        }
        const int blk0 = pack_blk(p, z, r, 0);
        const bool live = (r < p.R);
        float sc[PACK_CG];
#pragma unroll
        for (int j = 0; j < PACK_CG; ++j)
            sc[j] = (p.scales && cbeg + j < cend) ? p.scales[(cbeg + j) * p.sc_cs + blk0] : p.neg_scale;
        for (int j = 0; j < PACK_CG; ++j) {     // not unrolled: one candidate's body is already ~150 instructions
            const int c = cbeg + j;
            if (c >= cend) break;
            if (!((need >> j) & 1u)) continue;
            const long o = p.c_inner == 3 ? ((long)z * p.Rp * p.Kp + (((((long)(r >> 6) * (p.Kp >> 6) + (kc >> 2)) * 2 + ((r >> 5) & 1)) * 2 + ((kc >> 1) & 1)) * 64 + (kc & 1) * 32 + (r & 31)) * 16)
                         : p.c_inner == 2 ? ((((long)z * p.Rp + r) * ((p.C + 1) & ~1) + (c & ~1)) * p.Kp + (long)(kc >> 2) * 128 + (c & 1) * 64 + (kc & 3) * 16)
                         : p.c_inner ? ((((long)z * p.Rp + r) * p.C + c) * p.Kp + (long)kc * 16)
                                     : ((((long)c * p.Z + z) * p.Rp + r) * p.Kp + (long)kc * 16);
            if constexpr (sizeof(T) == 1) {
                const float s = sc[j];
                int w[4];
                if (p.mode == PACK_SYM && live && kc * 16 + 16 <= p.K && p.qbias != 0.0f) {
                    // hot path on the full symmetric 8-bit grid: 4.5 operations per element (quant16_sat8)
                    v4i o4;
                    quant16_sat8(x, s, 1.0f / s, p.qbias, o4);
                    w[0] = o4[0]; w[1] = o4[1]; w[2] = o4[2]; w[3] = o4[3];
                } else if (p.mode == PACK_SYM && live && kc * 16 + 16 <= p.K) {
                    // hot path: symmetric grid, no padding inside this 16-element run
                    const float rcp = 1.0f / s, flo = (float)p.lo, fhi = (float)p.hi;
                    unsigned qb[16];
                    float maxdev = 0.0f;
                    float magic = PACK_MAGIC;
                    asm volatile("" : "+v"(magic));
#pragma unroll
                    for (int e = 0; e < 16; ++e) qb[e] = quant_fast1(x[e], rcp, flo - 0.49f, fhi + 0.49f, magic, maxdev);
                    // 1/s overflowed, a grid wider than 8 bit, or a NaN: divide
                    const bool bad = !(maxdev <= 0.49996f) || !(rcp < 3.0e38f) || !(fmaxf(-flo, fhi) < 129.0f);
                    if (__any(bad)) {
                        float sd = s;
                        asm volatile("" : "+v"(sd));   // the divisions depend on this: they cannot be hoisted out of the rare branch
                        if (bad) {
#pragma unroll
                            for (int e = 0; e < 16; ++e)
                                qb[e] = __builtin_bit_cast(unsigned, fminf(fmaxf(rintf(x[e] / sd), flo), fhi) + PACK_MAGIC);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned lo16 = __builtin_amdgcn_perm(qb[q * 4 + 1], qb[q * 4], 0x0c0c0400u);
                        const unsigned hi16 = __builtin_amdgcn_perm(qb[q * 4 + 3], qb[q * 4 + 2], 0x0c0c0400u);
                        w[q] = (int)(lo16 | (hi16 << 16));   // (an OR of two perms with selector 0x04000c0c is mis-folded by the backend)
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        int acc = 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int kk = kc * 16 + q * 4 + e;
                            float v = (live && kk < p.K) ? pack_value(p, x[q * 4 + e], s) : 0.0f;
                            acc |= ((int)v & 0xff) << (8 * e);
                        }
                        w[q] = acc;
                    }
                }
                // streaming store: the plane is far larger than the L2 and is read back by a different kernel
                __builtin_nontemporal_store(v4i{w[0], w[1], w[2], w[3]}, reinterpret_cast<v4i*>(reinterpret_cast<int8_t*>(p.dst) + o));
            } else {
                float* d = reinterpret_cast<float*>(p.dst) + o;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v4f v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int kk = kc * 16 + q * 4 + e;
                        const float s = !p.scales ? 1.0f : per_k_blk ? p.scales[c * p.sc_cs + blk[q * 4 + e]] : sc[j];
                        v[e] = (live && kk < p.K) ? pack_value_f32(p, x[q * 4 + e], s) : 0.0f;
                    }
                    *reinterpret_cast<v4f*>(d + q * 4) = v;
                }
            }
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_pack(PackParams p) { k_pack_body<T>(p, P4V_BIDX, P4V_GDIM); }
template <typename T>
__global__ __launch_bounds__(256) void k_pack_g(GroupArgs<PackParams> a) { P4V_GROUP_ENTER(a); k_pack_body<T>(a.p[m_], vb_, vg_); }

// PACK_TWIN_I8: both grid indices of the post-GELU twin (linear.py:605-606) in ONE int8 plane, row-major [Z][Rp][Kp], zero
// padded.  A kernel of its own: the plane is small and fixed (one per weight-search pass) and k_pack's hot path is register
// sensitive (the extra branch there cost it an occupancy step: 112 -> 155 VGPRs).  IEEE divisions, as the reference divides;
// the supports are disjoint (x > 0 clamps the negative range to 0, x < 0 the positive one), so the sum is the index that is
// not zero and fits the int8 range.
__device__ __forceinline__ void k_pack_twin_body(const PackParams& p, const uint3 blockIdx, const uint3 gridDim) {
    const unsigned kchunks = p.Kp / 16;
    const unsigned total = (unsigned)p.Z * p.Rp * kchunks;
    const float s = p.scales[0], sn = p.neg_scale, flo = (float)p.lo, fhi = (float)p.hi;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned row = i / kchunks;
        const int kc = (int)(i - row * kchunks);
        const int z = (int)(row / (unsigned)p.Rp);
        const int r = (int)(row - (unsigned)z * p.Rp);
        const float* zbase = p.src + (long)z * p.s_z;
        int w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int acc = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = kc * 16 + q * 4 + e;
                const float x = pack_load(p, zbase, r, k);           // 0 outside the valid rows / columns
                const float v = fminf(fmaxf(rintf(x / s), 0.0f), fhi) + fminf(fmaxf(rintf(x / sn), flo), 0.0f);
                acc |= ((int)v & 0xff) << (8 * e);
            }
            w[q] = acc;
        }
        *reinterpret_cast<v4i*>(reinterpret_cast<int8_t*>(p.dst) + (((long)z * p.Rp + r) * p.Kp + (long)kc * 16)) = v4i{w[0], w[1], w[2], w[3]};
    }
}
__global__ __launch_bounds__(256) void k_pack_twin(PackParams p) { k_pack_twin_body(p, P4V_BIDX, P4V_GDIM); }
__global__ __launch_bounds__(256) void k_pack_twin_g(GroupArgs<PackParams> a) { P4V_GROUP_ENTER(a); k_pack_twin_body(a.p[m_], vb_, vg_); }


// Both int8 planes of a twin operand from ONE read of the source: the post-GELU twin's positive / negative range
// (linear.py:605-606: two PACK_SYM planes with clamps [0, hi] / [lo, 0]) or the split-of-softmax pair (matmul.py:595-598:
// PACK_SOS_HI / PACK_SOS_LO).  Same arithmetic as k_pack's per-element path (pack_value: IEEE division), two fixed planes
// (C = 1), one scale each (no blocks), row-major [Z][Rp][Kp].  Memory-bound: the source is read once instead of twice
// (quant_forward at batch 128: 310 MB per fc2, 238 MB per attention-probability operand).
struct PackDualParams { PackParams p; PackParams p2; };
__device__ __forceinline__ void k_pack_dual_body(const PackDualParams& a_, const uint3 blockIdx, const uint3 gridDim) {
    const auto& [p, p2] = a_;
    const unsigned kchunks = p.Kp / 16;
    const unsigned total = (unsigned)p.Z * p.Rp * kchunks;
    const float s1 = p.scales ? p.scales[0] : p.neg_scale, s2 = p2.scales ? p2.scales[0] : p2.neg_scale;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned row = i / kchunks;
        const int kc = (int)(i - row * kchunks);
        const int z = (int)(row / (unsigned)p.Rp);
        const int r = (int)(row - (unsigned)z * p.Rp);
        const float* zbase = p.zdiv > 0 ? p.src + (long)(z / p.zdiv) * p.s_z2 + (long)(z % p.zdiv) * p.s_z
                                        : p.src + (long)z * p.s_z;
        float x[16];
        const bool vec = p.s_k == 1 && r < p.R && kc * 16 + 16 <= p.K && (p.s_r & 3) == 0 && ((((unsigned long long)zbase) & 15) == 0);
        if (vec) {
            const v4f* src4 = reinterpret_cast<const v4f*>(zbase + (long)r * p.s_r + kc * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const v4f t = src4[q];
                x[q * 4 + 0] = t[0]; x[q * 4 + 1] = t[1]; x[q * 4 + 2] = t[2]; x[q * 4 + 3] = t[3];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) x[e] = pack_load(p, zbase, r, kc * 16 + e);
        }
        const bool live = r < p.R;
        int w1[4], w2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int a1 = 0, a2 = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool in = live && kc * 16 + q * 4 + e < p.K;
                const float v1 = in ? pack_value(p, x[q * 4 + e], s1) : 0.0f;
                const float v2 = in ? pack_value(p2, x[q * 4 + e], s2) : 0.0f;
                a1 |= ((int)v1 & 0xff) << (8 * e);
                a2 |= ((int)v2 & 0xff) << (8 * e);
            }
            w1[q] = a1; w2[q] = a2;
        }
        const long o = ((long)z * p.Rp + r) * p.Kp + (long)kc * 16;
        __builtin_nontemporal_store(v4i{w1[0], w1[1], w1[2], w1[3]}, reinterpret_cast<v4i*>(reinterpret_cast<int8_t*>(p.dst) + o));
        __builtin_nontemporal_store(v4i{w2[0], w2[1], w2[2], w2[3]}, reinterpret_cast<v4i*>(reinterpret_cast<int8_t*>(p2.dst) + o));
    }
}
__global__ __launch_bounds__(256) void k_pack_dual(PackDualParams p) { k_pack_dual_body(p, P4V_BIDX, P4V_GDIM); }
__global__ __launch_bounds__(256) void k_pack_dual_g(GroupArgs<PackDualParams> a) { P4V_GROUP_ENTER(a); k_pack_dual_body(a.p[m_], vb_, vg_); }

// Plain 2-D helpers behind p4v_quantize_i8 / p4v_fake_quant (quant_forward building blocks).
__global__ void k_fake_quant_rows(const float* x, long rows, long cols, const float* scales, long rows_per_scale,
                                  float lo, float hi, float* y) {
    const long n = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float s = scales[(i / cols) / rows_per_scale];
        y[i] = fminf(fmaxf(rintf(x[i] / s), lo), hi) * s;
    }
}

// ------------------------------------------------------------------------------------------
// k_multi_copy: the capture pass's "append to the cache" for ALL hooked tensors of a sub-batch in one launch
// ------------------------------------------------------------------------------------------
// After each replay of the captured sub-batch pass ~280 tensors (inputs, outputs, output gradients of every wrapped
// module) are copied from the graph's static buffers into slice i of their caches.  torch does that with one memcpy
// per tensor (2 260 copy kernels per ViT-B calibration); here a device table {src, dst base, bytes} drives one grid:
// blockIdx.y = tensor, blockIdx.x strides over its 16-byte words.  dst = dst base + index * bytes.
__global__ __launch_bounds__(256) void k_multi_copy(const long* table, long index) {
    const long* e = table + 3 * (long)blockIdx.y;
    const char* src = reinterpret_cast<const char*>(e[0]);
    const long bytes = e[2];
    char* dst = reinterpret_cast<char*>(e[1]) + index * bytes;
    if (((e[0] | (long)(size_t)dst | bytes) & 15) == 0) {
        const long n = bytes >> 4;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
            reinterpret_cast<v4i*>(dst)[i] = reinterpret_cast<const v4i*>(src)[i];
    } else {                                             // fp32 tensors: 4-byte granularity is always possible
        const long n = bytes >> 2;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
            reinterpret_cast<int*>(dst)[i] = reinterpret_cast<const int*>(src)[i];
    }
}

// ------------------------------------------------------------------------------------------
// k_export: integer image of a calibrated operand (the reference's utils/integer.py formats)
// ------------------------------------------------------------------------------------------
// One kernel for every export format: the source is a logical 4-D tensor [d0][d1][d2][d3] read through element
// strides (a weight exported once per V-block interval has stride 0 on d0; matmul operands keep torch's strides), the
// destination is contiguous.  Scale index of region r: sum_i (d_i / div_r[i]) * ss_r[i] -- the block geometry of the
// (n_G, n_V, n_H) padding view (integer.py:28-43: padding is cropped again, so only the block of each REAL element
// matters).  Arithmetic as the reference writes it: IEEE division, round-half-even, clamp; the split-of-softmax high
// range multiplies by (q-1) (integer.py:88) where every other range divides.  HBM-bound: 4 B read, 1 B written.
enum ExportMode {
    EXP_SYM_I8 = 0,     // int8  clamp(rint(x / s1), lo1, hi1)                                       integer.py:16,75,36
    EXP_SYM_F32 = 1,    // fp32  same value (quantize_matmul_input returns the float grid index)      integer.py:36
    EXP_GELU_U8 = 2,    // uint8 (clamp(rint(x / s1), 0, hi1) + 128) + |clamp(rint(x / s2), lo2, 0)|  integer.py:63-70
    EXP_SOS_U8 = 3      // uint8 (clamp(rint(clamp(x, s1, 1) * qm1), 0, qm1) + 128) + clamp(rint(clamp(x, 0, s1) / s2), 0, qm1)
                        //       with s1 = split, s2 = A_interval; uint8 wrap-around like the reference  integer.py:88-94
};

struct ExportParams {
    const float* src; long ss[4]; int d[4];
    const float* s1; long s1s[4]; int s1d[4];
    const float* s2; long s2s[4]; int s2d[4]; float s2_const;
    int mode, lo1, hi1, lo2, hi2; float qm1;
    void* dst;
};

__global__ __launch_bounds__(256) void k_export(ExportParams p) {
    const long n3 = p.d[3], n23 = (long)p.d[2] * n3, n123 = (long)p.d[1] * n23, total = (long)p.d[0] * n123;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int i0 = (int)(i / n123), i1 = (int)((i % n123) / n23), i2 = (int)((i % n23) / n3), i3 = (int)(i % n3);
        const float x = p.src[i0 * p.ss[0] + i1 * p.ss[1] + i2 * p.ss[2] + i3 * p.ss[3]];
        const float s1 = p.s1[(i0 / p.s1d[0]) * p.s1s[0] + (i1 / p.s1d[1]) * p.s1s[1] + (i2 / p.s1d[2]) * p.s1s[2] + (i3 / p.s1d[3]) * p.s1s[3]];
        const float s2 = p.s2 ? p.s2[(i0 / p.s2d[0]) * p.s2s[0] + (i1 / p.s2d[1]) * p.s2s[1] + (i2 / p.s2d[2]) * p.s2s[2] + (i3 / p.s2d[3]) * p.s2s[3]]
                              : p.s2_const;
        if (p.mode == EXP_SYM_I8 || p.mode == EXP_SYM_F32) {
            const float q = fminf(fmaxf(rintf(x / s1), (float)p.lo1), (float)p.hi1);
            if (p.mode == EXP_SYM_I8) reinterpret_cast<int8_t*>(p.dst)[i] = (int8_t)(int)q;
            else reinterpret_cast<float*>(p.dst)[i] = q;
        } else if (p.mode == EXP_GELU_U8) {
            const float pos = fminf(fmaxf(rintf(x / s1), 0.0f), (float)p.hi1);
            const float neg = fabsf(fminf(fmaxf(rintf(x / s2), (float)p.lo2), 0.0f));
            reinterpret_cast<uint8_t*>(p.dst)[i] = (uint8_t)(((int)pos + 128 + (int)neg) & 0xff);
        } else {
            const float hi = fminf(fmaxf(rintf(fminf(fmaxf(x, s1), 1.0f) * p.qm1), 0.0f), p.qm1);
            const float lo = fminf(fmaxf(rintf(fminf(fmaxf(x, 0.0f), s1) / s2), 0.0f), p.qm1);
            reinterpret_cast<uint8_t*>(p.dst)[i] = (uint8_t)(((int)hi + 128 + (int)lo) & 0xff);
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_sweep: the candidate-sweep GEMM with the similarity metric fused into the epilogue
// ------------------------------------------------------------------------------------------
enum EpiMode {
    EPI_SQ_W = 0,  // (w*d)^2     hessian (w = raw_grad), square_weighted_L2 (w = raw_out)
    EPI_SQ = 1,    // d^2         L2_norm
    EPI_ABS = 2,   // |d|         L1_norm
    EPI_W_SQ = 3,  // w*d^2       linear_weighted_L2 (w = |raw_out|)
    EPI_COS = 4,   // cosine: per-row dot / norm partials (rows of the MFMA tile = features)
    EPI_STORE = 5  // no metric: store raw_out - bias - scale*acc as an fp32 [M][N] tensor (candidate-invariant
                   // part of a twin operand folded into the target, see linear_impl)
    ,EPI_FWD = 6   // quant_forward: store scale*acc (+ scale2*acc2) + bias -- the quantised layer's output
    ,EPI_COS_T = 7 // k_sweep6 only: cosine with the SAMPLES on the stationary side (weight search): transposed MFMA output
};

// Device-side candidate range of a pruned pass (exact branch-and-bound, p4v_api.hip::run_pass_pruned): when `crange` is set
// a workgroup only evaluates the candidates of its group that lie in [crange[0], crange[1]); the others are provably not the
// argmax and k_finish gives them the score -inf.  The range lives in device memory: it is computed by the prune kernels from
// the scores of earlier stages of the same pass, stream-ordered, without a host round trip.
__device__ __forceinline__ void clip_crange(const int* crange, int& lo, int& hi, bool even = false) {
    if (!crange) return;
    int a = __builtin_amdgcn_readfirstlane(crange[0]), b = __builtin_amdgcn_readfirstlane(crange[1]);
    if (even) { a &= ~1; b = (b + 1) & ~1; }           // kernels that run candidate pairs: a superset on pair boundaries
    lo = max(lo, a); hi = min(hi, b);
}

// ... and per score block (prune_hull's rblk: 2 ints per block): the kernel's tile lies inside ONE score block `blk`
__device__ __forceinline__ void clip_crange_blk(const int* rblk, int blk, int& lo, int& hi) {
    if (!rblk) return;
    const int a = __builtin_amdgcn_readfirstlane(rblk[2 * blk]), b = __builtin_amdgcn_readfirstlane(rblk[2 * blk + 1]);
    lo = max(lo, a); hi = min(hi, b);
}

struct SweepParams {
    const int* crange;                 // optional device-side candidate range (see clip_crange)
    const int* crange_blk; int cb_div; // optional per-score-block ranges, block = z % cb_div (head-wise MatMul searches); k_sweep, k_sweep2,
                                       // k_sweep8, k_sweep9 honour them (k_sweep2g / k_bound never get them: run_pass)
    const void* A;  long a_cs, a_zs;   // byte strides between candidates / batch entries (0 = shared)
    const void* A2; long a2_cs, a2_zs; // twin second plane (post-GELU negative range / SoS low range)
    const void* B;  long b_cs, b_zs;
    int ldk;                           // bytes per operand row (Kp * sizeof(T)), multiple of 64
    int ktiles;                        // ldk / 64
    const float* S1; const float* S2;  // [C][nsb] combined scale of plane 1 / 2 (NULL -> 1)
    int s_cs;                          // nsb (scales per candidate)
    int sb_mode, sb_div;               // 0: sb = 0; 1: sb = n / sb_div; 2: sb = z % sb_div
    const float* bias;                 // per-column (n) bias, or per-row (m) when bias_axis = 1; NULL = none
    int bias_axis; long bias_zs;       // bias element offset per z (V-block batches of the swapped cosine sweep)
    const float* O;                    // raw_out
    const float* Wt;                   // metric weight source (raw_grad) or NULL
    int wt_mode;                       // 0: none, 1: Wt[idx] (hessian), 2: raw_out, 3: |raw_out|
    // element index = z*o_zs + (m / o_inner)*o_bs + (m % o_inner)*o_ms + (n / o_ninner)*o_nbs + (n % o_ninner)*o_ns
    long o_zs, o_bs, o_ms, o_nbs, o_ns;
    int o_inner, o_ninner;
    int M, N, Z;
    int c0, c1;                        // candidate range of this launch
    float* part;                       // [C][Z][MT][Np] per-column partial sums (MT = Mp/64 row slabs)
    long p_cs, p_zs;                   // element strides of `part`
    int Np;
    int mtiles, ntiles;
    int dbg;                           // tuning experiments only (0 in production): 1 = no operand loads, 2 = no MFMA
    float* store;                      // EPI_STORE output [M][N]
    int halves;                        // k_sweep9: workgroups per batch entry (part layout [C][Z][halves * 8]); 0 otherwise
    int rows_p_stream;                 // k_sweep9: padded rows of the streamed operand plane
    int bound;                         // 1: the one-candidate bound pass of a pruned search -> k_bound (no candidate loop)
    int b_rows;                        // k_sweep2: rows of the B plane per batch entry when it is NOT padded to the 128-column tile
                                       // (64: attn.v, N = head_dim); 0 = padded.  The tile's other rows re-read these (masked columns).
};

static constexpr int SW_BM = 128, SW_BN = 128, SW_BKB = 64, SW_ROW = 80;  // LDS row = 64 B + 16 B pad
static constexpr int SW_TILE_BYTES = SW_BM * SW_ROW;

// XCD-aware tile order: consecutive workgroup ids round-robin over the 8 XCDs (block b -> XCD b%8);
// give each XCD a contiguous run of tiles so that neighbouring tiles (sharing an operand panel of the
// same candidate) hit the same L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg / 8, r = nwg % 8;
    const int xcd = bid % 8, idx = bid / 8;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

template <typename T, bool TWIN, int EPI>
__device__ __forceinline__ void k_sweep_body(const SweepParams& p, const uint3 blockIdx, const uint3 gridDim) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NPL = TWIN ? 3 : 2;                 // planes per stage: A, (A2), B
    constexpr int STAGE = NPL * SW_TILE_BYTES;
    constexpr bool IS_I8 = (sizeof(T) == 1);
    typedef typename std::conditional<IS_I8, v16i, v16f>::type acc_t;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 2, wc = wid & 3;            // 2 x 4 waves, each 64 rows x 32 cols
    const int g = lane >> 5, l31 = lane & 31;

    const int nwg = p.mtiles * p.ntiles;
    const int t = xcd_remap(blockIdx.x, nwg);
    const int mt = t % p.mtiles, nt = t / p.mtiles;   // m fastest: tiles sharing a B (weight) panel are adjacent
    const int z = blockIdx.y;
    const int m0 = mt * SW_BM, n0 = nt * SW_BN;
    // candidate groups over gridDim.z (host: choose_cgroups) -- a sweep with few tiles fills the chip this way
    const int per = (p.c1 - p.c0 + gridDim.z - 1) / gridDim.z;
    int c_lo_ = p.c0 + blockIdx.z * per, c_hi_ = min(p.c1, c_lo_ + per);
    clip_crange(p.crange, c_lo_, c_hi_);
    if (p.crange_blk) clip_crange_blk(p.crange_blk, z % p.cb_div, c_lo_, c_hi_);
    const int c_lo = c_lo_, c_hi = c_hi_;
    if (c_lo >= c_hi) return;

    // ---- candidate-invariant epilogue operands, kept in registers for the whole sweep -------------
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    float u[2][16], w[2][16];
    const int n = n0 + wc * 32 + l31;
    const bool ncol_ok = n < p.N;
    const float* biasz = p.bias ? p.bias + (long)z * p.bias_zs : nullptr;
    const float bias_n = (biasz && ncol_ok && p.bias_axis == 0) ? biasz[n] : 0.0f;
    if constexpr (EPI != EPI_FWD) {
        // Phase 1: every load of the raw_out / weight tile issued back to back at clamped (always valid) addresses.
        // (A load inside a per-element branch costs one dependent memory round trip per element.)
        const int nc = min(n, p.N - 1);
        const long ncol_off = (long)z * p.o_zs + (long)(nc / p.o_ninner) * p.o_nbs + (long)(nc % p.o_ninner) * p.o_ns;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mc = min(m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g, p.M - 1);
                const long idx = ncol_off + (long)(mc / p.o_inner) * p.o_bs + (long)(mc % p.o_inner) * p.o_ms;
                u[i][r] = p.O[idx];
                w[i][r] = p.Wt[idx];   // host passes Wt = O when the metric has no weight tensor
            }
        // Phase 2: pure ALU; the metric switch is hoisted out of the element loops
        const int wm = EPI == EPI_COS ? 4 : p.wt_mode;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                const bool ok = ncol_ok && m < p.M;
                const float o = u[i][r], gw = w[i][r];
                float ov = o - (EPI == EPI_COS ? 0.0f : bias_n), wv;
                if (wm == 1) wv = gw; else if (wm == 2) wv = o; else if (wm == 3) wv = fabsf(o); else wv = 1.0f;
                u[i][r] = ok ? ov : 0.0f;
                w[i][r] = ok ? wv : 0.0f;
            }
        if (EPI == EPI_COS) {
            // cosine keeps the raw output and carries the bias (0 on padding rows) in w
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                    const bool ok = ncol_ok && m < p.M;
                    const float braw = (biasz && p.bias_axis) ? biasz[min(m, p.M - 1)] : bias_n;
                    w[i][r] = (ok && biasz) ? braw : 0.0f;
                }
        }
    }
    const int sb = p.sb_mode == 1 ? min(n / p.sb_div, p.s_cs - 1) : p.sb_mode == 2 ? z % p.sb_div : 0;

    // ---- global -> LDS staging: one 16-byte piece per thread per plane per k-tile ---------------------
    const int ld_row = tid >> 2, ld_col = (tid & 3) * 16;
    const char* gA = (const char*)p.A + (long)z * p.a_zs + (long)(m0 + ld_row) * p.ldk + ld_col;
    const char* gA2 = TWIN ? (const char*)p.A2 + (long)z * p.a2_zs + (long)(m0 + ld_row) * p.ldk + ld_col : nullptr;
    const char* gB = (const char*)p.B + (long)z * p.b_zs + (long)(n0 + ld_row) * p.ldk + ld_col;
    const int lds_st = ld_row * SW_ROW + ld_col;

    const int ncand = c_hi - c_lo;
    const int total = ncand * p.ktiles;
    v4i ra, ra2, rb;
    auto gload = [&](int it) {
        const int c = c_lo + it / p.ktiles, kt = it % p.ktiles;
        ra = *reinterpret_cast<const v4i*>(gA + (long)c * p.a_cs + kt * SW_BKB);
        if (TWIN) ra2 = *reinterpret_cast<const v4i*>(gA2 + (long)c * p.a2_cs + kt * SW_BKB);
        rb = *reinterpret_cast<const v4i*>(gB + (long)c * p.b_cs + kt * SW_BKB);
    };
    auto lstore = [&](int stage) {
        char* s = smem + stage * STAGE + lds_st;
        *reinterpret_cast<v4i*>(s) = ra;
        if (TWIN) *reinterpret_cast<v4i*>(s + SW_TILE_BYTES) = ra2;
        *reinterpret_cast<v4i*>(s + (NPL - 1) * SW_TILE_BYTES) = rb;
    };

    acc_t acc[2], acc2[2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][r] = 0; if (TWIN) acc2[i][r] = 0; }
    };
    zero_acc();

    // fragment base offsets inside a stage
    const int fa = (wr * 64 + l31) * SW_ROW;       // + i*32*SW_ROW
    const int fb = (NPL - 1) * SW_TILE_BYTES + (wc * 32 + l31) * SW_ROW;

    gload(0);
    lstore(0);
    __syncthreads();

    for (int it = 0; it < total; ++it) {
        const int stage = it & 1;
        if (it + 1 < total) gload(it + 1);
        const char* s = smem + stage * STAGE;
        if constexpr (IS_I8) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {          // two 32-deep K steps per 64-byte row
                const int off = h * 32 + g * 16;
                const v4i b = *reinterpret_cast<const v4i*>(s + fb + off);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const v4i a = *reinterpret_cast<const v4i*>(s + fa + i * 32 * SW_ROW + off);
                    acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
                    if (TWIN) {
                        const v4i a2 = *reinterpret_cast<const v4i*>(s + SW_TILE_BYTES + fa + i * 32 * SW_ROW + off);
                        acc2[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2, b, acc2[i], 0, 0, 0);
                    }
                }
            }
        } else {
            // 16 floats of K per row; lane group g owns floats [8g, 8g+8): 8 MFMA 32x32x2 steps.
            const int off = g * 32;
            const v4f b0 = *reinterpret_cast<const v4f*>(s + fb + off);
            const v4f b1 = *reinterpret_cast<const v4f*>(s + fb + off + 16);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const v4f a0 = *reinterpret_cast<const v4f*>(s + fa + i * 32 * SW_ROW + off);
                const v4f a1 = *reinterpret_cast<const v4f*>(s + fa + i * 32 * SW_ROW + off + 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[i], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc[i], 0, 0, 0);
                if (TWIN) {
                    const v4f c0 = *reinterpret_cast<const v4f*>(s + SW_TILE_BYTES + fa + i * 32 * SW_ROW + off);
                    const v4f c1 = *reinterpret_cast<const v4f*>(s + SW_TILE_BYTES + fa + i * 32 * SW_ROW + off + 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc2[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0[e], b0[e], acc2[i], 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc2[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1[e], b1[e], acc2[i], 0, 0, 0);
                }
            }
        }
        if (it + 1 < total) lstore(stage ^ 1);
        __syncthreads();

        if ((it + 1) % p.ktiles == 0) {
            // ---- fused similarity epilogue for candidate c ------------------------------------------
            const int c = c_lo + it / p.ktiles;
            const float s1 = p.S1 ? p.S1[c * p.s_cs + sb] : 1.0f;
            const float s2 = (TWIN && p.S2) ? p.S2[c * p.s_cs + sb] : 1.0f;
            if constexpr (EPI == EPI_STORE || EPI == EPI_FWD) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                        float o_sim = (float)acc[i][r] * s1;
                        if (TWIN) o_sim = fmaf((float)acc2[i][r], s2, o_sim);
                        const float v = (EPI == EPI_FWD) ? o_sim + bias_n : u[i][r] - o_sim;
                        if (ncol_ok && m < p.M) p.store[(long)z * p.M * p.N + (long)m * p.N + n] = v;
                    }
            } else if constexpr (EPI != EPI_COS) {
                float colsum = 0.0f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float o_sim = (float)acc[i][r] * s1;
                        if (TWIN) o_sim = fmaf((float)acc2[i][r], s2, o_sim);
                        const float d = u[i][r] - o_sim;
                        if (EPI == EPI_SQ_W) { const float tt = w[i][r] * d; colsum = fmaf(tt, tt, colsum); }
                        else if (EPI == EPI_SQ) colsum = fmaf(d, d, colsum);
                        else if (EPI == EPI_ABS) colsum += fabsf(d);
                        else colsum = fmaf(w[i][r] * d, d, colsum);
                    }
                colsum += __shfl_xor(colsum, 32);
                if (g == 0)
                    p.part[(long)c * p.p_cs + (long)z * p.p_zs + (long)(mt * 2 + wr) * p.Np + n0 + wc * 32 + l31] = colsum;
            } else {
                // cosine: the MFMA rows are the feature axis the cosine reduces over, the columns are
                // samples.  Per sample: partial dot(o, o_sim), |o_sim|^2, |o|^2 over this wave's 64 features.
                float dot = 0.0f, nn = 0.0f, oo = 0.0f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float o_sim = fmaf((float)acc[i][r], s1, w[i][r]);
                        if (TWIN) o_sim = fmaf((float)acc2[i][r], s2, o_sim);
                        dot = fmaf(u[i][r], o_sim, dot);
                        nn = fmaf(o_sim, o_sim, nn);
                        oo = fmaf(u[i][r], u[i][r], oo);
                    }
                dot += __shfl_xor(dot, 32);
                nn += __shfl_xor(nn, 32);
                oo += __shfl_xor(oo, 32);
                if (g == 0) {
                    float* q = p.part + (long)c * p.p_cs + (long)z * p.p_zs + ((long)(mt * 2 + wr) * p.Np + n0 + wc * 32 + l31) * 3;
                    q[0] = dot; q[1] = nn; q[2] = oo;
                }
            }
            zero_acc();
        }
    }
}
template <typename T, bool TWIN, int EPI>
__global__ __launch_bounds__(512, 2) void k_sweep(SweepParams p) { k_sweep_body<T, TWIN, EPI>(p, P4V_BIDX, P4V_GDIM); }
template <typename T, bool TWIN, int EPI>
__global__ __launch_bounds__(512, 2) void k_sweep_g(GroupArgs<SweepParams> a) { P4V_GROUP_ENTER(a); k_sweep_body<T, TWIN, EPI>(a.p[m_], vb_, vg_); }

// ------------------------------------------------------------------------------------------
// k_sweep2: the fast int8 candidate sweep (linear layers and attention matmuls, element-wise metrics)
// ------------------------------------------------------------------------------------------
// Same contract as k_sweep<int8_t,...> but built for the MI355X memory system:
//   * operand k-tiles (128 rows x 64 B) stream HBM/L2 -> LDS with direct-to-LDS loads
//     (global_load_lds_dwordx4: no VGPR staging), SW2_NS tiles deep, waited with counted vmcnt, so
//     the L2/HBM latency of tile t+3 hides under the MFMAs of tile t; the loop runs flat over
//     (candidate, k-tile) so the pipeline never drains between candidates;
//   * the LDS image is lane-linear (a requirement of LDS-DMA), so the 16-byte chunks of a row are
//     XOR-swizzled on the SOURCE address and on the ds_read_b128 address (conflict-free reads);
//   * every wave's 32 columns lie inside one scale / score block (checked by the host), so scales are
//     scalar loads (lgkmcnt, never vmcnt) and the per-candidate result is ONE float per wave, kept in LDS
//     and written once at the end: no vector-memory traffic besides the operand stream inside the loop.
static constexpr int SW2_NS = 4;
static constexpr int SW2_TILE = 128 * 64;   // bytes of one operand k-tile

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// Same with an immediate offset OFF (0 <= OFF < 4096) applied by the instruction to BOTH addresses: loads g + OFF into
// l + OFF -- saves the 64-bit VALU add of a compile-time k-tile offset.
template <int OFF> __device__ __forceinline__ void glds16_imm(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, OFF, 0);
}

// Wave-wide fp32 sum on the VALU (DPP), result valid in lane 63.  __shfl_xor lowers to ds_bpermute_b32: six LDS
// round trips with a full lgkmcnt(0) each -- about 0.4 us per reduction when nothing else runs on the SIMD.
// Fixed combination order: deterministic.
__device__ __forceinline__ float wave_sum_dpp(float v) {
    auto dpp = [](float x, auto ctrl, auto rmask) __attribute__((always_inline)) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value,
                                                                      decltype(rmask)::value, 0xf, false));
    };
    using std::integral_constant;
    v += dpp(v, integral_constant<int, 0xB1>{}, integral_constant<int, 0xf>{});    // quad_perm [1,0,3,2]
    v += dpp(v, integral_constant<int, 0x4E>{}, integral_constant<int, 0xf>{});    // quad_perm [2,3,0,1]
    v += dpp(v, integral_constant<int, 0x141>{}, integral_constant<int, 0xf>{});   // row_half_mirror
    v += dpp(v, integral_constant<int, 0x140>{}, integral_constant<int, 0xf>{});   // row_mirror: every lane = its row's sum
    v += dpp(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});   // row_bcast15 -> rows 1, 3
    v += dpp(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});   // row_bcast31 -> rows 2, 3
    return v;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool TWIN, int EPI>
__device__ __forceinline__ void k_sweep2_body(const SweepParams& p, const uint3 blockIdx, const uint3 gridDim) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NPL = TWIN ? 3 : 2;
    constexpr int STAGE = NPL * SW2_TILE;
    // tail of the LDS image: per-(candidate, wave) results and the scale tables (ONE __shared__ object: a
    // second one makes hipcc drain vmcnt(0) before every ds_read of an LDS-DMA pipeline)
    float* res = reinterpret_cast<float*>(smem + SW2_NS * STAGE);   // [per][8 waves], behind the NPL planes of SW2_NS stages

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31;

    // Tile order.  The ~32 workgroups that run concurrently on one XCD (consecutive t after xcd_remap) should touch as
    // few DISTINCT operand tiles as possible: every k-tile step they pull (distinct A tiles) x 8 KB (x 2 planes for the
    // twin) + (distinct B tiles) x 8 KB through their L2, and what misses comes from the Infinity Cache / HBM.  Grouped
    // ordering (GM rows of tiles, all columns, then the next GM rows): 32 workgroups cover ~GM x (32 / GM) tiles
    // instead of 32 x 1.  (fc2 weight search, twin: 32 x 2 + 1 = 65 tiles per step with m-fastest order vs 5 x 2 + 6 = 16
    // grouped: the A planes were re-streamed from the Infinity Cache once per workgroup and candidate.)
    const int nwg = p.mtiles * p.ntiles;
    const int t = xcd_remap(blockIdx.x, nwg);
    constexpr int GM = 4;
    const int per_group = GM * p.ntiles;
    const int first_m = (t / per_group) * GM;
    const int gsz = min(p.mtiles - first_m, GM);
    const int mt = first_m + (t % per_group) % gsz, nt = (t % per_group) / gsz;
    const int z = blockIdx.y;
    const int m0 = mt * SW_BM, n0 = nt * SW_BN;
    const int per = (p.c1 - p.c0 + gridDim.z - 1) / gridDim.z;
    int c_lo_ = p.c0 + blockIdx.z * per, c_hi_ = min(p.c1, c_lo_ + per);
    clip_crange(p.crange, c_lo_, c_hi_);
    if (p.crange_blk) clip_crange_blk(p.crange_blk, z % p.cb_div, c_lo_, c_hi_);
    const int c_lo = c_lo_, c_hi = c_hi_;
    if (c_lo >= c_hi) return;

    // Which 64 x 32 part of the tile this wave computes.  Parts that hold only padding (attn.v: N = 64 of a 128-column tile;
    // 197 tokens: the last 32-row / 32-column blocks) do no MFMA and no epilogue work -- their waves still move their share
    // of the operand stream and keep the barriers.  Waves w and w + 4 share a SIMD, so the parts WITH work are dealt to
    // wave ids 0, 1, 2, ... first: they spread over the four SIMDs instead of leaving whole SIMDs to the padding.
    int pos = wid;
    // (the one-candidate store passes: not worth their registers; cosine: every part writes its three sums per sample)
    constexpr bool SKIP = !(EPI == EPI_FWD || EPI == EPI_STORE || EPI == EPI_COS);
    if constexpr (SKIP) {
        auto useful = [&](int q) { return (n0 + (q & 3) * 32 < p.N) && (m0 + (q >> 2) * 64 < p.M); };
        int cnt = 0, found = -1;
        for (int q = 0; q < 8; ++q)
            if (useful(q)) { if (cnt == wid) found = q; ++cnt; }
        if (found < 0) {
            int k = wid - cnt;
            for (int q = 0; q < 8; ++q)
                if (!useful(q)) { if (k == 0) found = q; --k; }
        }
        pos = __builtin_amdgcn_readfirstlane(found);
    }
    const int wr = pos >> 2, wc = pos & 3;
    const bool act = !SKIP || ((n0 + wc * 32 < p.N) && (m0 + wr * 64 < p.M));
    const bool act1 = !SKIP || (act && (m0 + wr * 64 + 32 < p.M));   // second 32-row block of the part

    // ---- candidate-invariant epilogue operands (identical to k_sweep) -----------------------------
    float u[2][16], w[2][16];
    const int n = n0 + wc * 32 + l31;
    const bool ncol_ok = n < p.N;
    const float* biasz = p.bias ? p.bias + (long)z * p.bias_zs : nullptr;
    const float bias_n = (biasz && ncol_ok && p.bias_axis == 0) ? biasz[n] : 0.0f;
    constexpr bool STORES = (EPI == EPI_FWD || EPI == EPI_STORE);   // no metric: the tile itself is written out
    if constexpr (EPI != EPI_FWD) {
        // Phase 1: every load of the raw_out / weight tile issued back to back at clamped (always valid) addresses.
        // (A load inside a per-element branch costs one dependent memory round trip per element.)
        const int nc = min(n, p.N - 1);
        const long ncol_off = (long)z * p.o_zs + (long)(nc / p.o_ninner) * p.o_nbs + (long)(nc % p.o_ninner) * p.o_ns;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mc = min(m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g, p.M - 1);
                const long idx = ncol_off + (long)(mc / p.o_inner) * p.o_bs + (long)(mc % p.o_inner) * p.o_ms;
                u[i][r] = p.O[idx];
                w[i][r] = p.Wt[idx];   // host passes Wt = O when the metric has no weight tensor
            }
        // Phase 2: pure ALU; the metric switch is hoisted out of the element loops
        const int wm = EPI == EPI_COS ? 4 : p.wt_mode;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                const bool ok = ncol_ok && m < p.M;
                const float o = u[i][r], gw = w[i][r];
                float ov = o - (EPI == EPI_COS ? 0.0f : bias_n), wv;
                if (wm == 1) wv = gw; else if (wm == 2) wv = o; else if (wm == 3) wv = fabsf(o); else wv = 1.0f;
                u[i][r] = ok ? ov : 0.0f;
                w[i][r] = ok ? wv : 0.0f;
            }
        if (EPI == EPI_COS) {
            // cosine (as in k_sweep): u keeps the raw output, w carries the bias of the simulated output (0 on padding)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                    const bool ok = ncol_ok && m < p.M;
                    const float braw = (biasz && p.bias_axis) ? biasz[min(m, p.M - 1)] : bias_n;
                    w[i][r] = (ok && biasz) ? braw : 0.0f;
                }
        }
    }
    // cosine: |o|^2 of this lane's 32 outputs does not depend on the candidate (same order of additions as the per-candidate sum)
    float oo_fix = 0.0f;
    if constexpr (EPI == EPI_COS) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oo_fix = fmaf(u[i][r], u[i][r], oo_fix);
        oo_fix += __shfl_xor(oo_fix, 32);
    }
    // wave-uniform scale block of this wave's 32 columns (host guarantees they share one block)
    const int nw0 = n0 + wc * 32;
    const int sb = __builtin_amdgcn_readfirstlane(p.sb_mode == 1 ? min(nw0 / p.sb_div, p.s_cs - 1) : p.sb_mode == 2 ? z % p.sb_div : 0);
    // scales of this wave's block for every candidate of this workgroup -> LDS (ordinary loads must not appear
    // inside the LDS-DMA loop: hipcc would wait vmcnt(0) for them and drain the prefetch ring)
    float* s1tab = res + per * 8;
    float* s2tab = s1tab + per * 8;
    for (int i = lane; i < c_hi - c_lo; i += 64) {
        s1tab[i * 8 + pos] = p.S1 ? p.S1[(c_lo + i) * p.s_cs + sb] : 1.0f;
        if (TWIN) s2tab[i * 8 + pos] = p.S2 ? p.S2[(c_lo + i) * p.s_cs + sb] : 1.0f;
    }

    // ---- LDS-DMA addressing: wave `wid` fills rows [16*wid, 16*wid+16) of every plane ------------------
    // global address = wave-uniform 64-bit cursor (SGPRs, advanced with scalar adds) + per-lane 32-bit offset
    const int ld_row = wid * 16 + (lane >> 2);
    const int ld_chunk = (lane & 3) ^ ((ld_row >> 2) & 3);          // logical 16-B chunk landing in physical slot lane&3
    const unsigned voffA = (unsigned)(ld_row * p.ldk + ld_chunk * 16);
    // (a B plane of 64 rows -- attn.v: N = 64 of the 128-column tile, 100 candidate planes of V per module -- is not padded to
    // the tile: the DMA of the tile's upper rows re-reads rows 0..63 into the LDS, their columns are masked in the epilogue)
    const unsigned voffB = p.b_rows > 0 ? (unsigned)((ld_row & (p.b_rows - 1)) * p.ldk + ld_chunk * 16) : voffA;
    const char* curA = (const char*)p.A + (long)z * p.a_zs + (long)m0 * p.ldk + (long)c_lo * p.a_cs;
    const char* curA2 = TWIN ? (const char*)p.A2 + (long)z * p.a2_zs + (long)m0 * p.ldk + (long)c_lo * p.a2_cs : nullptr;
    const char* curB = (const char*)p.B + (long)z * p.b_zs + (long)n0 * p.ldk + (long)c_lo * p.b_cs;
    const int ktiles = p.ktiles;
    const long wrapA = p.a_cs - (long)ktiles * SW_BKB, wrapA2 = TWIN ? p.a2_cs - (long)ktiles * SW_BKB : 0,
               wrapB = p.b_cs - (long)ktiles * SW_BKB;   // cursor jump at the end of a candidate
    const int lds_wave = wid * 1024;
    const int total = (c_hi - c_lo) * ktiles;
    int ikt = 0;   // k-tile of the next tile to issue
    // LDS image: [plane][stage][8 KB] -- every fragment read is (per-lane base VGPR) + (immediate < 64 KB)
    constexpr int PLANE = SW2_NS * SW2_TILE;
    auto issue = [&](int stage) __attribute__((always_inline)) {
        char* s = smem + stage * SW2_TILE + lds_wave;
        glds16(curA + voffA, s);
        if (TWIN) glds16(curA2 + voffA, s + PLANE);
        glds16(curB + voffB, s + (NPL - 1) * PLANE);
        curA += SW_BKB; curB += SW_BKB;
        if (TWIN) curA2 += SW_BKB;
        if (++ikt == ktiles) { ikt = 0; curA += wrapA; curB += wrapB; if (TWIN) curA2 += wrapA2; }
    };

    v16i acc[2], acc2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0; if (TWIN) acc2[i][r] = 0; }

    // swizzled fragment addresses (per lane, stage-independent): row R, logical chunk c -> physical chunk c ^ ((R>>2)&3)
    const int ra0 = wr * 64 + l31, ra1 = ra0 + 32, rb = wc * 32 + l31;
    const int sa0 = (ra0 >> 2) & 3, sa1 = (ra1 >> 2) & 3, sbz = (rb >> 2) & 3;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const unsigned aA00 = lds0 + ra0 * 64 + ((g ^ sa0) << 4), aA01 = lds0 + ra0 * 64 + (((2 + g) ^ sa0) << 4);
    const unsigned aA10 = lds0 + ra1 * 64 + ((g ^ sa1) << 4), aA11 = lds0 + ra1 * 64 + (((2 + g) ^ sa1) << 4);
    const unsigned aB0 = lds0 + (NPL - 1) * PLANE + rb * 64 + ((g ^ sbz) << 4);
    const unsigned aB1 = lds0 + (NPL - 1) * PLANE + rb * 64 + (((2 + g) ^ sbz) << 4);

    const int npre = min(SW2_NS - 1, total);
    for (int i = 0; i < npre; ++i) issue(i);

    // Fragment reads are software pipelined one k-tile ahead with inline-asm ds_read_b128 and counted lgkmcnt waits
    // (the compiler's wait insertion degrades to lgkmcnt(0) while an LDS-DMA is pending and, left alone, re-serialises
    // read -> wait -> MFMA: a full LDS round trip exposed per k-tile with both waves of a SIMD in lock step).
#define P4V_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    struct Fr { v4i b0, b1, a00, a10, a01, a11, c00, c10, c01, c11; };
    constexpr int NRD = TWIN ? 10 : 6;                       // ds_reads per k-tile
    auto read_fr = [&](Fr& f, auto stage_c) __attribute__((always_inline)) {
        constexpr int SO = decltype(stage_c)::value * SW2_TILE;
        P4V_DSR(f.b0, aB0, SO); P4V_DSR(f.a00, aA00, SO); P4V_DSR(f.a10, aA10, SO);
        if (TWIN) { P4V_DSR(f.c00, aA00, SO + PLANE); P4V_DSR(f.c10, aA10, SO + PLANE); }
        P4V_DSR(f.b1, aB1, SO); P4V_DSR(f.a01, aA01, SO); P4V_DSR(f.a11, aA11, SO);
        if (TWIN) { P4V_DSR(f.c01, aA01, SO + PLANE); P4V_DSR(f.c11, aA11, SO + PLANE); }
    };
    Fr fa, fb;
    int kt = 0, c = c_lo;
    // step `it` (compile-time stage of tile it): tile it is in `cur` (read during step it-1).  Prove tile it+1 landed
    // (own pieces waited for, then the barrier), refill the stage of tile it-1 with tile it+3, start the reads of
    // tile it+1 into `nxt`, run the MFMAs of tile it.
    auto tile = [&](int it, auto stage_c, Fr& cur, Fr& nxt) __attribute__((always_inline)) {
        constexpr int ST = decltype(stage_c)::value;
        if (it + 2 < total) wait_vmcnt<NPL>(); else wait_vmcnt<0>();   // pieces of tile it+2 may stay in flight
        __builtin_amdgcn_s_barrier();
        if (it + SW2_NS - 1 < total) issue((ST + SW2_NS - 1) % SW2_NS);
        if (!act) {                                    // a part of pure padding: stream and barriers only
            if (++kt == ktiles) {
                if (!STORES && lane == 63) res[(c - c_lo) * 8 + pos] = 0.0f;
                kt = 0;
                ++c;
            }
            return;
        }
        if (it + 1 < total) {
            read_fr(nxt, std::integral_constant<int, (ST + 1) % SW2_NS>{});
            __builtin_amdgcn_s_waitcnt(0xC07F | (NRD << 8));          // the reads just issued stay in flight
        } else {
            __builtin_amdgcn_s_waitcnt(0xC07F);
        }
        asm volatile("" : "+v"(cur.b0), "+v"(cur.b1), "+v"(cur.a00), "+v"(cur.a10), "+v"(cur.a01), "+v"(cur.a11) :: "memory");
        if (TWIN) asm volatile("" : "+v"(cur.c00), "+v"(cur.c10), "+v"(cur.c01), "+v"(cur.c11));
        acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.a00, cur.b0, acc[0], 0, 0, 0);
        if (act1) acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.a10, cur.b0, acc[1], 0, 0, 0);
        if (TWIN) {
            acc2[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.c00, cur.b0, acc2[0], 0, 0, 0);
            if (act1) acc2[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.c10, cur.b0, acc2[1], 0, 0, 0);
        }
        acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.a01, cur.b1, acc[0], 0, 0, 0);
        if (act1) acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.a11, cur.b1, acc[1], 0, 0, 0);
        if (TWIN) {
            acc2[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.c01, cur.b1, acc2[0], 0, 0, 0);
            if (act1) acc2[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.c11, cur.b1, acc2[1], 0, 0, 0);
        }
        if (++kt == ktiles) {
            const float s1 = s1tab[(c - c_lo) * 8 + pos];
            const float s2 = TWIN ? s2tab[(c - c_lo) * 8 + pos] : 1.0f;
            if constexpr (STORES) {
                // quant_forward (EPI_FWD: scale * acc + bias) / folded twin target (EPI_STORE: raw_out - bias - scale * acc):
                // same arithmetic as the generic k_sweep, 128-byte coalesced rows
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                        float o_sim = (float)acc[i][r] * s1;
                        if (TWIN) o_sim = fmaf((float)acc2[i][r], s2, o_sim);
                        const float v = (EPI == EPI_FWD) ? o_sim + bias_n : u[i][r] - o_sim;
                        if (ncol_ok && m < p.M) p.store[(long)z * p.M * p.N + (long)m * p.N + n] = v;
                        acc[i][r] = 0;
                        if (TWIN) acc2[i][r] = 0;
                    }
                kt = 0;
                ++c;
                return;
            }
            if constexpr (EPI == EPI_COS) {
                // cosine: the MFMA rows are the feature axis the cosine reduces over, the columns are samples.  Per sample the
                // partial dot(o, o_sim), |o_sim|^2, |o|^2 over this wave's 64 features, in k_sweep's order and table layout
                // (k_finish_cos unchanged).  The three stores per candidate ride in the same vmcnt queue as the operand stream:
                // the counted waits of the ring then wait for MORE than they need, never for less.
                float dot = 0.0f, nn = 0.0f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float o_sim = fmaf((float)acc[i][r], s1, w[i][r]);
                        if (TWIN) o_sim = fmaf((float)acc2[i][r], s2, o_sim);
                        dot = fmaf(u[i][r], o_sim, dot);
                        nn = fmaf(o_sim, o_sim, nn);
                        acc[i][r] = 0;
                        if (TWIN) acc2[i][r] = 0;
                    }
                dot += __shfl_xor(dot, 32);
                nn += __shfl_xor(nn, 32);
                const float oo = oo_fix;
                if (g == 0) {
                    float* q = p.part + (long)c * p.p_cs + (long)z * p.p_zs + ((long)(mt * 2 + wr) * p.Np + n0 + wc * 32 + l31) * 3;
                    q[0] = dot; q[1] = nn; q[2] = oo;
                }
                kt = 0;
                ++c;
                return;
            }
            // ---- fused similarity epilogue of candidate c: one float per wave ---------------------------
            v2f sum2 = {0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i == 1 && !act1) break;              // (its accumulators were never touched: still zero)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const v2f a = {(float)acc[i][r], (float)acc[i][r + 1]};
                    const v2f uu = {u[i][r], u[i][r + 1]};
                    const v2f ww = {w[i][r], w[i][r + 1]};
                    v2f d = uu - a * s1;
                    if (TWIN) { const v2f a2 = {(float)acc2[i][r], (float)acc2[i][r + 1]}; d = d - a2 * s2; }
                    if (EPI == EPI_SQ_W) { const v2f t2 = ww * d; sum2 = t2 * t2 + sum2; }
                    else if (EPI == EPI_SQ) sum2 = d * d + sum2;
                    else if (EPI == EPI_ABS) sum2 += v2f{fabsf(d.x), fabsf(d.y)};
                    else sum2 = (ww * d) * d + sum2;
                    acc[i][r] = 0; acc[i][r + 1] = 0;
                    if (TWIN) { acc2[i][r] = 0; acc2[i][r + 1] = 0; }
                }
            }
            const float sum = wave_sum_dpp(sum2.x + sum2.y);           // fixed order: deterministic
            if (lane == 63) res[(c - c_lo) * 8 + pos] = sum;
            kt = 0;
            ++c;
        }
    };
    // tile 0
    if (total > 2) wait_vmcnt<NPL * 2>(); else if (total > 1) wait_vmcnt<NPL>(); else wait_vmcnt<0>();   // tiles 1, 2 may be in flight
    __builtin_amdgcn_s_barrier();
    read_fr(fa, std::integral_constant<int, 0>{});
    for (int it = 0; it < total; it += SW2_NS) {
        tile(it, std::integral_constant<int, 0>{}, fa, fb);
        if (it + 1 < total) tile(it + 1, std::integral_constant<int, 1>{}, fb, fa);
        if (it + 2 < total) tile(it + 2, std::integral_constant<int, 2>{}, fa, fb);
        if (it + 3 < total) tile(it + 3, std::integral_constant<int, 3>{}, fb, fa);
    }
#undef P4V_DSR
    __syncthreads();
    if constexpr (STORES || EPI == EPI_COS) return;
    // ---- one coalesced write of this workgroup's results: part[c][z][mt*2+wr][nt*4+wc] -----------------
    for (int i = tid; i < (c_hi - c_lo) * 8; i += 512) {
        const int cc = c_lo + i / 8, wv = i % 8;
        p.part[(long)cc * p.p_cs + (long)z * p.p_zs + (long)(mt * 2 + (wv >> 2)) * p.Np + nt * 4 + (wv & 3)] = res[i];
    }
}
template <bool TWIN, int EPI>
__global__ __launch_bounds__(512, 2) void k_sweep2(SweepParams p) { k_sweep2_body<TWIN, EPI>(p, P4V_BIDX, P4V_GDIM); }
template <bool TWIN, int EPI>
__global__ __launch_bounds__(512, 2) void k_sweep2_g(GroupArgs<SweepParams> a) { P4V_GROUP_ENTER(a); k_sweep2_body<TWIN, EPI>(a.p[m_], vb_, vg_); }

// ------------------------------------------------------------------------------------------
// k_bound: ONE candidate over ALL samples -- stage B1 of a pruned pass (p4v_api.hip::run_pass_pruned), the bound L*
// ------------------------------------------------------------------------------------------
// A pruned Linear pass evaluates one candidate (per score block) on every sample to get the bound that decides which candidates
// of the slice survive.  That is a plain int8 GEMM M x N x K with the metric epilogue and NO candidate loop to amortise
// anything over: on the sweep kernels it costs a full workgroup prologue per tile (k_sweep6: one workgroup per CU with 512
// registers per wave, 20 us of latency-bound prologue for 2.4 us of MFMAs; 83 us for ViT-B fc1 / qkv, the CUs blocked for every
// other stream meanwhile) where the bytes it must read -- raw_out and the metric weight, 8 B per output: 155 MB for fc1 -- take
// 30 us.  This kernel is built for that regime instead:
//   * no LDS, no barrier, no ring: every wave owns a 64 x 32 tile of outputs and loads its MFMA fragments straight from L2.
//     Both operand planes are packed for this pass in MFMA-fragment order (1 KB contiguous per fragment: 8 cache lines per
//     load instruction; from row-major planes an instruction touches 32 rows -- measured 104 us for ViT-B fc1, the dead end of
//     DESIGN 5.1) and the fragments of k-tile t + 2 are requested before the MFMAs of k-tile t (inline-asm loads, counted waits);
//   * 4 waves per workgroup (2 x 2 tiles: neighbours share rows in the L1), 4 workgroups per CU next to anything else:
//     the loads of one wave hide under the MFMAs and the epilogue of the others;
//   * raw_out / metric weight are read in place at the end (128 contiguous bytes per half wave and row), one float per
//     (64-row slab, 32-column group) goes to the partial-sum table k_sweep2 uses, k_finish reduces it as for any fast sweep.
// Its totals are summed in another order than the sweeps' -- so run_pass_pruned never lets a selection depend on them: they
// only set the bound (margin: prune_margin), and whenever more than the bound's own candidate survives, stage B2 re-evaluates
// ALL survivors with the unpruned kernels.
template <int EPI>
__device__ __forceinline__ void k_bound_body(const SweepParams& p, const uint3 blockIdx, const uint3 gridDim) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31;
    // workgroup tile 128 rows x 64 columns (2 x 2 waves of 64 x 32): p.ntiles counts 128-column tiles
    const int ntiles = p.ntiles * 2;
    const int nwg = p.mtiles * ntiles;
    const int t = xcd_remap(blockIdx.x, nwg);
    constexpr int GM = 4;                                  // tile order as k_sweep2: GM row tiles x all column tiles per group
    const int per_group = GM * ntiles;
    const int first_m = (t / per_group) * GM;
    const int gsz = min(p.mtiles - first_m, GM);
    const int mt = first_m + (t % per_group) % gsz, nt = (t % per_group) / gsz;
    int c = p.c0;
    if (p.crange) {
        const int a = __builtin_amdgcn_readfirstlane(p.crange[0]), b = __builtin_amdgcn_readfirstlane(p.crange[1]);
        if (a >= b) return;                                // empty range: k_finish writes -inf without reading the table
        c = max(c, a);
        if (c >= b) return;                                // (a chunked plane: the candidate sits in an earlier chunk -- nothing of this
                                                           // launch's chunk is in range, and its fragment image holds another candidate)
    }
    if (c >= p.c1) return;                                 // (a chunked plane: the candidate lives in another chunk's launch)
    const int wr = wid >> 1, wc = wid & 1;
    const int m0 = mt * 128 + wr * 64, n0 = nt * 64 + wc * 32;
    float* slot = p.part + (long)c * p.p_cs + (long)(mt * 2 + wr) * p.Np + nt * 2 + wc;
    if (m0 >= p.M || n0 >= p.N) {                          // a tile of pure padding
        if (lane == 63) slot[0] = 0.0f;
        return;
    }
    // Both planes are packed FOR this pass in MFMA-fragment order (k_pack layout c_inner = 3, the one k_sweep6's stationary
    // operand uses): [64-row slab][k-tile][32-row block][MFMA of the k-tile][lane] x 16 B -- every fragment is 1 KB contiguous,
    // a wave's load touches 8 cache lines instead of 32 rows.  Wave-uniform 64-bit bases (SGPR pairs, advanced per k-tile) +
    // the lane's 32-bit offset: no vector address arithmetic.
    const char* sA = (const char*)p.A + (long)(m0 >> 6) * p.ktiles * 4096;
    const char* sB = (const char*)p.B + (long)(n0 >> 6) * p.ktiles * 4096 + ((n0 >> 5) & 1) * 2048;
    const unsigned vo = (unsigned)lane * 16u;
    v16i acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    struct Fr { v4i a[2][2], b[2]; };                      // [32-row block][first / second 16 bytes of the lane's 32]
    // Inline-asm loads with counted waits: left to itself hipcc waits vmcnt(0) before the MFMAs of k-tile t, i.e. also for the
    // fragments of k-tile t + 1 it has just requested -- no overlap at all (seen in the ISA of the first version).
#define P4V_BLD(dst, voff, sbase, off) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(off) : "memory")
    auto load = [&](Fr& f) __attribute__((always_inline)) {            // 6 loads of 1 KB: the next k-tile of both operands
        P4V_BLD(f.a[0][0], vo, sA, 0); P4V_BLD(f.a[0][1], vo, sA, 1024);
        P4V_BLD(f.b[0], vo, sB, 0); P4V_BLD(f.b[1], vo, sB, 1024);
        P4V_BLD(f.a[1][0], vo, sA, 2048); P4V_BLD(f.a[1][1], vo, sA, 3072);
        sA += 4096; sB += 4096;
    };
    auto mma = [&](Fr& f) __attribute__((always_inline)) {
        asm volatile("" : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[1][0]), "+v"(f.a[1][1]), "+v"(f.b[0]), "+v"(f.b[1]) :: "memory");   // after the wait
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.a[i][h], f.b[h], acc[i], 0, 0, 0);
    };
    Fr f0, f1, f2;
    const int ktiles = p.ktiles;
    // three fragment sets, two k-tiles in flight (tile t + 2 is requested before the MFMAs of tile t), four waves per SIMD
    constexpr int LD = 6;
    load(f0);
    if (ktiles > 1) load(f1);
    auto step = [&](int kt, Fr& cur, Fr& refill) __attribute__((always_inline)) {
        const int ahead = min(ktiles - 1 - kt, 2);         // tiles after kt that are in flight once the refill is issued
        if (kt + 2 < ktiles) load(refill);
        if (ahead >= 2) wait_vmcnt<2 * LD>(); else if (ahead == 1) wait_vmcnt<LD>(); else wait_vmcnt<0>();
        mma(cur);
    };
    for (int kt = 0; kt < ((p.dbg & 1) ? 1 : ktiles); kt += 3) {     // (dbg 1: timing-only ablation, one k-tile)
        step(kt, f0, f2);
        if (kt + 1 < ktiles) step(kt + 1, f1, f0);
        if (kt + 2 < ktiles) step(kt + 2, f2, f1);
    }
#undef P4V_BLD
    // ---- metric epilogue: raw_out / weight read in place, one sum per (64-row slab, 32-column group) ----
    // The weight tensor is raw_grad (hessian) or raw_out itself (the host passes Wt = O for the raw_out-weighted metrics:
    // square-weighted (w d)^2 with w = raw_out, linear-weighted |w| d^2) -- no run-time switch in the element loop.
    const int n = n0 + l31;
    const bool ncol_ok = n < p.N;
    const int nc = min(n, p.N - 1);
    const float bias_n = (p.bias && ncol_ok) ? p.bias[nc] : 0.0f;
    const int sb = p.sb_mode == 1 ? min(nc / p.sb_div, p.s_cs - 1) : 0;
    const float s1 = p.S1 ? p.S1[(long)c * p.s_cs + sb] : 1.0f;
    const float* On = p.O + (long)nc * p.o_ns;
    const float* Wn = p.Wt + (long)nc * p.o_ns;
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {                          // one 32 x 32 block: 16 + 16 loads in flight per lane (the fragment registers are dead)
        float u[16], w[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {                     // every load at a clamped (valid) address, back to back
            const int mc = (p.dbg & 2) ? 0 : min(m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g, p.M - 1);   // (dbg 2: ablation, one row)
            u[r] = On[(long)mc * p.o_ms];
            if (EPI == EPI_SQ_W || EPI == EPI_W_SQ) w[r] = Wn[(long)mc * p.o_ms];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            const bool ok = ncol_ok && m < p.M;
            const float o = u[r];
            float wv = 1.0f;
            if (EPI == EPI_SQ_W) wv = w[r]; else if (EPI == EPI_W_SQ) wv = fabsf(w[r]);
            const float d = (o - bias_n) - (float)acc[i][r] * s1;
            float term;
            if (EPI == EPI_SQ_W) { const float t2 = wv * d; term = t2 * t2; }
            else if (EPI == EPI_SQ) term = d * d;
            else if (EPI == EPI_ABS) term = fabsf(d);
            else term = (wv * d) * d;
            sum += ok ? term : 0.0f;
        }
        asm volatile("" ::: "memory");
    }
    sum = wave_sum_dpp(sum);
    if (lane == 63) slot[0] = sum;
}
template <int EPI>
__global__ __launch_bounds__(256, 4) void k_bound(SweepParams p) { k_bound_body<EPI>(p, P4V_BIDX, P4V_GDIM); }
template <int EPI>
__global__ __launch_bounds__(256, 4) void k_bound_g(GroupArgs<SweepParams> a) { P4V_GROUP_ENTER(a); k_bound_body<EPI>(a.p[m_], vb_, vg_); }

// ------------------------------------------------------------------------------------------
// k_sweep8: k_sweep2 for K <= 64 (ONE k-tile: the q.k^T matmuls of every ViT / DeiT / Swin, head_dim <= 64)
// ------------------------------------------------------------------------------------------
// With a single k-tile a candidate is one ring step of k_sweep2: 16 KB of LDS-DMA, a barrier, 6 fragment reads, 4 MFMAs and a
// 32-element epilogue per lane -- and the step time is the DMA latency divided by the three tiles the ring keeps in flight
// (measured: 1.8 us per candidate and workgroup, two workgroups per CU; the matrix pipe idles).  Here the FIXED operand's
// fragments (8 or 16 VGPRs) are loaded once and stay in registers, only the candidate-expanded operand streams (8 KB per
// candidate: one piece per wave), and the ring is 8 candidates deep: half the bytes, 2.3x the steps per DMA latency.
// Same tile (128 x 128, 8 waves x (64 x 32)), same epilogue, same partial-sum table as k_sweep2.
static constexpr int SW8_NS = 8;

#ifndef P4V_SW8_DBG
#define P4V_SW8_DBG 0      // timing-only ablations: 1 no operand stream in the loop, 2 no MFMAs, 4 no epilogue arithmetic
#endif
template <bool ROWS_FIXED, int EPI, bool SKIP>
__device__ __forceinline__ void k_sweep8_body(const SweepParams& p, const uint3 blockIdx, const uint3 gridDim) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* res = reinterpret_cast<float*>(smem + SW8_NS * SW2_TILE);   // [per][8 waves]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31;

    const int nwg = p.mtiles * p.ntiles;
    const int t = xcd_remap(blockIdx.x, nwg);
    const int mt = t % p.mtiles, nt = t / p.mtiles;
    const int z = blockIdx.y;
    const int m0 = mt * SW_BM, n0 = nt * SW_BN;
    const int per = (p.c1 - p.c0 + gridDim.z - 1) / gridDim.z;
    int c_lo_ = p.c0 + blockIdx.z * per, c_hi_ = min(p.c1, c_lo_ + per);
    clip_crange(p.crange, c_lo_, c_hi_);
    if (p.crange_blk) clip_crange_blk(p.crange_blk, z % p.cb_div, c_lo_, c_hi_);
    const int c_lo = c_lo_, c_hi = c_hi_;
    if (c_lo >= c_hi) return;
    const int ncand = c_hi - c_lo;

    // the 64 x 32 part of this wave; the padding skip of k_sweep2 (parts of pure padding do no work, parts with work are dealt
    // to wave ids 0, 1, ... first):
    int pos = wid;
    // SKIP: A/B switch only (launch_sweep8_epi) -- slower at 197 and at 144 tokens: the candidate step of this kernel is
    // paced by the ring (DMA landing + barrier), not by the work the padding parts would skip
    if constexpr (SKIP) {
        auto useful = [&](int q) { return (n0 + (q & 3) * 32 < p.N) && (m0 + (q >> 2) * 64 < p.M); };
        int cnt = 0, found = -1;
        for (int q = 0; q < 8; ++q)
            if (useful(q)) { if (cnt == wid) found = q; ++cnt; }
        if (found < 0) {
            int k = wid - cnt;
            for (int q = 0; q < 8; ++q)
                if (!useful(q)) { if (k == 0) found = q; --k; }
        }
        pos = __builtin_amdgcn_readfirstlane(found);
    }
    const int wr = pos >> 2, wc = pos & 3;
    const bool act = !SKIP || ((n0 + wc * 32 < p.N) && (m0 + wr * 64 < p.M));
    const bool act1 = !SKIP || (act && (m0 + wr * 64 + 32 < p.M));   // second 32-row block of the part

    // ---- candidate-invariant epilogue operands (as k_sweep2) --------------------------------------------------------------
    float u[2][16], w[2][16];
    const int n = n0 + wc * 32 + l31;
    const bool ncol_ok = n < p.N;
    const float* biasz = p.bias ? p.bias + (long)z * p.bias_zs : nullptr;
    const float bias_n = (biasz && ncol_ok) ? biasz[n] : 0.0f;
    {
        const int nc = min(n, p.N - 1);
        const long ncol_off = (long)z * p.o_zs + (long)(nc / p.o_ninner) * p.o_nbs + (long)(nc % p.o_ninner) * p.o_ns;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mc = min(m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g, p.M - 1);
                const long idx = ncol_off + (long)(mc / p.o_inner) * p.o_bs + (long)(mc % p.o_inner) * p.o_ms;
                u[i][r] = p.O[idx];
                w[i][r] = p.Wt[idx];   // host passes Wt = O when the metric has no weight tensor
            }
        const int wm = p.wt_mode;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                const bool ok = ncol_ok && m < p.M;
                const float o = u[i][r], gw = w[i][r];
                float wv;
                if (wm == 1) wv = gw; else if (wm == 2) wv = o; else if (wm == 3) wv = fabsf(o); else wv = 1.0f;
                u[i][r] = ok ? o - bias_n : 0.0f;
                w[i][r] = ok ? wv : 0.0f;
            }
    }
    const int nw0 = n0 + wc * 32;
    const int sb = __builtin_amdgcn_readfirstlane(p.sb_mode == 1 ? min(nw0 / p.sb_div, p.s_cs - 1) : p.sb_mode == 2 ? z % p.sb_div : 0);
    float* s1tab = res + per * 8;
    for (int i = lane; i < ncand; i += 64) s1tab[i * 8 + pos] = p.S1 ? p.S1[(c_lo + i) * p.s_cs + sb] : 1.0f;

    // ---- the fixed operand's fragments: registers for the whole sweep (ldk = 64: one k-tile) ----------------------------------
    v4i fx[2][2];                                        // [32-row block (row side only)][k-half]
    if (ROWS_FIXED) {
        const char* gA = (const char*)p.A + (long)z * p.a_zs + (long)(m0 + wr * 64 + l31) * SW_BKB + g * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) fx[i][h] = *reinterpret_cast<const v4i*>(gA + i * 32 * SW_BKB + h * 32);
    } else {
        const char* gB = (const char*)p.B + (long)z * p.b_zs + (long)(n0 + wc * 32 + l31) * SW_BKB + g * 16;
#pragma unroll
        for (int h = 0; h < 2; ++h) { fx[0][h] = *reinterpret_cast<const v4i*>(gB + h * 32); fx[1][h] = fx[0][h]; }
    }

    // ---- the expanded operand streams: wave `wid` moves rows [16 wid, 16 wid + 16) of the 128-row tile, one candidate per stage ---
    const int ld_row = wid * 16 + (lane >> 2);
    const int ld_chunk = (lane & 3) ^ ((ld_row >> 2) & 3);
    const unsigned voff = (unsigned)(ld_row * SW_BKB + ld_chunk * 16);
    const long t_cs = ROWS_FIXED ? p.b_cs : p.a_cs;
    const char* cur = ROWS_FIXED ? (const char*)p.B + (long)z * p.b_zs + (long)n0 * SW_BKB + (long)c_lo * p.b_cs
                                 : (const char*)p.A + (long)z * p.a_zs + (long)m0 * SW_BKB + (long)c_lo * p.a_cs;
    const int lds_wave = wid * 1024;
    auto issue = [&](int stage) __attribute__((always_inline)) {
        glds16(cur + voff, smem + stage * SW2_TILE + lds_wave);
        cur += t_cs;
    };

    v16i acc[2];
    const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // a candidate is ONE k-tile: its first MFMAs start from zero

    // swizzled fragment addresses of the streamed tile: row side two 32-row blocks, column side one
    const int rs0 = ROWS_FIXED ? wc * 32 + l31 : wr * 64 + l31, rs1 = rs0 + 32;
    const int ss0 = (rs0 >> 2) & 3, ss1 = (rs1 >> 2) & 3;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const unsigned a00 = lds0 + rs0 * 64 + ((g ^ ss0) << 4), a01 = lds0 + rs0 * 64 + (((2 + g) ^ ss0) << 4);
    const unsigned a10 = lds0 + rs1 * 64 + ((g ^ ss1) << 4), a11 = lds0 + rs1 * 64 + (((2 + g) ^ ss1) << 4);
#define P4V_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    struct Fr { v4i t00, t01, t10, t11; };               // [block][k-half] of the streamed operand (column side: block 0 only)
    constexpr int NRD = ROWS_FIXED ? 2 : 4;
    auto read_fr = [&](Fr& f, auto stage_c) __attribute__((always_inline)) {
        constexpr int SO = decltype(stage_c)::value * SW2_TILE;
        const unsigned b00 = a00, b01 = a01, b10 = a10, b11 = a11;   // (locals: clang rejects captured names that appear only in asm operands)
        P4V_DSR(f.t00, b00, SO); P4V_DSR(f.t01, b01, SO);
        if constexpr (!ROWS_FIXED) { P4V_DSR(f.t10, b10, SO); P4V_DSR(f.t11, b11, SO); }
    };

    const int npre = min(SW8_NS - 1, ncand);
    for (int i = 0; i < npre; ++i) issue(i);
    Fr fa, fb;
    int c = c_lo;
    // step `it` (compile-time stage): `cur` holds candidate it (read during step it-1).  Prove candidate it+1 landed (own piece
    // waited for, then the barrier), refill the stage of candidate it-1 with candidate it+7, start the reads of it+1, run the
    // MFMAs and the epilogue of candidate it.
    auto step = [&](int it, auto stage_c, Fr& curf, Fr& nxt) __attribute__((always_inline)) {
        constexpr int ST = decltype(stage_c)::value;
        // pieces of candidates .. it+6 have been issued; the fragments of candidate it+1 are read below, so at most the FIVE
        // youngest (it+2 .. it+6) may still be in flight.  (The first version waited for vmcnt <= 6, i.e. only proved
        // candidate `it`: harmless while every step took a microsecond, a race once workgroups of mostly-padding parts
        // began to step faster than the DMA latency / 6.)
        if (it + SW8_NS - 1 < ncand) wait_vmcnt<SW8_NS - 3>(); else wait_vmcnt<0>();   // (tail: no younger pieces are counted on)
        __builtin_amdgcn_s_barrier();
        if constexpr (!(P4V_SW8_DBG & 1)) if (it + SW8_NS - 1 < ncand) issue((ST + SW8_NS - 1) % SW8_NS);
        if (!act) {                                    // a part of pure padding: stream and barriers only
            if (lane == 63) res[(c - c_lo) * 8 + pos] = 0.0f;
            ++c;
            return;
        }
        if (it + 1 < ncand) {
            read_fr(nxt, std::integral_constant<int, (ST + 1) % SW8_NS>{});
            __builtin_amdgcn_s_waitcnt(0xC07F | (NRD << 8));
        } else {
            __builtin_amdgcn_s_waitcnt(0xC07F);
        }
        asm volatile("" : "+v"(curf.t00), "+v"(curf.t01) :: "memory");
        if (!ROWS_FIXED) asm volatile("" : "+v"(curf.t10), "+v"(curf.t11));
        if constexpr ((P4V_SW8_DBG & 2) != 0) { acc[0] = zero16; acc[1] = zero16; acc[0][0] = curf.t00[0] + curf.t01[1]; }
        else if (ROWS_FIXED) {      // A (rows) in registers, B (columns) streamed
            acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fx[0][0], curf.t00, zero16, 0, 0, 0);
            if (act1) acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fx[1][0], curf.t00, zero16, 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fx[0][1], curf.t01, acc[0], 0, 0, 0);
            if (act1) acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fx[1][1], curf.t01, acc[1], 0, 0, 0);
        } else {               // A (rows) streamed, B (columns) in registers
            acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(curf.t00, fx[0][0], zero16, 0, 0, 0);
            if (act1) acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(curf.t10, fx[0][0], zero16, 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(curf.t01, fx[0][1], acc[0], 0, 0, 0);
            if (act1) acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(curf.t11, fx[0][1], acc[1], 0, 0, 0);
        }
        // ---- fused similarity epilogue of candidate c: one float per wave -----------------------------------------------
        const float s1 = s1tab[(c - c_lo) * 8 + pos];
        v2f sum2 = {0.0f, 0.0f};
        if constexpr ((P4V_SW8_DBG & 4) != 0) sum2.x = (float)(acc[0][0] + acc[1][5]) * s1;
        else
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i == 1 && !act1) break;                  // a block of pure padding rows
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const v2f a = {(float)acc[i][r], (float)acc[i][r + 1]};
                const v2f uu = {u[i][r], u[i][r + 1]};
                const v2f ww = {w[i][r], w[i][r + 1]};
                const v2f d = uu - a * s1;
                if (EPI == EPI_SQ_W) { const v2f t2 = ww * d; sum2 = t2 * t2 + sum2; }
                else if (EPI == EPI_SQ) sum2 = d * d + sum2;
                else if (EPI == EPI_ABS) sum2 += v2f{fabsf(d.x), fabsf(d.y)};
                else sum2 = (ww * d) * d + sum2;
            }
        }
        const float sum = wave_sum_dpp(sum2.x + sum2.y);           // fixed order: deterministic
        if (lane == 63) res[(c - c_lo) * 8 + pos] = sum;
        ++c;
    };
    if (ncand >= SW8_NS) wait_vmcnt<SW8_NS - 2>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    read_fr(fa, std::integral_constant<int, 0>{});
    for (int it = 0; it < ncand; it += SW8_NS) {
        step(it, std::integral_constant<int, 0>{}, fa, fb);
        if (it + 1 < ncand) step(it + 1, std::integral_constant<int, 1>{}, fb, fa);
        if (it + 2 < ncand) step(it + 2, std::integral_constant<int, 2>{}, fa, fb);
        if (it + 3 < ncand) step(it + 3, std::integral_constant<int, 3>{}, fb, fa);
        if (it + 4 < ncand) step(it + 4, std::integral_constant<int, 4>{}, fa, fb);
        if (it + 5 < ncand) step(it + 5, std::integral_constant<int, 5>{}, fb, fa);
        if (it + 6 < ncand) step(it + 6, std::integral_constant<int, 6>{}, fa, fb);
        if (it + 7 < ncand) step(it + 7, std::integral_constant<int, 7>{}, fb, fa);
    }
#undef P4V_DSR
    __syncthreads();
    for (int i = tid; i < ncand * 8; i += 512) {
        const int cc = c_lo + i / 8, wv = i % 8;
        p.part[(long)cc * p.p_cs + (long)z * p.p_zs + (long)(mt * 2 + (wv >> 2)) * p.Np + nt * 4 + (wv & 3)] = res[i];
    }
}
template <bool ROWS_FIXED, int EPI, bool SKIP>
__global__ __launch_bounds__(512, 2) void k_sweep8(SweepParams p) { k_sweep8_body<ROWS_FIXED, EPI, SKIP>(p, P4V_BIDX, P4V_GDIM); }
template <bool ROWS_FIXED, int EPI, bool SKIP>
__global__ __launch_bounds__(512, 2) void k_sweep8_g(GroupArgs<SweepParams> a) { P4V_GROUP_ENTER(a); k_sweep8_body<ROWS_FIXED, EPI, SKIP>(a.p[m_], vb_, vg_); }

// ------------------------------------------------------------------------------------------
// k_sweep9: single-k-tile sweeps (q.k^T, K <= 64) on 16 x 16 blocks
// ------------------------------------------------------------------------------------------
// k_sweep8 is bound by its epilogue (profiles/r2_sweep8_ablation.txt: 60 % of the launch), and its 128 x 128 tiles compute 1.69 x
// the valid outputs at 197 tokens, 3.16 x at the 144 tokens of a Swin window.  Here the score matrix of one batch entry is cut into
// 16 x 16 blocks (mfma_i32_16x16x64_i8: one instruction per block and candidate; 197 -> 13 x 13 blocks = 1.11 x, 144 -> 9 x 9 = 1.0 x),
// the blocks are dealt in row-major order to the 8 waves of `halves` workgroups (<= 12 blocks per wave), and every wave keeps per
// block: the fixed operand's fragment, raw_out / metric weight (4 + 4 values per lane) and its accumulator.  The expanded operand
// streams as in k_sweep8 (8-deep ring, one stage = the whole operand of one candidate: 256 rows x 64 B, two 1 KB pieces per wave).
// P4V_SW9_NW waves per workgroup: 8 = one workgroup per CU behind an 8-deep ring; 4 = TWO workgroups per CU (4-deep rings of
// 64 KB each): the candidate step is a latency chain (barrier, fragment reads, MFMAs, epilogue) and the two waves of a SIMD now
// belong to different workgroups -- different barriers -- so one's VALU phase runs under the other's waits instead of both
// waiting and both computing in lock step.
#ifndef P4V_SW9_NW
#define P4V_SW9_NW 4
#endif
static constexpr int SW9_NW = P4V_SW9_NW, SW9_NS = (SW9_NW == 4 ? 4 : 8), SW9_STAGE = 256 * 64, SW9_NB = 12;
static constexpr int SW9_PIECES = 16 / SW9_NW;            // 1 KB LDS-DMA pieces (16 rows) per wave and candidate

template <bool ROWS_FIXED, int EPI>
__device__ __forceinline__ void k_sweep9_body(const SweepParams& p, const uint3 blockIdx, const uint3 gridDim) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* res = reinterpret_cast<float*>(smem + SW9_NS * SW9_STAGE);   // [per][SW9_NW waves]
    typedef int v4i_ __attribute__((ext_vector_type(4)));

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int half = blockIdx.x, z = blockIdx.y;
    const int per = (p.c1 - p.c0 + gridDim.z - 1) / gridDim.z;
    int c_lo_ = p.c0 + blockIdx.z * per, c_hi_ = min(p.c1, c_lo_ + per);
    clip_crange(p.crange, c_lo_, c_hi_);
    if (p.crange_blk) clip_crange_blk(p.crange_blk, z % p.cb_div, c_lo_, c_hi_);
    const int c_lo = c_lo_, c_hi = c_hi_;
    if (c_lo >= c_hi) return;
    const int ncand = c_hi - c_lo;

    // ---- this wave's blocks: a contiguous run of the row-major block list ---------------------------------------------------
    const int CBk = (p.N + 15) / 16, nb = ((p.M + 15) / 16) * CBk;
    const int nbh = (nb + p.halves - 1) / p.halves;
    const int hb0 = half * nbh, hb1 = min(nb, hb0 + nbh);
    const int nbw = (max(0, hb1 - hb0) + SW9_NW - 1) / SW9_NW;
    const int b0 = hb0 + wid * nbw;
    const int nblk = __builtin_amdgcn_readfirstlane(max(0, min(hb1, b0 + nbw) - b0));

    const float* biasz = p.bias ? p.bias + (long)z * p.bias_zs : nullptr;
    float u[SW9_NB][4], w[SW9_NB][4];
    v4i_ fxb[SW9_NB];
    unsigned saddr[SW9_NB];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const int wm = p.wt_mode;
#pragma unroll
    for (int j = 0; j < SW9_NB; ++j) {
        const int bj = min(b0 + j, nb - 1);
        const int rb = bj / CBk, cb = bj - rb * CBk;
        const int n = cb * 16 + l15;
        const int nc = min(n, p.N - 1);
        const long ncol_off = (long)z * p.o_zs + (long)(nc / p.o_ninner) * p.o_nbs + (long)(nc % p.o_ninner) * p.o_ns;
        const float bias_n = (biasz && n < p.N) ? biasz[n] : 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = rb * 16 + 4 * l4 + e;
            const int mc = min(m, p.M - 1);
            const long idx = ncol_off + (long)(mc / p.o_inner) * p.o_bs + (long)(mc % p.o_inner) * p.o_ms;
            const float o = p.O[idx], gw = p.Wt[idx];
            const bool ok = j < nblk && n < p.N && m < p.M;
            float wv;
            if (wm == 1) wv = gw; else if (wm == 2) wv = o; else if (wm == 3) wv = fabsf(o); else wv = 1.0f;
            u[j][e] = ok ? o - bias_n : 0.0f;
            w[j][e] = ok ? wv : 0.0f;
        }
        // fragments: 16 rows x 64 B, lane (l4, l15) holds bytes [16 l4, 16 l4 + 16) of row l15 -- 1 KB contiguous per wave
        const int fr = (ROWS_FIXED ? rb : cb) * 16 + l15, sr = (ROWS_FIXED ? cb : rb) * 16 + l15;
        const char* gF = ROWS_FIXED ? (const char*)p.A + (long)z * p.a_zs : (const char*)p.B + (long)z * p.b_zs;
        fxb[j] = *reinterpret_cast<const v4i_*>(gF + (long)fr * SW_BKB + l4 * 16);
        saddr[j] = lds0 + sr * SW_BKB + l4 * 16;
    }
    const int sb = __builtin_amdgcn_readfirstlane(p.sb_mode == 2 ? z % p.sb_div : 0);
    float* s1tab = res + per * SW9_NW;
    for (int i = lane; i < ncand; i += 64) s1tab[i * SW9_NW + wid] = p.S1 ? p.S1[(c_lo + i) * p.s_cs + sb] : 1.0f;

    // ---- the expanded operand streams: one candidate per stage, wave `wid` moves rows [256 / SW9_NW * wid, + 256 / SW9_NW) ---------------------
    const long t_cs = ROWS_FIXED ? p.b_cs : p.a_cs;
    const char* cur = ROWS_FIXED ? (const char*)p.B + (long)z * p.b_zs + (long)c_lo * p.b_cs
                                 : (const char*)p.A + (long)z * p.a_zs + (long)c_lo * p.a_cs;
    // (rows beyond the padded plane are clamped: they are never read as fragments)
    unsigned voff[SW9_PIECES];
#pragma unroll
    for (int k = 0; k < SW9_PIECES; ++k)
        voff[k] = (unsigned)(min(wid * (16 * SW9_PIECES) + 16 * k + (lane >> 2), p.rows_p_stream - 1) * SW_BKB + (lane & 3) * 16);
    const int lds_wave = wid * (1024 * SW9_PIECES);
    auto issue = [&](int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < SW9_PIECES; ++k) glds16(cur + voff[k], smem + stage * SW9_STAGE + lds_wave + k * 1024);
        cur += t_cs;
    };
    const int npre = min(SW9_NS - 1, ncand);
    for (int i = 0; i < npre; ++i) issue(i);

#define P4V_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    const v4i_ zero4 = {0, 0, 0, 0};
    int c = c_lo;
    // step `it` (compile-time stage): prove candidate `it` landed (own two pieces, then the barrier: candidates it+1 .. it+6 --
    // twelve pieces -- may stay in flight), refill the stage of candidate it-1 with it+7, read the fragments, one MFMA per block,
    // epilogue.
    auto step = [&](int it, auto stage_c) __attribute__((always_inline)) {
        constexpr int ST = decltype(stage_c)::value;
        if (it + SW9_NS - 1 <= ncand) wait_vmcnt<(SW9_NS - 2) * SW9_PIECES>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (it + SW9_NS - 1 < ncand) issue((ST + SW9_NS - 1) % SW9_NS);
        v4i_ sf[SW9_NB];
        // (stages 4 .. 7 lie beyond the 16-bit offset field of ds_read: second base)
        constexpr unsigned HI = (ST >> 2) * 65536u;
        constexpr int SO = (ST & 3) * SW9_STAGE;
        float s1;
        {   // the candidate's scale first: LDS returns in order, so it is covered by the first counted wait below
            const unsigned sa = lds0 + SW9_NS * SW9_STAGE + (per * SW9_NW + (c - c_lo) * SW9_NW + wid) * 4;
            asm volatile("ds_read_b32 %0, %1" : "=v"(s1) : "v"(sa));
        }
#pragma unroll
        for (int j = 0; j < SW9_NB; ++j) {
            const unsigned a_ = saddr[j] + HI;
            v4i_ t_;
            P4V_DSR(t_, a_, SO);
            sf[j] = t_;
        }
        v2f sum2 = {0.0f, 0.0f};
        auto block = [&](int j) __attribute__((always_inline)) {
            const v4i_ acc = ROWS_FIXED ? __builtin_amdgcn_mfma_i32_16x16x64_i8(fxb[j], sf[j], zero4, 0, 0, 0)
                                        : __builtin_amdgcn_mfma_i32_16x16x64_i8(sf[j], fxb[j], zero4, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const v2f a = {(float)acc[e], (float)acc[e + 1]};
                const v2f uu = {u[j][e], u[j][e + 1]};
                const v2f ww = {w[j][e], w[j][e + 1]};
                const v2f d = uu - a * s1;
                if (EPI == EPI_SQ_W) { const v2f t2 = ww * d; sum2 = t2 * t2 + sum2; }
                else if (EPI == EPI_SQ) sum2 = (ww * d) * d + sum2;
                else if (EPI == EPI_ABS) sum2 += ww * v2f{fabsf(d.x), fabsf(d.y)};
                else sum2 = (ww * d) * d + sum2;
            }
        };
        // two halves: the fragments of blocks 6 .. 11 are still in flight while blocks 0 .. 5 run their MFMAs and epilogue
        constexpr int H1 = SW9_NB / 2;
        __builtin_amdgcn_s_waitcnt(0xC07F | ((SW9_NB - H1) << 8));          // lgkmcnt(6): the scale and fragments 0 .. 5 have landed
        asm volatile("" : "+v"(s1));
#pragma unroll
        for (int j = 0; j < H1; ++j) { v4i_ t_ = sf[j]; asm volatile("" : "+v"(t_)); sf[j] = t_; }
#pragma unroll
        for (int j = 0; j < H1; ++j) {
            if (j >= nblk) break;
            block(j);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                                  // lgkmcnt(0)
#pragma unroll
        for (int j = H1; j < SW9_NB; ++j) { v4i_ t_ = sf[j]; asm volatile("" : "+v"(t_)); sf[j] = t_; }
#pragma unroll
        for (int j = H1; j < SW9_NB; ++j) {
            if (j >= nblk) break;
            block(j);
        }
        const float sum = wave_sum_dpp(sum2.x + sum2.y);           // fixed order: deterministic
        if (lane == 63) res[(c - c_lo) * SW9_NW + wid] = sum;
        ++c;
    };
    for (int it = 0; it < ncand; it += SW9_NS) {
        step(it, std::integral_constant<int, 0>{});
        if (it + 1 < ncand) step(it + 1, std::integral_constant<int, 1>{});
        if (it + 2 < ncand) step(it + 2, std::integral_constant<int, 2>{});
        if (it + 3 < ncand) step(it + 3, std::integral_constant<int, 3>{});
        if constexpr (SW9_NS == 8) {
            if (it + 4 < ncand) step(it + 4, std::integral_constant<int, 4>{});
            if (it + 5 < ncand) step(it + 5, std::integral_constant<int, 5>{});
            if (it + 6 < ncand) step(it + 6, std::integral_constant<int, 6>{});
            if (it + 7 < ncand) step(it + 7, std::integral_constant<int, 7>{});
        }
    }
#undef P4V_DSR
    __syncthreads();
    for (int i = tid; i < ncand * SW9_NW; i += SW9_NW * 64)
        p.part[(long)(c_lo + i / SW9_NW) * p.p_cs + (long)z * p.p_zs + half * SW9_NW + (i % SW9_NW)] = res[i];
}
template <bool ROWS_FIXED, int EPI>
__global__ __launch_bounds__(SW9_NW * 64, 2) void k_sweep9(SweepParams p) { k_sweep9_body<ROWS_FIXED, EPI>(p, P4V_BIDX, P4V_GDIM); }
template <bool ROWS_FIXED, int EPI>
__global__ __launch_bounds__(SW9_NW * 64, 2) void k_sweep9_g(GroupArgs<SweepParams> a) { P4V_GROUP_ENTER(a); k_sweep9_body<ROWS_FIXED, EPI>(a.p[m_], vb_, vg_); }

// ------------------------------------------------------------------------------------------
// k_slice_b: stage A of a pruned MatMul B search -- all candidates of B on the 16-row sample slice, B quantised IN the kernel
// ------------------------------------------------------------------------------------------
// Stage A of the B search of an attention matmul (p4v_api.hip::run_pass_pruned) scores the 100 candidate scales of the WHOLE
// column operand (the keys of q.k^T, the values of attn.v) against the 16 heaviest query rows of every (image, head).  On the sweep
// kernels that meant materialising 100 int8 planes of B (629 MB per ViT-B module for the keys, 630 MB for the values: k_pack at
// 4.6 TB/s) only to stream them once through a GEMM with 16 useful rows (k_sweep9 / k_sweep2 at the HBM rate, or below it: the
// fixed twin planes of attn.v were re-streamed per candidate) -- both memory-bound on bytes that exist for this one read.
// Here a workgroup owns one (image, head): B (fp32, 50 KB) is read ONCE into the LDS (transposed to [n][k], padded rows), every
// wave takes every fourth candidate and quantises B's 16 x 64 fragments in registers with k_pack's own arithmetic
// (quant_fast1 + the exactness check + the IEEE division for flagged elements: the same integers), feeds them to
// mfma_i32_16x16x64_i8 against the slice's fragments, which stay in registers with raw_out / the metric weight, and writes one
// float per (candidate, batch entry).  No candidate plane is written or read; what is left is the VALU work of the quantisation.
// The planes stage B2 needs (the few surviving candidates, all rows) are packed on demand as before.
struct SliceBParams {
    const int8_t* A; const int8_t* A2;     // int8 planes [Z][16][Kp] of the fixed row operand (its slice); A2: twin second plane
    const float* B; long b_z2, b_z, b_n, b_k; int zdiv;     // fp32 column operand: element (z, n, k) at B[(z / zdiv) b_z2 + (z % zdiv) b_z + n b_n + k b_k]
    const float* bscale; int bs_cs, bs_div;                 // candidate scales: bscale[c * bs_cs + z % bs_div]
    int lo, hi;                                             // grid clamp of B
    const float* S1; const float* S2; int s_cs, s_div;      // combined output scales [C][s_cs] of plane 1 / 2, block z % s_div
    const float* O; const float* Wt; int wt_mode;           // slice tiles [Z][16][N] fp32 (dense): raw_out, metric weight source
    int Z, M, K, Kp, N, C;                                  // M <= 16 valid slice rows
    float* part;                                            // [C][Z]
};
template <bool TWIN, int KTM, int NBM, int EPI>
__device__ __forceinline__ void k_slice_b_body(const SliceBParams& p, const uint3 blockIdx, const uint3 gridDim) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Bt = reinterpret_cast<float*>(smem);                         // [NB * 16][Kp + 4] fp32, zero padded
    typedef int v4i_ __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int z = blockIdx.x;
    const int ktn = p.Kp / 64, nb = (p.N + 15) / 16, ldb = p.Kp + 4;
    const int per = (p.C + gridDim.y - 1) / gridDim.y;
    const int c_lo = blockIdx.y * per, c_hi = min(p.C, c_lo + per);
    // ---- B of this batch entry -> LDS as [n][k]; the faster-varying index of the copy follows the contiguous source stride --------
    const float* Bz = p.B + (p.zdiv > 0 ? (long)(z / p.zdiv) * p.b_z2 + (long)(z % p.zdiv) * p.b_z : (long)z * p.b_z);
    const int rows = nb * 16;
    if (p.b_k == 1) {
        for (int i = tid; i < rows * p.Kp; i += 256) {
            const int n = i / p.Kp, k = i - n * p.Kp;
            Bt[n * ldb + k] = (n < p.N && k < p.K) ? Bz[(long)n * p.b_n + k] : 0.0f;
        }
    } else {
        for (int i = tid; i < rows * p.Kp; i += 256) {
            const int k = i / rows, n = i - k * rows;
            Bt[n * ldb + k] = (n < p.N && k < p.K) ? Bz[(long)n * p.b_n + (long)k * p.b_k] : 0.0f;
        }
    }
    // ---- fixed fragments of the slice (16 rows x 64 B per k-tile: lane (l4, l15) holds bytes [16 l4, +16) of row l15) -----------
    v4i_ fa[KTM], fa2[KTM];
#pragma unroll
    for (int kt = 0; kt < KTM; ++kt) {
        const long off = ((long)z * 16 + l15) * p.Kp + (long)min(kt, ktn - 1) * 64 + l4 * 16;
        fa[kt] = *reinterpret_cast<const v4i_*>(p.A + off);
        if (TWIN) fa2[kt] = *reinterpret_cast<const v4i_*>(p.A2 + off);
    }
    // ---- raw_out / metric weight of the 16 x N tile in accumulator layout: lane -> column l15 of a block, rows 4 l4 + e ---------
    float u[NBM][4], w[NBM][4];
    const int wm = p.wt_mode;
#pragma unroll
    for (int j = 0; j < NBM; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = j * 16 + l15, m = 4 * l4 + e;
            const bool ok = j < nb && n < p.N && m < p.M;
            const long idx = ((long)z * 16 + min(m, 15)) * p.N + min(n, p.N - 1);
            const float o = p.O[idx], gw = p.Wt[idx];
            float wv;
            if (wm == 1) wv = gw; else if (wm == 2) wv = o; else if (wm == 3) wv = fabsf(o); else wv = 1.0f;
            u[j][e] = ok ? o : 0.0f;
            w[j][e] = ok ? wv : 0.0f;
        }
    __syncthreads();
    const int sbz = p.bs_div > 0 ? z % p.bs_div : 0, ssz = p.s_div > 0 ? z % p.s_div : 0;
    const float flo = (float)p.lo, fhi = (float)p.hi;
    const bool wide = !(fmaxf(-flo, fhi) < 129.0f);
    const float* brow = Bt + l15 * ldb + l4 * 16;                        // + (block * 16) * ldb + k-tile * 64
    for (int c = c_lo + wid; c < c_hi; c += 4) {
        const float s = p.bscale[(long)c * p.bs_cs + sbz];
        const float rcp = 1.0f / s;
        const float s1 = p.S1 ? p.S1[(long)c * p.s_cs + ssz] : 1.0f;
        const float s2 = (TWIN && p.S2) ? p.S2[(long)c * p.s_cs + ssz] : 1.0f;
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < NBM; ++j) {
            if (j < nb) {                                            // (uniform guard, no break: the loop must unroll -- u / w are registers)
            v4i_ acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
#pragma unroll
            for (int kt = 0; kt < KTM; ++kt) {
                if (kt < ktn) {
                const v4f* src = reinterpret_cast<const v4f*>(brow + (j * 16) * ldb + kt * 64);
                float x[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) { const v4f t = src[q]; x[q * 4] = t[0]; x[q * 4 + 1] = t[1]; x[q * 4 + 2] = t[2]; x[q * 4 + 3] = t[3]; }
                // k_pack's hot path (symmetric grid): x * (1 / s) where provably equal to the division, the division otherwise
                unsigned qb[16];
                float maxdev = 0.0f, magic = PACK_MAGIC;
                asm volatile("" : "+v"(magic));
#pragma unroll
                for (int e = 0; e < 16; ++e) qb[e] = quant_fast1(x[e], rcp, flo - 0.49f, fhi + 0.49f, magic, maxdev);
                const bool bad = !(maxdev <= 0.49996f) || !(rcp < 3.0e38f) || wide;
                if (__any(bad)) {
                    float sd = s;
                    asm volatile("" : "+v"(sd));
                    if (bad) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) qb[e] = __builtin_bit_cast(unsigned, fminf(fmaxf(rintf(x[e] / sd), flo), fhi) + PACK_MAGIC);
                    }
                }
                v4i_ fb;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned lo16 = __builtin_amdgcn_perm(qb[q * 4 + 1], qb[q * 4], 0x0c0c0400u);
                    const unsigned hi16 = __builtin_amdgcn_perm(qb[q * 4 + 3], qb[q * 4 + 2], 0x0c0c0400u);
                    fb[q] = (int)(lo16 | (hi16 << 16));
                }
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[kt], fb, acc, 0, 0, 0);
                if (TWIN) acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa2[kt], fb, acc2, 0, 0, 0);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float d = u[j][e] - (float)acc[e] * s1;
                if (TWIN) d -= (float)acc2[e] * s2;
                const float ww = w[j][e];
                if (EPI == EPI_SQ_W) { const float t2 = ww * d; sum = fmaf(t2, t2, sum); }
                else if (EPI == EPI_ABS) sum = fmaf(ww, fabsf(d), sum);
                else sum = fmaf(ww * d, d, sum);                        // EPI_SQ (w = validity mask) and EPI_W_SQ
            }
            }
        }
        sum = wave_sum_dpp(sum);
        if (lane == 63) p.part[(long)c * p.Z + z] = sum;
    }
}
template <bool TWIN, int KTM, int NBM, int EPI>
__global__ __launch_bounds__(256, 2) void k_slice_b(SliceBParams p) { k_slice_b_body<TWIN, KTM, NBM, EPI>(p, P4V_BIDX, P4V_GDIM); }
template <bool TWIN, int KTM, int NBM, int EPI>
__global__ __launch_bounds__(256, 2) void k_slice_b_g(GroupArgs<SliceBParams> a) { P4V_GROUP_ENTER(a); k_slice_b_body<TWIN, KTM, NBM, EPI>(a.p[m_], vb_, vg_); }

// k_slice_b2 (round 5): k_slice_b with B in REGISTERS.  k_slice_b dealt the candidates over the four waves: every wave
// re-read all of B (fp32) from the LDS for each of its candidates -- 4 KB of ds_read_b128 and the address arithmetic per 1 KB
// fragment, ~200 instructions per fragment, VALU-bound at 0.015 of the matrix peak.  Here the waves deal the 16-COLUMN BLOCKS
// of B instead (q.k^T: 13 blocks of one k-tile -> 4 / 3 / 3 / 3; attn.v: 4 blocks of 4 k-tiles -> one each): a lane keeps its
// 16 fp32 values of each of its (at most 4) fragments for the whole kernel, every wave works on EVERY candidate and quantises
// only its own fragments (quant16_sat8 on the symmetric 8-bit grid: 4.5 operations per element), and writes one float per
// (candidate, batch entry, wave); k_finish adds the four.  No LDS, no candidate planes.
struct SliceB2Params {
    SliceBParams b;
    float qbias;            // bias of quant16_sat8's conversion (k_probe_cvt); 0: the conversion is not usable -> SAT8 must be false
};
template <bool TWIN, int KTM, int NBW, int EPI, bool SAT8>
__device__ __forceinline__ void k_slice_b2_body(const SliceB2Params& pp, const uint3 blockIdx, const uint3 gridDim) {
    const SliceBParams& p = pp.b;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int z = blockIdx.x;
    const int ktn = p.Kp / 64, nb = (p.N + 15) / 16;
    const int per = (p.C + gridDim.y - 1) / gridDim.y;
    const int c_lo = blockIdx.y * per, c_hi = min(p.C, c_lo + per);
    const float* Bz = p.B + (p.zdiv > 0 ? (long)(z / p.zdiv) * p.b_z2 + (long)(z % p.zdiv) * p.b_z : (long)z * p.b_z);
    // ---- this wave's fragments of B (fp32): block j = wid + 4 i, k-tile kt; lane (l4, l15) holds k = 64 kt + 16 l4 + e of column 16 j + l15
    float xb[NBW][KTM][16];
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = (wid + 4 * i) * 16 + l15, k = kt * 64 + l4 * 16 + e;
                const bool ok = n < p.N && k < p.K;
                const float v = Bz[(long)min(n, p.N - 1) * p.b_n + (long)min(k, p.K - 1) * p.b_k];
                xb[i][kt][e] = ok ? v : 0.0f;
            }
    // ---- fixed fragments of the slice (16 rows x 64 B per k-tile: lane (l4, l15) holds bytes [16 l4, +16) of row l15) -----------
    v4i fa[KTM], fa2[KTM];
#pragma unroll
    for (int kt = 0; kt < KTM; ++kt) {
        const long off = ((long)z * 16 + l15) * p.Kp + (long)min(kt, ktn - 1) * 64 + l4 * 16;
        fa[kt] = *reinterpret_cast<const v4i*>(p.A + off);
        if (TWIN) fa2[kt] = *reinterpret_cast<const v4i*>(p.A2 + off);
    }
    // ---- raw_out / metric weight of this wave's 16 x 16 blocks in accumulator layout: lane -> column l15, rows 4 l4 + e ----------
    float u[NBW][4], w[NBW][4];
    const int wm = p.wt_mode;
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = wid + 4 * i;
            const int n = j * 16 + l15, m = 4 * l4 + e;
            const bool ok = j < nb && n < p.N && m < p.M;
            const long idx = ((long)z * 16 + min(m, 15)) * p.N + min(n, p.N - 1);
            const float o = p.O[idx], gw = p.Wt[idx];
            float wv;
            if (wm == 1) wv = gw; else if (wm == 2) wv = o; else if (wm == 3) wv = fabsf(o); else wv = 1.0f;
            u[i][e] = ok ? o : 0.0f;
            w[i][e] = ok ? wv : 0.0f;
        }
    const int sbz = p.bs_div > 0 ? z % p.bs_div : 0, ssz = p.s_div > 0 ? z % p.s_div : 0;
    const float flo = (float)p.lo, fhi = (float)p.hi;
    const bool wide = !(fmaxf(-flo, fhi) < 129.0f);
    for (int c = c_lo; c < c_hi; ++c) {
        const float s = p.bscale[(long)c * p.bs_cs + sbz];
        const float rcp = 1.0f / s;
        const float s1 = p.S1 ? p.S1[(long)c * p.s_cs + ssz] : 1.0f;
        const float s2 = (TWIN && p.S2) ? p.S2[(long)c * p.s_cs + ssz] : 1.0f;
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            if (wid + 4 * i < nb) {                                  // (wave-uniform guard, no break: the loop must unroll -- xb / u / w are registers)
                v4i acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
#pragma unroll
                for (int kt = 0; kt < KTM; ++kt) {
                    if (kt < ktn) {
                        v4i fb;
                        if constexpr (SAT8) quant16_sat8(xb[i][kt], s, rcp, pp.qbias, fb);
                        else quant16_any(xb[i][kt], s, rcp, flo, fhi, wide, fb);
                        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[kt], fb, acc, 0, 0, 0);
                        if (TWIN) acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa2[kt], fb, acc2, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float d = u[i][e] - (float)acc[e] * s1;
                    if (TWIN) d -= (float)acc2[e] * s2;
                    const float ww = w[i][e];
                    if (EPI == EPI_SQ_W) { const float t2 = ww * d; sum = fmaf(t2, t2, sum); }
                    else if (EPI == EPI_ABS) sum = fmaf(ww, fabsf(d), sum);
                    else sum = fmaf(ww * d, d, sum);                        // EPI_SQ (w = validity mask) and EPI_W_SQ
                }
            }
        }
        sum = wave_sum_dpp(sum);
        if (lane == 63) p.part[((long)c * p.Z + z) * 4 + wid] = sum;
    }
}
template <bool TWIN, int KTM, int NBW, int EPI, bool SAT8>
__global__ __launch_bounds__(256, 2) void k_slice_b2(SliceB2Params p) { k_slice_b2_body<TWIN, KTM, NBW, EPI, SAT8>(p, P4V_BIDX, P4V_GDIM); }
template <bool TWIN, int KTM, int NBW, int EPI, bool SAT8>
__global__ __launch_bounds__(256, 2) void k_slice_b2_g(GroupArgs<SliceB2Params> a) { P4V_GROUP_ENTER(a); k_slice_b2_body<TWIN, KTM, NBW, EPI, SAT8>(a.p[m_], vb_, vg_); }

// k_slice_a: the same for a MatMul A search with K <= 64 (q.k^T): the EXPANDED operand is the 16-row slice itself -- a lane keeps
// its 16 fp32 values of the slice in registers and re-quantises them per candidate (one fragment), the fixed operand B (int8,
// packed once: [Z][NB * 16][64]) stays in registers as NB fragments.  One workgroup per (image, head), candidates dealt over
// the four waves; replaces the slice's candidate planes + a k_sweep9 launch (78 -> ~30 us per pass).
struct SliceAParams {
    const float* A;                        // fp32 slice [Z][16][K] (dense)
    const int8_t* B;                       // int8 plane [Z][NB * 16][64] of the fixed column operand
    const float* ascale; int as_cs, as_div;                 // candidate scales: ascale[c * as_cs + z % as_div]
    int lo, hi;
    const float* S1; int s_cs, s_div;
    const float* O; const float* Wt; int wt_mode;           // [Z][16][N]
    int Z, M, K, N, C;
    float* part;                                            // [C][Z]
    float qbias;                                            // != 0: quant16_sat8 (symmetric 8-bit grid; bias from k_probe_cvt), else quant_fast1
};
template <int NBM, int EPI>
__device__ __forceinline__ void k_slice_a_body(const SliceAParams& p, const uint3 blockIdx, const uint3 gridDim) {
    typedef int v4i_ __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int z = blockIdx.x;
    const int nb = (p.N + 15) / 16;
    const int per = (p.C + gridDim.y - 1) / gridDim.y;
    const int c_lo = blockIdx.y * per, c_hi = min(p.C, c_lo + per);
    float x[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int k = l4 * 16 + e;
        x[e] = (l15 < p.M && k < p.K) ? p.A[((long)z * 16 + l15) * p.K + k] : 0.0f;
    }
    v4i_ fb[NBM];
    float u[NBM][4], w[NBM][4];
    const int wm = p.wt_mode;
#pragma unroll
    for (int j = 0; j < NBM; ++j) {
        fb[j] = *reinterpret_cast<const v4i_*>(p.B + ((long)z * nb * 16 + min(j, nb - 1) * 16 + l15) * 64 + l4 * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = j * 16 + l15, m = 4 * l4 + e;
            const bool ok = j < nb && n < p.N && m < p.M;
            const long idx = ((long)z * 16 + min(m, 15)) * p.N + min(n, p.N - 1);
            const float o = p.O[idx], gw = p.Wt[idx];
            float wv;
            if (wm == 1) wv = gw; else if (wm == 2) wv = o; else if (wm == 3) wv = fabsf(o); else wv = 1.0f;
            u[j][e] = ok ? o : 0.0f;
            w[j][e] = ok ? wv : 0.0f;
        }
    }
    const int saz = p.as_div > 0 ? z % p.as_div : 0, ssz = p.s_div > 0 ? z % p.s_div : 0;
    const float flo = (float)p.lo, fhi = (float)p.hi;
    const bool wide = !(fmaxf(-flo, fhi) < 129.0f);
    for (int c = c_lo + wid; c < c_hi; c += 4) {
        const float s = p.ascale[(long)c * p.as_cs + saz];
        const float rcp = 1.0f / s;
        const float s1 = p.S1 ? p.S1[(long)c * p.s_cs + ssz] : 1.0f;
        v4i fa;
        if (p.qbias != 0.0f) quant16_sat8(x, s, rcp, p.qbias, fa);
        else quant16_any(x, s, rcp, flo, fhi, wide, fa);
        float sum = 0.0f;
        const v4i_ zero4 = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < NBM; ++j) {
            if (j < nb) {
                const v4i_ acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb[j], zero4, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = u[j][e] - (float)acc[e] * s1;
                    const float ww = w[j][e];
                    if (EPI == EPI_SQ_W) { const float t2 = ww * d; sum = fmaf(t2, t2, sum); }
                    else if (EPI == EPI_ABS) sum = fmaf(ww, fabsf(d), sum);
                    else sum = fmaf(ww * d, d, sum);
                }
            }
        }
        sum = wave_sum_dpp(sum);
        if (lane == 63) p.part[(long)c * p.Z + z] = sum;
    }
}
template <int NBM, int EPI>
__global__ __launch_bounds__(256, 2) void k_slice_a(SliceAParams p) { k_slice_a_body<NBM, EPI>(p, P4V_BIDX, P4V_GDIM); }
template <int NBM, int EPI>
__global__ __launch_bounds__(256, 2) void k_slice_a_g(GroupArgs<SliceAParams> a) { P4V_GROUP_ENTER(a); k_slice_a_body<NBM, EPI>(a.p[m_], vb_, vg_); }

// ------------------------------------------------------------------------------------------
// k_sweep2g: k_sweep2 for LARGE K with two candidates per pass (weight search of fc2-like layers)
// ------------------------------------------------------------------------------------------
// PMC (profiles/r1_pmc_fc2_sweep2.txt): at K = 3072 k_sweep2 moves 16 KB (24 KB twin) per k-tile from L2 into LDS for
// 128 x 128 x 64 MACs and runs at the L2->CU bandwidth (~18.6 TB/s of the ~20 the LDS-DMA micro-benchmark reaches),
// the matrix pipe 39 % busy.  In the weight search the ROW operand (activations; two planes for the post-GELU twin)
// is candidate-invariant: this kernel feeds TWO candidates of the column operand from one pass over the row planes --
// 16 + 2 x 8 = 32 KB per k-tile for twice the MACs (-33 % L2 bytes per MAC, -40 % fragment reads per MFMA).
// Four accumulator sets (2 candidates x 2 planes) leave no registers for the raw_out / raw_grad tile, so that tile is
// re-read once per candidate PAIR in the epilogue: 128 KB against 1.5 MB of operands per pair at K = 3072 (+8 %).
// Requirements (host): column operand expanded (b_cs != 0), row operand invariant (a_cs == 0), p.Z == 1 or z handled
// by blockIdx.y as in k_sweep2.  An odd candidate count runs its last candidate twice (second result dropped).
template <bool TWIN, int EPI>
__device__ __forceinline__ void k_sweep2g_body(const SweepParams& p, const uint3 blockIdx, const uint3 gridDim) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NPL = TWIN ? 4 : 3;                    // planes: A, (A2), B(c), B(c+1)
    constexpr int PLANE = SW2_NS * SW2_TILE;
    float* res = reinterpret_cast<float*>(smem + NPL * PLANE);   // [per][8 waves]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int g = lane >> 5, l31 = lane & 31;

    const int nwg = p.mtiles * p.ntiles;
    const int t = xcd_remap(blockIdx.x, nwg);
    constexpr int GM = 4;
    const int per_group = GM * p.ntiles;
    const int first_m = (t / per_group) * GM;
    const int gsz = min(p.mtiles - first_m, GM);
    const int mt = first_m + (t % per_group) % gsz, nt = (t % per_group) / gsz;
    const int z = blockIdx.y;
    const int m0 = mt * SW_BM, n0 = nt * SW_BN;
    const int per = 2 * ((p.c1 - p.c0 + 2 * gridDim.z - 1) / (2 * gridDim.z));   // even: groups start on a pair
    int c_lo_ = p.c0 + blockIdx.z * per, c_hi_ = min(p.c1, c_lo_ + per);
    clip_crange(p.crange, c_lo_, c_hi_, true);
    const int c_lo = c_lo_, c_hi = c_hi_;
    if (c_lo >= c_hi) return;
    const int ncand = c_hi - c_lo, npairs = (ncand + 1) >> 1;

    const int n = n0 + wc * 32 + l31;
    const bool ncol_ok = n < p.N;
    const float* biasz = p.bias ? p.bias + (long)z * p.bias_zs : nullptr;
    const float bias_n = (biasz && ncol_ok) ? biasz[n] : 0.0f;
    // plain [M][N] output layout only (host: o_bs == o_nbs == 0): element (m, n) at z*o_zs + m*o_ms + n*o_ns
    const int nc = min(n, p.N - 1);
    const int mlane = m0 + wr * 64 + 4 * g;               // row of element (i = 0, r = 0) of this lane
    // Epilogue addressing: wave-uniform row base (SGPRs) + a per-lane offset (column, row group); every lane reads a
    // valid element, elements outside the matrix are masked by `ok` below.
    // (Per-element 64-bit addresses in VGPRs spill and make hipcc serialise the 64 loads of a candidate pair.)
    const int m0w = m0 + wr * 64;
    const int gl_max = min(4, p.M - 1), gl = min(4 * g, p.M - 1);
    const unsigned lane_off0 = (unsigned)(nc * (int)p.o_ns);                 // this lane's column
    const unsigned lane_off = lane_off0 + (unsigned)(gl * (int)p.o_ms);     // ... and its row group (+0 / +4 rows)
    const float* Ou = p.O + (long)z * p.o_zs;
    const float* Wu = p.Wt + (long)z * p.o_zs;
    const unsigned m_g = p.wt_mode == 1 ? 0xffffffffu : 0u;
    const unsigned m_o = p.wt_mode == 2 ? 0xffffffffu : p.wt_mode == 3 ? 0x7fffffffu : 0u;
    const unsigned m_1 = p.wt_mode == 0 ? 0x3f800000u : 0u;

    const int nw0 = n0 + wc * 32;
    const int sb = __builtin_amdgcn_readfirstlane(p.sb_mode == 1 ? min(nw0 / p.sb_div, p.s_cs - 1) : p.sb_mode == 2 ? z % p.sb_div : 0);
    float* s1tab = res + per * 8;
    float* s2tab = s1tab + per * 8;
    for (int i = lane; i < per; i += 64) {
        const int cc = min(c_lo + i, c_hi - 1);
        s1tab[i * 8 + wid] = p.S1 ? p.S1[cc * p.s_cs + sb] : 1.0f;
        if (TWIN) s2tab[i * 8 + wid] = p.S2 ? p.S2[cc * p.s_cs + sb] : 1.0f;
    }

    // ---- LDS-DMA: wave `wid` fills rows [16*wid, 16*wid+16) of every plane ------------------------------------------
    const int ld_row = wid * 16 + (lane >> 2);
    const int ld_chunk = (lane & 3) ^ ((ld_row >> 2) & 3);
    const unsigned voff = (unsigned)(ld_row * p.ldk + ld_chunk * 16);
    const char* curA = (const char*)p.A + (long)z * p.a_zs + (long)m0 * p.ldk;
    const char* curA2 = TWIN ? (const char*)p.A2 + (long)z * p.a2_zs + (long)m0 * p.ldk : nullptr;
    const char* curB0 = (const char*)p.B + (long)z * p.b_zs + (long)n0 * p.ldk + (long)c_lo * p.b_cs;
    long b1_off = (c_lo + 1 < c_hi) ? p.b_cs : 0;         // second candidate of the pair (the last odd one: itself)
    const int ktiles = p.ktiles;
    const long krow = (long)ktiles * SW_BKB;
    const int lds_wave = wid * 1024;
    const int total = npairs * ktiles;
    int ikt = 0, ipair = 0;
    auto issue = [&](int stage) __attribute__((always_inline)) {
        char* s = smem + stage * SW2_TILE + lds_wave;
        glds16(curA + voff, s);
        if (TWIN) glds16(curA2 + voff, s + PLANE);
        glds16(curB0 + voff, s + (NPL - 2) * PLANE);
        glds16(curB0 + b1_off + voff, s + (NPL - 1) * PLANE);
        curA += SW_BKB; curB0 += SW_BKB;
        if (TWIN) curA2 += SW_BKB;
        if (++ikt == ktiles) {                            // next pair: rewind the invariant planes, advance two candidates
            ikt = 0; ++ipair;
            curA -= krow; if (TWIN) curA2 -= krow;
            curB0 += 2 * p.b_cs - krow;
            b1_off = (c_lo + 2 * ipair + 1 < c_hi) ? p.b_cs : 0;
        }
    };

    v16i acc[2][2], acc2[2][2];                           // [candidate of the pair][32-row half]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[j][i][r] = 0; if (TWIN) acc2[j][i][r] = 0; }

    const int ra0 = wr * 64 + l31, ra1 = ra0 + 32, rb = wc * 32 + l31;
    const int sa0 = (ra0 >> 2) & 3, sa1 = (ra1 >> 2) & 3, sbz = (rb >> 2) & 3;
    const char* fA0 = smem + ra0 * 64;
    const char* fA1 = smem + ra1 * 64;
    const char* fB = smem + (NPL - 2) * PLANE + rb * 64;
    const int oa00 = (g ^ sa0) << 4, oa01 = ((2 + g) ^ sa0) << 4;
    const int oa10 = (g ^ sa1) << 4, oa11 = ((2 + g) ^ sa1) << 4;
    const int ob0 = (g ^ sbz) << 4, ob1 = ((2 + g) ^ sbz) << 4;

    const int npre = min(SW2_NS - 1, total);
    for (int i = 0; i < npre; ++i) issue(i);

    int kt = 0, pr = 0;
    auto tile = [&](int it, auto stage_c) __attribute__((always_inline)) {
        constexpr int ST = decltype(stage_c)::value;
        constexpr int SO = ST * SW2_TILE;
        if (it + 2 < total) wait_vmcnt<2 * NPL>(); else if (it + 1 < total) wait_vmcnt<NPL>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();   // tile `it` is in LDS for everyone; stage (it-1)%NS is free for everyone
        if (it + SW2_NS - 1 < total) issue((ST + SW2_NS - 1) % SW2_NS);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int oa0 = h ? oa01 : oa00, oa1 = h ? oa11 : oa10, ob = h ? ob1 : ob0;
            const v4i a0 = *reinterpret_cast<const v4i*>(fA0 + SO + oa0);
            const v4i a1 = *reinterpret_cast<const v4i*>(fA1 + SO + oa1);
            const v4i b0 = *reinterpret_cast<const v4i*>(fB + SO + ob);
            const v4i b1 = *reinterpret_cast<const v4i*>(fB + PLANE + SO + ob);
            acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, acc[1][1], 0, 0, 0);
            if (TWIN) {
                const v4i c0 = *reinterpret_cast<const v4i*>(fA0 + PLANE + SO + oa0);
                const v4i c1 = *reinterpret_cast<const v4i*>(fA1 + PLANE + SO + oa1);
                acc2[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(c0, b0, acc2[0][0], 0, 0, 0);
                acc2[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(c1, b0, acc2[0][1], 0, 0, 0);
                acc2[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(c0, b1, acc2[1][0], 0, 0, 0);
                acc2[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(c1, b1, acc2[1][1], 0, 0, 0);
            }
        }
        if (++kt == ktiles) {
            // ---- epilogue of the candidate pair: the raw_out / weight tile is streamed in, shared by both candidates
            const int ci = 2 * pr;
            const float s1a = s1tab[ci * 8 + wid], s1b = s1tab[(ci + 1) * 8 + wid];
            const float s2a = TWIN ? s2tab[ci * 8 + wid] : 1.0f, s2b = TWIN ? s2tab[(ci + 1) * 8 + wid] : 1.0f;
            float suma = 0.0f, sumb = 0.0f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 2; ++q) {                    // 8 elements at a time: all loads first, then the math
                    float uo[8], gw[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int r = q * 8 + e;
                        // uniform row of the g = 0 lanes; the g = 1 lanes sit 4 rows below.  Where those 4 rows would leave
                        // the matrix (last rows of a ragged tile) every lane reads the g = 0 row: the g = 1 elements are
                        // masked by `ok`, the g = 0 ones still get their own row
                        const int br = m0w + i * 32 + (r & 3) + 8 * (r >> 2);
                        const bool whole = br + gl_max <= p.M - 1;
                        const long rowb = (long)min(br, p.M - 1) * p.o_ms;
                        const unsigned lo = whole ? lane_off : lane_off0;
                        unsigned long long ob = (unsigned long long)(Ou + rowb), wb = (unsigned long long)(Wu + rowb);
                        asm volatile("" : "+s"(ob), "+s"(wb));   // SGPR pairs, computed here (not hoisted into 64 VGPR pairs)
                        typedef const __attribute__((address_space(1))) float* gptr_t;   // (an integer -> pointer cast is a FLAT pointer otherwise)
                        uo[e] = reinterpret_cast<gptr_t>(ob)[lo];
                        gw[e] = reinterpret_cast<gptr_t>(wb)[lo];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int r = q * 8 + e;
                        const bool ok = ncol_ok && (mlane + i * 32 + (r & 3) + 8 * (r >> 2)) < p.M;
                        const unsigned wbits = (__builtin_bit_cast(unsigned, gw[e]) & m_g) | (__builtin_bit_cast(unsigned, uo[e]) & m_o) | m_1;
                        const float uu = ok ? uo[e] - bias_n : 0.0f;
                        const float ww = ok ? __builtin_bit_cast(float, wbits) : 0.0f;
                        float da = uu - (float)acc[0][i][r] * s1a, db = uu - (float)acc[1][i][r] * s1b;
                        if (TWIN) { da -= (float)acc2[0][i][r] * s2a; db -= (float)acc2[1][i][r] * s2b; }
                        if (EPI == EPI_SQ_W) { const float ta = ww * da, tb = ww * db; suma = fmaf(ta, ta, suma); sumb = fmaf(tb, tb, sumb); }
                        else if (EPI == EPI_SQ) { suma = fmaf(da, da, suma); sumb = fmaf(db, db, sumb); }
                        else if (EPI == EPI_ABS) { suma += fabsf(da); sumb += fabsf(db); }
                        else { suma = fmaf(ww * da, da, suma); sumb = fmaf(ww * db, db, sumb); }
                        acc[0][i][r] = 0; acc[1][i][r] = 0;
                        if (TWIN) { acc2[0][i][r] = 0; acc2[1][i][r] = 0; }
                    }
                }
            suma = wave_sum_dpp(suma);
            sumb = wave_sum_dpp(sumb);
            if (lane == 63) { res[ci * 8 + wid] = suma; res[(ci + 1) * 8 + wid] = sumb; }
            kt = 0;
            ++pr;
        }
    };
    for (int it = 0; it < total; it += SW2_NS) {
        tile(it, std::integral_constant<int, 0>{});
        if (it + 1 < total) tile(it + 1, std::integral_constant<int, 1>{});
        if (it + 2 < total) tile(it + 2, std::integral_constant<int, 2>{});
        if (it + 3 < total) tile(it + 3, std::integral_constant<int, 3>{});
    }
    __syncthreads();
    for (int i = tid; i < ncand * 8; i += 512) {
        const int cc = c_lo + i / 8, wv = i % 8;
        p.part[(long)cc * p.p_cs + (long)z * p.p_zs + (long)(mt * 2 + (wv >> 2)) * p.Np + nt * 4 + (wv & 3)] = res[i];
    }
}
template <bool TWIN, int EPI>
__global__ __launch_bounds__(512, 2) void k_sweep2g(SweepParams p) { k_sweep2g_body<TWIN, EPI>(p, P4V_BIDX, P4V_GDIM); }
template <bool TWIN, int EPI>
__global__ __launch_bounds__(512, 2) void k_sweep2g_g(GroupArgs<SweepParams> a) { P4V_GROUP_ENTER(a); k_sweep2g_body<TWIN, EPI>(a.p[m_], vb_, vg_); }

// ------------------------------------------------------------------------------------------
// Stationary-operand int8 sweep (k_sweep4 below): parameter block
// ------------------------------------------------------------------------------------------
// The candidate-invariant operand (activations in the weight search, weights in the activation search) is
// re-used by all ~100 candidates.  k_sweep2 re-streams its 128 x K tile from L2 for every candidate; here the
// whole tile (K/64 k-tiles of 8 KB, <= 96 KB) is loaded into LDS ONCE per workgroup and only the
// candidate-expanded operand streams through a 3-stage LDS-DMA ring.  Tile = 128 stationary rows x 256
// streaming rows, 8 waves as 2 x 4, 64 x 64 per wave (2 x 2 MFMA 32x32x32): half the L2->LDS bytes per MAC and
// two thirds of the LDS reads per MFMA of k_sweep2, at one workgroup (8 waves, ~240 VGPRs) per CU.
// The MFMA rows are the stationary rows: in the activation search the output tile is therefore transposed
// (rows = output features, columns = samples); the raw_out / raw_grad tile is gathered through strides.
struct Sweep3Params {
    const int* crange;                  // optional device-side candidate range (clip_crange)
    const void* S; long s_zs;           // stationary plane [rows_p][ldk] (never candidate-expanded)
    const void* T; long t_cs, t_zs;     // streaming plane  [rows_p][C][ldk]: t_rs = bytes between rows (= C_chunk * ldk)
    long t_rs;
    int ldk, ktiles;
    const float* S1;                    // [C][s_cs] combined scales
    int s_cs, sb_on_t, sb_div;          // scale block = (stationary or streaming row) / sb_div
    const float* bias; int bias_on_t;   // bias indexed by the streaming (1) or stationary (0) row
    const float* O; const float* Wt; int wt_mode;
    long o_ss, o_ts;                    // element index = srow * o_ss + trow * o_ts
    int SR, TR;                         // valid stationary / streaming rows
    int c0, c1;
    float* part; long p_cs; int NG;     // part[c*p_cs + (st*2+wr)*NG + tt*4+wc]
    int stiles, ttiles;
    int dbg;
    int tile0, ntile;                   // k_sweep6: this launch covers tiles [tile0, tile0 + ntile) (ntile == 0: all of them)
    const float* E;                     // k_sweep6: epilogue operands in fragment order (k_prep_epi6); S is in fragment order too
    const int* crange_blk;              // k_sweep6: optional per-score-block candidate ranges (block = scale block of the tile)
#ifdef P4V_TRACE
    unsigned long long* trace;          // tuning builds only: [workgroup][8] timestamps (100 MHz) + hw id
#endif
};

// ------------------------------------------------------------------------------------------
// k_sweep4: int8 sweep with the candidate-INVARIANT operand stationary in LDS (Linear layers, K <= 768)
// ------------------------------------------------------------------------------------------
// The candidate-invariant operand (activations in the weight search, weights in the activation search) is
// re-used by all ~100 candidates: its whole 128 x K tile (K/64 k-tiles of 8 KB, <= 96 KB) is loaded into LDS
// ONCE per workgroup and only the candidate-expanded operand streams, through a 6-deep LDS-DMA ring.
// Tile 128 (stationary rows) x 128 (streaming rows), 8 waves as 2 x 4, 64 x 32 per wave.  What profiling the
// earlier variants taught (profiles/):
//   * every wave keeps TWO fragment sets: the ds_reads of k-tile t+1 are in flight while the MFMAs of k-tile t
//     run, and the barrier only has to prove that tile t+1 has landed;
//   * the loop is scalar-instruction bound if it does any bookkeeping: the expanded plane is therefore laid out
//     [row][candidate][K] so the stream cursor advances by a constant 64 B per tile (no candidate wrap), the
//     ring always issues (the plane has slack behind it; stale tiles are never consumed) so the vmcnt wait is a
//     constant, and ring / k-tile offsets wrap with one s_cselect each.
// The MFMA rows are the stationary rows: in the activation search the output tile is transposed (rows = output
// features, columns = samples); the raw_out / raw_grad tile is gathered through strides.
static constexpr int SW4_NS = 6;

template <int EPI>
__device__ __forceinline__ void k_sweep4_body(const Sweep3Params& p, const uint3 blockIdx, const uint3 gridDim) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ktiles = p.ktiles;
    const int ring0 = ktiles * SW2_TILE;                 // LDS byte offset of the ring
    float* res = reinterpret_cast<float*>(smem + ring0 + SW4_NS * SW2_TILE);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int g = lane >> 5, l31 = lane & 31;

    const int nwg = p.stiles * p.ttiles;
    const int t = xcd_remap(blockIdx.x, nwg);
    const int st = t % p.stiles, tt = t / p.stiles;    // neighbours share the streaming tile
    const int s0 = st * 128, t0 = tt * 128;
    const int per = (p.c1 - p.c0 + gridDim.z - 1) / gridDim.z;
    int c_lo_ = p.c0 + blockIdx.z * per, c_hi_ = min(p.c1, c_lo_ + per);
    clip_crange(p.crange, c_lo_, c_hi_);
    const int c_lo = c_lo_, c_hi = c_hi_;
    if (c_lo >= c_hi) return;

    // ---- stationary operand: all k-tiles of this workgroup's 128 rows, once ------------------------------
    const int ld_row = wid * 16 + (lane >> 2);
    const int ld_chunk = (lane & 3) ^ ((ld_row >> 2) & 3);
    {
        const char* gS = (const char*)p.S + (long)(s0 + ld_row) * p.ldk + ld_chunk * 16;
        for (int kt = 0; kt < ktiles; ++kt) glds16(gS + kt * SW_BKB, smem + kt * SW2_TILE + wid * 1024);
    }
    // ---- streaming operand [row][candidate][K]: one 16-row piece per wave per tile, cursor += 64 B per tile --------
    const char* curT = (const char*)p.T + (long)(t0 + ld_row) * p.t_rs + (long)(c_lo - p.c0) * p.ldk + ld_chunk * 16;
    const int total = (c_hi - c_lo) * ktiles;
#pragma unroll
    for (int i = 0; i < SW4_NS - 1; ++i) { glds16(curT, smem + ring0 + i * SW2_TILE + wid * 1024); curT += SW_BKB; }

    // ---- candidate-invariant epilogue operands: 2 MFMA tiles of 32 x 32 (rows = stationary, cols = streaming) --
    float u[2][16], w[2][16];
    {
        const int tr = t0 + wc * 32 + l31;
        const long toff = (long)min(tr, p.TR - 1) * p.o_ts;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long idx = toff + (long)min(s0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g, p.SR - 1) * p.o_ss;
                u[i][r] = p.O[idx];
                w[i][r] = p.Wt[idx];
            }
        float bias_s[2][16];
        const float bias_t = p.bias[p.bias_on_t ? min(tr, p.TR - 1) : 0];
        if (!p.bias_on_t) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) bias_s[i][r] = p.bias[min(s0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g, p.SR - 1)];
        }
        const bool t_ok = tr < p.TR;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = t_ok && (s0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g) < p.SR;
                const float o = u[i][r], gw = w[i][r];
                const float b = p.bias_on_t ? bias_t : bias_s[i][r];
                float wv;
                if (p.wt_mode == 1) wv = gw; else if (p.wt_mode == 2) wv = o; else if (p.wt_mode == 3) wv = fabsf(o); else wv = 1.0f;
                u[i][r] = ok ? o - b : 0.0f;
                w[i][r] = ok ? wv : 0.0f;
            }
    }
    const int blk_row = p.sb_on_t ? (t0 + wc * 32) : (s0 + wr * 64);
    const int sb = __builtin_amdgcn_readfirstlane(min(blk_row / p.sb_div, p.s_cs - 1));
    float* s1tab = res + per * 8;
    for (int i = lane; i < c_hi - c_lo; i += 64) s1tab[i * 8 + wid] = p.S1 ? p.S1[(c_lo + i) * p.s_cs + sb] : 1.0f;

    v16i acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;

    // fragment addresses (LDS byte offsets): stationary rows + k-tile offset, streaming rows + stage offset
    const int rs0 = wr * 64 + l31, rs1 = rs0 + 32, rt = wc * 32 + l31;
    const int ss0 = (rs0 >> 2) & 3, ss1 = (rs1 >> 2) & 3, stz = (rt >> 2) & 3;
    const int aS00 = rs0 * 64 + ((g ^ ss0) << 4), aS01 = rs0 * 64 + (((2 + g) ^ ss0) << 4);
    const int aS10 = rs1 * 64 + ((g ^ ss1) << 4), aS11 = rs1 * 64 + (((2 + g) ^ ss1) << 4);
    const int aT0 = ring0 + rt * 64 + ((g ^ stz) << 4), aT1 = ring0 + rt * 64 + (((2 + g) ^ stz) << 4);
    const int issue_base = ring0 + wid * 1024;

    struct Frag { v4i s00, s10, s01, s11, t0, t1; };
    Frag fa, fb;
    // wave-uniform loop state (SGPRs): byte offsets of the stage / stationary k-tile of the tile to READ next, the
    // stage to ISSUE into next, and the k-tile counter of the tile to COMPUTE next
    int rd_stage = 0, is_stage = (SW4_NS - 1) * SW2_TILE, rd_koff = 0;
    const int koff_end = ktiles * SW2_TILE;
    int kt = 0, cidx = 0;
    auto read_frags = [&](Frag& f) __attribute__((always_inline)) {
        f.s00 = *reinterpret_cast<const v4i*>(smem + rd_koff + aS00);
        f.s10 = *reinterpret_cast<const v4i*>(smem + rd_koff + aS10);
        f.t0 = *reinterpret_cast<const v4i*>(smem + rd_stage + aT0);
        f.s01 = *reinterpret_cast<const v4i*>(smem + rd_koff + aS01);
        f.s11 = *reinterpret_cast<const v4i*>(smem + rd_koff + aS11);
        f.t1 = *reinterpret_cast<const v4i*>(smem + rd_stage + aT1);
        rd_stage = (rd_stage + SW2_TILE == SW4_NS * SW2_TILE) ? 0 : rd_stage + SW2_TILE;
        rd_koff = (rd_koff + SW2_TILE == koff_end) ? 0 : rd_koff + SW2_TILE;
    };
    auto mma = [&](const Frag& f) __attribute__((always_inline)) {
        acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.s00, f.t0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.s10, f.t0, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.s01, f.t1, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.s11, f.t1, acc[1], 0, 0, 0);
        if (++kt == ktiles) {
            const float s1 = s1tab[cidx * 8 + wid];
            v2f sum2 = {0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const v2f a = {(float)acc[i][r], (float)acc[i][r + 1]};
                    const v2f uu = {u[i][r], u[i][r + 1]};
                    const v2f ww = {w[i][r], w[i][r + 1]};
                    const v2f d = uu - a * s1;
                    if (EPI == EPI_SQ_W) { const v2f t2 = ww * d; sum2 = t2 * t2 + sum2; }
                    else if (EPI == EPI_SQ) sum2 = d * d + sum2;
                    else if (EPI == EPI_ABS) sum2 += v2f{fabsf(d.x), fabsf(d.y)};
                    else sum2 = (ww * d) * d + sum2;
                    acc[i][r] = 0;
                    acc[i][r + 1] = 0;
                }
            float sum = sum2.x + sum2.y;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            if (lane == 0) res[cidx * 8 + wid] = sum;
            kt = 0;
            ++cidx;
        }
    };
    // step: tile `it` is in `cur`.  Prove tile it+1 landed (own piece waited for, then the barrier), issue tile
    // it+NS-1 into the stage of tile it-1 (everybody finished reading it one step ago), start the ds_reads of tile
    // it+1 into the idle fragment set and run the MFMAs of tile it.  No tail conditionals: the ring keeps issuing
    // (slack behind the plane) so exactly NS-3 younger pieces are outstanding at every wait.
    auto step = [&](Frag& cur, Frag& nxt) __attribute__((always_inline)) {
        wait_vmcnt<SW4_NS - 3>();
        __builtin_amdgcn_s_barrier();
        glds16(curT, smem + issue_base + is_stage);
        curT += SW_BKB;
        is_stage = (is_stage + SW2_TILE == SW4_NS * SW2_TILE) ? 0 : is_stage + SW2_TILE;
        read_frags(nxt);
        mma(cur);
    };
    // tile 0 (older loads -- the stationary operand -- complete first: vmcnt is in order)
    wait_vmcnt<SW4_NS - 2>();
    __builtin_amdgcn_s_barrier();
    read_frags(fa);
    int it = 0;
    for (; it + 1 < total; it += 2) { step(fa, fb); step(fb, fa); }
    if (it < total) step(fa, fb);
    __syncthreads();   // also drains the over-issued (never consumed) ring pieces before the LDS is released
    for (int i = tid; i < (c_hi - c_lo) * 8; i += 512) {
        const int cc = c_lo + i / 8, wv = i % 8;
        p.part[(long)cc * p.p_cs + (long)(st * 2 + (wv >> 2)) * p.NG + tt * 4 + (wv & 3)] = res[i];
    }
}
template <int EPI>
__global__ __launch_bounds__(512, 2) void k_sweep4(Sweep3Params p) { k_sweep4_body<EPI>(p, P4V_BIDX, P4V_GDIM); }
template <int EPI>
__global__ __launch_bounds__(512, 2) void k_sweep4_g(GroupArgs<Sweep3Params> a) { P4V_GROUP_ENTER(a); k_sweep4_body<EPI>(a.p[m_], vb_, vg_); }

// ------------------------------------------------------------------------------------------
// k_sweep5: k_sweep4 with TWO candidates per pass (candidate pair = one "pair-step" per k-tile)
// ------------------------------------------------------------------------------------------
// k_sweep4's inner loop is bound by LDS bandwidth, not by the matrix pipe: per k-tile every wave reads 4 KB of
// stationary and 2 KB of streaming fragments for 4 MFMAs (1.5 KB / MFMA; 8 waves -> 48 KB + 8 KB of LDS-DMA
// writes = 448 clk of the 128 B/clk LDS against 256 clk of MFMA; measured 458, tools/ubench_mfma.hip).  The
// candidate-invariant state -- stationary fragments and the raw_out / raw_grad registers -- can be shared by
// several candidates: here each wave keeps two accumulator sets and feeds both candidates of a pair from ONE read
// of the stationary fragments (1 KB / MFMA -> 320 clk per candidate-step).  The expanded plane is laid out
// [row][pair][k-tile][2][64 B] (k_pack c_inner = 2) so that the stream cursor still advances by a constant
// (128 B per pair-step).  Ring = 3 pair-stages of 2 x 8 KB; every step ends with lgkmcnt(0), so the stage whose
// fragments were read during step t-1 can be refilled right after the barrier of step t (two steps of L2 latency
// covered with three stages).  An odd candidate count runs one padding candidate whose score is dropped.
static constexpr int SW5_NP = 3;

template <int EPI>
__device__ __forceinline__ void k_sweep5_body(const Sweep3Params& p, const uint3 blockIdx, const uint3 gridDim) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef P4V_TRACE
    unsigned long long* trc = p.trace + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 16;
    if (threadIdx.x == 0) { trc[0] = __builtin_amdgcn_s_memrealtime(); trc[8] = __builtin_amdgcn_s_memtime(); trc[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); trc[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); }
#endif
    const int ktiles = p.ktiles;
    const int ring0 = ktiles * SW2_TILE;                 // LDS byte offset of the ring
    float* res = reinterpret_cast<float*>(smem + ring0 + SW5_NP * 2 * SW2_TILE);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int g = lane >> 5, l31 = lane & 31;

    const int nwg = p.stiles * p.ttiles;
    const int t = xcd_remap(blockIdx.x, nwg);
    const int st = t % p.stiles, tt = t / p.stiles;    // neighbours share the streaming tile
    const int s0 = st * 128, t0 = tt * 128;
    const int per = 2 * ((p.c1 - p.c0 + 2 * gridDim.z - 1) / (2 * gridDim.z));   // even: groups start on a pair
    int c_lo_ = p.c0 + blockIdx.z * per, c_hi_ = min(p.c1, c_lo_ + per);
    clip_crange(p.crange, c_lo_, c_hi_, true);
    const int c_lo = c_lo_, c_hi = c_hi_;
    if (c_lo >= c_hi) return;

    const int ld_row = wid * 16 + (lane >> 2);
    const int ld_chunk = (lane & 3) ^ ((ld_row >> 2) & 3);
    {
        const char* gS = (const char*)p.S + (long)(s0 + ld_row) * p.ldk + ld_chunk * 16;
        for (int kt = 0; kt < ktiles; ++kt) glds16(gS + kt * SW_BKB, smem + kt * SW2_TILE + wid * 1024);
    }
    // streaming operand [row][pair][k-tile][2][64 B]: two 16-row pieces per wave per pair-step, cursor += 128 B
    const char* curT = (const char*)p.T + (long)(t0 + ld_row) * p.t_rs + (long)(c_lo - p.c0) * p.ldk + ld_chunk * 16;
    const int total = ((c_hi - c_lo + 1) >> 1) * ktiles;
#pragma unroll
    for (int i = 0; i < SW5_NP; ++i) {
        glds16(curT, smem + ring0 + i * 2 * SW2_TILE + wid * 1024);
        glds16(curT + SW_BKB, smem + ring0 + i * 2 * SW2_TILE + SW2_TILE + wid * 1024);
        curT += 2 * SW_BKB;
    }

    // ---- candidate-invariant epilogue operands: 2 MFMA tiles of 32 x 32 (rows = stationary, cols = streaming) --
    float u[2][16], w[2][16];
    {
        const int tr = t0 + wc * 32 + l31;
        const long toff = (long)min(tr, p.TR - 1) * p.o_ts;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long idx = toff + (long)min(s0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g, p.SR - 1) * p.o_ss;
                u[i][r] = p.O[idx];
                w[i][r] = p.Wt[idx];
            }
        float bias_s[2][16];
        const float bias_t = p.bias[p.bias_on_t ? min(tr, p.TR - 1) : 0];
        if (!p.bias_on_t) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) bias_s[i][r] = p.bias[min(s0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g, p.SR - 1)];
        }
        const bool t_ok = tr < p.TR;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = t_ok && (s0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g) < p.SR;
                const float o = u[i][r], gw = w[i][r];
                const float b = p.bias_on_t ? bias_t : bias_s[i][r];
                float wv;
                if (p.wt_mode == 1) wv = gw; else if (p.wt_mode == 2) wv = o; else if (p.wt_mode == 3) wv = fabsf(o); else wv = 1.0f;
                u[i][r] = ok ? o - b : 0.0f;
                w[i][r] = ok ? wv : 0.0f;
            }
    }
    const int blk_row = p.sb_on_t ? (t0 + wc * 32) : (s0 + wr * 64);
    const int sb = __builtin_amdgcn_readfirstlane(min(blk_row / p.sb_div, p.s_cs - 1));
    float* s1tab = res + per * 8;
    for (int i = lane; i < per; i += 64) s1tab[i * 8 + wid] = (p.S1 && c_lo + i < c_hi) ? p.S1[(c_lo + i) * p.s_cs + sb] : 1.0f;

    v16i acc[2][2];   // [candidate of the pair][stationary 32-row half]
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][i][r] = 0;

    const int rs0 = wr * 64 + l31, rs1 = rs0 + 32, rt = wc * 32 + l31;
    const int ss0 = (rs0 >> 2) & 3, ss1 = (rs1 >> 2) & 3, stz = (rt >> 2) & 3;
    const int aS00 = rs0 * 64 + ((g ^ ss0) << 4), aS01 = rs0 * 64 + (((2 + g) ^ ss0) << 4);
    const int aS10 = rs1 * 64 + ((g ^ ss1) << 4), aS11 = rs1 * 64 + (((2 + g) ^ ss1) << 4);
    const int aT0 = ring0 + rt * 64 + ((g ^ stz) << 4), aT1 = ring0 + rt * 64 + (((2 + g) ^ stz) << 4);
    const int issue_base = ring0 + wid * 1024;

    struct Frag { v4i s00, s10, s01, s11, a0, a1, b0, b1; };
    Frag fa, fb;
    constexpr int PST = 2 * SW2_TILE;                    // bytes of one pair-stage
    int rd_stage = 0, is_stage = 0, rd_koff = 0;
    const int koff_end = ktiles * SW2_TILE;
    int kt = 0, cidx = 0;
    auto read_frags = [&](Frag& f) __attribute__((always_inline)) {
        f.s00 = *reinterpret_cast<const v4i*>(smem + rd_koff + aS00);
        f.s10 = *reinterpret_cast<const v4i*>(smem + rd_koff + aS10);
        f.a0 = *reinterpret_cast<const v4i*>(smem + rd_stage + aT0);
        f.b0 = *reinterpret_cast<const v4i*>(smem + rd_stage + SW2_TILE + aT0);
        f.s01 = *reinterpret_cast<const v4i*>(smem + rd_koff + aS01);
        f.s11 = *reinterpret_cast<const v4i*>(smem + rd_koff + aS11);
        f.a1 = *reinterpret_cast<const v4i*>(smem + rd_stage + aT1);
        f.b1 = *reinterpret_cast<const v4i*>(smem + rd_stage + SW2_TILE + aT1);
        rd_stage = (rd_stage + PST == SW5_NP * PST) ? 0 : rd_stage + PST;
        rd_koff = (rd_koff + SW2_TILE == koff_end) ? 0 : rd_koff + SW2_TILE;
    };
    auto epilogue = [&](v16i (&a2)[2], int ci) __attribute__((always_inline)) {
        const float s1 = s1tab[ci * 8 + wid];
        v2f sum2 = {0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const v2f a = {(float)a2[i][r], (float)a2[i][r + 1]};
                const v2f uu = {u[i][r], u[i][r + 1]};
                const v2f ww = {w[i][r], w[i][r + 1]};
                const v2f d = uu - a * s1;
                if (EPI == EPI_SQ_W) { const v2f t2 = ww * d; sum2 = t2 * t2 + sum2; }
                else if (EPI == EPI_SQ) sum2 = d * d + sum2;
                else if (EPI == EPI_ABS) sum2 += v2f{fabsf(d.x), fabsf(d.y)};
                else sum2 = (ww * d) * d + sum2;
                a2[i][r] = 0;
                a2[i][r + 1] = 0;
            }
        float sum = sum2.x + sum2.y;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (lane == 0) res[ci * 8 + wid] = sum;
    };
    auto mma = [&](const Frag& f) __attribute__((always_inline)) {
        acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.s00, f.a0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.s10, f.a0, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.s00, f.b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.s10, f.b0, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.s01, f.a1, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.s11, f.a1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.s01, f.b1, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.s11, f.b1, acc[1][1], 0, 0, 0);
        // issue order: the 8 fragment reads of the NEXT pair-step go out in the shadow of the first MFMAs
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 x ds_read
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 x MFMA
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        if (++kt == ktiles) {
            epilogue(acc[0], cidx);
            epilogue(acc[1], cidx + 1);
            kt = 0;
            cidx += 2;
        }
    };
    // step: pair-step `it` is in `cur`.  Own pieces of pair-step it+1 waited for (only the 2 pieces of it+2 may be
    // in flight), barrier, refill the stage of pair-step `it` (its fragment reads completed before every wave's
    // previous lgkmcnt(0)) with pair-step it+3, start the ds_reads of it+1 and run the 8 MFMAs of `it`.
    auto step = [&](Frag& cur, Frag& nxt) __attribute__((always_inline)) {
        wait_vmcnt<2>();
        __builtin_amdgcn_s_barrier();
        glds16(curT, smem + issue_base + is_stage);
        glds16(curT + SW_BKB, smem + issue_base + is_stage + SW2_TILE);
        curT += 2 * SW_BKB;
        is_stage = (is_stage + PST == SW5_NP * PST) ? 0 : is_stage + PST;
        read_frags(nxt);
        mma(cur);
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0); a real S_WAITCNT so that the compiler's own wait insertion sees it
    };
#ifdef P4V_TRACE
    if (threadIdx.x == 0) trc[1] = __builtin_amdgcn_s_memrealtime();
#endif
    wait_vmcnt<4>();      // stationary operand + pair-step 0 (vmcnt completes in order)
    __builtin_amdgcn_s_barrier();
#ifdef P4V_TRACE
    if (threadIdx.x == 0) trc[6] = __builtin_amdgcn_s_memrealtime();
#endif
    read_frags(fa);
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0); a real S_WAITCNT so that the compiler's own wait insertion sees it
    int it = 0;
    for (; it + 1 < total; it += 2) { step(fa, fb); step(fb, fa); }
    if (it < total) step(fa, fb);
#ifdef P4V_TRACE
    if (threadIdx.x == 0) trc[2] = __builtin_amdgcn_s_memrealtime();
#endif
    __syncthreads();   // also drains the over-issued (never consumed) ring pieces before the LDS is released
    for (int i = tid; i < (c_hi - c_lo) * 8; i += 512) {
        const int cc = c_lo + i / 8, wv = i % 8;
        p.part[(long)cc * p.p_cs + (long)(st * 2 + (wv >> 2)) * p.NG + tt * 4 + (wv & 3)] = res[i];
    }
#ifdef P4V_TRACE
    if (threadIdx.x == 0) { trc[3] = __builtin_amdgcn_s_memrealtime(); trc[7] = (unsigned long long)total; trc[9] = __builtin_amdgcn_s_memtime(); }
#endif
}
template <int EPI>
__global__ __launch_bounds__(512, 2) void k_sweep5(Sweep3Params p) { k_sweep5_body<EPI>(p, P4V_BIDX, P4V_GDIM); }
template <int EPI>
__global__ __launch_bounds__(512, 2) void k_sweep5_g(GroupArgs<Sweep3Params> a) { P4V_GROUP_ENTER(a); k_sweep5_body<EPI>(a.p[m_], vb_, vg_); }

// ------------------------------------------------------------------------------------------
// k_sweep6: stationary operand in REGISTERS, one wave per SIMD (K = KT * 64 bytes, KT <= 12)
// ------------------------------------------------------------------------------------------
// What bounds k_sweep4/5 (tools/ubench_mfma.hip, profiles/r1_ubench.txt): not the matrix pipe but the LDS --
// 6 (4) ds_read_b128 per 4 MFMAs plus the LDS-DMA writes of the streamed tile, which interfere badly with the
// reads (MFMA only 133 ns per k-tile step, + reads 162 ns, + LDS-DMA 231 ns, + epilogue 261 ns = the kernel).
// The cure is fewer LDS bytes per MFMA on BOTH paths:
//   * 4 waves per workgroup, one per SIMD, 512 registers each.  Every wave keeps its 64 stationary rows for the
//     WHOLE K in registers (KT * 16 VGPRs = 192 at K = 768) -- the stationary operand never touches the LDS;
//   * workgroup tile = 256 stationary rows x 64 streaming rows: per k-tile step a wave reads the 64 x 64 B
//     streaming tile (4 x ds_read_b128) for 8 MFMAs (0.5 KB / MFMA instead of 1.5), and the tile that has to be
//     streamed in is 4 KB per step instead of 8 KB (half the LDS-DMA and half the L2 traffic per MAC);
//   * the ring holds whole candidates (KT x 4 KB per stage, 3 stages): ONE barrier and one counted vmcnt wait per
//     candidate instead of per k-tile; the epilogue of candidate c-1 runs after the barrier of candidate c, under
//     the latency of its first fragment reads.
// Output tile orientation, scales, bias, partial-sum table and plane layout ([row][candidate][K], c_inner = 1)
// are those of k_sweep4, so k_pack / k_finish are unchanged.
// RB = 32-row blocks of the stationary operand per wave: RB = 2 -> 4 waves (one per SIMD, 512 registers each);
// RB = 1 -> 8 waves (two per SIMD, 256 registers each): half the rows per wave, so one wave's epilogue VALU work and
// LDS waits hide under the other wave's MFMAs, at twice the fragment reads per MFMA.
// Epilogue operands of k_sweep6 in fragment order, written once per (module, search orientation) and read by every pass
// of that orientation (raw_out, raw_grad and the bias do not change during calibration_step2).  For tile t = tt * stiles + st,
// 32-row block b (of the tile's 256 stationary rows), column block cb, quarter q and lane (g, l31): the four values of
// stationary rows st * 256 + b * 32 + 8 q + 4 g + 0..3 at streaming row tt * 64 + cb * 32 + l31 -- chunk
// ((((t * 8 + b) * 2 + cb) * 4 + q) * 2 + k) * 64 + lane of 16 bytes, k = 0: raw_out - bias, k = 1: the metric weight
// (raw_grad | raw_out | |raw_out| | 1), zero where either row is padding.  The sweep's prologue is then 64 dwordx4 loads
// of 1 KB contiguous per wave; gathered in place (k_sweep6 until round 2) every load touched 32-64 cache lines and the
// prologue took 10 % of the launch (profiles/r2_sweep6_ablation.txt).
struct PrepEpi6Params {
    const float* O; const float* Wt; const float* bias;
    long o_ss, o_ts; int SR, TR, bias_on_t, wt_mode;   // wt_mode 4 (cosine): k = 0 holds raw_out itself, k = 1 the bias of the element
    int stiles, ttiles;
    float* E;
    int transposed;     // cosine weight search (EPI_COS_T): a lane's four values are four STREAMING rows of one stationary row
};
__device__ __forceinline__ void k_prep_epi6_body(const PrepEpi6Params& p, const uint3 blockIdx, const uint3 gridDim) {
    const long total = (long)p.stiles * p.ttiles * 8 * 2 * 4 * 2 * 64;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int lane = (int)(i & 63), k = (int)((i >> 6) & 1), q = (int)((i >> 7) & 3), cb = (int)((i >> 9) & 1), b = (int)((i >> 10) & 7);
        const long t = i >> 13;
        const int st = (int)(t % p.stiles), tt = (int)(t / p.stiles);
        const int g = lane >> 5, l31 = lane & 31;
        // plain: stationary rows st*256 + b*32 + 8q + 4g + e at streaming row tt*64 + cb*32 + l31;
        // transposed: streaming rows tt*64 + cb*32 + 8q + 4g + e at stationary row st*256 + b*32 + l31
        const int tr0 = tt * 64 + cb * 32 + (p.transposed ? 8 * q + 4 * g : l31);
        const int sr0 = st * 256 + b * 32 + (p.transposed ? l31 : 8 * q + 4 * g);
        v4f v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int sr = sr0 + (p.transposed ? 0 : e), tr = tr0 + (p.transposed ? e : 0);
            if (sr < p.SR && tr < p.TR) {
                const long idx = (long)sr * p.o_ss + (long)tr * p.o_ts;
                const float o = p.O[idx];
                const float bs = p.bias[p.bias_on_t ? tr : sr];
                if (k == 0) v[e] = p.wt_mode == 4 ? o : o - bs;
                else v[e] = p.wt_mode == 1 ? p.Wt[idx] : p.wt_mode == 2 ? o : p.wt_mode == 3 ? fabsf(o) : p.wt_mode == 4 ? bs : 1.0f;
            }
        }
        reinterpret_cast<v4f*>(p.E)[i] = v;
    }
}
__global__ __launch_bounds__(256) void k_prep_epi6(PrepEpi6Params p) { k_prep_epi6_body(p, P4V_BIDX, P4V_GDIM); }
__global__ __launch_bounds__(256) void k_prep_epi6_g(GroupArgs<PrepEpi6Params> a) { P4V_GROUP_ENTER(a); k_prep_epi6_body(a.p[m_], vb_, vg_); }

// Timing-only ablations (tools/build_ablation_libs.sh, never shipped): -DP4V_SW6_DBG = 1 no operand stream in the loop,
// 2 no MFMAs, 4 no epilogue, 8 no fragment reads, 16 no epilogue-operand loads in the prologue, 32 no stationary-operand
// loads (bit mask).
#ifndef P4V_SW6_DBG
#define P4V_SW6_DBG 0
#endif
template <int EPI, int KT, int RB>
__device__ __forceinline__ void k_sweep6_body(const Sweep3Params& p, const uint3 blockIdx, const uint3 gridDim) {
    constexpr int NW = 8 / RB;                           // waves per workgroup
    // Cosine (round 6): a sample's dot(raw, sim) and |sim|^2 are sums over FEATURES, so the samples must sit on the lanes of the
    // MFMA output (columns) and the features in a lane's 16 registers.  Activation search (EPI_COS): the stationary operand is
    // the weights -- rows = features, as in every other epilogue.  Weight search (EPI_COS_T): the stationary operand is the
    // samples, so the two MFMA operands swap places (both are 16 bytes of K per lane: D' = D^T) and a lane's registers run over
    // the 64 streaming features of the tile.  Either way a wave writes (dot, |sim|^2, |raw|^2) per (candidate, 64-feature slab,
    // sample) straight to k_finish_cos's table [slab][sample][3] (p.NG = padded samples): there is no LDS left for 100
    // candidates x 256 samples, and the stores ride behind the ring barrier, a whole candidate before the next counted wait.
    constexpr bool COS = EPI == EPI_COS || EPI == EPI_COS_T, TR = EPI == EPI_COS_T;
    static_assert(!COS || RB == 2, "cosine epilogue: one wave per SIMD only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef P4V_TRACE
    unsigned long long* trc = p.trace + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 16;
    if (threadIdx.x == 0) { trc[0] = __builtin_amdgcn_s_memrealtime(); trc[8] = __builtin_amdgcn_s_memtime(); trc[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); }
#endif
    constexpr int KT_TILE = 64 * 64;                     // bytes of one k-tile of the 64-row streaming tile
    constexpr int STG = KT * KT_TILE;                    // one candidate
    float* res = reinterpret_cast<float*>(smem + 3 * STG);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31;

    const int nwg = p.ntile > 0 ? p.ntile : p.stiles * p.ttiles;
    const int t = p.tile0 + xcd_remap(blockIdx.x, nwg);
    const int st = t % p.stiles, tt = t / p.stiles;    // neighbours share the streaming tile
    const int s0 = st * 256 + wid * (32 * RB), t0 = tt * 64;
    const int per = (p.c1 - p.c0 + gridDim.z - 1) / gridDim.z;
    int c_lo_ = p.c0 + blockIdx.z * per, c_hi_ = min(p.c1, c_lo_ + per);
    clip_crange(p.crange, c_lo_, c_hi_);
    if (p.crange_blk) clip_crange_blk(p.crange_blk, min((p.sb_on_t ? tt * 64 : st * 256) / p.sb_div, p.s_cs - 1), c_lo_, c_hi_);
    const int c_lo = c_lo_, c_hi = c_hi_;
    if (c_lo >= c_hi) return;
    const int ncand = c_hi - c_lo;

    // ---- streaming operand [row][candidate][K]: wave w moves rows w*16 .. w*16+15 of every k-tile ----------------
    // a candidate's tile = 4 * KT pieces of 1 KB (16 rows x 64 B); wave w moves pieces w, w + NW, ...: piece q is
    // row group q % 4 of k-tile q / 4
    constexpr int NPC = 4 * KT;                          // pieces per candidate
    constexpr int PPW = (NPC + NW - 1) / NW;             // pieces per wave
    const int ld_row = (wid & 3) * 16 + (lane >> 2);
    const int ld_chunk = (lane & 3) ^ ((ld_row >> 2) & 3);
    const char* curT = (const char*)p.T + (long)(t0 + ld_row) * p.t_rs + (long)(c_lo - p.c0) * p.ldk + ld_chunk * 16;
    const int kt_w = wid >> 2;                           // first k-tile of this wave's pieces (NW = 8: 0 or 1)
    curT += kt_w * SW_BKB;                               // (NW = 8: the odd waves start one k-tile in)
    auto piece = [&](const char* src, int stage_off, auto j_c) __attribute__((always_inline)) {
        // j-th piece of this wave: k-tile j * (NW / 4) + kt_w; the compile-time part of the k offset rides in the
        // instruction (it is added to the LDS address as well, hence the "- OFF" on the destination)
        constexpr int j = decltype(j_c)::value;
        constexpr int OFF = j * (NW / 4) * SW_BKB;
        const int kt = j * (NW / 4) + kt_w;
        if (NPC % NW == 0 || NW * j + wid < NPC)
            glds16_imm<OFF>(src, smem + stage_off + kt * KT_TILE + (wid & 3) * 1024 - OFF);
    };
    auto issue = [&](int stage_off) __attribute__((always_inline)) {
        [&]<int... J>(std::integer_sequence<int, J...>) __attribute__((always_inline)) {
            (piece(curT, stage_off, std::integral_constant<int, J>{}), ...);
        }(std::make_integer_sequence<int, PPW>{});
        curT += p.ldk;
    };
    issue(0);
    issue(STG);      // always two candidates ahead (slack behind the plane; stale stages are never consumed)

    // ---- stationary operand: 64 rows x K bytes of this wave, MFMA A-fragments, registers for the whole kernel -----
    v4i sfr[KT][RB][2];   // [k-tile][32-row block][32-byte half]
    {
        // fragment order (k_pack c_inner == 3): every load of a wave is 1 KB contiguous
        const v4i* gS = reinterpret_cast<const v4i*>(p.S) + (long)(s0 >> 6) * (KT * 4 * 64) + lane;
        const int ib = (s0 >> 5) & 1;                        // RB == 1: the wave's single 32-row block within its 64-row slab
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if constexpr ((P4V_SW6_DBG & 32) != 0) sfr[kt][i][h] = v4i{lane, kt, i, h};     // ablation: no stationary loads
                    else sfr[kt][i][h] = gS[((kt * 2 + (RB == 2 ? i : ib)) * 2 + h) * 64];
                }
        // cosine instances: the fragments are pinned to the ACCUMULATION file (the MFMAs read them there).  Left alone the
        // allocator spreads them over both files and shuffles them inside the candidate loop (112 v_accvgpr_mov + 3 scratch
        // reloads per candidate at KT = 12; the difference-metric instances do not show it and stay as they were tuned).
        if constexpr (COS && KT * RB * 2 * 4 > 128) {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) asm volatile("" : "+a"(sfr[kt][i][h]));
        }
    }

    // ---- candidate-invariant epilogue operands: 2 x 2 MFMA tiles of 32 x 32 (rows = stationary, cols = streaming), in
    // fragment order (k_prep_epi6: bias, padding and the choice of the metric weight are folded in) ---------------------
    float u[RB][2][16], w[RB][2][16];
    if constexpr (COS) {
        // cosine: u = raw_out itself, w = the bias of the element's FEATURE -- the same for both column blocks (plain: features
        // on the rows) or both 32-row blocks (transposed: features on the columns), so only w[.][0][.] / w[0][.][.] is loaded and
        // kept: the 256 architectural registers of the wave are full (64 accumulators, the fragment ring, 64 + 32 here)
        const v4f* gE = reinterpret_cast<const v4f*>(p.E) + ((long)t * 8 + wid * RB) * (2 * 4 * 2 * 64) + lane;
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4f u4 = gE[(((i * 2 + cb) * 4 + q) * 2 + 0) * 64];
#pragma unroll
                    for (int e = 0; e < 4; ++e) u[i][cb][q * 4 + e] = u4[e];
                    if ((TR && i == 0) || (!TR && cb == 0)) {
                        const v4f w4 = gE[(((i * 2 + cb) * 4 + q) * 2 + 1) * 64];
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[i][cb][q * 4 + e] = w4[e];
                    }
                }
    } else if (p.E) {
        const v4f* gE = reinterpret_cast<const v4f*>(p.E) + ((long)t * 8 + wid * RB) * (2 * 4 * 2 * 64) + lane;
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v4f u4, w4;
                    if constexpr ((P4V_SW6_DBG & 16) != 0) { u4 = v4f{(float)lane, 1.f, 2.f, 3.f}; w4 = v4f{1.f, 1.f, 1.f, 1.f}; }
                    else {
                        u4 = gE[(((i * 2 + cb) * 4 + q) * 2 + 0) * 64];
                        w4 = gE[(((i * 2 + cb) * 4 + q) * 2 + 1) * 64];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) { u[i][cb][q * 4 + e] = u4[e]; w[i][cb][q * 4 + e] = w4[e]; }
                }
    } else {
        // in place (weight search: the streaming rows are the output features, the contiguous dimension of raw_out, so the 32
        // lanes of a half wave read 128 contiguous bytes per load): every load issued unconditionally at clamped addresses,
        // then branch-free masking / weight selection
        const unsigned m_g = p.wt_mode == 1 ? 0xffffffffu : 0u;
        const unsigned m_o = p.wt_mode == 2 ? 0xffffffffu : p.wt_mode == 3 ? 0x7fffffffu : 0u;
        const unsigned m_1 = p.wt_mode == 0 ? 0x3f800000u : 0u;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int tr = t0 + cb * 32 + l31;
            const long toff = (long)min(tr, p.TR - 1) * p.o_ts;
            const float bias_t = p.bias[p.bias_on_t ? min(tr, p.TR - 1) : 0];
            const bool t_ok = tr < p.TR;
            float bs[RB][16];
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int src = min(s0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g, p.SR - 1);
                    const long idx = toff + (long)src * p.o_ss;
                    u[i][cb][r] = p.O[idx];
                    w[i][cb][r] = p.Wt[idx];
                    bs[i][r] = p.bias[p.bias_on_t ? 0 : src];
                }
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = t_ok && (s0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g) < p.SR;
                    const float o = u[i][cb][r], gw = w[i][cb][r];
                    const float b = p.bias_on_t ? bias_t : bs[i][r];
                    const unsigned wbits = (__builtin_bit_cast(unsigned, gw) & m_g) | (__builtin_bit_cast(unsigned, o) & m_o) | m_1;
                    u[i][cb][r] = ok ? o - b : 0.0f;
                    w[i][cb][r] = ok ? __builtin_bit_cast(float, wbits) : 0.0f;
                }
        }
    }
    const int blk_row = p.sb_on_t ? t0 : s0;
    const int sb = __builtin_amdgcn_readfirstlane(min(blk_row / p.sb_div, p.s_cs - 1));
    // LDS behind the ring: res [(per + 1) candidates][8] (slot 0 is a dump for the warm-up epilogue), s1tab [per][4],
    // dump [64] (target of the lanes that do not hold the wave sum)
    float* s1tab = res + (per + 1) * (2 * NW);
    for (int i = lane; i < ncand; i += 64) s1tab[i * NW + wid] = p.S1 ? p.S1[(c_lo + i) * p.s_cs + sb] : 1.0f;

    // cosine: |raw|^2 of a lane's sample over the features this wave sees of it (plain: the wave's 64 stationary features, per
    // column block; transposed: the tile's 64 streaming features, per 32-row block), in the order of the per-candidate sums
    float oo_fix[2] = {0.0f, 0.0f};
    auto half_sum = [](float x) __attribute__((always_inline)) {
        // x(lane) + x(lane ^ 32): v_permlane32_swap exchanges the upper half of its first operand with the lower half of its second
        float y = x;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
        return x + y;
    };
    if constexpr (COS) {
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
            float t2 = 0.0f;
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float v = TR ? u[sg][o][r] : u[o][sg][r]; t2 = fmaf(v, v, t2); }
            oo_fix[sg] = half_sum(t2);
        }
    }
    float cd[2] = {0.0f, 0.0f}, cn[2] = {0.0f, 0.0f};    // cosine: running dot / |sim|^2 of the epilogue in flight
    float sd[2] = {0.0f, 0.0f}, sn[2] = {0.0f, 0.0f};    // ... finished, waiting for their store slot behind the ring barrier
    const int Sp = p.NG;
    // table rows of this lane: plain -- slab = the wave's 64 features, samples t0 + cb * 32 + l31; transposed -- slab = the
    // streaming tile, samples s0 + i * 32 + l31
    // (one candidate's table is well below 4 GB: wave-uniform 64-bit base + per-lane 32-bit byte offset)
    unsigned cos_row = 0;
    if constexpr (COS) cos_row = (unsigned)((TR ? tt * Sp + s0 + l31 : (st * 4 + wid) * Sp + t0 + l31) * 12);
    // Branch-free: both half waves hold the same sums behind half_sum and store them to the same address (a branch here -- or on
    // "is there a previous candidate" below -- splits the basic block, and the pure arithmetic of the epilogue slices in front of
    // it sinks across the split, out from between the MFMAs).
    auto cos_store = [&](int cand, int sgrp, float d, float n2) __attribute__((always_inline)) {
        float* q = reinterpret_cast<float*>(reinterpret_cast<char*>(p.part + (long)cand * p.p_cs) + (cos_row + sgrp * 384));
        q[0] = d; q[1] = n2; q[2] = oo_fix[sgrp];
    };

    // ---- main loop ----------------------------------------------------------------------------------------------
    // One candidate = 2 phases (column block cb = 0, then 1) of KT steps; a step = 2 fragment reads + 4 MFMAs.  With a
    // single wave per SIMD nothing hides behind another wave, so everything is interleaved by construction:
    //   * fragment reads run two steps ahead of their MFMAs (three fragment buffers, counted lgkmcnt(4) waits;
    //     inline-asm ds_reads, because the compiler's own wait insertion falls back to lgkmcnt(0) whenever an
    //     LDS-DMA is pending) -- also across the candidate boundary, which therefore has no bubble;
    //   * the epilogue of a column block (cvt, fma, DPP wave sum, one LDS store) is cut into KT slices that ride in
    //     the VALU slots between the MFMAs of the OTHER column block: block 0 of candidate c during phase 1 of c,
    //     block 1 during phase 0 of c+1.  Element granularity, scalar ops: KT-1 slices of ceil(32 / (KT-1)) elements
    //     and one slice for the reduction, so that every MFMA gap carries about the same 3..5 filler instructions
    //     (one wave per SIMD hides at most ~5 single-issue instructions per 32x32x32 MFMA; packed f32 VALU beside
    //     MFMAs costs more than two scalar ops -- MI355X_MICROARCH.md, instruction table);
    //   * the ring barrier sits in the middle of phase 1: every wave's pieces of candidate c+1 have landed (issued
    //     one candidate earlier), and the stage of candidate c-1 is refilled with c+2.
    const int sw0 = (l31 >> 2) & 3;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const unsigned tbase0 = lds0 + l31 * 64 + ((g ^ sw0) << 4);          // half 0; column block 1 is +2048
    const unsigned tbase1 = lds0 + l31 * 64 + (((2 + g) ^ sw0) << 4);    // half 1
    const unsigned s1addr0 = lds0 + 3 * STG + (per + 1) * (8 * NW) + wid * 4;   // &s1tab[0 * NW + wid]
    float* dump = s1tab + per * NW;
#define P4V_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    struct TF2 { v4i f[2]; };                 // the two 32-byte halves of one column block of one k-tile
    constexpr int PD = ((2 * KT) % 3 == 0) ? 2 : 3;   // fragment reads run PD steps ahead of their MFMAs (2, 3, 5 measured the same)
    constexpr int NB = PD + 1;                // fragment buffers, ring indexed by (step % NB)
    TF2 tf[NB];
    static_assert((2 * KT) % NB == 0 && KT >= 3, "fragment ring needs 2 * KT divisible by the buffer count");
    constexpr int NSTEP = 2 * KT;
    // Step before which the ring barrier of a candidate sits: in the middle of phase 1, but never later than the first
    // read of the NEXT candidate's stage (step NSTEP - PD) -- those fragments are only guaranteed to have landed, and
    // to be visible to every wave, behind this candidate's wait + barrier.  (KT = 4: PD = 3 -> step 5, not 6: with the
    // barrier at 6 the read at step 5 raced with the other waves' pieces -- practically always landed, never ordered.)
    constexpr int SBAR = (KT + KT / 2 < NSTEP - PD) ? KT + KT / 2 : NSTEP - PD;
    static_assert(SBAR >= KT && SBAR <= NSTEP - PD, "ring barrier must sit in phase 1, before the next candidate's first read");
    constexpr int NEL = RB * 16;                             // accumulator elements per lane and column block
    constexpr int NSL = KT - 1;                              // slices that carry element math (the last one reduces)
    constexpr int EPS = (NEL + NSL - 1) / NSL;               // elements per slice
    v16i acc[RB][2];
    const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < RB; ++i) acc[i][1] = zero16;        // consumed by the warm-up epilogue of candidate "-1"
    float esum0 = 0.0f, esum1 = 0.0f;                        // even / odd accumulator registers (fixed order)
    float ered = 0.0f, es1 = 1.0f, es1_next = 1.0f;
    // epilogue slice `sl` of column block cbE; result goes to res slot `slot` (= candidate + 1)
    auto epi_slice = [&](auto sl_c, auto cb_c, int slot) __attribute__((always_inline)) {
        constexpr int sl = decltype(sl_c)::value, cbE = decltype(cb_c)::value;
        if constexpr ((P4V_SW6_DBG & 4) != 0) {     // keep the accumulators (and with them the MFMAs) alive
            if constexpr (sl == 0) {
#pragma unroll
                for (int i = 0; i < RB; ++i) asm volatile("" : "+v"(acc[i][cbE]));
            }
            return;
        }
        if constexpr (COS) {
            // plain: block cbE holds samples cbE*32 + l31, summed over both 32-feature blocks i; transposed: block cbE holds
            // features cbE*32 + .., sample group = i, summed over both column blocks (block 0 first, then block 1)
            if constexpr (sl == 0 && (!TR || cbE == 0)) { cd[0] = cd[1] = cn[0] = cn[1] = 0.0f; }
            if constexpr (sl < NSL) {
#pragma unroll
                for (int e = sl * EPS; e < (sl + 1) * EPS && e < NEL; ++e) {
                    const int i = e >> 4, r = e & 15;
                    const int sgi = TR ? i : 0;
                    const float o = fmaf((float)acc[i][cbE][r], es1, TR ? w[0][cbE][r] : w[i][0][r]);
                    cd[sgi] = fmaf(u[i][cbE][r], o, cd[sgi]);
                    cn[sgi] = fmaf(o, o, cn[sgi]);
                }
                // (pins the slice between the MFMAs of its step: nothing else orders pure arithmetic against the asm statements)
                if constexpr (TR) asm volatile("" : "+v"(cd[0]), "+v"(cd[1]), "+v"(cn[0]), "+v"(cn[1]));
                else asm volatile("" : "+v"(cd[0]), "+v"(cn[0]));
            } else if constexpr (!TR) {
                const float d = half_sum(cd[0]), n2 = half_sum(cn[0]);
                if constexpr (cbE == 1) { sd[0] = d; sn[0] = n2; }      // block 1 of candidate slot-1: stored with the next block 0
                else {
                    // behind the ring barrier of this candidate: block 0 of this candidate, block 1 of the previous one
                    // (first candidate: the warm-up epilogue's sums go to the candidate's own block-1 entry, which the real ones
                    // overwrite one candidate later -- same lane, same address, program order)
                    cos_store(c_lo + slot - 1, 0, d, n2);
                    cos_store(c_lo + max(slot - 2, 0), 1, sd[0], sn[0]);
                }
            } else {
                if constexpr (cbE == 1) {                                // candidate slot-1 complete
                    sd[0] = half_sum(cd[0]); sn[0] = half_sum(cn[0]); sd[1] = half_sum(cd[1]); sn[1] = half_sum(cn[1]);
                } else {                                                 // the previous candidate, behind this one's ring barrier
                    // (first candidate: warm-up sums into its own entries, overwritten one candidate later)
                    cos_store(c_lo + max(slot - 2, 0), 0, sd[0], sn[0]);
                    cos_store(c_lo + max(slot - 2, 0), 1, sd[1], sn[1]);
                }
            }
            return;
        }
        if constexpr (sl == 0) { esum0 = 0.0f; esum1 = 0.0f; }
        if constexpr (sl < NSL) {
#pragma unroll
            for (int e = sl * EPS; e < (sl + 1) * EPS && e < NEL; ++e) {
                const int i = e >> 4, r = e & 15;
                const float a = (float)acc[i][cbE][r];
                const float d = u[i][cbE][r] - a * es1;
                const float ww = w[i][cbE][r];
                float& es = (r & 1) ? esum1 : esum0;
                if (EPI == EPI_SQ_W) { const float t2 = ww * d; es = t2 * t2 + es; }
                else if (EPI == EPI_SQ) es = d * d + es;
                else if (EPI == EPI_ABS) es += fabsf(d);
                else es = (ww * d) * d + es;
            }
        } else {
            ered = wave_sum_dpp(esum0 + esum1);
            float* dst = (lane == 63) ? res + slot * (2 * NW) + wid * 2 + cbE : dump + lane;   // branch-free: every lane stores
            *dst = ered;
        }
    };
    const char* fillT = curT - p.ldk;                     // candidate being streamed in (warm-up: candidate 1 again)
    int fill_stage = STG;
    // one step: prefetch the fragments of step s+2, wait for those of step s, 4 MFMAs, one epilogue slice
    auto step = [&](auto s_c, unsigned ad0, unsigned ad1, unsigned adn0, unsigned adn1, int ci) __attribute__((always_inline)) {
        constexpr int s = decltype(s_c)::value;
        constexpr int cb = s / KT, kt = s % KT;
        TF2& cur = tf[s % NB];
        TF2& pre = tf[(s + PD) % NB];
        constexpr int t = s + PD;
        if constexpr ((P4V_SW6_DBG & 8) != 0) {
        } else if constexpr (t < NSTEP) {
            P4V_DSR(pre.f[0], ad0, (t % KT) * KT_TILE + (t / KT) * 2048); P4V_DSR(pre.f[1], ad1, (t % KT) * KT_TILE + (t / KT) * 2048);
        } else {   // first PD steps of the next candidate (its stage landed before the barrier of the previous phase 1)
            P4V_DSR(pre.f[0], adn0, (t - NSTEP) * KT_TILE); P4V_DSR(pre.f[1], adn1, (t - NSTEP) * KT_TILE);
        }
        if constexpr ((P4V_SW6_DBG & 8) != 0) {
            if constexpr (s == 0) { asm volatile("ds_read_b32 %0, %1" : "=v"(es1_next) : "v"(s1addr0 + ci * (4 * NW))); __builtin_amdgcn_s_waitcnt(0xC07F); }
        } else if constexpr (s == 0) {   // scale of this candidate for the epilogues that start in phase 1
            asm volatile("ds_read_b32 %0, %1" : "=v"(es1_next) : "v"(s1addr0 + ci * (4 * NW)));
            __builtin_amdgcn_s_waitcnt(0xC07F | ((2 * PD + 1) << 8));   // the scale read is newer than the fragments of this step
        } else {
            __builtin_amdgcn_s_waitcnt(0xC07F | ((2 * PD) << 8));       // lgkmcnt(2 * PD): the fragments of steps s+1 .. s+PD stay in flight
        }
        // the MFMAs below may not be hoisted above the reads / the wait: their fragments pass through this fence
        asm volatile("" : "+v"(cur.f[0]), "+v"(cur.f[1]) :: "memory");
        if constexpr (s == KT) { asm volatile("" : "+v"(es1_next)); es1 = es1_next; }   // landed KT waits ago
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                if constexpr ((P4V_SW6_DBG & 2) == 0) {
                    if constexpr (TR) acc[i][cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.f[h], sfr[kt][i][h], (kt == 0 && h == 0) ? zero16 : acc[i][cb], 0, 0, 0);
                    else acc[i][cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(sfr[kt][i][h], cur.f[h], (kt == 0 && h == 0) ? zero16 : acc[i][cb], 0, 0, 0);
                }
                if (h == 1 && i == 0) {
                    // the streamed tile of candidate ci+2 trickles in one 1 KB piece per wave and step (a burst right after
                    // the barrier stalls the fragment reads).  Issue point: after the third MFMA of the step, fenced so that
                    // only VALU / SALU work may cross -- left to the scheduler the LDS-DMA lands in the gap right behind the
                    // fragment reads, its most expensive place (measured: -2 % per launch; after the second MFMA: no change)
                    constexpr int P1 = NSTEP - SBAR;                                // steps left in this candidate after the barrier
                    constexpr int j = (s >= SBAR) ? s - SBAR : s + P1;              // piece index of this wave
                    __builtin_amdgcn_sched_barrier(0x6);
                    if constexpr (j < PPW && (P4V_SW6_DBG & 1) == 0) piece(fillT, fill_stage, std::integral_constant<int, j>{});
                    __builtin_amdgcn_sched_barrier(0x6);
                }
            }
        // phase 0 carries the epilogue of block 1 of the previous candidate (slot ci), phase 1 that of block 0 of this one
        if constexpr (cb == 0) epi_slice(std::integral_constant<int, kt>{}, std::integral_constant<int, 1>{}, ci);
        else epi_slice(std::integral_constant<int, kt>{}, std::integral_constant<int, 0>{}, ci + 1);
        __builtin_amdgcn_sched_barrier(0);
    };

    // both prologue candidates (and the register operands) have landed; make them visible to every wave
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    int stage = 0;                                        // byte offset of the stage holding the current candidate
    auto pro_read = [&](auto s_c) __attribute__((always_inline)) {
        constexpr int S = decltype(s_c)::value;
        TF2& d = tf[S];
        const unsigned b0 = tbase0, b1 = tbase1;
        P4V_DSR(d.f[0], b0, (S % KT) * KT_TILE + (S / KT) * 2048);
        P4V_DSR(d.f[1], b1, (S % KT) * KT_TILE + (S / KT) * 2048);
    };
    [&]<int... S>(std::integer_sequence<int, S...>) __attribute__((always_inline)) {
        (pro_read(std::integral_constant<int, S>{}), ...);
    }(std::make_integer_sequence<int, PD>{});
#ifdef P4V_TRACE
    if (threadIdx.x == 0) { trc[1] = __builtin_amdgcn_s_memrealtime(); trc[6] = trc[1]; }
#endif
    for (int ci = 0; ci < ncand; ++ci) {
        const int stage_n = (stage + STG == 3 * STG) ? 0 : stage + STG;
        const unsigned ad0 = tbase0 + stage, ad1 = tbase1 + stage, adn0 = tbase0 + stage_n, adn1 = tbase1 + stage_n;
        auto run = [&](auto lo_c, auto hi_c) __attribute__((always_inline)) {
            // compile-time loop over steps [lo, hi)
            [&]<int... S>(std::integer_sequence<int, S...>) __attribute__((always_inline)) {
                (step(std::integral_constant<int, decltype(lo_c)::value + S>{}, ad0, ad1, adn0, adn1, ci), ...);
            }(std::make_integer_sequence<int, decltype(hi_c)::value - decltype(lo_c)::value>{});
        };
        constexpr int SB = SBAR;                          // the ring barrier sits in the middle of phase 1
        run(std::integral_constant<int, 0>{}, std::integral_constant<int, SB>{});
        wait_vmcnt<0>();                                  // own pieces of candidate ci+1 (issued one candidate ago)
        __builtin_amdgcn_s_barrier();                     // ci+1 visible to all; nobody reads the stage of ci-1 any more
        fill_stage = (stage_n + STG == 3 * STG) ? 0 : stage_n + STG;   // candidate ci+2 -> stage of ci-1, piece by piece
        fillT = curT;
        curT += p.ldk;
        run(std::integral_constant<int, SB>{}, std::integral_constant<int, NSTEP>{});
        stage = stage_n;
    }
    // block 1 of the last candidate
    [&]<int... S>(std::integer_sequence<int, S...>) __attribute__((always_inline)) {
        (epi_slice(std::integral_constant<int, S>{}, std::integral_constant<int, 1>{}, ncand), ...);
    }(std::make_integer_sequence<int, KT>{});
#ifdef P4V_TRACE
    if (threadIdx.x == 0) trc[2] = __builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (COS) {
        // the last candidate: plain -- its block 1 (block 0 went out behind its ring barrier); transposed -- both sample groups
        if constexpr (!TR) cos_store(c_lo + ncand - 1, 1, sd[0], sn[0]);
        else { cos_store(c_lo + ncand - 1, 0, sd[0], sn[0]); cos_store(c_lo + ncand - 1, 1, sd[1], sn[1]); }
    }
    __syncthreads();   // also drains the over-issued (never consumed) ring pieces before the LDS is released
    if constexpr (COS) return;
    // part[c][64-row slab][32-column group]; with RB = 1 two waves share a slab: fixed-order sum of their results
    for (int i = tid; i < ncand * 8; i += 64 * NW) {
        const int ci = i / 8, wv = (i % 8) >> 1, cb = i & 1;
        const float* r = res + (ci + 1) * (2 * NW);
        const float v = (RB == 2) ? r[wv * 2 + cb] : r[(2 * wv) * 2 + cb] + r[(2 * wv + 1) * 2 + cb];
        p.part[(long)(c_lo + ci) * p.p_cs + (long)(st * 4 + wv) * p.NG + tt * 2 + cb] = v;
    }
#undef P4V_DSR
#ifdef P4V_TRACE
    if (threadIdx.x == 0) { trc[3] = __builtin_amdgcn_s_memrealtime(); trc[7] = (unsigned long long)ncand; trc[9] = __builtin_amdgcn_s_memtime(); }
#endif
}
template <int EPI, int KT, int RB>
__global__ __launch_bounds__(512 / RB, RB == 2 ? 1 : 2) void k_sweep6(Sweep3Params p) { k_sweep6_body<EPI, KT, RB>(p, P4V_BIDX, P4V_GDIM); }
template <int EPI, int KT, int RB>
__global__ __launch_bounds__(512 / RB, RB == 2 ? 1 : 2) void k_sweep6_g(GroupArgs<Sweep3Params> a) { P4V_GROUP_ENTER(a); k_sweep6_body<EPI, KT, RB>(a.p[m_], vb_, vg_); }

// ------------------------------------------------------------------------------------------
// k_sweep7: int8 candidate sweep for LARGE K (K >= 1024: fc2 of every ViT, every Linear of ViT-L / Swin stage 4)
// ------------------------------------------------------------------------------------------
// At K = 3072 neither operand can be stationary (k_sweep6 holds K <= 768 in registers), so both stream through the LDS.
// What bounded k_sweep2 / k_sweep2g there (round-1 PMC, profiles/r1_pmc_fc2_v7.txt: matrix pipe 39 % busy, 47 % of the
// wave cycles in s_waitcnt / s_barrier; round-2 ablations, profiles/r2_sweep7_ablation.txt): with 128 x 128 tiles a wave
// has 4 MFMAs (128 clk) per k-tile against 2-3 LDS-DMA issues (60-180 clk EACH, MI355X_MICROARCH.md), 6 fragment reads and
// a barrier, and the two waves of a SIMD run these phases in lock step -- matrix time, DMA issue time and epilogue time
// ADD UP instead of overlapping.  This kernel:
//   * workgroup tile 256 features x 256 samples (twin: 256 x 128 samples x 2 planes), 8 waves as 2 x 4, wave tile
//     128 x 64 (twin 128 x 32 x 2 planes): 16 MFMAs (512 clk) per wave and k-tile for 4 LDS-DMA pieces and 12 fragment
//     reads; half the L2 -> LDS bytes per MAC of k_sweep2 (1 B / 128 MACs);
//   * PING-PONG: the two waves of a SIMD (waves w and w + 4 = the two feature halves of the tile) alternate roles with two
//     barriers per k-tile -- while group A issues the 16 MFMAs of k-tile t from registers, group B issues its LDS-DMA pieces
//     for tile t + 3 and reads its 12 fragments of tile t; then they swap.  Every instruction that is not an MFMA runs in
//     the shadow of the partner's MFMAs (the "compute segment / load segment" pairing of MI355X_MICROARCH.md, "Two waves
//     per SIMD"); 4-stage LDS ring, counted vmcnt, raw s_barrier;
//   * the epilogue operands of a candidate -- raw_out - bias and the metric weight, 128 + 128 values per lane, far more
//     than fit in registers next to 128 accumulators -- are PRE-PACKED once per pass in fragment order (k_prep_epi: what
//     k_pack is for the MFMA operands), so that every dwordx4 load of the epilogue is 1 KB contiguous per wave (read in
//     place from [sample][feature] memory the same load touches 32 cache lines: 26 % of the kernel, measured) and bias,
//     validity masks and the weight choice of the metric are gone from the inner loop.  The loads are inline asm with
//     counted vmcnt (hipcc drains vmcnt(0) for every ordinary load while an LDS-DMA is in flight) and run two sub-blocks
//     ahead of the arithmetic through a ring of three register sets;
//   * per candidate ONE float per wave and 32-feature block, kept in LDS, written once at the end (k_finish unchanged).
// Workgroup order: feature tile fastest, then sample tile, candidate group slowest -- the workgroups that share a tile
// of the candidate-expanded operand are neighbours on one XCD and pull it through that L2 once.
struct Sweep7Params {
    const int* crange;                  // optional device-side candidate range (clip_crange)
    const void* R;  long r_cs;          // feature-side plane(s) [C or 1][Np][ldk] (weights): rows -> MFMA rows
    const void* Cp; long c_cs;          // sample-side plane(s)  [C or 1][Mp][ldk] (activations): rows -> MFMA columns
    const void* C2;                     // twin: second sample-side plane (never candidate-expanded)
    int ldk, ktiles;                    // ktiles % 4 == 0
    const float* S1; const float* S2;   // [candidate][s_cs] scale of plane 1 / 2
    int s_cs, sb_div;                   // scale block of feature n = min(n / sb_div, s_cs - 1)
    const float* E;                     // pre-packed epilogue operands (k_prep_epi)
    int c0, c1;
    float* part; long p_cs; int NG;     // part[c * p_cs + (ct * 4 + wc) * NG + rt * 8 + wr * 4 + j]
    int rtiles, ctiles, cgroups;
    int order;                          // workgroup order on the 1-D grid (see the kernel)
};

// timing-only ablation builds (never in production): -DP4V_SW7_DBG=1 no operand stream, 2 no MFMA, 4 no epilogue (bits add)
#ifndef P4V_SW7_DBG
#define P4V_SW7_DBG 0
#endif
// P4V_SW7_FILL: LDS-DMA pieces of a k-tile that a wave issues between its own MFMAs (the rest in its load phase)
#ifndef P4V_SW7_FILL
#define P4V_SW7_FILL 4
#endif
static constexpr int SW7_NS = 4;
static constexpr int SW7_REGION = 256 * 64;       // one operand side of a k-tile: 256 rows x 64 B
static constexpr int SW7_STAGE = 2 * SW7_REGION;

// Epilogue operands in fragment order.  For workgroup tile (ct, rt), wave w = wr * 4 + wc, sub-block sb and k = 0..3:
// chunk ((tile * 8 + w) * NSB + sb) * 4 + k holds, for each of the 64 lanes, the four values of accumulator rows
// rq * 4 .. rq * 4 + 3 (rq = 2 h2 + (k >> 1)) of block (j, q): k even = raw_out - bias, k odd = the metric weight
// (raw_grad | raw_out | |raw_out| | 1); zero where the feature or the sample is padding.  NSB = 8 (twin) / 16 sub-blocks:
// sb = (j * NQ + q) * 2 + h2.
struct PrepEpiParams {
    const float* O; const float* G; const float* bias;
    long ldo; int M, N, wt_mode;
    int rtiles, ctiles, twin;
    float* E;
};
__device__ __forceinline__ void k_prep_epi_body(const PrepEpiParams& p, const uint3 blockIdx, const uint3 gridDim) {
    const int NQ = p.twin ? 1 : 2, NSB = 8 * NQ;
    const long total = (long)p.rtiles * p.ctiles * 8 * NSB * 4 * 64;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int lane = (int)(i & 63), k = (int)((i >> 6) & 3);
        long rest = i >> 8;
        const int sb = (int)(rest % NSB); rest /= NSB;
        const int w = (int)(rest & 7); rest >>= 3;
        const int rt = (int)(rest % p.rtiles), ct = (int)(rest / p.rtiles);
        const int h2 = sb & 1, q = (sb >> 1) % NQ, j = (sb >> 1) / NQ;
        const int wr = w >> 2, wc = w & 3, g = lane >> 5, l31 = lane & 31;
        const int rq = 2 * h2 + (k >> 1);
        const int n = rt * 256 + wr * 128 + j * 32 + 8 * rq + 4 * g;
        const int m = ct * (p.twin ? 128 : 256) + (p.twin ? wc * 32 : wc * 64 + q * 32) + l31;
        v4f v = {0.f, 0.f, 0.f, 0.f};
        if (n < p.N && m < p.M) {                        // N % 4 == 0: the four features are valid together
            const v4f o = *reinterpret_cast<const v4f*>(p.O + (long)m * p.ldo + n);
            if (!(k & 1)) {
                v = o;
                if (p.bias && p.wt_mode != 4) { const v4f b = *reinterpret_cast<const v4f*>(p.bias + n); v = o - b; }
            } else if (p.wt_mode == 4) {       // cosine: raw_out itself above, the bias of the simulated output here
                if (p.bias) v = *reinterpret_cast<const v4f*>(p.bias + n);
            } else if (p.wt_mode == 1) v = *reinterpret_cast<const v4f*>(p.G + (long)m * p.ldo + n);
            else if (p.wt_mode == 2) v = o;
            else if (p.wt_mode == 3) v = v4f{fabsf(o[0]), fabsf(o[1]), fabsf(o[2]), fabsf(o[3])};
            else v = v4f{1.f, 1.f, 1.f, 1.f};
        }
        reinterpret_cast<v4f*>(p.E)[i] = v;
    }
}
__global__ __launch_bounds__(256) void k_prep_epi(PrepEpiParams p) { k_prep_epi_body(p, P4V_BIDX, P4V_GDIM); }
__global__ __launch_bounds__(256) void k_prep_epi_g(GroupArgs<PrepEpiParams> a) { P4V_GROUP_ENTER(a); k_prep_epi_body(a.p[m_], vb_, vg_); }

// TW = 0: one sample-side plane.  TW = 1: twin, two planes streamed side by side (128 samples x 2 planes per tile).
// TW = 2: twin whose two ranges have DISJOINT supports (post-GELU, linear.py:605-606), streamed as ONE merged int8 plane
// k_pos + k_neg (PACK_TWIN_I8) and split into its two MFMA fragments in registers -- per byte max(v, 0) / min(v, 0), six
// VALU operations per dword in the wave's load phase: a quarter less LDS-DMA (three 1 KB pieces per wave and k-tile instead
// of four) and 10 instead of 12 fragment reads per k-tile; the MFMAs, accumulators and epilogue are those of TW = 1.
template <int TW, int EPI>
__device__ __forceinline__ void k_sweep7_body(const Sweep7Params& p, const uint3 blockIdx, const uint3 gridDim) {
    constexpr bool TWIN = TW != 0, MERGED = TW == 2;
    constexpr int PPT = MERGED ? 3 : 4;                  // LDS-DMA pieces per wave and k-tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* res = reinterpret_cast<float*>(smem + SW7_NS * SW7_STAGE);   // [per][8 waves][4 feature blocks]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;               // wave grid: 2 (features; = ping-pong group) x 4 (samples)
    const int g = lane >> 5, l31 = lane & 31;

    // 1-D grid of rtiles x ctiles x cgroups workgroups; block b runs on XCD b % 8 and xcd_remap hands every XCD a
    // contiguous run of the order below, so the ~32 workgroups resident on an XCD at any time are neighbours in it:
    //   order 0: feature tile fastest, then sample tile, candidate group slowest
    //   order 1: feature tile fastest, then candidate group, sample tile slowest -- the workgroups of one sample tile
    //            (all feature tiles x all candidate groups) run together: they share the fixed operand's tiles, the epilogue
    //            operands and, per candidate group, the expanded tile of the sample side
    //   order 2: candidate group fastest, then feature tile, then sample tile
    //   order 3: feature tile, then 4 sample tiles, then candidate group, then groups of 4 sample tiles
    const int nwg = p.rtiles * p.ctiles * p.cgroups;
    const int t = xcd_remap(blockIdx.x, nwg);
    int rt, ct, cg;
    if (p.order == 1) { rt = t % p.rtiles; cg = (t / p.rtiles) % p.cgroups; ct = t / (p.rtiles * p.cgroups); }
    else if (p.order == 2) { cg = t % p.cgroups; rt = (t / p.cgroups) % p.rtiles; ct = t / (p.rtiles * p.cgroups); }
    else if (p.order == 3) {
        const int grp = 4 * p.rtiles * p.cgroups, g0 = (t / grp) * 4, gsz = min(4, p.ctiles - g0), tt = t % grp;
        rt = tt % p.rtiles; ct = g0 + (tt / p.rtiles) % gsz; cg = tt / (p.rtiles * gsz);
    } else { rt = t % p.rtiles; ct = (t / p.rtiles) % p.ctiles; cg = t / (p.rtiles * p.ctiles); }
    constexpr int CM = TWIN ? 128 : 256;                 // samples per workgroup tile
    const int r0 = rt * 256, m0 = ct * CM;
    const int per = (p.c1 - p.c0 + p.cgroups - 1) / p.cgroups;
    int c_lo_ = p.c0 + cg * per, c_hi_ = min(p.c1, c_lo_ + per);
    clip_crange(p.crange, c_lo_, c_hi_);
    const int c_lo = c_lo_, c_hi = c_hi_;
    if (c_lo >= c_hi) return;
    const int ncand = c_hi - c_lo;

    // ---- tables behind the ring: scales per (candidate, 32-feature block) ---------------------------------------------
    float* s1tab = res + per * 32;
    float* s2tab = s1tab + per * 8;
    for (int i = tid; i < ncand * 8; i += 512) {
        const int cc = c_lo + (i >> 3);
        const int sb = min((r0 + (i & 7) * 32) / p.sb_div, p.s_cs - 1);
        s1tab[i] = p.S1 ? p.S1[cc * p.s_cs + sb] : 1.0f;
        if (TWIN) s2tab[i] = p.S2 ? p.S2[cc * p.s_cs + sb] : 1.0f;
    }
    __syncthreads();                                     // (no LDS-DMA in flight yet: a plain barrier)

    // ---- LDS-DMA: wave w fills rows [32 w, 32 w + 32) of both regions, two 16-row pieces each ------------------------
    const int ld_row = lane >> 2;
    const int ld_chunk = (lane & 3) ^ ((ld_row >> 2) & 3);          // logical 16-B chunk landing in physical slot lane & 3
    const unsigned voff0 = (unsigned)(ld_row * p.ldk + ld_chunk * 16);
    const unsigned voff1 = voff0 + 16u * (unsigned)p.ldk;
    // (timing-only ablation 8: every workgroup streams the bytes of tile (0, 0), candidate group 0 -- what is left of the
    // stream's cost when it always hits in the L2)
    const int r0a = (P4V_SW7_DBG & 8) ? 0 : r0, m0a = (P4V_SW7_DBG & 8) ? 0 : m0, c_loa = (P4V_SW7_DBG & 8) ? p.c0 : c_lo;
    const char* curR = (const char*)p.R + (long)(r0a + wid * 32) * p.ldk + (long)c_loa * p.r_cs;
    const char* cbase = (TW == 1 && wid >= 4) ? (const char*)p.C2 : (const char*)p.Cp;
    // (merged twin: the 128 sample rows of the tile are one plane; wave w fills rows [16 w, 16 w + 16) with ONE piece)
    const char* curC = cbase + (long)(m0a + (MERGED ? wid * 16 : TWIN ? (wid & 3) * 32 : wid * 32)) * p.ldk + (TWIN ? 0L : (long)c_loa * p.c_cs);
    const int ktiles = p.ktiles;
    const long wrapR = p.r_cs - (long)ktiles * SW_BKB, wrapC = (TWIN ? 0L : p.c_cs) - (long)ktiles * SW_BKB;
    const int total = ncand * ktiles;
    int ikt = 0;
    const int lds_w = wid * 2048;
    // the four pieces of a tile: k = 0, 1 feature side, k = 2, 3 sample side (the cursors advance behind the last one)
    auto piece = [&](auto stage_c, auto k_c) __attribute__((always_inline)) {
        constexpr int ST = decltype(stage_c)::value, k = decltype(k_c)::value;
        if constexpr (MERGED && k == 3) return;          // three pieces per tile
        char* s = smem + ST * SW7_STAGE + ((MERGED && k == 2) ? wid * 1024 : lds_w) + (k >> 1) * SW7_REGION + (k & 1) * 1024;
        if constexpr (!(P4V_SW7_DBG & 1)) glds16((k >> 1 ? curC : curR) + ((k & 1) ? voff1 : voff0), s);
        if constexpr (k == PPT - 1) {
            curR += SW_BKB; curC += SW_BKB;
            if (++ikt == ktiles) { ikt = 0; curR += wrapR; curC += wrapC; }
        }
    };
    auto issue = [&](auto stage_c) __attribute__((always_inline)) {
        piece(stage_c, std::integral_constant<int, 0>{}); piece(stage_c, std::integral_constant<int, 1>{});
        piece(stage_c, std::integral_constant<int, 2>{}); piece(stage_c, std::integral_constant<int, 3>{});
    };

    v16i acc[4][2];                                      // [32-feature block j][sample block q (twin: plane q)]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][q][r] = 0;

    // ---- fragment addresses: row R, logical chunk c -> physical chunk c ^ ((R >> 2) & 3); stages 2, 3 lie beyond the
    // 16-bit offset field of ds_read, hence a second set of bases -----------------------------------------------------
    const int sw = (l31 >> 2) & 3;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const unsigned aR0 = lds0 + (wr * 128 + l31) * 64 + ((g ^ sw) << 4), aR1 = lds0 + (wr * 128 + l31) * 64 + (((2 + g) ^ sw) << 4);
    const unsigned crow = (TWIN ? wc * 32 : wc * 64) + l31;
    const unsigned aC0 = lds0 + SW7_REGION + crow * 64 + ((g ^ sw) << 4), aC1 = lds0 + SW7_REGION + crow * 64 + (((2 + g) ^ sw) << 4);
    const unsigned aR0h = aR0 + 2 * SW7_STAGE, aR1h = aR1 + 2 * SW7_STAGE, aC0h = aC0 + 2 * SW7_STAGE, aC1h = aC1 + 2 * SW7_STAGE;
    constexpr int QOFF = TW == 1 ? 128 * 64 : 32 * 64;   // second sample block: the other plane / the next 32 samples

#define P4V_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    struct Fr { v4i r[2][4], c[2][2]; };                 // [k-half][block]: the 12 fragments of one k-tile
    Fr f;
    auto read_tile_ = [&](Fr& f, auto stage_c) __attribute__((always_inline)) {
        constexpr int ST = decltype(stage_c)::value;
        constexpr int SO = (ST & 1) * SW7_STAGE;
        const unsigned bR0 = (ST >> 1) ? aR0h : aR0, bR1 = (ST >> 1) ? aR1h : aR1;
        const unsigned bC0 = (ST >> 1) ? aC0h : aC0, bC1 = (ST >> 1) ? aC1h : aC1;
        P4V_DSR(f.c[0][0], bC0, SO); P4V_DSR(f.r[0][0], bR0, SO); P4V_DSR(f.r[0][1], bR0, SO + 2048);
        if constexpr (!MERGED) P4V_DSR(f.c[0][1], bC0, SO + QOFF);
        P4V_DSR(f.r[0][2], bR0, SO + 4096); P4V_DSR(f.r[0][3], bR0, SO + 6144);
        P4V_DSR(f.c[1][0], bC1, SO); P4V_DSR(f.r[1][0], bR1, SO); P4V_DSR(f.r[1][1], bR1, SO + 2048);
        if constexpr (!MERGED) P4V_DSR(f.c[1][1], bC1, SO + QOFF);
        P4V_DSR(f.r[1][2], bR1, SO + 4096); P4V_DSR(f.r[1][3], bR1, SO + 6144);
    };
    auto read_tile = [&](auto stage_c) __attribute__((always_inline)) { read_tile_(f, stage_c); };
    // all 12 fragments have landed, and no MFMA below is scheduled above this point
    auto frags_ready = [&]() __attribute__((always_inline)) {
        if constexpr (MERGED) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.r[0][0]), "+v"(f.r[0][1]), "+v"(f.r[0][2]), "+v"(f.r[0][3]), "+v"(f.c[0][0]),
                                                   "+v"(f.r[1][0]), "+v"(f.r[1][1]), "+v"(f.r[1][2]), "+v"(f.r[1][3]), "+v"(f.c[1][0]) :: "memory");
            // split the merged plane: byte-wise negative part n = v & mask (mask = 0xFF where the sign bit is set), positive
            // part v ^ n; plane 0 = positive range, plane 1 = negative range (the order of S1 / S2 and of acc[.][0 / 1])
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned v = (unsigned)f.c[h][0][e];
                    const unsigned sg = v & 0x80808080u;
                    const unsigned mask = (sg - (sg >> 7)) | sg;
                    const unsigned ng = v & mask;
                    f.c[h][1][e] = (int)ng;
                    f.c[h][0][e] = (int)(v ^ ng);
                }
            // (keep the split in this load phase: nothing else orders it against the barrier in front of the MFMAs)
            asm volatile("" : "+v"(f.c[0][0]), "+v"(f.c[0][1]), "+v"(f.c[1][0]), "+v"(f.c[1][1]));
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.r[0][0]), "+v"(f.r[0][1]), "+v"(f.r[0][2]), "+v"(f.r[0][3]), "+v"(f.c[0][0]), "+v"(f.c[0][1]),
                                                   "+v"(f.r[1][0]), "+v"(f.r[1][1]), "+v"(f.r[1][2]), "+v"(f.r[1][3]), "+v"(f.c[1][0]), "+v"(f.c[1][1]) :: "memory");
        }
    };
    // The 16 MFMAs of a k-tile.  `fill` (compile-time stage, or none): pieces k0, k0 + 1 of the tile being streamed in are
    // issued between them -- an LDS-DMA issue costs a wave ~60 clk next to bare MFMAs but 100-185 clk in a phase that also
    // carries fragment reads (MI355X_MICROARCH.md), so two of a wave's four pieces ride here and only two in its load phase.
    auto compute = [&](bool fill, auto stage_c, auto k0_c) __attribute__((always_inline)) {
        constexpr int k0 = decltype(k0_c)::value;
        if constexpr (P4V_SW7_DBG & 2) { if (fill) { piece(stage_c, std::integral_constant<int, k0>{}); piece(stage_c, std::integral_constant<int, k0 + 1>{}); } return; }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    acc[j][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.r[h][j], f.c[h][q], acc[j][q], 0, 0, 0);
                    if constexpr (P4V_SW7_FILL == 4) {
                        if (j % 2 == 0 && q == 1) {      // behind the 2nd / 6th / 10th / 14th MFMA: all four pieces ride here
                            __builtin_amdgcn_sched_barrier(0x6);
                            if (fill) {
                                if (h == 0 && j == 0) piece(stage_c, std::integral_constant<int, 0>{});
                                else if (h == 0) piece(stage_c, std::integral_constant<int, 1>{});
                                else if (j == 0) piece(stage_c, std::integral_constant<int, 2>{});
                                else piece(stage_c, std::integral_constant<int, 3>{});
                            }
                            __builtin_amdgcn_sched_barrier(0x6);
                        }
                    } else if (j == 1 && q == 1) {       // behind the 4th / 12th MFMA
                        __builtin_amdgcn_sched_barrier(0x6);
                        if (fill) { if (h == 0) piece(stage_c, std::integral_constant<int, k0>{}); else piece(stage_c, std::integral_constant<int, k0 + 1>{}); }
                        __builtin_amdgcn_sched_barrier(0x6);
                    }
                }
        }
        // MFMAs are register-only instructions: nothing but their operands orders them against barriers and asm statements,
        // and left alone the instruction selector sinks them below the partner's phases (the ping-pong collapses into
        // "three load phases, then three compute phases").  An empty volatile asm that consumes the accumulators keeps the
        // 16 MFMAs of a k-tile between the fragment wait above and the barrier below.
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]),
                          "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]));
    };

    // ---- epilogue of one candidate ------------------------------------------------------------------------------------
    // A sub-block = 8 of a lane's 16 elements of one 32 x 32 block (row quads rq = 2 h2, 2 h2 + 1): 2 + 2 dwordx4 loads of
    // 1 KB per wave each (fragment order, see k_prep_epi).  Loads run two sub-blocks ahead of the arithmetic through a ring
    // of three register sets; wave-uniform base (SGPR pair) + per-lane 32-bit offset: no vector address arithmetic.
    constexpr int NQ = TWIN ? 1 : 2;
    constexpr int NSB = 8 * NQ;
    constexpr bool COS = EPI == EPI_COS;
    static_assert(!COS || TW == 0, "cosine epilogue: one sample-side plane");
    constexpr bool NEEDW = (EPI == EPI_SQ_W || EPI == EPI_W_SQ || COS);
    struct Hb { v4f u[2], w[2]; };
    const unsigned e_voff = (unsigned)lane * 16u;
    const float* e_wave = p.E + ((long)(ct * p.rtiles + rt) * 8 + wid) * (NSB * 4 * 256);   // 256 floats per 1 KB chunk
#define P4V_GLD(dst, voff, sbase, off) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(off) : "memory")
    auto load_sb = [&](Hb& b, auto sb_c) __attribute__((always_inline)) {
        constexpr int sb = decltype(sb_c)::value;
        const float* se = e_wave + sb * 1024;
        P4V_GLD(b.u[0], e_voff, se, 0);
        if (NEEDW) P4V_GLD(b.w[0], e_voff, se, 1024);
        P4V_GLD(b.u[1], e_voff, se, 2048);
        if (NEEDW) P4V_GLD(b.w[1], e_voff, se, 3072);
    };
    constexpr int LPS = NEEDW ? 4 : 2;                   // loads per sub-block
    auto epilogue = [&](int ci) __attribute__((always_inline)) {
        const v4f s1v = *reinterpret_cast<const v4f*>(s1tab + ci * 8 + wr * 4);
        v4f s2v = {0.f, 0.f, 0.f, 0.f};
        if (TWIN) s2v = *reinterpret_cast<const v4f*>(s2tab + ci * 8 + wr * 4);
        float sumj[4] = {0.f, 0.f, 0.f, 0.f};
        // cosine (round 6): the MFMA columns are samples, so a lane owns ONE sample per sample block q and sums dot(raw, sim),
        // |sim|^2, |raw|^2 over the wave's 128 features (j-major, the order of the sub-blocks); the two half waves are added and
        // the triple goes straight to k_finish_cos's table [128-feature slab][sample][3] (p.NG = padded samples)
        float dq[2] = {0.f, 0.f}, nq[2] = {0.f, 0.f}, oq[2] = {0.f, 0.f};
        // ring of RD register sets, loads RD - 1 sub-blocks ahead of the arithmetic: the 48 fragment registers are dead during
        // the epilogue, so six sets (96 VGPRs) fit next to the 128 accumulators; the operands come from L2 / Infinity Cache
        // at ~1-2 us per access and only the depth of this ring hides that
        constexpr int RD = 3;
        Hb hb[RD];
        [&]<int... SB>(std::integer_sequence<int, SB...>) __attribute__((always_inline)) {
            (load_sb(hb[SB], std::integral_constant<int, SB>{}), ...);
        }(std::make_integer_sequence<int, (RD - 1 < NSB ? RD - 1 : NSB)>{});
        auto do_sb = [&](auto sb_c) __attribute__((always_inline)) {
            constexpr int sb = decltype(sb_c)::value;
            constexpr int j = sb / (2 * NQ), q = (sb / 2) % NQ, h2 = sb % 2;
            Hb& cur = hb[sb % RD];
            if constexpr (sb + RD - 1 < NSB) load_sb(hb[(sb + RD - 1) % RD], std::integral_constant<int, sb + RD - 1>{});
            constexpr int ahead = (NSB - 1 - sb < RD - 1) ? NSB - 1 - sb : RD - 1;             // sub-blocks loaded after this one
            constexpr int younger = ahead * LPS;
            if constexpr (NEEDW) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(cur.u[0]), "+v"(cur.u[1]), "+v"(cur.w[0]), "+v"(cur.w[1]) : "n"(younger) : "memory");
            else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(cur.u[0]), "+v"(cur.u[1]) : "n"(younger) : "memory");
            const float s1 = s1v[j], s2 = s2v[j];
            if constexpr (COS) {
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = (2 * h2 + rr) * 4 + e;
                        const float uu = cur.u[rr][e];
                        const float o = fmaf((float)acc[j][q][r], s1, cur.w[rr][e]);
                        dq[q] = fmaf(uu, o, dq[q]);
                        nq[q] = fmaf(o, o, nq[q]);
                        oq[q] = fmaf(uu, uu, oq[q]);
                    }
                asm volatile("" : "+v"(dq[q]), "+v"(nq[q]), "+v"(oq[q]));     // (pinned here, as `sum` below)
                __builtin_amdgcn_sched_barrier(0);
                return;
            }
            float sum = sumj[j];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = (2 * h2 + rr) * 4 + e;
                    float d = cur.u[rr][e] - (float)acc[j][TWIN ? 0 : q][r] * s1;
                    if (TWIN) d -= (float)acc[j][1][r] * s2;
                    if (EPI == EPI_SQ_W) { const float tt = cur.w[rr][e] * d; sum = fmaf(tt, tt, sum); }
                    else if (EPI == EPI_SQ) sum = fmaf(d, d, sum);
                    else if (EPI == EPI_ABS) sum += fabsf(d);
                    else sum = fmaf(cur.w[rr][e] * d, d, sum);
                }
            // pin the arithmetic of this sub-block HERE: nothing orders it against the asm statements, so the instruction
            // selector sinks it towards its only use at the end of the epilogue -- every loaded value of the candidate live
            // at once, 800 B of scratch per lane.  An empty volatile asm that consumes `sum` is chained to the loads.
            asm volatile("" : "+v"(sum));
            sumj[j] = sum;
            __builtin_amdgcn_sched_barrier(0);
        };
        [&]<int... SB>(std::integer_sequence<int, SB...>) __attribute__((always_inline)) {
            (do_sb(std::integral_constant<int, SB>{}), ...);
        }(std::make_integer_sequence<int, NSB>{});
        if constexpr (COS) {
            auto half_sum = [](float x) __attribute__((always_inline)) {     // x(lane) + x(lane ^ 32), as in k_sweep6
                float y = x;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
                return x + y;
            };
            // both half waves store the same triple to the same address (no exec-mask branch in the k-tile pipeline)
            char* row = reinterpret_cast<char*>(p.part + (long)(c_lo + ci) * p.p_cs) +
                        (unsigned)(((rt * 2 + wr) * p.NG + m0 + wc * 64 + l31) * 12);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float* dst = reinterpret_cast<float*>(row + q * 384);
                dst[0] = half_sum(dq[q]); dst[1] = half_sum(nq[q]); dst[2] = half_sum(oq[q]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = COS ? 0.0f : wave_sum_dpp(sumj[j]);       // fixed order: deterministic
            if (!COS && lane == 63) res[(ci * 8 + wid) * 4 + j] = s;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][q][r] = 0;
        }
    };

    // ---- main loop: flat over (candidate, k-tile); tile `it` sits in stage it % 4 = k-tile % 4 (ktiles % 4 == 0) ----------
    // Per k-tile two barriers B1, B2 and two phases.  Group A (waves 0-3): MFMAs of tile `it` | B1 | DMA of tile it+3, fragments
    // of tile it+1 | B2.  Group B (waves 4-7): DMA of tile it+3, fragments of tile `it` | B1 | MFMAs of tile `it` | B2.
    //   landed:  every wave waits for its own pieces of tile it+1 before B1(it), so behind B1(it) tile it+1 is in the LDS for
    //            everybody (A reads it in the second phase of `it`, B in the first phase of it+1);
    //   free:    tile it+3 goes into the stage of tile it-1, whose last fragment reads (A: second phase of it-2, B: first
    //            phase of it-1) were waited for before B1(it-1).
    // The younger pieces in flight at the landed-wait are tile it+2 (A) and tiles it+2, it+3 (B has just issued it+3).
    const int npre = min(SW7_NS - 1, total);
    if (npre > 0) issue(std::integral_constant<int, 0>{});
    if (npre > 1) issue(std::integral_constant<int, 1>{});
    if (npre > 2) issue(std::integral_constant<int, 2>{});
    if (total > 2) wait_vmcnt<2 * PPT>(); else if (total > 1) wait_vmcnt<PPT>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    // Tile it+3 is streamed in during k-tile `it`, two pieces under each group's MFMAs and two in its load phase (A: feature
    // side while computing, then sample side; B: feature side in its load phase, sample side while computing).  At the
    // landed-wait before B1(it) the younger pieces of a wave are therefore tile it+2 (4) and the first two of tile it+3.
    int it = 0;
    auto wait_landed = [&](int it_) __attribute__((always_inline)) {      // own pieces of tile it_+1
        if constexpr (P4V_SW7_FILL == 4) {
            // group A has issued all of tile it+3 (under the MFMAs of this k-tile), group B none of it yet
            if (wr == 0) { if (it_ + 3 < total) wait_vmcnt<2 * PPT>(); else if (it_ + 2 < total) wait_vmcnt<PPT>(); else wait_vmcnt<0>(); }
            else { if (it_ + 2 < total) wait_vmcnt<PPT>(); else wait_vmcnt<0>(); }
        } else {
            if (it_ + 3 < total) wait_vmcnt<PPT + 2>(); else if (it_ + 2 < total) wait_vmcnt<PPT>(); else wait_vmcnt<0>();
        }
    };
    if (wr == 0) {
        read_tile(std::integral_constant<int, 0>{});
        // k-tile of group A: MFMAs (+ 2 pieces) | B1 | fragments of the next tile, 2 pieces | B2
        auto first = [&](int it_, auto stage_c) __attribute__((always_inline)) {
            constexpr int ST = decltype(stage_c)::value;
            compute(it_ + 3 < total, std::integral_constant<int, (ST + 3) % SW7_NS>{}, std::integral_constant<int, 0>{});
        };
        auto second = [&](int it_, auto stage_c, int epi_ci = -1) __attribute__((always_inline)) {
            constexpr int ST = decltype(stage_c)::value;
            wait_landed(it_);
            __builtin_amdgcn_s_barrier();                                                     // B1
            // a candidate's epilogue runs HERE, in the phase in which group B computes the candidate's last tile and then
            // runs its own epilogue: the two groups' epilogues (memory-latency bound) overlap instead of following each other
            if constexpr (!(P4V_SW7_DBG & 4)) if (epi_ci >= 0) epilogue(epi_ci);
            // fragment reads FIRST: their LDS latency passes under the DMA issues behind them
            if (it_ + 1 < total) read_tile(std::integral_constant<int, (ST + 1) % SW7_NS>{});
            if (P4V_SW7_FILL != 4 && it_ + 3 < total) {
                piece(std::integral_constant<int, (ST + 3) % SW7_NS>{}, std::integral_constant<int, 2>{});
                piece(std::integral_constant<int, (ST + 3) % SW7_NS>{}, std::integral_constant<int, 3>{});
            }
            frags_ready();
            __builtin_amdgcn_s_barrier();                                                     // B2
        };
        frags_ready();
        for (int ci = 0; ci < ncand; ++ci) {
            for (int kq = 0; kq < ktiles; kq += 4, it += 4) {
                first(it, std::integral_constant<int, 0>{}); second(it, std::integral_constant<int, 0>{});
                first(it + 1, std::integral_constant<int, 1>{}); second(it + 1, std::integral_constant<int, 1>{});
                first(it + 2, std::integral_constant<int, 2>{}); second(it + 2, std::integral_constant<int, 2>{});
                first(it + 3, std::integral_constant<int, 3>{});
                if (kq + 4 < ktiles) second(it + 3, std::integral_constant<int, 3>{});
            }
            second(it - 1, std::integral_constant<int, 3>{}, ci);
        }
    } else {
        // k-tile of group B: fragments of this tile, 2 pieces | B1 | MFMAs (+ 2 pieces) | B2
        auto first = [&](int it_, auto stage_c) __attribute__((always_inline)) {
            constexpr int ST = decltype(stage_c)::value;
            read_tile(stage_c);
            if (P4V_SW7_FILL != 4 && it_ + 3 < total) {
                piece(std::integral_constant<int, (ST + 3) % SW7_NS>{}, std::integral_constant<int, 0>{});
                piece(std::integral_constant<int, (ST + 3) % SW7_NS>{}, std::integral_constant<int, 1>{});
            }
            wait_landed(it_);
            frags_ready();
            __builtin_amdgcn_s_barrier();                                                     // B1
        };
        auto second = [&](int it_, auto stage_c) __attribute__((always_inline)) {
            constexpr int ST = decltype(stage_c)::value;
            compute(it_ + 3 < total, std::integral_constant<int, (ST + 3) % SW7_NS>{}, std::integral_constant<int, 2>{});
        };
        for (int ci = 0; ci < ncand; ++ci) {
            for (int kq = 0; kq < ktiles; kq += 4, it += 4) {
                first(it, std::integral_constant<int, 0>{}); second(it, std::integral_constant<int, 0>{}); __builtin_amdgcn_s_barrier();
                first(it + 1, std::integral_constant<int, 1>{}); second(it + 1, std::integral_constant<int, 1>{}); __builtin_amdgcn_s_barrier();
                first(it + 2, std::integral_constant<int, 2>{}); second(it + 2, std::integral_constant<int, 2>{}); __builtin_amdgcn_s_barrier();
                first(it + 3, std::integral_constant<int, 3>{}); second(it + 3, std::integral_constant<int, 3>{});
                if (kq + 4 < ktiles) __builtin_amdgcn_s_barrier();
            }
            if constexpr (!(P4V_SW7_DBG & 4)) epilogue(ci);
            __builtin_amdgcn_s_barrier();                                                     // B2 of the candidate's last tile
        }
    }
#undef P4V_DSR
#undef P4V_GLD
    __syncthreads();
    if constexpr (COS) return;
    // ---- one coalesced write of this workgroup's results ---------------------------------------------------------------
    for (int i = tid; i < ncand * 32; i += 512) {
        const int cc = c_lo + (i >> 5), wv = (i >> 2) & 7, j = i & 3;
        p.part[(long)cc * p.p_cs + (long)(ct * 4 + (wv & 3)) * p.NG + rt * 8 + (wv >> 2) * 4 + j] = res[i];
    }
}
template <int TW, int EPI>
__global__ __launch_bounds__(512, 2) void k_sweep7(Sweep7Params p) { k_sweep7_body<TW, EPI>(p, P4V_BIDX, P4V_GDIM); }
template <int TW, int EPI>
__global__ __launch_bounds__(512, 2) void k_sweep7_g(GroupArgs<Sweep7Params> a) { P4V_GROUP_ENTER(a); k_sweep7_body<TW, EPI>(a.p[m_], vb_, vg_); }

// ------------------------------------------------------------------------------------------
// k_sos_split: the split search of the split-of-softmax matmul (reference matmul.py:600-631) in one kernel
// ------------------------------------------------------------------------------------------
// For each of the 20 splits s = 2^-i the reference quantises the post-softmax operand A to
//     A_sim = clamp(rint(clamp(A, s, 1) * (q-1)), 0, q-1) / (q-1)  +  clamp(rint(clamp(A, 0, s) / a), 0, q-1) * a,   a = s / (q-1)
// and scores A_sim @ B against raw_out with the UNQUANTISED B -- fp32 by definition.  Through the generic path this is 20
// fp32 planes of A (2 GB for ViT-B, 32 images) written by k_pack<float> and streamed back by the fp32 sweep, 1.9 ms per module.
// Here A never leaves the registers: a wave owns 32 rows of one (image, head) for the whole K (K <= 2 KS), as operands of
// mfma_f32_32x32x2 (lane (g, l31): row l31, k = 2 ks + g), B lies in the LDS, and the 20 candidates are quantised in place.
// What makes that cheap is that every rounding step above is monotone, so clamps commute with them exactly:
//     high part = med3(hv, fl(rint(fl(s (q-1))) / (q-1)), 1)        hv = fl(rint(fl(A (q-1))) / (q-1))   -- per element, once
//     low index = med3(rint(y * 2^i), 0, min(rint(fl(1 / c)), q-1))   y = fl(A / c), c = fl(1 / (q-1))    -- per element, once
// (a = 2^-i c exactly, so A / a = (A / c) 2^i exactly; fl(s / a) = fl(1 / c)), i.e. the two IEEE divisions per element are
// candidate-invariant and one candidate costs 6 VALU operations per element next to its two MFMAs.  mul and add of the low part
// stay separate instructions: the reference rounds the product before the sum.
// Workgroup = 4 waves = 128 rows of one z; 512 registers per wave (one wave per SIMD); scores leave as one float per
// (candidate, wave) and go through k_finish / k_select like every other sweep.
struct SosSplitParams {
    const float* A; long a_z2, a_z, a_r, a_k; int zdiv;
    const float* B; long b_z2, b_z, b_k, b_n;
    const float* O; const float* G;        // [Z][M][N] contiguous
    int Z, M, K, N, wt_mode, C;
    const float* splits;                   // [C], powers of two
    float qm1, c_inv, lo_top;              // q-1, fl(1 / (q-1)), min(rint(fl(1 / c_inv)), q-1)
    float* part;                           // [C][Z][halves * 4]
    int halves;
    const int* crange;                     // optional device-side candidate range (clip_crange)
};

#ifndef P4V_SOS_DBG
#define P4V_SOS_DBG 0          // timing-only ablations: 1 no quantisation arithmetic, 2 no B fragment reads, 4 no MFMAs
#endif
template <int KS, int EPI>
__device__ __forceinline__ void k_sos_split_body(const SosSplitParams& p, const uint3 blockIdx, const uint3 gridDim) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Bt = reinterpret_cast<float*>(smem);                    // [2 KS][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31;
    const int half = blockIdx.x, z = blockIdx.y;
    const long zoffA = p.zdiv > 0 ? (long)(z / p.zdiv) * p.a_z2 + (long)(z % p.zdiv) * p.a_z : (long)z * p.a_z;
    const long zoffB = p.zdiv > 0 ? (long)(z / p.zdiv) * p.b_z2 + (long)(z % p.zdiv) * p.b_z : (long)z * p.b_z;

    // ---- B tile -> LDS, zero padded -------------------------------------------------------------------------------
    for (int i = tid; i < 2 * KS * 64; i += 256) {
        const int k = i >> 6, n = i & 63;
        Bt[i] = (k < p.K && n < p.N) ? p.B[zoffB + (long)k * p.b_k + (long)n * p.b_n] : 0.0f;
    }
    // ---- this wave's 32 rows: the two candidate-invariant images of every element ----------------------------------
    // (a problem of at most 32 rows -- the 16-row sample slice of a pruned pass -- would leave three of the four waves without
    // rows: there all four take the SAME rows and every fourth candidate each)
    const bool share = p.M <= 32 && p.halves == 1;
    const int row0 = share ? 0 : half * 128 + wid * 32;
    const int row = row0 + l31;
    float hv[KS], yv[KS];
    {
        const float* ap = p.A + zoffA + (long)min(row, p.M - 1) * p.a_r;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = 2 * ks + g;
            const float x = (row < p.M && k < p.K) ? ap[(long)k * p.a_k] : 0.0f;
            hv[ks] = rintf(x * p.qm1) / p.qm1;
            yv[ks] = x / p.c_inv;
        }
    }
    // ---- raw_out / metric weight of the wave's 32 x 64 outputs, accumulator layout -----------------------------------
    float u[2][16], w[2][16];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = row0 + (r & 3) + 8 * (r >> 2) + 4 * g, n = cb * 32 + l31;
            const bool ok = m < p.M && n < p.N;
            const long idx = ((long)z * p.M + min(m, p.M - 1)) * p.N + min(n, p.N - 1);
            const float o = p.O[idx];
            const float gw = p.wt_mode == 1 ? p.G[idx] : p.wt_mode == 2 ? o : p.wt_mode == 3 ? fabsf(o) : 1.0f;
            u[cb][r] = ok ? o : 0.0f;
            w[cb][r] = ok ? gw : 0.0f;         // also the validity mask of the unweighted metrics (padding rows quantise to != 0)
        }
    __syncthreads();
    if (row0 >= p.M) return;                  // a wave of pure padding (no barrier below)

    typedef float v16f __attribute__((ext_vector_type(16)));
    const float* b0 = Bt + g * 64 + l31;      // B fragment of k-step ks, column block cb: b0[ks * 128 + cb * 32]
    int c_lo_ = 0, c_hi_ = p.C;
    clip_crange(p.crange, c_lo_, c_hi_);
    for (int c = c_lo_ + (share ? wid : 0); c < c_hi_; c += share ? 4 : 1) {
        const float s = p.splits[c];
        const float inv_s = 1.0f / s;                           // 2^i, exact
        const float a_int = s / p.qm1;                          // matmul.py:609
        const float cl = rintf(s * p.qm1) / p.qm1;
        v16f acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
        // B fragments and the quantised A value run two k-steps ahead of their MFMAs (the LDS latency and the VALU work pass
        // under the matrix pipe's 128 cycles per k-step; a wave is alone on its SIMD)
        float bq0[3], bq1[3], aq[3];
        auto stage = [&](auto ks_c) __attribute__((always_inline)) {
            constexpr int ks = decltype(ks_c)::value;
            if constexpr (ks < KS) {
                if constexpr ((P4V_SOS_DBG & 2) != 0) { bq0[ks % 3] = cl; bq1[ks % 3] = inv_s; }
                else { bq0[ks % 3] = b0[ks * 128]; bq1[ks % 3] = b0[ks * 128 + 32]; }
                if constexpr ((P4V_SOS_DBG & 1) != 0) { aq[ks % 3] = hv[ks] + yv[ks]; return; }
                const float hi = __builtin_amdgcn_fmed3f(hv[ks], cl, 1.0f);
                const float li = __builtin_amdgcn_fmed3f(rintf(yv[ks] * inv_s), 0.0f, p.lo_top);
                float lo = li * a_int;
                asm volatile("" : "+v"(lo));                      // the product is rounded before the sum (no fma contraction)
                aq[ks % 3] = hi + lo;
            }
        };
        stage(std::integral_constant<int, 0>{});
        stage(std::integral_constant<int, 1>{});
        [&]<int... S>(std::integer_sequence<int, S...>) __attribute__((always_inline)) {
            ([&] {
                stage(std::integral_constant<int, S + 2>{});
                if constexpr ((P4V_SOS_DBG & 4) != 0) { acc0[S % 16] += aq[S % 3] * bq0[S % 3]; acc1[S % 16] += aq[S % 3] * bq1[S % 3]; }
                else {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[S % 3], bq0[S % 3], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[S % 3], bq1[S % 3], acc1, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, KS>{});
        float sum = 0.0f;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = u[cb][r] - (cb ? acc1[r] : acc0[r]);
                const float ww = w[cb][r];
                if (EPI == EPI_SQ_W) { const float t2 = ww * d; sum = fmaf(t2, t2, sum); }
                else if (EPI == EPI_SQ) sum = fmaf(ww * d, d, sum);
                else if (EPI == EPI_ABS) sum = fmaf(ww, fabsf(d), sum);
                else sum = fmaf(ww * d, d, sum);
            }
        sum = wave_sum_dpp(sum);
        if (lane == 63) p.part[((long)c * p.Z + z) * (p.halves * 4) + half * 4 + wid] = sum;
    }
}
template <int KS, int EPI>
__global__ __launch_bounds__(256, 1) void k_sos_split(SosSplitParams p) { k_sos_split_body<KS, EPI>(p, P4V_BIDX, P4V_GDIM); }
template <int KS, int EPI>
__global__ __launch_bounds__(256, 1) void k_sos_split_g(GroupArgs<SosSplitParams> a) { P4V_GROUP_ENTER(a); k_sos_split_body<KS, EPI>(a.p[m_], vb_, vg_); }

// ------------------------------------------------------------------------------------------
// k_finish / k_select
// ------------------------------------------------------------------------------------------
// torch.argmax(dim=0) over the candidates of one score block: first maximum, NaN counts as the maximum (the first NaN wins).
// `a` beats `b` if it is NaN and b is not, or both are / neither is NaN and (its value is larger, or equal with a lower index).
__device__ __forceinline__ bool score_beats(float av, int ai, float bv, int bi) {
    const bool an = av != av, bn = bv != bv;
    if (an != bn) return an;
    if (an) return ai < bi;
    return av > bv || (av == bv && ai < bi);
}
// all threads of the block (a power of two <= 256) call it with their running best; returns the block's winner to every thread
__device__ __forceinline__ int block_argmax(float v, int i, float* sv, int* si) {
    const int t = threadIdx.x;
    sv[t] = v; si[t] = i;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if (t < o && score_beats(sv[t + o], si[t + o], sv[t], si[t])) { sv[t] = sv[t + o]; si[t] = si[t + o]; }
        __syncthreads();
    }
    const int r = si[0];
    __syncthreads();
    return r;
}

struct SelectParams {
    const float* scores; int C, nj;
    const float* cands; int cand_cs, cand_js, cand_off;  // cands[best*cand_cs + j*cand_js + cand_off]
    float* interval; int out_js, out_off;
    float* aux_out; float aux_div;                        // optional: aux_out[0] = selected / aux_div (SoS A_interval)
    float* scores_out;  // optional copy [C][scores_out_ld]
    int scores_out_ld;
    int32_t* best_out;  // optional [nj]
    float* iv_host; float* aux_host;   // optional mirrors of `interval` / `aux_out` in mapped host memory (same indexing): the pass
                                       // memo reads them after the stream synchronisation it does anyway -- no copy command
};
// the selection of score block j over the candidates [c_lo, c_hi) (all threads of the workgroup; sv / si: blockDim.x entries)
__device__ __forceinline__ void select_block(const SelectParams& p, int j, int c_lo, int c_hi, float* sv, int* si) {
    float bv = -__builtin_inff();
    int bi = 0x7fffffff;
    for (int c = c_lo + threadIdx.x; c < c_hi; c += blockDim.x) {
        const float v = p.scores[(long)c * p.nj + j];
        if (p.scores_out) p.scores_out[(long)c * p.scores_out_ld + j] = v;
        if (bi == 0x7fffffff || score_beats(v, c, bv, bi)) { bv = v; bi = c; }
    }
    const int best = block_argmax(bv, bi, sv, si);
    if (threadIdx.x == 0) {
        const float sel = p.cands[(long)best * p.cand_cs + (long)j * p.cand_js + p.cand_off];
        p.interval[(long)j * p.out_js + p.out_off] = sel;
        if (p.iv_host) p.iv_host[(long)j * p.out_js + p.out_off] = sel;
        if (p.aux_out) p.aux_out[j] = sel / p.aux_div;
        if (p.aux_out && p.aux_host) p.aux_host[j] = sel / p.aux_div;
        if (p.best_out) p.best_out[j] = best;
    }
}
__device__ __forceinline__ void k_select_body(const SelectParams& p, const uint3 blockIdx, const uint3 gridDim) {          // one workgroup per score block
    __shared__ float sv[128];
    __shared__ int si[128];
    select_block(p, blockIdx.x, 0, p.C, sv, si);
}
__global__ __launch_bounds__(128) void k_select(SelectParams p) { k_select_body(p, P4V_BIDX, P4V_GDIM); }
__global__ __launch_bounds__(128) void k_select_g(GroupArgs<SelectParams> a) { P4V_GROUP_ENTER(a); k_select_body(a.p[m_], vb_, vg_); }

// ---- exact candidate pruning: the kernels between the stages (run_pass_pruned, p4v_api.hip) ------------------------------------
// virt (several score blocks): stage B1 evaluates ONE synthetic candidate whose scale in block j is the scale of block j's own
// stage-A winner (the blocks -- heads of a matmul, V blocks of a Linear -- are scored independently, so its score in block j IS
// that winner's total) instead of the hull of the winners (a dozen heads: 20-30 candidates).  vrow = that candidate's row of the
// candidate table, best[j] = the winners.
struct PruneParams { const float* SA; const float* SB; int C, nj; float margin; const int* r_in; int* r_out;
                     int virt; int* best; const float* cands; int cand_cs, cand_js, cand_off; float* vrow;
                     int* r_host;         // optional mirror of r_out in mapped host memory (read by the host after its stream sync)
                     int* rblk;           // optional per-score-block survivor ranges [2 * nj] (see prune_hull)
                     int* rblk_host; };   // optional mirror of rblk in mapped host memory, written when nj <= 32 (64 ints)
// (one workgroup of 256 threads; the score blocks one after the other, the candidates of a block across the threads)
// r_out = hull over the blocks of stage A's first maxima
#define PRUNE_WIDE_NJ 64        // from this many score blocks on: one THREAD per block (channel-wise weights: hundreds of blocks)
__device__ __forceinline__ void prune_pick(const PruneParams& p, float* sv, int* si) {
    int lo = p.C, hi = 0;
    if (p.nj >= PRUNE_WIDE_NJ) {
        for (int j = threadIdx.x; j < p.nj; j += 256) {
            float bv = p.SA[j];
            int best = 0;
            for (int c = 1; c < p.C; ++c) {
                const float v = p.SA[(long)c * p.nj + j];
                if (score_beats(v, c, bv, best)) { bv = v; best = c; }
            }
            lo = min(lo, best); hi = max(hi, best + 1);
            if (p.virt) {
                p.best[j] = best;
                p.vrow[j * p.cand_js + p.cand_off] = p.cands[(long)best * p.cand_cs + j * p.cand_js + p.cand_off];
            }
        }
        si[threadIdx.x] = lo; si[256 + threadIdx.x] = hi;       // (si: 512 ints in the callers' shared arrays, see k_prune_pick)
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) { si[threadIdx.x] = min(si[threadIdx.x], si[threadIdx.x + o]); si[256 + threadIdx.x] = max(si[256 + threadIdx.x], si[256 + threadIdx.x + o]); }
            __syncthreads();
        }
        if (threadIdx.x == 0) { p.r_out[0] = si[0]; p.r_out[1] = si[256]; }
        return;
    }
    for (int j = 0; j < p.nj; ++j) {
        float bv = -__builtin_inff();
        int bi = 0x7fffffff;
        for (int c = threadIdx.x; c < p.C; c += 256) {
            const float v = p.SA[(long)c * p.nj + j];
            if (bi == 0x7fffffff || score_beats(v, c, bv, bi)) { bv = v; bi = c; }
        }
        const int best = block_argmax(bv, bi, sv, si);
        lo = min(lo, best); hi = max(hi, best + 1);
        if (p.virt && threadIdx.x == 0) {
            p.best[j] = best;
            p.vrow[j * p.cand_js + p.cand_off] = p.cands[(long)best * p.cand_cs + j * p.cand_js + p.cand_off];
        }
    }
    if (threadIdx.x == 0) { p.r_out[0] = lo; p.r_out[1] = hi; }
}
__device__ __forceinline__ void k_prune_pick_body(const PruneParams& p, const uint3 blockIdx, const uint3 gridDim) {
    __shared__ float sv[256];
    __shared__ int si[512];
    prune_pick(p, sv, si);
}
__global__ __launch_bounds__(256) void k_prune_pick(PruneParams p) { k_prune_pick_body(p, P4V_BIDX, P4V_GDIM); }
__global__ __launch_bounds__(256) void k_prune_pick_g(GroupArgs<PruneParams> a) { P4V_GROUP_ENTER(a); k_prune_pick_body(a.p[m_], vb_, vg_); }
// r_out = what stage B2 has to evaluate: the hull of the candidates whose stage-A bound reaches the best complete score, or the
// empty range when stage B1 already evaluated all of them.  Returns (to every thread) whether the range is empty.
// rblk (optional, 2 ints per score block): the same PER BLOCK -- the blocks of a pass are scored independently, so block j only
// needs ITS survivors [rblk[2j], rblk[2j+1]) re-evaluated; a block whose only survivor is its own stage-A winner (virt) is closed:
// empty range, the winner stands without any total (the q block of every ViT qkv layer, whose class-token rows hold its whole
// weight: one survivor, while the flat optima of the k / v blocks keep 10-20 -- a third of stages A2 / B2 of the largest sweep
// family).  r_out is then the hull of the OPEN blocks' ranges (what k_pack has to provide); kernels that cannot take per-block
// ranges sweep r_out for every block, a superset.  sblk: 3 * 64 ints of shared memory (narrow path).
__device__ __forceinline__ bool prune_hull(const PruneParams& p, float* sv, int* sh, int* sblk) {
    int &lo_s = sh[0], &hi_s = sh[1], &bad_s = sh[2], &more_s = sh[3], &empty_s = sh[4];
    const int a = p.r_in[0], b = p.r_in[1];
    if (threadIdx.x == 0) { lo_s = p.C; hi_s = 0; bad_s = 0; more_s = 0; }
    const bool narrow = p.nj < PRUNE_WIDE_NJ;
    if (narrow && (int)threadIdx.x < p.nj) { sblk[3 * threadIdx.x] = p.C; sblk[3 * threadIdx.x + 1] = 0; sblk[3 * threadIdx.x + 2] = 0; }
    __syncthreads();
    if (!narrow) {          // one thread per score block
        bool bad = false;
        for (int j = threadIdx.x; j < p.nj; j += 256) {
            float L = -__builtin_inff();
            bool nan = false, more = false;
            int l = p.C, h = 0;
            if (p.virt) { L = p.SB[j]; nan = L != L; }
            else for (int c = a; c < b; ++c) { const float v = p.SB[(long)c * p.nj + j]; nan |= v != v; L = fmaxf(L, v); }
            const float thr = L - p.margin * fabsf(L);
            const int bj = p.virt ? p.best[j] : -1;
            for (int c = 0; c < p.C; ++c) {
                const float v = p.SA[(long)c * p.nj + j];
                nan |= v != v;
                if (!(v < thr)) { l = min(l, c); h = max(h, c + 1); more |= p.virt && c != bj; }
            }
            bad |= nan || !(L > -__builtin_inff());
            const bool open = p.virt ? more : h > 0;
            if (open) { atomicMin(&lo_s, l); atomicMax(&hi_s, h); atomicOr(&more_s, more ? 1 : 0); }
            if (p.rblk) { p.rblk[2 * j] = open ? l : 0; p.rblk[2 * j + 1] = open ? h : 0; }
        }
        if (bad) atomicOr(&bad_s, 1);
    } else
    for (int j = 0; j < p.nj; ++j) {
        // L* = the best complete score among stage B1's candidates (virt: the one synthetic candidate's score in this block)
        float L = -__builtin_inff();
        bool nan = false;
        if (p.virt) { L = p.SB[j]; nan = L != L; }
        else for (int c = a + threadIdx.x; c < b; c += 256) { const float v = p.SB[(long)c * p.nj + j]; nan |= v != v; L = fmaxf(L, v); }
        sv[threadIdx.x] = L;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sv[threadIdx.x] = fmaxf(sv[threadIdx.x], sv[threadIdx.x + o]); __syncthreads(); }
        L = sv[0];
        __syncthreads();
        const float thr = L - p.margin * fabsf(L);
        int l = p.C, h = 0;
        bool more = false;
        for (int c = threadIdx.x; c < p.C; c += 256) {
            const float v = p.SA[(long)c * p.nj + j];
            nan |= v != v;
            if (!(v < thr)) {
                l = min(l, c); h = max(h, c + 1);
                more |= p.virt && c != p.best[j];     // a survivor besides the block's winner: stage B2 decides
            }
        }
        if (nan || !(L > -__builtin_inff())) atomicOr(&bad_s, 1);
        if (h > 0) { atomicMin(&sblk[3 * j], l); atomicMax(&sblk[3 * j + 1], h); }
        if (more) atomicOr(&sblk[3 * j + 2], 1);
    }
    __syncthreads();
    if (narrow && threadIdx.x == 0) {
        for (int j = 0; j < p.nj; ++j) {
            const bool open = p.virt ? sblk[3 * j + 2] != 0 : sblk[3 * j + 1] > 0;
            if (open) { lo_s = min(lo_s, sblk[3 * j]); hi_s = max(hi_s, sblk[3 * j + 1]); more_s |= sblk[3 * j + 2]; }
            else { sblk[3 * j] = 0; sblk[3 * j + 1] = 0; }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int l = bad_s ? 0 : min(lo_s, hi_s), h = bad_s ? p.C : hi_s;
        // nothing survives outside what stage B1 evaluated: its totals decide
        if (p.virt ? (!bad_s && !more_s) : (!bad_s && l >= a && h <= b && h > l)) l = h = 0;
        if (!p.virt && !bad_s && h > l) { l = min(l, a); h = max(h, b); }     // stage B2 re-evaluates stage B1's candidates with the other survivors
        p.r_out[0] = l; p.r_out[1] = h;
        if (p.r_host) { p.r_host[0] = l; p.r_host[1] = h; }
        lo_s = l; hi_s = h;
        empty_s = l >= h;
    }
    __syncthreads();
    if (p.rblk) {
        // per-block ranges: the block's own hull (virt, healthy); the global range otherwise (NaN anywhere: everything)
        const bool per_blk = p.virt && !bad_s && !empty_s;
        if (narrow) {
            for (int j = threadIdx.x; j < p.nj; j += 256) {
                const int l = per_blk ? sblk[3 * j] : lo_s, h = per_blk ? sblk[3 * j + 1] : hi_s;
                p.rblk[2 * j] = l; p.rblk[2 * j + 1] = h;
                if (p.rblk_host && p.nj <= 32) { p.rblk_host[2 * j] = l; p.rblk_host[2 * j + 1] = h; }
            }
        }
        else if (!per_blk) { for (int j = threadIdx.x; j < p.nj; j += 256) { p.rblk[2 * j] = lo_s; p.rblk[2 * j + 1] = hi_s; } }
    }
    return empty_s != 0;
}
// ... and, when the range is empty, the pass's selection (sl.interval != nullptr) from stage B1's totals: what k_select would pick
// from the table that holds them and -inf elsewhere.  (Tried and dropped: running these one-workgroup steps as the tail of the last
// workgroup of k_finish -- the agent-scope release/acquire it needs writes back and invalidates the L2 of every XCD per workgroup;
// +20 us per k_finish, 3 ms per calibration slower than the separate launches.)
struct HullParams { PruneParams p; SelectParams sl; };
__device__ __forceinline__ void k_prune_hull_body(const HullParams& a_, const uint3 blockIdx, const uint3 gridDim) {
    const auto& [p, sl] = a_;
    __shared__ float sv[256];
    __shared__ int si[256];
    __shared__ int sh[8];
    __shared__ int sblk[3 * PRUNE_WIDE_NJ];
    const bool empty = prune_hull(p, sv, sh, sblk);
    if (!empty || !sl.interval) return;
    if (p.virt) {                                  // every block's only survivor is its stage-A winner
        for (int j = threadIdx.x; j < sl.nj; j += 256) {
            const int best = p.best[j];
            const float sel = sl.cands[(long)best * sl.cand_cs + (long)j * sl.cand_js + sl.cand_off];
            sl.interval[(long)j * sl.out_js + sl.out_off] = sel;
            if (sl.iv_host) sl.iv_host[(long)j * sl.out_js + sl.out_off] = sel;
            if (sl.aux_out) sl.aux_out[j] = sel / sl.aux_div;
            if (sl.aux_out && sl.aux_host) sl.aux_host[j] = sel / sl.aux_div;
            if (sl.best_out) sl.best_out[j] = best;
        }
    } else {                                       // (the caller leaves many-block non-virt selections to k_select)
        const int a = p.r_in[0], b = p.r_in[1];
        for (int j = 0; j < sl.nj; ++j) select_block(sl, j, a, b, sv, si);
    }
}
__global__ __launch_bounds__(256) void k_prune_hull(HullParams p) { k_prune_hull_body(p, P4V_BIDX, P4V_GDIM); }
__global__ __launch_bounds__(256) void k_prune_hull_g(GroupArgs<HullParams> a) { P4V_GROUP_ENTER(a); k_prune_hull_body(a.p[m_], vb_, vg_); }

struct FinishParams {
    const float* part; long p_cs, p_zs; int Np, MT, Z, N, C;
    int j_mode, j_div;     // 0: one block; 1: j = n / j_div; 2: j = z % j_div; 3: j = n
    int nj;
    double norm;           // score = -norm * sum
    float* scores;         // [C][nj]
    const int* crange;     // optional: candidates outside [crange[0], crange[1]) were not evaluated -> score -inf
    unsigned char* mark_done; int mark_n;        // optional, with crange: the candidates this pass packed into the module's plane (flags, count)
    const int* crange_blk;                       // optional, with crange: per-score-block ranges [2 * nj] (the sweep evaluated block j on those only)
};

// One workgroup per (candidate, block): fixed thread->element assignment, double accumulation,
// fixed-shape tree: the result does not depend on scheduling.
__device__ __forceinline__ void k_finish_body(const FinishParams& p, const uint3 blockIdx, const uint3 gridDim) {
    const int c = blockIdx.x, j = blockIdx.y;
    __shared__ double red[256];
    if (p.mark_done && c == 0 && j == 0) {
        // (k_pack read the flags earlier on this stream; the next reader is a later launch)
        const int a = p.crange[0], b = p.crange[1];
        for (int cc = max(a, 0) + (int)threadIdx.x; cc < min(b, p.mark_n); cc += 256) p.mark_done[cc] = 1;
    }
    if (p.crange && (c < p.crange[0] || c >= p.crange[1] || (p.crange_blk && (c < p.crange_blk[2 * j] || c >= p.crange_blk[2 * j + 1])))) {
        if (threadIdx.x == 0) p.scores[(long)c * p.nj + j] = -__builtin_inff();
    } else {
        int nlo = 0, nhi = p.N, zstep = 1, zlo = 0;
        if (p.j_mode == 1) { nlo = j * p.j_div; nhi = min(p.N, nlo + p.j_div); if (j == p.nj - 1) nhi = p.N; }
        else if (p.j_mode == 3) { nlo = j; nhi = j + 1; }
        else if (p.j_mode == 2) { zlo = j; zstep = p.j_div; }
        const int wn = nhi - nlo;
        const int nz = (p.Z - zlo + zstep - 1) / zstep;
        const long total = (long)nz * p.MT * wn;
        double s = 0.0;
        if (total < (1L << 31)) {
            // 32-bit index arithmetic (the 64-bit divisions of the general loop were most of this kernel's 8-10 us -- one
            // workgroup of a B1 finish does all the work, 37 dependent iterations for a ViT-B fc1); same order of summation
            const int total32 = (int)total;
            const float* pc = p.part + (long)c * p.p_cs + nlo;
            for (int i = threadIdx.x; i < total32; i += 256) {
                const int q = i / wn, nn = i - q * wn;
                const int zi = q / p.MT, mt = q - zi * p.MT;
                s += (double)pc[(long)(zlo + zi * zstep) * p.p_zs + (long)mt * p.Np + nn];
            }
        } else
        for (long i = threadIdx.x; i < total; i += 256) {
            const int nn = (int)(i % wn);
            const int mt = (int)((i / wn) % p.MT);
            const int zz = zlo + (int)(i / ((long)wn * p.MT)) * zstep;
            s += (double)p.part[(long)c * p.p_cs + (long)zz * p.p_zs + (long)mt * p.Np + nlo + nn];
        }
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) p.scores[(long)c * p.nj + j] = (float)(-p.norm * red[0]);
    }
}
__global__ __launch_bounds__(256) void k_finish(FinishParams p) { k_finish_body(p, P4V_BIDX, P4V_GDIM); }
__global__ __launch_bounds__(256) void k_finish_g(GroupArgs<FinishParams> a) { P4V_GROUP_ENTER(a); k_finish_body(a.p[m_], vb_, vg_); }

// Cosine finish.  `part` holds triples (dot, |sim|^2, |raw|^2) laid out [C][ZB][ZV][FS][Sp][3]:
// ZB real batch entries, ZV feature blocks swept as separate GEMMs, FS 64-feature slabs, Sp padded samples.
// cos(sample) = dot / (max(|raw|,eps) * max(|sim|,eps)) over the features of the score block
// (torch cosine_similarity, linear.py:406-407), then the mean/sum over samples (linear.py:483-487).
struct FinishCosParams {
    const float* part; long p_cs, p_zs; int Sp, FS, ZB, ZV, S, C;
    int j_mode, j_div;     // 0: one block, all zv; 1: j = zv; 2: j = zb % j_div; 3: j = sample (sum over zb)
    int nj;
    double norm;
    float* scores;
};
__device__ __forceinline__ float cos_item(const FinishCosParams& p, int c, int zb, int zv0, int zv1, int smp) {
    float dot = 0.f, nn = 0.f, oo = 0.f;
    for (int zv = zv0; zv < zv1; ++zv)
        for (int f = 0; f < p.FS; ++f) {
            const float* q = p.part + (long)c * p.p_cs + (long)(zb * p.ZV + zv) * p.p_zs + ((long)f * p.Sp + smp) * 3;
            dot += q[0]; nn += q[1]; oo += q[2];
        }
    const float na = fmaxf(sqrtf(oo), 1e-8f), nb = fmaxf(sqrtf(nn), 1e-8f);
    return dot / (na * nb);
}
__device__ __forceinline__ void k_finish_cos_body(const FinishCosParams& p, const uint3 blockIdx, const uint3 gridDim) {
    // block size: 256 for j_mode 3 (thread = sample), 1024 otherwise (one workgroup per (candidate, score block) reads the
    // candidate's whole table -- 92 MB per pass for a ViT-B proj layer, 370 MB for fc1: 16 waves keep enough loads in flight)
    const int c = blockIdx.x, j = blockIdx.y, nt = blockDim.x;
    double acc = 0.0;
    if (p.j_mode == 3) {
        // one workgroup per (candidate, chunk of samples): thread = sample, serial over zb
        const int smp = j * nt + threadIdx.x;
        if (smp < p.S) {
            for (int zb = 0; zb < p.ZB; ++zb) acc += (double)cos_item(p, c, zb, 0, p.ZV, smp);
            p.scores[(long)c * p.nj + smp] = (float)(p.norm * acc);
        }
        return;
    }
    int zv0 = 0, zv1 = p.ZV, zlo = 0, zstep = 1;
    if (p.j_mode == 1) { zv0 = j; zv1 = j + 1; }
    if (p.j_mode == 2) { zlo = j; zstep = p.j_div; }
    const int nz = (p.ZB - zlo + zstep - 1) / zstep;
    const long total = (long)nz * p.S;
    for (long i = threadIdx.x; i < total; i += nt) {
        const int smp = (int)(i % p.S);
        const int zb = zlo + (int)(i / p.S) * zstep;
        acc += (double)cos_item(p, c, zb, zv0, zv1, smp);
    }
    __shared__ double red[1024];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = nt >> 1; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) p.scores[(long)c * p.nj + j] = (float)(p.norm * red[0]);
}
__global__ __launch_bounds__(1024) void k_finish_cos(FinishCosParams p) { k_finish_cos_body(p, P4V_BIDX, P4V_GDIM); }
__global__ __launch_bounds__(1024) void k_finish_cos_g(GroupArgs<FinishCosParams> a) { P4V_GROUP_ENTER(a); k_finish_cos_body(a.p[m_], vb_, vg_); }

// argmax over candidates per block (torch.argmax semantics: first maximum, NaN is the maximum) and
// gather of the winning candidate interval (linear.py:493-494).
// ---- exact candidate pruning (branch and bound on the score's non-negative terms) -------------------------------------------
// Every difference metric scores a candidate with MINUS a sum of non-negative terms over the samples, so a partial sum over a
// subset of the samples is an upper bound of the candidate's final score.  Stage A scores all candidates on a slice of the
// samples (SA); stage B1 scores, on ALL samples, the candidates that won stage A (one per score block: the range r1) -- the best
// of those totals, L*_j, is a lower bound of the final maximum; a candidate whose stage-A score is already below L*_j (by a
// relative margin that covers the rounding of the two sums) cannot be the argmax of block j.  k_prune_hull writes the hull of
// the survivors over all blocks; stage B2 evaluates exactly that range with the unpruned kernels on the unpruned tiles, so the
// totals of the survivors -- and the selection -- are bit-identical to the unpruned pass.  NaN anywhere disables the pruning
// (torch.argmax treats NaN as the maximum, linear.py:493).
// Which samples form the slice of stage A: the ones that carry the most of the metric's weight.  In a ViT calibrated with the
// Hessian-guided metric the weight raw_grad^2 is concentrated on a handful of samples (the class-token rows: measured on
// ViT-B/224 x 32, the heaviest 1/8 of a Linear's 6304 samples hold 99.9 % of the mass, the first 1/8 hold 2 %,
// tools/row_mass.py), so the scores of a small slice are very tight upper bounds and only a few candidates survive.
//   k_row_mass:  mass[r] = sum over the row's `cols` elements of the metric weight (g^2 | o^2 | |o| | 1)
//   k_topk_rows: indices of the k heaviest rows, ascending (radix select on the float bits + ordered compaction; one
//                workgroup; deterministic)
//   k_gather:    dst[i][...] = src[idx[i]][...] for a 3-D strided inner block (dense destination)
struct RowMassParams { const float* W; const float* O; long rows; long cols; int wt_mode; float* mass; };
__device__ __forceinline__ void k_row_mass_body(const RowMassParams& a_, const uint3 blockIdx, const uint3 gridDim) {
    const auto& [W, O, rows, cols, wt_mode, mass] = a_;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* src = (wt_mode == 1 ? W : O) + r * cols;
    float s = 0.0f;
    if (wt_mode == 0) s = lane == 0 ? 1.0f : 0.0f;
    else
        for (long i = lane; i < cols; i += 64) {
            const float v = src[i];
            s += wt_mode == 2 ? fabsf(v) : v * v;
        }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) mass[r] = s;
}
__global__ __launch_bounds__(256) void k_row_mass(RowMassParams p) { k_row_mass_body(p, P4V_BIDX, P4V_GDIM); }
__global__ __launch_bounds__(256) void k_row_mass_g(GroupArgs<RowMassParams> a) { P4V_GROUP_ENTER(a); k_row_mass_body(a.p[m_], vb_, vg_); }
// mass2[i] = sum of `group` consecutive masses (matmul: the heads of one image)
__global__ void k_group_mass(const float* mass, int n_groups, int group, float* out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    float s = 0.0f;
    for (int i = 0; i < group; ++i) s += mass[(long)g * group + i];
    out[g] = s;
}
struct TopkParams { const float* mass_all; int n; int k; int* idx_all; };
__device__ __forceinline__ void k_topk_rows_body(const TopkParams& a_, const uint3 blockIdx, const uint3 gridDim) {
    const auto& [mass_all, n, k, idx_all] = a_;
    // one workgroup per segment (matmul: the rows of one (image, head); Linear: a single segment); indices are segment-local.
    // Radix select, one byte of the key per pass (256-bin histogram in the LDS, a wave whose keys share the digit adds its count
    // once), then an ordered compaction with wave ballots: 4 + 1 passes over the masses and one barrier per 1024 rows
    // (round 3: 32 one-bit passes and a 10-step scan per 1024 rows -- 50 us for the 6304 samples of a ViT-B Linear, 1-5 ms for
    // the 1.2 M samples of a Swin stage-1 Linear).
    const float* mass = mass_all + (long)blockIdx.x * n;
    int* idx = idx_all + (long)blockIdx.x * k;
    __shared__ int hist[256];
    __shared__ int wtot[4];
    __shared__ int wg[2][16], we[2][16];
    __shared__ unsigned prefix_s;
    __shared__ int krem_s;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    auto key = [&](int i) -> unsigned { const unsigned b = __float_as_uint(mass[i]); return (b & 0x80000000u) ? 0u : b; };   // negative / -0: lightest
    if (t == 0) { prefix_s = 0; krem_s = k; }
    for (int shift = 24; shift >= 0; shift -= 8) {       // the k-th largest key, a byte at a time
        if (t < 256) hist[t] = 0;
        __syncthreads();
        const unsigned hi_mask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
        const unsigned pre = prefix_s;
        const int kr = krem_s;
        for (int i0 = 0; i0 < n; i0 += 1024) {
            const int i = i0 + t;
            const unsigned kv = i < n ? key(i) : 0u;
            const bool in = i < n && (kv & hi_mask) == pre;
            const int d = (int)((kv >> shift) & 255u);
            const unsigned long long m = __ballot(in);
            if (m) {
                const int d0 = __shfl(d, __ffsll((long long)m) - 1);
                if (__ballot(in && d != d0) == 0ull) { if (lane == __ffsll((long long)m) - 1) atomicAdd(&hist[d0], __popcll(m)); }
                else if (in) atomicAdd(&hist[d], 1);
            }
        }
        __syncthreads();
        // suffix sums over the bins (threads 0..255): bin b holds the kr-th largest iff  count(bins > b) < kr <= count(bins >= b)
        const int h = t < 256 ? hist[t] : 0;
        int sfx = h;
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_down(sfx, o); if (lane + o < 64) sfx += v; }
        if (t < 256 && lane == 0) wtot[wv] = sfx;
        __syncthreads();
        if (t < 256) {
            int above = 0;
            for (int q = wv + 1; q < 4; ++q) above += wtot[q];
            const int incl = sfx + above, excl = incl - h;
            if (excl < kr && kr <= incl) { prefix_s = pre | ((unsigned)t << shift); krem_s = kr - excl; }
        }
        __syncthreads();
    }
    const unsigned T = prefix_s;
    const int eq_want = krem_s;                          // rows AT the threshold to take (the first ones by index); k - eq_want rows are above it
    // ordered compaction, 1024 rows at a time: rows above the threshold, and rows AT it until k are taken
    int base = 0, eq_taken = 0;
    for (int i0 = 0, it = 0; i0 < n; i0 += 1024, ++it) {
        const int i = i0 + t;
        const unsigned kv = i < n ? key(i) : 0u;
        const bool is_gt = i < n && kv > T, is_eq = i < n && kv == T;
        const unsigned long long bg = __ballot(is_gt), be = __ballot(is_eq);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (lane == 0) { wg[it & 1][wv] = __popcll(bg); we[it & 1][wv] = __popcll(be); }
        __syncthreads();                                 // (one barrier per chunk: the tables alternate)
        int gt_before = __popcll(bg & below), eq_before = __popcll(be & below), gt_tot = 0, eq_tot = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int a_ = wg[it & 1][q], b_ = we[it & 1][q];
            if (q < wv) { gt_before += a_; eq_before += b_; }
            gt_tot += a_; eq_tot += b_;
        }
        const int eq_room = max(eq_want - eq_taken, 0);
        const bool take = is_gt || (is_eq && eq_before < eq_room);
        if (take) idx[base + gt_before + min(eq_before, eq_room)] = i;
        const int eq_take = min(eq_tot, eq_room);
        base += gt_tot + eq_take;
        eq_taken += eq_take;
    }
}
__global__ __launch_bounds__(1024) void k_topk_rows(TopkParams p) { k_topk_rows_body(p, P4V_BIDX, P4V_GDIM); }
__global__ __launch_bounds__(1024) void k_topk_rows_g(GroupArgs<TopkParams> a) { P4V_GROUP_ENTER(a); k_topk_rows_body(a.p[m_], vb_, vg_); }
// dst[r][a][b][c] = src[seg_off(r / seg) + idx[r] * s0 + a * s1 + b * s2 + c * s3]; seg = rows per segment (0: one segment),
// seg_off(z) = (z / zdiv) * sz2 + (z % zdiv) * sz (two-level batch stride: image, head)
// frac[0] = (weight of the selected rows) / (weight of all rows): how tight the slice's bounds are
struct MassFracParams { const float* mass; long n; const int* idx; int segs; int seg_rows; int k; float* frac; float* frac_host; };
__device__ __forceinline__ void k_mass_fraction_body(const MassFracParams& a_, const uint3 blockIdx, const uint3 gridDim) {
    const auto& [mass, n, idx, segs, seg_rows, k, frac, frac_host] = a_;
    __shared__ double red[1024];
    double tot = 0.0, sel = 0.0;
    for (long i = threadIdx.x; i < n; i += 1024) tot += (double)mass[i];
    for (long i = threadIdx.x; i < (long)segs * k; i += 1024) sel += (double)mass[(i / k) * seg_rows + idx[i]];
    for (int pass = 0; pass < 2; ++pass) {
        red[threadIdx.x] = pass ? sel : tot;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
        if (threadIdx.x == 0) { if (pass) sel = red[0]; else tot = red[0]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float f = tot > 0.0 ? (float)(sel / tot) : 0.0f;
        frac[0] = f;
        if (frac_host) frac_host[0] = f;               // mapped host memory: read after the stream sync, no copy command
    }
}
__global__ __launch_bounds__(1024) void k_mass_fraction(MassFracParams p) { k_mass_fraction_body(p, P4V_BIDX, P4V_GDIM); }
__global__ __launch_bounds__(1024) void k_mass_fraction_g(GroupArgs<MassFracParams> a) { P4V_GROUP_ENTER(a); k_mass_fraction_body(a.p[m_], vb_, vg_); }

struct GatherParams { const float* src; long s0, s1, s2, s3; int d1, d2, d3; const int* idx; int k; float* dst; int seg, zdiv; long sz2, sz; };
__device__ __forceinline__ void k_gather_body(const GatherParams& p, const uint3 blockIdx, const uint3 gridDim) {
    const long inner = (long)p.d1 * p.d2 * p.d3, total = inner * p.k;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / inner);
        long rem = i - (long)r * inner;
        const int a = (int)(rem / ((long)p.d2 * p.d3)); rem -= (long)a * p.d2 * p.d3;
        const int b = (int)(rem / p.d3), cidx = (int)(rem - (long)b * p.d3);
        long off = 0;
        if (p.seg > 0) { const int z = r / p.seg; off = (long)(z / p.zdiv) * p.sz2 + (long)(z % p.zdiv) * p.sz; }
        p.dst[i] = p.src[off + (long)p.idx[r] * p.s0 + (long)a * p.s1 + (long)b * p.s2 + (long)cidx * p.s3];
    }
}
__global__ __launch_bounds__(256) void k_gather(GatherParams p) { k_gather_body(p, P4V_BIDX, P4V_GDIM); }
__global__ __launch_bounds__(256) void k_gather_g(GroupArgs<GatherParams> a) { P4V_GROUP_ENTER(a); k_gather_body(a.p[m_], vb_, vg_); }

// rows idx[0..k) of the im2col matrix of a conv input (PackParams' conv fields; flat layout, Z = 1), dense [k][K]
struct GatherIm2colParams { PackParams p; const int* idx; int k; float* dst; };
__device__ __forceinline__ void k_gather_im2col_body(const GatherIm2colParams& a_, const uint3 blockIdx, const uint3 gridDim) {
    const auto& [p, idx, k, dst] = a_;
    const long total = (long)k * p.K;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / p.K), kk = (int)(i - (long)r * p.K);
        dst[i] = pack_load(p, p.src, idx[r], kk);
    }
}
__global__ __launch_bounds__(256) void k_gather_im2col(GatherIm2colParams p) { k_gather_im2col_body(p, P4V_BIDX, P4V_GDIM); }
__global__ __launch_bounds__(256) void k_gather_im2col_g(GroupArgs<GatherIm2colParams> a) { P4V_GROUP_ENTER(a); k_gather_im2col_body(a.p[m_], vb_, vg_); }
// conv output / gradient [b][oc][L] -> rows of the im2col GEMM [b * L][oc] (32 x 32 tiles through LDS; grid (L/32, oc/32, b))
struct NchwRowsParams { const float* src; int oc; int L; float* dst; };
__device__ __forceinline__ void k_nchw_to_rows_body(const NchwRowsParams& a_, const uint3 blockIdx, const uint3 gridDim) {
    const auto& [src, oc, L, dst] = a_;
    __shared__ float t[32][33];
    const int l0 = blockIdx.x * 32, c0 = blockIdx.y * 32, bi = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < oc && l0 + tx < L) t[i][tx] = src[((long)bi * oc + c0 + i) * L + l0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (l0 + i < L && c0 + tx < oc) dst[((long)bi * L + l0 + i) * oc + c0 + tx] = t[tx][i];
}
__global__ __launch_bounds__(256) void k_nchw_to_rows(NchwRowsParams p) { k_nchw_to_rows_body(p, P4V_BIDX, P4V_GDIM); }
__global__ __launch_bounds__(256) void k_nchw_to_rows_g(GroupArgs<NchwRowsParams> a) { P4V_GROUP_ENTER(a); k_nchw_to_rows_body(a.p[m_], vb_, vg_); }
// memset of a small device range as a kernel of this library (hipMemsetAsync is a runtime kernel of its own per call and cannot
// join a grouped launch): flags, ordered-max accumulators, zero biases, partial-sum tables
struct FillBytesParams { void* dst; int value; long bytes; };
__device__ __forceinline__ void k_fill_bytes_body(const FillBytesParams& p, const uint3 blockIdx, const uint3 gridDim) {
    const unsigned b = (unsigned)p.value & 0xffu;
    if (((unsigned long long)p.dst & 3) == 0 && (p.bytes & 3) == 0) {
        const unsigned w = b * 0x01010101u;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (p.bytes >> 2); i += (long)gridDim.x * 256) reinterpret_cast<unsigned*>(p.dst)[i] = w;
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.bytes; i += (long)gridDim.x * 256) reinterpret_cast<unsigned char*>(p.dst)[i] = (unsigned char)b;
    }
}
__global__ __launch_bounds__(256) void k_fill_bytes(FillBytesParams p) { k_fill_bytes_body(p, P4V_BIDX, P4V_GDIM); }
__global__ __launch_bounds__(256) void k_fill_bytes_g(GroupArgs<FillBytesParams> a) { P4V_GROUP_ENTER(a); k_fill_bytes_body(a.p[m_], vb_, vg_); }
struct FillParams { float* p; float v; int n; };
__device__ __forceinline__ void k_fill_f32_body(const FillParams& a_, const uint3 blockIdx, const uint3 gridDim) {
    const auto& [p, v, n] = a_;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void k_fill_f32(FillParams p) { k_fill_f32_body(p, P4V_BIDX, P4V_GDIM); }
__global__ void k_fill_f32_g(GroupArgs<FillParams> a) { P4V_GROUP_ENTER(a); k_fill_f32_body(a.p[m_], vb_, vg_); }
// virt: the final table is stage B2's; where B2 did not run a block's winner (empty B2), the synthetic candidate's score stands in
struct MergeVirtParams { float* S2; const float* SB; const int* best; int nj; };
__device__ __forceinline__ void k_merge_virtual_body(const MergeVirtParams& a_, const uint3 blockIdx, const uint3 gridDim) {
    const auto& [S2, SB, best, nj] = a_;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nj) return;
    float* d = S2 + (long)best[j] * nj + j;
    if (*d == -__builtin_inff()) *d = SB[j];
}
__global__ void k_merge_virtual(MergeVirtParams p) { k_merge_virtual_body(p, P4V_BIDX, P4V_GDIM); }
__global__ void k_merge_virtual_g(GroupArgs<MergeVirtParams> a) { P4V_GROUP_ENTER(a); k_merge_virtual_body(a.p[m_], vb_, vg_); }
// final score table of a pruned pass: stage B2's where it evaluated the candidate, stage B1's otherwise (the two agree bit for bit
// where both did)
struct MergeParams { float* S2; const float* SB; int n; };
__device__ __forceinline__ void k_merge_scores_body(const MergeParams& a_, const uint3 blockIdx, const uint3 gridDim) {
    const auto& [S2, SB, n] = a_;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && S2[i] == -__builtin_inff()) S2[i] = SB[i];
}
__global__ void k_merge_scores(MergeParams p) { k_merge_scores_body(p, P4V_BIDX, P4V_GDIM); }
__global__ void k_merge_scores_g(GroupArgs<MergeParams> a) { P4V_GROUP_ENTER(a); k_merge_scores_body(a.p[m_], vb_, vg_); }

}  // namespace p4v
