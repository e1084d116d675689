// icp_amd/csrc/search_mfma.hip -- the reduced-precision pre-filter of the brute-force search on the matrix pipe (gfx950).
//
// Same search, same exact slow path as search_f32_kernel (kernels.hip); only the fast path moves.  The filter value
//     s(i, j) = |q_j|^2 - 2 p_i.q_j        ( = d^2(p_i, q_j) - |p_i|^2 )
// is a small contraction of a target record with a source record, i.e. matrix-pipe work; the VALU is left with half a
// v_min3_f32 per pair and the event checks.  (Round 1 also carried the same contraction on the fp32 MFMA,
// v_mfma_f32_32x32x2_f32: it does not co-execute with VALU work on gfx950 and ran at 74 ms against 67 ms for the packed-VALU
// filter and 28 ms for the fp16 form below -- profiles/r1_v8_mfma_experiment.txt -- so it was removed in round 2.)
#include "kernels.h"
#include "search_common.h"

#include <math.h>
#include <stdlib.h>

namespace icpk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------
// the filter on v_mfma_f32_32x32x16_f16: one MFMA per 32 targets x 32 source rows
// ------------------------------------------------------------------------------------------------
// The K = 16 half-precision MFMA of gfx950 runs at 16 times the fp32 matrix rate, and K = 16 is room to carry every
// operand in TWO fp16 pieces (hi + lo, 22 bits) -- after scaling both clouds by a power of two sigma that puts the largest
// coordinate at 2^6..2^7, far from fp16's overflow and with the lo pieces mostly above its subnormals:
//     s' = sigma^2 (|q|^2 - 2 p.q) = sum over c of (Th + Tl)(Ph + Pl) + (W0 + W1 + W2),   T = -2 sigma q_c, P = sigma p_c, W = sigma^2 |q|^2
// = 15 products (the 16th slot is 0): half 0 of the wave carries x and y, half 1 carries z and the three pieces of W
// against ones.  Every product of two fp16 values is exact in fp32, so the error is
//  * the pieces' truncation: |v - hi - lo| <= 2^-22 |v| for each operand, i.e. <= 2^-21 |P||T| over the three products;
//  * the fp32 accumulation of 15 terms: <= 15 * 2^-23 (|P||T| + W) even for a pipe that truncates instead of rounding;
//  * a lo piece in fp16's subnormal range (below 2^-14), should the pipe flush it: <= 2^-14 times its partner, over all
//    pieces <= 2^-13 (|P| + Tmax), and 2^-13 for the pieces of W:
//     |s'_computed - s'| <= 2^-18 (|P| Tmax + Wmax) + 2^-13 (|P| + Tmax) + 2^-12 =: E'    (all in scaled units),
// with a factor two in hand on the first term.  Everything else is the fp32 filter: threshold
// sigma^2 (bd (1 + 2^-50) - |p|^2 (1 - 2^-50)) + E', exact re-evaluation of every block of 64 targets that trips it.  A
// source row whose scaled coordinates leave fp16's range multiplies by zero instead and takes the exact path on every
// block.  Measured (1M x 1M, one MI355X): 28.7 ms per iteration against 66 ms for the packed-VALU filter; the fp16 MFMA
// (32 cycles each) co-executes with the VALU, which is left with half a v_min3_f32 per pair.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
double search_mfma16_scale(double qmax);

// LEAN operands: 11 of the 16 K slots carry a product, 5 are zero in BOTH operands -- the three lo x lo products
// (each <= 2^-22 |T_c| |P_c|) and the third piece of W (W0 + W1 hold 22 bits) are dropped.  Their sum stays inside the SAME
// bound E': piece truncation 2^-21 |P||T| + dropped lo x lo 2^-22 |P||T| + fp32 accumulation of 11 terms 11 * 2^-23
// (|P||T| + W) + W's truncation 2^-22 W  =  8.5 * 2^-22 |P||T| + 7.5 * 2^-22 W  <  2^-18 (|P||T| + W).
// Why: the kernel is POWER bound, not issue bound -- the chip lowers its clock to 1.6-1.9 GHz under random fp16 MFMA
// operands and holds 2.39 GHz on zeros (tools/ubench_mfma.hip, profiles/r2_ubench_mfma.txt), so every multiplier that
// sees a zero buys clock.
#ifndef ICPK_LEAN
#define ICPK_LEAN 1
#endif
constexpr bool kLean = ICPK_LEAN != 0;

struct __attribute__((aligned(16))) Half8 { _Float16 v[8]; };

__device__ __forceinline__ void split16(double v, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)(float)v;
    lo = (_Float16)(float)(v - (double)(float)hi);
}

// rec16[0 .. m_pad) = half 0 of every target {Txh,Txh,Txl,Txl,Tyh,Tyh,Tyl,Tyl}, rec16[m_pad .. 2 m_pad) = half 1
// {Tzh,Tzh,Tzl,Tzl,W0,W1,W2,0}; pad targets: W0 = +inf
__global__ __launch_bounds__(kWG) void build_qrec16_kernel(const double *__restrict__ qx, const double *__restrict__ qy,
                                                           const double *__restrict__ qz, size_t m, size_t m_pad, double sigma,
                                                           Half8 *__restrict__ rec16)
{
    const size_t i = (size_t)blockIdx.x * kWG + threadIdx.x;
    if (i >= m_pad) return;
    Half8 a, b;
    for (int k = 0; k < 8; ++k) a.v[k] = b.v[k] = (_Float16)0.0f;
    if (i < m) {
        const double x = qx[i], y = qy[i], z = qz[i];
        _Float16 h, l;
        split16(-2.0 * sigma * x, h, l); a.v[0] = a.v[1] = h; a.v[2] = l; a.v[3] = kLean ? (_Float16)0.0f : l;
        split16(-2.0 * sigma * y, h, l); a.v[4] = a.v[5] = h; a.v[6] = l; a.v[7] = kLean ? (_Float16)0.0f : l;
        split16(-2.0 * sigma * z, h, l); b.v[0] = b.v[1] = h; b.v[2] = l; b.v[3] = kLean ? (_Float16)0.0f : l;
        const double w = sigma * sigma * ((x * x + y * y) + z * z);
        const _Float16 w0 = (_Float16)(float)w;
        const double r1 = w - (double)(float)w0;
        const _Float16 w1 = (_Float16)(float)r1;
        const _Float16 w2 = (_Float16)(float)(r1 - (double)(float)w1);
        b.v[4] = w0; b.v[5] = w1; b.v[6] = kLean ? (_Float16)0.0f : w2;
    } else {
        b.v[4] = (_Float16)INFINITY;
    }
    rec16[i] = a;
    rec16[m_pad + i] = b;
}

// E' of the header comment; pl = sigma |p|, tmax = 2 sigma qmax, wmax = sigma^2 qmax^2
__device__ __forceinline__ double mfma16_bound(double pl, double tmax, double wmax)
{
    return 0x1p-18 * (pl * tmax + wmax) + 0x1p-13 * (pl + tmax) + 0x1p-12;
}

constexpr int kTileQH = 1024;   // targets per LDS tile (2 x 16 KB)
#ifndef ICPK_ROWSH
#define ICPK_ROWSH 64
#endif
constexpr int kRowsH = ICPK_ROWSH;   // targets between event checks of the seeded launches (64 or 128; <= kSplitGran)
constexpr int kRowsCoop = 64;        // ... of the un-seeded launches: the cooperative exact path maps lane u to target u
static_assert((kRowsH == 64 || kRowsH == 128) && (size_t)kRowsH <= kSplitGran, "event-check blocks must tile a target split");

#ifndef ICPK_COOPMAX
#define ICPK_COOPMAX 8
#endif
constexpr int kCoopMax = ICPK_COOPMAX;     // rows of a source group flagged in one block up to which the wave handles them one by one

// minimum over the wave, NaN ignored (fmin); every lane gets the result
__device__ __forceinline__ double wave_min_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = __builtin_fmin(v, __shfl_xor(v, o, 64));
    return v;
}

// COOP selects how the rare exact path is entered.  false (seeded launches, i.e. every iteration after the first): each
// flagged lane re-scans the block for its own row -- events are so rare that only the filter loop's register budget
// matters.  true (unseeded launches: the first iteration, the stage-level correspondence call): the running minimum of a
// row drops ~ln(M/64) times during the scan, so the wave handles flagged rows cooperatively, one lane per target.
template <int G, bool COOP>
__global__ __launch_bounds__(kWG) __attribute__((amdgpu_waves_per_eu(4))) void search_mfma16_kernel(
    const double *__restrict__ px, const double *__restrict__ py, const double *__restrict__ pz, const double *__restrict__ pc,
    const double *__restrict__ qx, const double *__restrict__ qy, const double *__restrict__ qz, const Half8 *__restrict__ rec16,
    double qmax, double sigma, unsigned n_pad, unsigned m, unsigned m_pad, unsigned m_scan, unsigned q_per_split,
    const uint32_t *__restrict__ idx_prev, double *__restrict__ out_key, uint32_t *__restrict__ out_idx)
{
    __shared__ Half8 st[2][kTileQH];
    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63u, wave = tid >> 6;
    const unsigned col = lane & 31u, half = lane >> 5;
    const unsigned base = blockIdx.x * (4u * 32u * G) + wave * (32u * G) + col;
    const double c0 = pc[0], c1 = pc[1], c2 = pc[2];
    const unsigned q0 = blockIdx.y * q_per_split;
    const unsigned q1 = (q0 + q_per_split < m_scan) ? (q0 + q_per_split) : m_scan;
    const double s2 = sigma * sigma, tmax = 2.0 * sigma * qmax, wmax = s2 * qmax * qmax;

    constexpr int ROWS = COOP ? kRowsCoop : kRowsH;
    f16x8 b[G];
    float ms[G], thr[G];
    double bd[G], bk[G];
    uint32_t bj[G], sblk[G];
    // threshold in scaled units from the exact running bound
    auto threshold = [&](double bdv, double pn) -> float {
        const double pl = sigma * sqrt(pn);
        const double E = mfma16_bound(pl, tmax, wmax);
        const bool usable = pn < 1e30 && pl < 16384.0;
        return f32_above(usable ? s2 * ((bdv - pn) + (bdv + pn) * 0x1p-50) + E : INFINITY);
    };
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const unsigned row = base + g * 32u;
        const double ax = px[row] - c0, ay = py[row] - c1, az = pz[row] - c2;
        const double pn = (ax * ax + ay * ay) + az * az;
        const bool usable = pn < 1e30 && sigma * sqrt(pn) < 16384.0;
        _Float16 xh, xl, yh, yl, zh, zl;
        split16(usable ? sigma * ax : 0.0, xh, xl);
        split16(usable ? sigma * ay : 0.0, yh, yl);
        split16(usable ? sigma * az : 0.0, zh, zl);
        const _Float16 one = (_Float16)1.0f, zero = (_Float16)0.0f;
        const f16x8 b0 = {xh, xl, xh, kLean ? zero : xl, yh, yl, yh, kLean ? zero : yl};
        const f16x8 b1 = {zh, zl, zh, kLean ? zero : zl, one, one, kLean ? zero : one, zero};
        b[g] = half ? b1 : b0;
        const uint32_t js = idx_prev ? idx_prev[row] : 0xFFFFFFFFu;
        bd[g] = seed_init<ROWS>(ax, ay, az, qx, qy, qz, js, m, q0, q1, bk[g], bj[g], sblk[g]);
        thr[g] = threshold(bd[g], pn);
        ms[g] = thr[g];
    }
    const Half8 *sa = st[half];

    for (unsigned t0 = q0; t0 < q1; t0 += kTileQH) {
        const int cnt = (q1 - t0 < (unsigned)kTileQH) ? (int)(q1 - t0) : kTileQH;   // multiple of kSplitGran
        __syncthreads();
        for (int k = tid; k < cnt; k += kWG) {
            st[0][k] = rec16[t0 + k];
            st[1][k] = rec16[(size_t)m_pad + t0 + k];
        }
        __syncthreads();
        // ROWS targets between event checks, in sub-blocks of 64: two MFMA rounds per source group and sub-block, the operand
        // of the next round already on its way from LDS while this one is reduced
        f16x8 a_next = *(const f16x8 *)&sa[col];
        for (int j0 = 0; j0 < cnt; j0 += ROWS) {
#pragma unroll
            for (int sub = 0; sub < ROWS; sub += 64) {
                const f16x8 a0 = a_next;
                const f16x8 a1 = *(const f16x8 *)&sa[j0 + sub + 32 + col];
                a_next = *(const f16x8 *)&sa[(j0 + sub + 64 < cnt ? j0 + sub + 64 : 0) + col];
                f32x16 acc[G];
                const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[g], zero, 0, 0, 0);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    float v = ms[g];
#pragma unroll
                    for (int r = 0; r < 16; r += 2) v = vmin3f(v, acc[g][r], acc[g][r + 1]);
                    ms[g] = v;
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[g], zero, 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    float v = ms[g];
#pragma unroll
                    for (int r = 0; r < 16; r += 2) v = vmin3f(v, acc[g][r], acc[g][r + 1]);
                    ms[g] = v;
                }
            }
            const unsigned blk = (t0 + (unsigned)j0) / ROWS;
            bool ev = false;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (blk == sblk[g]) ms[g] = thr[g];   // the seed's own block was evaluated exactly by seed_init
                ev |= (ms[g] < thr[g]);
            }
            if constexpr (COOP) {
                if (__ballot(ev)) {   // wave-uniform: the cooperative path below needs every lane
                    const unsigned jg = t0 + (unsigned)j0;
                    const double tqx = qx[jg + lane], tqy = qy[jg + lane], tqz = qz[jg + lane];   // lane u <-> target jg + u
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const bool mine = ms[g] < thr[g];
                        unsigned long long mk = __ballot(mine);
                        if (mk == 0) continue;
                        const unsigned row = base + g * 32u;
                        if (__popcll(mk) > kCoopMax) {
                            // many rows at once (the first blocks of the scan): each lane re-scans for its own row
                            if (mine) {
                                const double ax = px[row] - c0, ay = py[row] - c1, az = pz[row] - c2;
                                const double pn = (ax * ax + ay * ay) + az * az;
                                double dbest;
                                rescan_block<ROWS>(ax, ay, az, qx + jg, qy + jg, qz + jg, 0, jg, bd[g], bk[g], bj[g], dbest);
                                thr[g] = threshold(bd[g], pn);
                                ms[g] = thr[g];
                            }
                            continue;
                        }
                        // few rows: the wave takes them one at a time, lane u evaluating target u of the block.  With a
                        // unique minimum outside the tie band of the second smallest value the outcome of the in-order
                        // scan is known without running it (rescan_block's own shortcut, evaluated across the wave);
                        // anything else -- ties, NaN -- is left to that scan.
                        while (mk) {
                            const int L = __builtin_ctzll(mk);
                            mk &= mk - 1;
                            const unsigned rowL = row - col + ((unsigned)L & 31u);
                            const double ax = px[rowL] - c0, ay = py[rowL] - c1, az = pz[rowL] - c2;
                            const double pn = (ax * ax + ay * ay) + az * az;
                            const double d = d2_exact(ax, ay, az, tqx, tqy, tqz);
                            const unsigned long long bad = __ballot(d != d);
                            const double bdL = __shfl(bd[g], L, 64);
                            const unsigned long long lt = __ballot(d < bdL);   // targets of the block that beat the row's bound
                            if (!bad && lt == 0) {   // false alarm of the filter
                                if ((int)lane == L) ms[g] = thr[g];
                                continue;
                            }
                            double dm;
                            int um;
                            bool direct;
                            if (!bad && (lt & (lt - 1)) == 0) {
                                // exactly one: it is the block's minimum, and unique; no wave reduction needed
                                um = __builtin_ctzll(lt);
                                dm = __shfl(d, um, 64);
                                direct = __popcll(__ballot(d <= tie_band(dm))) == 1;   // nothing else inside its tie band
                            } else {
                                dm = wave_min_f64(d);
                                const unsigned long long eq = __ballot(d == dm);
                                um = eq ? __builtin_ctzll(eq) : 0;
                                const double d2nd = (__popcll(eq) == 1) ? wave_min_f64((int)lane == um ? INFINITY : d) : dm;
                                direct = !bad && d2nd > tie_band(dm);
                            }
                            if ((int)lane == L) {
                                if (direct) {
                                    bd[g] = dm;
                                    const double key = key_of(dm);
                                    const uint32_t j = jg + (unsigned)um;
                                    if (key < bk[g] || (key == bk[g] && j < bj[g])) {
                                        bk[g] = key;
                                        bj[g] = j;
                                    }
                                } else {
                                    double dbest;
                                    rescan_block<ROWS>(ax, ay, az, qx + jg, qy + jg, qz + jg, 0, jg, bd[g], bk[g], bj[g], dbest);
                                }
                                thr[g] = threshold(bd[g], pn);
                                ms[g] = thr[g];
                            }
                        }
                    }
                }
            } else {
                if (ev) {
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        if (ms[g] < thr[g]) {
                            const unsigned row = base + g * 32u;
                            const double ax = px[row] - c0, ay = py[row] - c1, az = pz[row] - c2;
                            const double pn = (ax * ax + ay * ay) + az * az;
                            const unsigned jg = t0 + (unsigned)j0;
                            double dbest;
                            rescan_block<ROWS>(ax, ay, az, qx + jg, qy + jg, qz + jg, 0, jg, bd[g], bk[g], bj[g], dbest);
                            thr[g] = threshold(bd[g], pn);
                            ms[g] = thr[g];
                        }
                    }
                }
            }
        }
    }
    const size_t o = (size_t)blockIdx.y * n_pad + base;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const double ok = __shfl_xor(bk[g], 32, 64);
        const uint32_t oj = __shfl_xor(bj[g], 32, 64);
        if (ok < bk[g] || (ok == bk[g] && oj < bj[g])) {
            bk[g] = ok;
            bj[g] = oj;
        }
        if (half == 0) {
            out_key[o + g * 32u] = bk[g];
            out_idx[o + g * 32u] = bj[g];
        }
    }
}

// Test hook: the filter values of the kernel above against the exact ones.  The targets come in runs of 32 identical
// points, so that all 16 values a lane gets back for a block belong to one (source, target) pair whatever the row order of
// the instruction's result layout.  *worst = max over all pairs of |s'_computed - s'_exact| / E'(p): the bound holds on the
// data iff it is <= 1.  P is taken as already centred; one wave per (32 source rows, one block of 32 targets).
__global__ __launch_bounds__(64) void mfma16_probe_kernel(const double *__restrict__ px, const double *__restrict__ py,
                                                          const double *__restrict__ pz, const double *__restrict__ qx,
                                                          const double *__restrict__ qy, const double *__restrict__ qz,
                                                          const Half8 *__restrict__ rec16, double qmax, double sigma,
                                                          unsigned m_pad, unsigned long long *__restrict__ worst_bits)
{
    const unsigned lane = threadIdx.x, col = lane & 31u, half = lane >> 5;
    const unsigned row = blockIdx.x * 32u + col, j0 = blockIdx.y * 32u;
    const double ax = px[row], ay = py[row], az = pz[row];
    const double pn = (ax * ax + ay * ay) + az * az, pl = sigma * sqrt(pn);
    if (!(pl < 16384.0)) return;   // whole-wave decision is not needed: such lanes simply do not report
    _Float16 xh, xl, yh, yl, zh, zl;
    split16(sigma * ax, xh, xl);
    split16(sigma * ay, yh, yl);
    split16(sigma * az, zh, zl);
    const _Float16 one = (_Float16)1.0f, zero = (_Float16)0.0f;
    const f16x8 b0 = {xh, xl, xh, kLean ? zero : xl, yh, yl, yh, kLean ? zero : yl};
    const f16x8 b1 = {zh, zl, zh, kLean ? zero : zl, one, one, kLean ? zero : one, zero};
    const f16x8 a = *(const f16x8 *)&rec16[(size_t)half * m_pad + j0 + col];
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, half ? b1 : b0, zero16, 0, 0, 0);
    const double tx = qx[j0], ty = qy[j0], tz = qz[j0];   // the block's one target
    const double exact = sigma * sigma * (((tx * tx + ty * ty) + tz * tz) - 2.0 * ((ax * tx + ay * ty) + az * tz));
    const double E = mfma16_bound(pl, 2.0 * sigma * qmax, sigma * sigma * qmax * qmax);
    double worst = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) worst = fmax(worst, fabs((double)acc[r] - exact) / E);
    atomicMax(worst_bits, (unsigned long long)__double_as_longlong(worst));   // non-negative doubles order like their bits
}

hipError_t mfma16_probe(hipStream_t s, Soa3 P, size_t n32, Soa3 Qz, size_t m32, size_t m_pad, double qmax, const void *rec16,
                        unsigned long long *worst_bits)
{
    if (n32 == 0 || m32 == 0 || n32 % 32 || m32 % 32 || !(qmax > 0.0)) return hipErrorInvalidValue;
    return launch(mfma16_probe_kernel, dim3((unsigned)(n32 / 32), (unsigned)(m32 / 32)), dim3(64), 0, s, P.x, P.y, P.z, Qz.x, Qz.y, Qz.z,
                  (const Half8 *)rec16, qmax, search_mfma16_scale(qmax), (unsigned)m_pad, worst_bits);
}

// sigma: the power of two that puts qmax (the largest |q| of the centred target) into [2^6, 2^7)
double search_mfma16_scale(double qmax)
{
    int e = 0;
    (void)frexp(qmax, &e);   // qmax = f * 2^e, f in [0.5, 1)
    return ldexp(1.0, 7 - e);
}

hipError_t build_qrec16(hipStream_t s, Soa3 Qz, size_t m, size_t m_pad, double qmax, void *rec16)
{
    if (m_pad == 0 || !(qmax > 0.0)) return hipErrorInvalidValue;
    return launch(build_qrec16_kernel, dim3((unsigned)((m_pad + kWG - 1) / kWG)), dim3(kWG), 0, s, Qz.x, Qz.y, Qz.z, m, m_pad,
                  search_mfma16_scale(qmax), (Half8 *)rec16);
}

hipError_t search_mfma16(hipStream_t s, int groups, Soa3 P, size_t n_pad, const double *pc, Soa3 Qz, const void *rec16, double qmax,
                         size_t m, size_t m_pad, int q_splits, const uint32_t *idx_prev, double *out_key, uint32_t *out_idx)
{
    if (n_pad == 0 || m_pad == 0 || n_pad % 2048 || m_pad % kTileQH || q_splits < 1 || m > m_pad || !(qmax > 0.0))
        return hipErrorInvalidValue;
    const unsigned m_scan = (unsigned)search_scan_rows(m);
    const unsigned q_per_split = (unsigned)search_rows_per_split(m, q_splits);
    const double sigma = search_mfma16_scale(qmax);
    if (!idx_prev) groups = 2;   // the cooperative exact path fits beside two accumulator groups without spilling, not beside four
#define ICPK_CASE(G_) \
    if (groups == G_) { \
        dim3 grid((unsigned)(n_pad / (128 * G_)), (unsigned)q_splits); \
        if (idx_prev) \
            return launch(search_mfma16_kernel<G_, false>, grid, dim3(kWG), 0, s, P.x, P.y, P.z, pc, Qz.x, Qz.y, Qz.z, \
                          (const Half8 *)rec16, qmax, sigma, (unsigned)n_pad, (unsigned)m, (unsigned)m_pad, m_scan, q_per_split, \
                          idx_prev, out_key, out_idx); \
        /* unseeded: many exact-path entries, handled by the wave cooperatively */ \
        return launch(search_mfma16_kernel<G_, true>, grid, dim3(kWG), 0, s, P.x, P.y, P.z, pc, Qz.x, Qz.y, Qz.z, \
                      (const Half8 *)rec16, qmax, sigma, (unsigned)n_pad, (unsigned)m, (unsigned)m_pad, m_scan, q_per_split, \
                      idx_prev, out_key, out_idx); \
    }
    ICPK_CASE(2) ICPK_CASE(4)
#undef ICPK_CASE
    return hipErrorInvalidValue;
}

int search_mfma_rows_per_workgroup(int groups) { return 128 * groups; }

}  // namespace icpk
