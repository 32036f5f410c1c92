// factor.hip -- tile Cholesky (potrf) and triangular solve (trsm) for gfx950.
//
//   npw_dpotrf_lower  replaces kernels.chol (reference numpywren/kernels.py:225-226)
//   npw_dtrsm_rltn    replaces kernels.trsm (reference numpywren/kernels.py:254-257)
//
// The flops live in the MFMA GEMM of gemm.hip; this file supplies the latency-critical pieces and the order:
//
//   potrf(A):   right-looking over NB-wide block columns, three launches each:
//               diagonal block (factor + invert, one workgroup, LDS-resident)  ->  panel P <- P inv(L_jj)^T
//               (one in-place GEMM)  ->  trailing update A22 -= P P^T (one GEMM over the lower tiles only).
//   trsm(X,L):  recursive:  X1 = trsm(X1, L11);  X2 -= X1 * L21^T;  X2 = trsm(X2, L22)
//               leaf:  X_j = X_j * inv(L_jj)^T   (GEMM with the cached inverse of a diagonal block: 1024-wide groups
//               as two independent products, 512-wide halves, NB-wide blocks next to ragged ends / in place)
//
// The NB x NB (128) diagonal blocks are handled by one workgroup each, entirely in LDS, as a blocked
// algorithm over 16 x 16 sub-blocks with look-ahead between the waves (see potrf_diag_kernel).  The 16 x 16
// factorisations keep one matrix row per lane and move multipliers with DPP row broadcasts; the sub-block
// updates and the block inverse run on v_mfma_f64_16x16x4_f64.  Multiplying by the explicit inverse of a
// small diagonal block instead of substituting is the standard GPU trsm formulation (the error grows with
// cond(L_jj) of the 128-wide block, not of the tile).
#include "npw_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>

#include <type_traits>

namespace npw {
namespace {

constexpr int NB = 128;   // diagonal block size handled by one workgroup
constexpr int JB = 16;    // sub-block size inside the diagonal block (one MFMA tile)
constexpr int NJB = NB / JB;
constexpr int SLD = NB + 2;  // LDS row stride (doubles): 2*SLD mod 64 == 4 -> conflict-free b64 columns
constexpr int WLD = JB + 1;
constexpr int S_ELEMS = NB * SLD;
constexpr int W_ELEMS = NJB * JB * WLD;
constexpr size_t DIAG_LDS_BYTES = (size_t)(S_ELEMS + W_ELEMS + 2) * sizeof(double);  // 150,544 B
constexpr int DIAG_THREADS = 512;
// The inverses of L's diagonal blocks ("Winv") live in groups of LW x LW (row-major, ld = LW): group g covers
// the diagonal blocks WPG g .. WPG g + WPG - 1 of size NB at (q NB, q NB) inside it.  The diagonal kernels fill the
// NB x NB blocks; complete_groups() then fills the rest of the lower triangle by doubling (every 2^k NB-wide pair that
// lies inside the matrix), so a full group is inv(L[g LW : (g+1) LW, same]) and each of its HW-wide halves the
// inverse of its own diagonal block.  The solve multiplies a whole group at a time (trsm_rec) or, next to a ragged
// end, an HW-wide half.  Everything above the diagonal is zero (memset once).  After the groups comes scratch for
// complete_groups (LW/2 x LW/2 per group).  (The error of a group's solve grows with cond of that 1024-wide block of
// L, not of the tile: tests/test_kernels_gpu.py::test_chol_trsm_ill_conditioned.)
constexpr int LW = 1024;
constexpr int HW = LW / 2;    // a group's two diagonal halves are complete inverses of their own (the solve's leaves)
constexpr int WPG = LW / NB;  // diagonal blocks per group

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2v_t __attribute__((ext_vector_type(2)));  // 16-byte chunk (HIP's double2 struct defeats SROA in register arrays)


// ---- cross-lane arithmetic inside a row of 16 lanes (DPP row_newbcast: every lane of the row reads lane K) ----
// gfx950 offers DPP on the 64-bit ALU only for v_fmac / v_mov (v_rsq_f64_dpp assembles but returns garbage on
// hardware) and only with row_newbcast, which is exactly what a 16 x 16 factorisation with one matrix row per
// lane needs: the multiplier l_kj lives in lane k.  One instruction replaces the v_readlane x2 + wait state +
// v_fma of the SGPR route.  Everything on the pivot chain is `asm volatile`: a wave issues in order, so the
// only way to hide the ~10 dependent fp64 operations between two pivots is to place them *between* the
// independent broadcast-FMAs by hand; left to the scheduler they end up back to back.
// Hazards the assembler does not see inside inline asm: a DPP source written by the preceding VALU needs two
// wait states (NOP = true where that can happen); the consumer of a transcendental result needs one.
template <int K, bool NOP>
__device__ inline void fnma_bcast(double& acc, double from_lane_k, double mine) {  // acc -= lane_K(from) * mine
    if constexpr (NOP)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc)
                     : "v"(from_lane_k), "v"(mine), "n"(K));
    else
        asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc)
                     : "v"(from_lane_k), "v"(mine), "n"(K));
}
template <int K>
__device__ inline double mov_bcast(double x) {
    double y;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(y) : "v"(x), "n"(K));
    return y;
}
__device__ inline double op_rsq(double x) {
    double y;
    asm volatile("v_rsq_f64 %0, %1\n\ts_nop 0" : "=v"(y) : "v"(x));
    return y;
}
__device__ inline double op_mul(double a, double b) {
    double y;
    asm volatile("v_mul_f64 %0, %1, %2" : "=v"(y) : "v"(a), "v"(b));
    return y;
}
__device__ inline double op_half(double a) {
    double y;
    asm volatile("v_mul_f64 %0, %1, 0.5" : "=v"(y) : "v"(a));
    return y;
}
__device__ inline double op_fma(double a, double b, double c) {
    double y;
    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(y) : "v"(a), "v"(b), "v"(c));
    return y;
}
__device__ inline double op_one_minus(double a, double b) {  // 1 - a * b
    double y;
    asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(y) : "v"(a), "v"(b));
    return y;
}

template <int I, int N, typename F>
__device__ inline void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// 1/sqrt(d) from the hardware estimate y0 by two coupled (Goldschmidt) steps, cut into stages so that the
// caller can spread them between independent instructions:
//     g0 = d y0 (~sqrt d)   r0 = 1 - g0 y0   y1 = y0 + (y0/2) r0   g1 = g0 + (g0/2) r0
//     r1 = 1 - g1 y1        rinv = y1 + (y1/2) r1
// g and y carry the same relative error (g/y = d up to rounding), so r measures it and each step squares it:
// ~1 ulp after two steps from any estimate better than 2^-14.  Dependent depth: rsq + 6.
struct PivotChain {
    double d, y0, g0, hh0, r0, gg, y1, g1, hh1, r1, rinv;
    static constexpr int STAGES = 9;
    template <int S>
    __device__ inline void stage() {
        if constexpr (S == 0) g0 = op_mul(d, y0);
        if constexpr (S == 1) hh0 = op_half(y0);
        if constexpr (S == 2) r0 = op_one_minus(g0, y0);
        if constexpr (S == 3) gg = op_half(g0);
        if constexpr (S == 4) y1 = op_fma(hh0, r0, y0);
        if constexpr (S == 5) g1 = op_fma(gg, r0, g0);
        if constexpr (S == 6) hh1 = op_half(y1);
        if constexpr (S == 7) r1 = op_one_minus(g1, y1);
        if constexpr (S == 8) rinv = op_fma(hh1, r1, y1);
    }
};

// Cholesky of the 16 x 16 diagonal sub-block *and* the solve of 64 rows below it, in one instruction stream.
// Lane (g, i) (g = lane >> 4, i = lane & 15) holds row i of the diagonal sub-block in a[0..15] -- the four
// 16-lane rows carry identical copies, so every DPP row sees the whole sub-block -- and one row of the panel
// in x[0..15].  Column step j:
//     pivot d = a_jj (lane j), rinv = 1/sqrt(d);   a[j] *= rinv;  x[j] *= rinv       (l_jj = d * rinv)
//     a[k] -= l_kj * a[j],  x[k] -= l_kj * x[j]   for k > j,   l_kj = a[j] of lane k  (DPP broadcast)
// which is right-looking Cholesky on a[] and the forward substitution  X L^T = P  on x[].  Column j+1 is
// updated first, its pivot chain is then issued in stages between the remaining updates of step j.
// `bad` / `bad_col` are per-lane copies of wave-uniform values (first non-positive pivot).
__device__ inline void chol16_panel(double (&a)[JB], double (&x)[JB], bool& bad, int& bad_col) {
    bad = false;
    bad_col = 0;
    PivotChain c;
    c.d = mov_bcast<0>(a[0]);
    c.y0 = op_rsq(c.d);
    static_for<0, PivotChain::STAGES>([&](auto S) { c.template stage<decltype(S)::value>(); });
    static_for<0, JB>([&](auto J) {
        constexpr int j = decltype(J)::value;
        if (!bad && !(c.d > 0.0)) {
            bad = true;
            bad_col = j;
        }
        a[j] = op_mul(a[j], c.rinv);
        x[j] = op_mul(x[j], c.rinv);
        if constexpr (j + 1 < JB) {
            fnma_bcast<j + 1, true>(a[j + 1], a[j], a[j]);
            fnma_bcast<j + 1, false>(x[j + 1], a[j], x[j]);
            c.d = mov_bcast<j + 1>(a[j + 1]);
            c.y0 = op_rsq(c.d);
            static_for<j + 2, JB>([&](auto Kc) {
                constexpr int k = decltype(Kc)::value;
                fnma_bcast<k, false>(a[k], a[j], a[j]);
                fnma_bcast<k, false>(x[k], a[j], x[j]);
                constexpr int s = k - (j + 2);
                if constexpr (s < PivotChain::STAGES) c.template stage<s>();
            });
            constexpr int done = (JB - (j + 2)) < 0 ? 0 : JB - (j + 2);
            static_for<(done < PivotChain::STAGES ? done : PivotChain::STAGES), PivotChain::STAGES>(
                [&](auto S) { c.template stage<decltype(S)::value>(); });
        }
    });
}

// Wd[jb] = inverse of the (final) diagonal sub-block jb of L in S; one wave (all four DPP rows compute the
// same thing, row 0 stores).  Lane c builds column c of W = inv(L) right-looking:
//     w = e_c;   for k: w[k] /= l_kk;  w[r] -= l_rk * w[k]  (r > k),   l_rk = a[k] of lane r  (DPP broadcast)
__device__ inline void invert_diag16(const double* S, double* Wd, int jb, int lane) {
    const int li = lane & 15;
    double a[JB], w[JB];
    const double2* p = reinterpret_cast<const double2*>(S + (jb * JB + li) * SLD + jb * JB);
#pragma unroll
    for (int k = 0; k < JB / 2; ++k) {
        const double2 v = p[k];
        a[2 * k] = v.x;
        a[2 * k + 1] = v.y;
    }
#pragma unroll
    for (int k = 0; k < JB; ++k) w[k] = (k == li) ? 1.0 : 0.0;
    static_for<0, JB>([&](auto Kc) {
        constexpr int k = decltype(Kc)::value;
        const double d = mov_bcast<k>(a[k]);
        double y = __builtin_amdgcn_rcp(d);
        y = y * fma(-d, y, 2.0);
        y = y * fma(-d, y, 2.0);
        w[k] *= y;
        static_for<k + 1, JB>([&](auto R) {
            constexpr int r = decltype(R)::value;
            fnma_bcast<r, (r == k + 1)>(w[r], a[k], w[k]);
        });
    });
    if (lane < JB) {
#pragma unroll
        for (int k = 0; k < JB; ++k) Wd[(jb * JB + k) * WLD + li] = w[k];  // W[r=k][c=li]
    }
}

// The NB x NB block at `src` (lower triangle) -> S, identity-padded to the full NB x NB when n < NB.
// All of a thread's loads are issued before the first LDS store (one memory round trip, not 32).
__device__ inline void load_lower_block(double* S, const double* src, int64_t ld, int n, int tid) {
    constexpr int PER = NB * NB / DIAG_THREADS;  // 32
    constexpr int RSTEP = DIAG_THREADS / NB;     // 4
    const int c = tid & (NB - 1), r0 = tid >> 7;
    double v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int r = r0 + RSTEP * i;
        v[i] = (r == c) ? 1.0 : 0.0;
        if (c <= r && r < n) v[i] = src[(int64_t)r * ld + c];
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) S[(r0 + RSTEP * i) * SLD + c] = v[i];
}

// NP sub-block updates  A(ib,kb) -= L(ib,jb) L(kb,jb)^T  by one wave: every operand is fetched before the
// first MFMA and the NP accumulation chains are interleaved.
template <int NP>
__device__ inline void update_tiles(double* S, int jb, const int (&ib)[NP], const int (&kb)[NP], const bool (&on)[NP],
                                    int li, int lg) {
    d4_t acc[NP];
    double a[NP][4], b[NP][4];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        if (!on[q]) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[q][r] = S[(ib[q] * JB + lg + 4 * r) * SLD + kb[q] * JB + li];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            a[q][st] = -S[(ib[q] * JB + li) * SLD + jb * JB + 4 * st + lg];
            b[q][st] = S[(kb[q] * JB + li) * SLD + jb * JB + 4 * st + lg];
        }
    }
#pragma unroll
    for (int st = 0; st < 4; ++st) {
#pragma unroll
        for (int q = 0; q < NP; ++q)
            if (on[q]) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][st], b[q][st], acc[q], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        if (!on[q]) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(ib[q] * JB + lg + 4 * r) * SLD + kb[q] * JB + li] = acc[q][r];
    }
}

// (row, column) of the p-th entry of a lower triangle enumerated row by row
__constant__ unsigned char TRI_ROW[21] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5};
__constant__ unsigned char TRI_COL[21] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 5};

// Block column jb of the diagonal block: factor its 16 x 16 diagonal sub-block and solve the rows below.
// Waves 0 and 1 each factor the diagonal sub-block redundantly (see chol16_panel) next to 64 panel rows, so
// the two instruction streams never have to talk to each other.
// The diagonal rows (lanes 0..15 of wave 0) stay in a[] on return: the caller stores them after the next
// barrier, because wave 1 may still be loading the unfactored diagonal sub-block.
__device__ inline void panel_step(double* S, int jb, int wave, int lane, int32_t* info, int base, int* flag,
                                  double (&a)[JB]) {
    const int first = (jb + 1) * JB + wave * 64;  // first panel row of this wave
    if (wave > 0 && first >= NB) return;
    const int li = lane & 15;
    const int row = first + lane;
    const bool valid = row < NB;
    const double2* pd = reinterpret_cast<const double2*>(S + (jb * JB + li) * SLD + jb * JB);
    double2* px = reinterpret_cast<double2*>(S + (valid ? row : jb * JB) * SLD + jb * JB);
    double x[JB];
#pragma unroll
    for (int k = 0; k < JB / 2; ++k) {
        const double2 v = pd[k], w = px[k];
        a[2 * k] = v.x;
        a[2 * k + 1] = v.y;
        x[2 * k] = w.x;
        x[2 * k + 1] = w.y;
    }
    bool bad;
    int bad_col;
    chol16_panel(a, x, bad, bad_col);
    if (bad) {
        if (wave == 0 && lane == 0) {
            atomicCAS(info, 0, base + jb * JB + bad_col + 1);
            *flag = 1;
        }
        return;
    }
    if (!valid) return;
#pragma unroll
    for (int k = 0; k < JB / 2; ++k) px[k] = make_double2(x[2 * k], x[2 * k + 1]);
}

// the deferred half of panel_step: L(jb, jb) from the registers of wave 0's diagonal lanes
__device__ inline void store_diag_rows(double* S, int jb, int lane, const double (&a)[JB]) {
    double2* p = reinterpret_cast<double2*>(S + (jb * JB + lane) * SLD + jb * JB);
#pragma unroll
    for (int k = 0; k < JB / 2; ++k)
        p[k] = make_double2(2 * k <= lane ? a[2 * k] : 0.0, 2 * k + 1 <= lane ? a[2 * k + 1] : 0.0);
}

// One 16 x 16 tile of  L21 * X11  (stage 1 of a doubling step), written to the mirror position (j, i) in
// the unused upper triangle of S.  X11 = inverse of the diagonal block that spans sub-blocks [.., t_end).
__device__ inline void inv_stage1(double* S, const double* Wd, int i, int j, int t_end, int li, int lg) {
    d4_t acc = {0, 0, 0, 0};
    for (int k = j; k < t_end; ++k) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const double a = S[(i * JB + li) * SLD + k * JB + 4 * st + lg];
            const double b = (k == j) ? Wd[(j * JB + 4 * st + lg) * WLD + li] : S[(k * JB + 4 * st + lg) * SLD + j * JB + li];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) S[(j * JB + lg + 4 * r) * SLD + i * JB + li] = acc[r];
}

// One tile of  X21 = -X22 * (L21 X11)  (stage 2): X22 spans sub-blocks [b0, ..], the product comes from
// the mirror tiles; the result overwrites L(i, j).
__device__ inline void inv_stage2(double* S, const double* Wd, int i, int j, int b0, int li, int lg) {
    d4_t acc = {0, 0, 0, 0};
    for (int k = b0; k <= i; ++k) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const double a = (k == i) ? Wd[(i * JB + li) * WLD + 4 * st + lg] : S[(i * JB + li) * SLD + k * JB + 4 * st + lg];
            const double b = S[(j * JB + 4 * st + lg) * SLD + k * JB + li];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a, b, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) S[(i * JB + lg + 4 * r) * SLD + j * JB + li] = acc[r];
}

// Given L (lower, in S) and the inverses of its eight 16 x 16 diagonal sub-blocks (in Wd), overwrite the
// strictly-lower sub-blocks of S with those of X = inv(L) by recursive doubling:
//   inv [L11 0; L21 L22] = [X11 0; -X22 L21 X11, X22]       16 -> 32 -> 64 -> 128
// Five barrier-separated stages; the strict upper triangle of S is scratch.  All 8 waves, block-uniform.
__device__ inline void block_trtri(double* S, const double* Wd, int wave, int li, int lg) {
    // 16 -> 32: four pairs, both products in registers (the D layout of the first is the B layout of the second)
    if (wave < 4) {
        const int j = 2 * wave, i = j + 1;
        d4_t acc = {0, 0, 0, 0}, res = {0, 0, 0, 0};
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const double a = S[(i * JB + li) * SLD + j * JB + 4 * st + lg];
            const double b = Wd[(j * JB + 4 * st + lg) * WLD + li];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const double a = -Wd[(i * JB + li) * WLD + 4 * st + lg];
            res = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[st], res, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(i * JB + lg + 4 * r) * SLD + j * JB + li] = res[r];
    }
    __syncthreads();
    // 32 -> 64: two pairs x (2 x 2) tiles, one per wave
    {
        const int t0 = 4 * (wave >> 2), b0 = t0 + 2;
        const int i = b0 + ((wave >> 1) & 1), j = t0 + (wave & 1);
        inv_stage1(S, Wd, i, j, b0, li, lg);
        __syncthreads();
        inv_stage2(S, Wd, i, j, b0, li, lg);
    }
    __syncthreads();
    // 64 -> 128: (4 x 4) tiles, two per wave, paired so that every wave sums five sub-block products
    {
        const int j = wave & 3, hi = wave >> 2;
        inv_stage1(S, Wd, 4 + hi, j, 4, li, lg);
        inv_stage1(S, Wd, 6 + hi, 3 - j, 4, li, lg);
        __syncthreads();
        inv_stage2(S, Wd, 4 + hi, j, 4, li, lg);
        inv_stage2(S, Wd, 7 - hi, j, 4, li, lg);
    }
    __syncthreads();
}

__host__ __device__ inline size_t w_block_offset(int64_t b) {  // element offset of diagonal block b inside Winv
    return (size_t)(b / WPG) * LW * LW + (size_t)(b % WPG) * NB * (LW + 1);
}

// write inv(L) (diagonal sub-blocks from Wd, strictly-lower ones from S) row-major, ld = LW; rows >= n are zero
// coherent: write through to device-coherent memory (relaxed agent-scope atomics) so that workgroups on other
// XCDs can read the block *during this kernel* without an L2 write-back fence (see potrf_panel_kernel).
__device__ inline void store_inverse(const double* S, const double* Wd, int n, double* Winv, int tid, bool coherent = false) {
    const int c = tid & (NB - 1), r0 = tid >> 7;
#pragma unroll 8
    for (int i = 0; i < NB * NB / DIAG_THREADS; ++i) {
        const int r = r0 + (DIAG_THREADS / NB) * i;
        double v = 0.0;
        if (r < n && c <= r) v = ((r ^ c) < JB) ? Wd[r * WLD + (c & (JB - 1))] : S[r * SLD + c];
        if (coherent)
            __hip_atomic_store(&Winv[r * LW + c], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            Winv[r * LW + c] = v;
    }
}

#ifdef NPW_DIAG_STAMPS
__device__ long long* g_diag_stamps = nullptr;
#endif

// Factor the n x n (n <= NB) diagonal block A (lower triangle used) in place: on exit the lower triangle
// holds L, the strict upper triangle is zero; Winv receives inv(L).  One workgroup of 8 waves, everything
// in LDS.  Per 16-wide block column jb the serial chain is only
//     U_col(jb-1) -> barrier -> chol16 + panel solve (waves 0-1) -> barrier
// the rest of the rank-16 update (waves 2-6) and the inversion of the finished diagonal sub-block
// (wave 7) run beside the next column's factorisation (look-ahead inside the workgroup).
typedef double pslot_fwd_t __attribute__((ext_vector_type(2)));
__device__ inline void publish_step(const double* S, const double* Wd, int jb, int lane, pslot_fwd_t* msg, unsigned long long tag);
__device__ inline void publish_step_l(const double* S, int jb, int lane, pslot_fwd_t* msg, unsigned long long tag);

__device__ inline void diag_block(int n, double* A, int64_t lda, int32_t* info, int base, double* Winv, double* S,
                                  bool coherent, pslot_fwd_t* msg = nullptr, unsigned long long msg_tag = 0) {
    double* Wd = S + S_ELEMS;
    int* flag = reinterpret_cast<int*>(Wd + W_ELEMS);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int nbk = (n + JB - 1) / JB;
#ifdef NPW_DIAG_STAMPS  // developer timing: wall-clock stamps (10 ns units) of the phases, see tools/
#define NPW_STAMP(i) if (g_diag_stamps && tid == 0) g_diag_stamps[i] = wall_clock64();
#else
#define NPW_STAMP(i)
#endif
    NPW_STAMP(0)
#ifdef NPW_DIAG_STAMPS
    if (g_diag_stamps && tid == 0) g_diag_stamps[22] = clock64();
#endif

    load_lower_block(S, A, lda, n, tid);
    if (tid == 0) *flag = (*info != 0) ? 1 : 0;
    __syncthreads();
    bool failed = (*flag != 0);  // an earlier block of the same matrix already failed
    NPW_STAMP(1)

    double a[JB];
    if (!failed) {
        if (wave < 2) panel_step(S, 0, wave, lane, info, base, flag, a);
        for (int jb = 0; jb < nbk; ++jb) {
            __syncthreads();  // column jb is final; all updates of step jb-1 have landed
            NPW_STAMP(2 + 2 * jb)
            if (*flag != 0) {
                failed = true;
                break;
            }
            if (wave == 0 && lane < JB) store_diag_rows(S, jb, lane, a);
            const int t = nbk - jb - 1;  // sub-block rows below the diagonal one
            if (wave < t) {              // U_col: block column jb+1 only (t <= 7 tiles, one per wave)
                const int ib[1] = {jb + 1 + wave}, kb[1] = {jb + 1};
                const bool on[1] = {true};
                update_tiles<1>(S, jb, ib, kb, on, li, lg);
            }
            // wave 7 is idle in this half step: it sends the L[jb, jb-1] half of message jb now (final since the
            // previous step; the panel workgroups act on a message only when all of it, inv(D_jb) included, is there)
            if (wave == 7 && msg != nullptr) publish_step_l(S, jb, lane, msg, msg_tag);
            // progressive mode: the stores of block column jb-1 (issued in the previous half step) have been
            // acknowledged before anybody publishes message jb
            if (msg != nullptr) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            NPW_STAMP(3 + 2 * jb)
            if (wave < 2) {
                if (t > 0) panel_step(S, jb + 1, wave, lane, info, base, flag, a);
            } else if (wave < 7) {       // U_rest: block columns >= jb+2, beside the next panel
                const int t2 = t - 1, tw = wave - 2;
                const int npairs = t2 * (t2 + 1) / 2;
                for (int q0 = 0; q0 < 6 && tw + 5 * q0 < npairs; q0 += 3) {
                    int ib[3], kb[3];
                    bool on[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const int pr = tw + 5 * (q0 + q);
                        on[q] = pr < npairs;
                        const int prc = on[q] ? pr : 0;
                        ib[q] = jb + 2 + TRI_ROW[prc];
                        kb[q] = jb + 2 + TRI_COL[prc];
                    }
                    update_tiles<3>(S, jb, ib, kb, on, li, lg);
                }
            } else {
                invert_diag16(S, Wd, jb, lane);
                if (msg != nullptr) publish_step(S, Wd, jb, lane, msg, msg_tag);
            }
            if (wave >= 2) {
                // Block column jb of the 128 x 128 block is final (L below the diagonal sub-block, zeros above it):
                // waves 2..7 send it home now, 8-row groups dealt round robin.  Asynchronous stores beside the next
                // column's factorisation -- a separate store pass at the end cost 2.5 us per block.
                const int cp = jb * JB + (lane & 7) * 2;
                const bool vec = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && ((lda & 1) == 0);
                for (int r = (wave - 2) * 8 + (lane >> 3); r < n; r += 48) {
                    if (vec && cp + 1 < n) {
                        const double2 v2 = *reinterpret_cast<const double2*>(&S[r * SLD + cp]);
                        if (msg != nullptr) {  // read by the panel workgroups during this kernel: write through
                            pslot_fwd_t x;
                            x[0] = v2.x;
                            x[1] = v2.y;
                            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(&A[(int64_t)r * lda + cp]), "v"(x) : "memory");
                        } else {
                            *reinterpret_cast<double2*>(&A[(int64_t)r * lda + cp]) = v2;
                        }
                    } else if (msg != nullptr) {  // unaligned rows: scalar write-through stores
                        if (cp < n) __hip_atomic_store(&A[(int64_t)r * lda + cp], S[r * SLD + cp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (cp + 1 < n)
                            __hip_atomic_store(&A[(int64_t)r * lda + cp + 1], S[r * SLD + cp + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        if (cp < n) A[(int64_t)r * lda + cp] = S[r * SLD + cp];
                        if (cp + 1 < n) A[(int64_t)r * lda + cp + 1] = S[r * SLD + cp + 1];
                    }
                }
            }
        }
    }
    __syncthreads();
    NPW_STAMP(18)
    if (failed) {
        for (int idx = tid; idx < NB * NB; idx += DIAG_THREADS)
            __hip_atomic_store(&Winv[(idx / NB) * LW + (idx % NB)], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (msg != nullptr) {
            // the panel workgroups are waiting for messages that will not come: release them (zeros; the matrix is
            // reported as not positive definite, its contents are unspecified)
            for (int k = 0; k < NJB; ++k) {
                pslot_fwd_t x;
                x[0] = 0.0;
                x[1] = __longlong_as_double((long long)(msg_tag + k));
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(msg + (size_t)k * 2 * JB * JB + tid), "v"(x) : "memory");
            }
        }
        return;
    }
    // Progressive mode: nobody in this launch needs inv(L_jj) -- potrf_right inverts all diagonal blocks of the finished
    // factor with ONE trtri_diag_kernel launch instead of 5.4 + 1.9 us at the end of every block column.
    if (msg != nullptr) return;
    if (wave >= nbk && lane < JB) {  // identity padding blocks
#pragma unroll
        for (int k = 0; k < JB; ++k) Wd[(wave * JB + k) * WLD + li] = (k == li) ? 1.0 : 0.0;
    }
    __syncthreads();
    NPW_STAMP(19)
    block_trtri(S, Wd, wave, li, lg);
    NPW_STAMP(20)
    store_inverse(S, Wd, n, Winv, tid, coherent);
    NPW_STAMP(21)
#ifdef NPW_DIAG_STAMPS
    if (g_diag_stamps && tid == 0) g_diag_stamps[23] = clock64();
#endif
}

__global__ __launch_bounds__(DIAG_THREADS) void potrf_diag_kernel(int n, double* A, int64_t lda, int32_t* info, int base,
                                                                  double* Winv) {
    extern __shared__ __attribute__((aligned(16))) double S[];
    diag_block(n, A, lda, info, base, Winv, S, false);
}

// ---- fused panel:  rows below the diagonal block  <-  rows * inv(L_jj)^T  -------------------------------------
// Workgroups 1.. of potrf_panel_kernel each own PROWS rows of the block column.  They pull their rows into LDS
// while workgroup 0 is still factoring the diagonal block, spin on a flag until inv(L_jj) is in memory, then
// multiply from LDS (A operand) and straight from L2 (B operand = inv(L_jj), 128 KiB shared by all of them).
// Compared with a separate GEMM launch this removes the launch gap, the first global round trip and the
// LDS staging of a matrix that is used once: ~4 us after the flag instead of ~14 us.
constexpr int PROWS = 32;  // rows per panel workgroup: 16 output sub-blocks, two per wave (the matrix pipes of one CU
                           // need 64 cycles per fp64 MFMA, so 64-row strips spent 3.8 us on 576 of them)

constexpr int WPB = JB * (JB + 1);              // one packed 16 x 16 block of inv(L_jj), row stride 17
constexpr int WP_BLOCKS = NJB * (NJB + 1) / 2;  // 36 blocks on or below the diagonal
__constant__ unsigned char WP_ROW[WP_BLOCKS] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4, 5, 5, 5,
                                                5, 5, 5, 6, 6, 6, 6, 6, 6, 6, 7, 7, 7, 7, 7, 7, 7, 7};
__constant__ unsigned char WP_COL[WP_BLOCKS] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4, 0, 1, 2,
                                                3, 4, 5, 0, 1, 2, 3, 4, 5, 6, 0, 1, 2, 3, 4, 5, 6, 7};

// X(rb, NBLK) = sum_{kb <= NBLK} P(rb, kb) * W(NBLK, kb)^T, both operands from LDS
template <int NBLK>
__device__ inline void panel_tile(const double* S, const double* Wp, double* P, int64_t lda, int rows, int rb, int li, int lg) {
    d4_t acc = {0, 0, 0, 0};
    static_for<0, NBLK + 1>([&](auto KB) {
        constexpr int kb = decltype(KB)::value;
        constexpr int blk = NBLK * (NBLK + 1) / 2 + kb;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const double a = S[(rb * JB + li) * SLD + kb * JB + 4 * st + lg];
            const double b = Wp[blk * WPB + li * (JB + 1) + 4 * st + lg];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
    });
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = rb * JB + lg + 4 * r;
        if (row < rows) P[(int64_t)row * lda + NBLK * JB + li] = acc[r];
    }
}

__device__ inline void panel_rows(double* S, double* P, int64_t lda, int rows, const double* W, const int* ready,
                                  int ready_val) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    double* Wp = S + PROWS * SLD;
    {
        constexpr int PER = PROWS * NB / DIAG_THREADS;  // 16
        const int c = tid & (NB - 1), r0 = tid >> 7;
        double v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int r = r0 + (DIAG_THREADS / NB) * i;
            v[i] = (r < rows) ? P[(int64_t)r * lda + c] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) S[(r0 + (DIAG_THREADS / NB) * i) * SLD + c] = v[i];
    }
    if (tid == 0) {
        while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ready_val) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
    // inv(L_jj) -> LDS, lower 16 x 16 blocks only, 128-byte row segments (a strided read straight into the MFMA
    // operand registers puts every 32-byte sector of a 4 KiB-strided row on the same L2 channel: 19 us).
    // Plain loads are safe without an acquire: nothing on this CU or XCD has touched these lines since the
    // kernel started (caches are invalidated at kernel boundaries), and workgroup 0 wrote them through.
    {
        constexpr int PER = WP_BLOCKS * JB * JB / DIAG_THREADS;  // 18
        double v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = tid + DIAG_THREADS * i;
            const int blk = e >> 8, r = (e >> 4) & 15, c = e & 15;
            v[i] = W[(int64_t)(WP_ROW[blk] * JB + r) * LW + WP_COL[blk] * JB + c];
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = tid + DIAG_THREADS * i;
            Wp[(e >> 8) * WPB + ((e >> 4) & 15) * (JB + 1) + (e & 15)] = v[i];
        }
    }
    __syncthreads();
    // wave -> one 16-row block and two 16-column blocks (nb, 7 - nb): every wave sums 9 sub-block products
    const int rb = wave & 1;
    switch (wave >> 1) {
        case 0:
            panel_tile<0>(S, Wp, P, lda, rows, rb, li, lg);
            panel_tile<7>(S, Wp, P, lda, rows, rb, li, lg);
            break;
        case 1:
            panel_tile<1>(S, Wp, P, lda, rows, rb, li, lg);
            panel_tile<6>(S, Wp, P, lda, rows, rb, li, lg);
            break;
        case 2:
            panel_tile<2>(S, Wp, P, lda, rows, rb, li, lg);
            panel_tile<5>(S, Wp, P, lda, rows, rb, li, lg);
            break;
        default:
            panel_tile<3>(S, Wp, P, lda, rows, rb, li, lg);
            panel_tile<4>(S, Wp, P, lda, rows, rb, li, lg);
            break;
    }
}

// ------------------------------------------------------------------------------------------------
// Progressive hand-off (the fused kernel's default): the panel workgroups do not wait for inv(L_jj).  After step k of the
// diagonal block's factorisation workgroup 0 publishes a MESSAGE k = { inv(D_k) (the 16 x 16 diagonal sub-block's
// inverse), L[k, k-1] } as 512 tagged 16-byte slots ({value, tag}: arrival of the data is the synchronisation, as in
// qr.hip), and the panel rows advance by substitution,
//     X_k = (B_k - sum_{i<k} X_i L[k,i]^T) inv(D_k)^T,
// one 16-column block behind the factorisation instead of a whole block inverse (5.4 us) + a 128-wide product behind
// it.  L[k, 0..k-2] is prefetched with plain loads one step earlier: message k-1 is published after every wave of
// workgroup 0 has seen the stores of column blocks <= k-2 acknowledged (s_waitcnt before the barrier that precedes it).
// ------------------------------------------------------------------------------------------------
typedef double pslot_t __attribute__((ext_vector_type(2)));  // {value, tag bits}
constexpr int MSG_SLOTS = 2 * JB * JB;                        // inv(D_k) then L[k, k-1]
constexpr int PGROWS = 64;                                    // rows per panel workgroup in this mode (4 computing waves)
constexpr int LRD = NB - JB + 1;                              // row stride of the prefetched L[k, 0..k-1) rows (113)

// (s_nop after the store: see st_slot in qr.hip -- a wide store still reads its data registers after issue)
__device__ inline void st_pslot(pslot_t* p, double v, unsigned long long tag) {
    pslot_t x;
    x[0] = v;
    x[1] = __longlong_as_double((long long)tag);
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}
__device__ inline double ld_pslot(const pslot_t* p, unsigned long long tag) {
    for (int spin = 0; spin < (1 << 22); ++spin) {  // bounded: a lost message must not hang the GPU
        pslot_t x;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(p) : "memory");
        if (__double_as_longlong(x[1]) == (long long)tag) return x[0];
        __builtin_amdgcn_s_sleep(1);
    }
    return 0.0;
}

// workgroup 0, wave 7, step jb: message jb in two halves -- L[jb, jb-1] (final since the previous step) before the
// wave inverts the diagonal sub-block, inv(D_jb) right after
__device__ inline void publish_step_l(const double* S, int jb, int lane, pslot_t* msg, unsigned long long tag) {
    if (jb == 0) return;
    pslot_t* m = msg + (size_t)jb * MSG_SLOTS + JB * JB;
    double v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = lane + 64 * i;
        v[i] = S[(jb * JB + (e >> 4)) * SLD + (jb - 1) * JB + (e & 15)];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) st_pslot(m + lane + 64 * i, v[i], tag + jb);
}
__device__ inline void publish_step(const double* S, const double* Wd, int jb, int lane, pslot_t* msg, unsigned long long tag) {
    pslot_t* m = msg + (size_t)jb * MSG_SLOTS;
    double v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = lane + 64 * i;
        v[i] = Wd[(jb * JB + (e >> 4)) * WLD + (e & 15)];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) st_pslot(m + lane + 64 * i, v[i], tag + jb);
}

__device__ inline void panel_rows_prog(double* S, double* P, int64_t lda, int rows, const double* Ldiag, const pslot_t* msg,
                                       unsigned long long tag) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    double* Lrow = S + PGROWS * SLD;            // [2][JB][LRD]: L[k, 0 .. k-1) of the current / next step
    double* Msg = Lrow + 2 * JB * LRD;          // [2][2][JB][WLD]: inv(D_k), L[k, k-1]
    {
        constexpr int PER = PGROWS * NB / DIAG_THREADS;  // 16
        const int c = tid & (NB - 1), r0 = tid >> 7;
        double v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int r = r0 + (DIAG_THREADS / NB) * i;
            v[i] = (r < rows) ? P[(int64_t)r * lda + c] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) S[(r0 + (DIAG_THREADS / NB) * i) * SLD + c] = v[i];
    }
    for (int k = 0; k < NJB; ++k) {
        const int par = k & 1;
        double* Mk = Msg + par * 2 * JB * WLD;
        {   // one slot per thread: 256 of inv(D_k), 256 of L[k, k-1]
            const int e = tid & (JB * JB - 1), half = tid >> 8;
            if (half == 0 || k > 0) {
                const double v = ld_pslot(msg + (size_t)k * MSG_SLOTS + tid, tag + k);
                Mk[half * JB * WLD + (e >> 4) * WLD + (e & 15)] = v;
            }
        }
        __syncthreads();
#ifdef NPW_PP_DEBUG_L
        if (k > 0 && tid >= 256) {
            const int e = tid & 255;
            const double g = Ldiag[(int64_t)(k * JB + (e >> 4)) * lda + (k - 1) * JB + (e & 15)];
            const double mv = Mk[JB * WLD + (e >> 4) * WLD + (e & 15)];
            if (blockIdx.x == 1 && k <= 2 && mv != g)
                printf("k=%d r=%d c=%d msg=%.17g global=%.17g diff=%.3e\n", k, e >> 4, e & 15, mv, g, mv - g);
        }
        __syncthreads();
#endif
        if (wave < PGROWS / JB) {
            const int rb = wave;
            d4_t acc = {0, 0, 0, 0};
            const double* Lr = Lrow + par * JB * LRD;
            for (int i = 0; i + 1 < k; ++i) {   // prefetched blocks L[k, i], i <= k - 2
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const double a = S[(rb * JB + li) * SLD + i * JB + 4 * st + lg];
                    const double b = Lr[li * LRD + i * JB + 4 * st + lg];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                }
            }
            if (k > 0) {
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const double a = S[(rb * JB + li) * SLD + (k - 1) * JB + 4 * st + lg];
                    const double b = Mk[JB * WLD + li * WLD + 4 * st + lg];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                }
            }
            // T = B_k - acc, back into the strip (this wave's rows only: LDS is in order within a wave)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double* q = &S[(rb * JB + lg + 4 * r) * SLD + k * JB + li];
                *q = *q - acc[r];
            }
            d4_t x = {0, 0, 0, 0};
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const double a = S[(rb * JB + li) * SLD + k * JB + 4 * st + lg];
                const double b = Mk[li * WLD + 4 * st + lg];   // inv(D_k)[c = li][j]
                x = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, x, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rb * JB + lg + 4 * r;
                S[row * SLD + k * JB + li] = x[r];
                if (row < rows) P[(int64_t)row * lda + k * JB + li] = x[r];
            }
        } else if (k + 1 < NJB && k > 0) {
            // waves 4..7: L[k+1, 0 .. k) for the next step -- visible now that message k is here
            double* Ln = Lrow + (par ^ 1) * JB * LRD;
            const int t = tid - (PGROWS / JB) * 64;    // 0..255
            for (int e = t; e < JB * JB * k; e += DIAG_THREADS - (PGROWS / JB) * 64) {
                const int r = e / (JB * k), c = e - r * (JB * k);
                Ln[r * LRD + c] = Ldiag[(int64_t)((k + 1) * JB + r) * lda + c];
            }
        }
    }
}

// One block column of the right-looking factorisation in one launch: workgroup 0 = diagonal block (factor +
// invert), workgroups 1.. = the rows below it.  All workgroups are resident at once (<= 63, one per CU; the
// dispatcher starts workgroup 0 first), so the flag wait cannot deadlock.
__global__ __launch_bounds__(DIAG_THREADS) void potrf_panel_kernel(int n, double* A, int64_t lda, int32_t* info, int base,
                                                                   double* Winv, int m_below, pslot_t* msg,
                                                                   unsigned long long msg_tag) {
    extern __shared__ __attribute__((aligned(16))) double S[];
    if (blockIdx.x == 0) {
        // factor + publish a message per 16-column step; then inv(L_jj) for the trsm tasks (nobody in this kernel waits
        // for it)
        diag_block(n, A, lda, info, base, Winv, S, false, msg, msg_tag);
        return;
    }
    const int r0 = (blockIdx.x - 1) * PGROWS;
    double* P = A + (int64_t)(n + r0) * lda;  // rows below the diagonal block, same columns
    panel_rows_prog(S, P, lda, min(PGROWS, m_below - r0), A, msg, msg_tag);
#ifdef NPW_DIAG_STAMPS
    if (g_diag_stamps && threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) g_diag_stamps[24] = wall_clock64();
#endif
}

// ================================================================================================
// Look-ahead block column: ONE launch per 128-wide block column that overlaps the latency-bound chain
// (diagonal block + panel rows) of column j with the rank-128 trailing update left over from column j-1.
//
//   role A  "strips"   (every workgroup, first): block column j itself still lacks the update with panel j-1.  It is cut
//                      into 16-row strips, one per workgroup (<= 248 for a 4096^2 tile: all run at once),
//                      C[16 x 128] -= P[strip rows] * P[rows of diagonal block j]^T,  K = 128, one MFMA tile per wave.
//                      Results are written through (sc1) and announced per 64-row group with one counter increment.
//   role B  "chain"    (workgroups 0 .. m/64): exactly the fused kernel above -- workgroup 0 factors the diagonal
//                      block and publishes a message per 16-column step, the others follow with their 64 panel rows
//                      by substitution -- after the counters of their own rows have reached 4.
//   role C  "trailing" (every workgroup, when it has nothing else to do): 128 x 128 tiles of the lower triangle right
//                      of block column j,  C -= P[tile rows] * P[tile cols]^T  with panel j-1, pulled off a ticket
//                      counter.  This is the 2/3 of the tile's flops; it runs beside the chain instead of after it.
//
// With one workgroup per CU (148 KiB of LDS each) all roles of a launch are resident together, which the chain's
// waits rely on (as potrf_panel_kernel's do); every spin is bounded and a timeout is reported through `info`.
// Critical path per block column: strips (~4 us) + chain (31 us) + kernel boundary, instead of
// chain + boundary + trailing GEMM (25 us) + boundary.
// ================================================================================================
constexpr int STRIP = 16;                 // rows per look-ahead strip (one MFMA tile row)
constexpr int STRIPS_PER_GROUP = PGROWS / STRIP;
constexpr int KC = 32;                    // k chunk of the trailing tiles
constexpr int LA_TIMEOUT_INFO = -7777;    // written to *info when a bounded wait expired

__device__ inline void st_through(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// role A: strip `strip` of block column j0 (rows j0 + 16 strip ...), K = 128 panel at columns j0 - 128 .. j0
__device__ inline void la_strip(double* S, double* A, int64_t lda, int j0, int strip, int* group_cnt) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    double* Bs = S;                    // [128][SLD]  P[rows of the diagonal block], k contiguous
    double* As = S + NB * SLD;         // [16][SLD]   P[strip rows]
    const double* Pb = A + (int64_t)j0 * lda + (j0 - NB);
    const int r0 = j0 + strip * STRIP;
    const double* Pa = A + (int64_t)r0 * lda + (j0 - NB);
    double* C = A + (int64_t)r0 * lda + j0;
    {
        // all global loads first (one memory round trip), 16-byte chunks
        constexpr int BCH = NB * NB / 2 / DIAG_THREADS;  // 16
        d2v_t vb[BCH];
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int e = tid + DIAG_THREADS * i;        // chunk id: row = e / 64, chunk = e % 64
            vb[i] = *reinterpret_cast<const d2v_t*>(Pb + (int64_t)(e >> 6) * lda + 2 * (e & 63));
        }
        d2v_t va[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + DIAG_THREADS * i;        // 1024 chunks of the 16 x 128 strip
            va[i] = *reinterpret_cast<const d2v_t*>(Pa + (int64_t)(e >> 6) * lda + 2 * (e & 63));
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int e = tid + DIAG_THREADS * i;
            *reinterpret_cast<d2v_t*>(&Bs[(e >> 6) * SLD + 2 * (e & 63)]) = vb[i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + DIAG_THREADS * i;
            *reinterpret_cast<d2v_t*>(&As[(e >> 6) * SLD + 2 * (e & 63)]) = va[i];
        }
    }
    // the product is summed from zero and subtracted once (as the GEMM epilogue does): accumulating into C would
    // round 32 times at the magnitude of C
    double cv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cv[r] = C[(int64_t)(lg + 4 * r) * lda + wave * JB + li];
    d4_t acc = {0, 0, 0, 0};
    __syncthreads();
#pragma unroll 8
    for (int st = 0; st < NB / 4; ++st) {
        const double a = As[li * SLD + 4 * st + lg];
        const double b = Bs[(wave * JB + li) * SLD + 4 * st + lg];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) st_through(&C[(int64_t)(lg + 4 * r) * lda + wave * JB + li], cv[r] - acc[r]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its stores have been acknowledged
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(&group_cnt[strip / STRIPS_PER_GROUP], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wait until the strips of the 64-row groups [g0, g1) have all arrived (thread 0 polls; bounded)
__device__ inline bool la_wait_groups(const int* group_cnt, int g0, int g1, int nstrips, int32_t* info) {
    __shared__ int ok_flag;
    if (threadIdx.x == 0) {
        int ok = 1;
        for (int g = g0; g < g1; ++g) {
            const int want = min(STRIPS_PER_GROUP, nstrips - g * STRIPS_PER_GROUP);
            if (want <= 0) break;
            int spin = 0;
            while (__hip_atomic_load(&group_cnt[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(2);
                if (++spin > (1 << 22)) {
                    ok = 0;
                    break;
                }
            }
            if (!ok) break;
        }
        if (!ok) atomicCAS(info, 0, LA_TIMEOUT_INFO);
        ok_flag = ok;
    }
    __syncthreads();
    return ok_flag != 0;
}

// role C: the trailing update with the K = 128 panel at columns pc .. pc + 128, as 128 x 128 tiles of the lower
// triangle of the t x t tile grid whose corner is (base, base):
//   C[rm .. rm+128, cn .. cn+128] -= A[rm .., pc ..] * A[cn .., pc ..]^T
// pulled off a ticket counter until it runs dry.  8 waves as 4 x 2, wave tile 32 x 64; k in four chunks of 32 through
// a double-buffered LDS stage (one barrier per chunk).  One workgroup per CU means nobody else covers this
// workgroup's memory phases, so the loop is software-pipelined ACROSS tiles: the next ticket is drawn at the start of
// a tile, the next tile's first k chunk is loaded during the last chunk's products, and the tile's own C values are
// fetched one chunk before they are needed -- the matrix pipes only idle for the LDS stores between chunks.
__device__ inline void la_trailing(double* S, double* A, int64_t lda64, int pc, int base, int t, int* ticket_ctr) {
    __shared__ int s_next[2];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 64;
    constexpr int STAGE = 2 * NB * KC;               // [A chunk | B chunk], each [128][KC] with XOR-swizzled 16-byte chunks
    constexpr int CH = NB * KC / 2 / DIAG_THREADS;   // 4 16-byte chunks per thread and operand
    const int ntiles = t * (t + 1) / 2;
    const int grow = tid >> 4, gcol = 2 * (tid & 15);  // chunk i of this thread: row grow + 32 i, columns gcol, gcol + 1
    // LDS element offset of (row, 16-byte chunk c) inside an operand's [128][KC] block: rows are unpadded and the chunk
    // index is XORed with the row number (gemm.hip's kc_off for 16 chunks per row): a fragment for TWO k steps is one
    // conflict-free ds_read_b128 (lane group lg, step pair p: k = 8 p + 2 lg + {0, 1} -- any k order works as long as
    // both operands use the same one), and the 16-byte stores of the staging copy are conflict-free as well.
    auto sw = [](int row, int chunk) { return row * KC + ((chunk ^ (row & 15)) << 1); };
    const int64_t goff = (int64_t)grow * lda64 + gcol;
    const int64_t coff = (int64_t)(wm0 + lg) * lda64 + (wn0 + li);
    // column-major walk of the lower triangle (the tall left columns first): ticket -> absolute (row, column)
    auto tile_of = [&](int id, int& rm, int& cn) {
        int tn = 0, rem = id;
        while (rem >= t - tn) {
            rem -= t - tn;
            ++tn;
        }
        rm = base + (tn + rem) * NB;
        cn = base + tn * NB;
    };
    __syncthreads();               // the previous user of S is done
    if (tid == 0) s_next[0] = __hip_atomic_fetch_add(ticket_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    int id = __builtin_amdgcn_readfirstlane(s_next[0]);
    if (id >= ntiles) return;
    int rm, cn;
    tile_of(id, rm, cn);
    const double* pa = A + (int64_t)rm * lda64 + pc + goff;
    const double* pb = A + (int64_t)cn * lda64 + pc + goff;
    d2v_t ra[CH], rb[CH];
#define LA_GLOAD(kc)                                                                    \
    _Pragma("unroll") for (int i = 0; i < CH; ++i) {                                    \
        ra[i] = *reinterpret_cast<const d2v_t*>(pa + (int64_t)(32 * i) * lda64 + (kc)); \
        rb[i] = *reinterpret_cast<const d2v_t*>(pb + (int64_t)(32 * i) * lda64 + (kc)); \
    }
#define LA_LSTORE(buf)                                                                            \
    _Pragma("unroll") for (int i = 0; i < CH; ++i) {                                              \
        *reinterpret_cast<d2v_t*>(&S[(buf) * STAGE + sw(grow + 32 * i, tid & 15)]) = ra[i];           \
        *reinterpret_cast<d2v_t*>(&S[(buf) * STAGE + NB * KC + sw(grow + 32 * i, tid & 15)]) = rb[i]; \
    }
    LA_GLOAD(0)
    LA_LSTORE(0)
    int slot = 0;
    for (;;) {
        double* const c0 = A + (int64_t)rm * lda64 + cn + coff;   // one pointer per accumulator row, column blocks as immediates
        __syncthreads();           // stage 0 of this tile is in LDS; everybody is done with the previous tile's stages
        d4_t acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
        double cv[2][4][4];
        int nid = ntiles, ticket = 0;
#pragma unroll
        for (int c = 0; c < NB / KC; ++c) {
            // the next ticket: drawn while nothing of this wave's is in flight but k chunks (a returning atomic drains
            // the wave's memory queue), parked in a register for a chunk, published through LDS before the last chunk
            if (c == 1 && tid == 0) ticket = __hip_atomic_fetch_add(ticket_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c + 1 < NB / KC) {
                LA_GLOAD((c + 1) * KC)
            } else {
                nid = __builtin_amdgcn_readfirstlane(s_next[slot ^ 1]);
                if (nid < ntiles) {
                    tile_of(nid, rm, cn);
                    pa = A + (int64_t)rm * lda64 + pc + goff;
                    pb = A + (int64_t)cn * lda64 + pc + goff;
                    LA_GLOAD(0)
                }
            }
            if (c == NB / KC - 2) {
                if (tid == 0) s_next[slot ^ 1] = ticket;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double* crow = c0 + (int64_t)(16 * i + 4 * r) * lda64;
#pragma unroll
                        for (int j = 0; j < 4; ++j) cv[i][r][j] = crow[16 * j];
                    }
            }
            const double* Sc = S + (c & 1) * STAGE;
            // fragments of step pair p + 1 are fetched while pair p is multiplied
            d2v_t af[2][2], bf[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[0][i] = *reinterpret_cast<const d2v_t*>(&Sc[sw(wm0 + 16 * i + li, lg)]);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[0][j] = *reinterpret_cast<const d2v_t*>(&Sc[NB * KC + sw(wn0 + 16 * j + li, lg)]);
#pragma unroll
            for (int pr = 0; pr < KC / 8; ++pr) {
                if (pr + 1 < KC / 8) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        af[(pr + 1) & 1][i] = *reinterpret_cast<const d2v_t*>(&Sc[sw(wm0 + 16 * i + li, 4 * (pr + 1) + lg)]);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        bf[(pr + 1) & 1][j] = *reinterpret_cast<const d2v_t*>(&Sc[NB * KC + sw(wn0 + 16 * j + li, 4 * (pr + 1) + lg)]);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[pr & 1][i][h], bf[pr & 1][j][h], acc[i][j], 0, 0, 0);
            }
            if (c + 1 < NB / KC) {
                LA_LSTORE((c + 1) & 1)
                __syncthreads();
            }
        }
        // the next tile's first chunk goes to stage 0 (last read two barriers ago) BEFORE this tile's stores are issued:
        // a wait for those loads placed behind the stores would wait for the stores as well
        if (nid < ntiles) {
            LA_LSTORE(0)
        }
        // C - product, the product summed from zero (one rounding at the magnitude of C, as in the GEMM epilogue)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double* crow = c0 + (int64_t)(16 * i + 4 * r) * lda64;
#pragma unroll
                for (int j = 0; j < 4; ++j) crow[16 * j] = cv[i][r][j] - acc[i][j][r];
            }
        if (nid >= ntiles) break;
        id = nid;
        slot ^= 1;
    }
#undef LA_GLOAD
#undef LA_LSTORE
}

// ctl: [0] = ticket counter of role C, [1 ..] = arrival counters of the 64-row groups of block column j0 (zeroed by the host)
__global__ __launch_bounds__(DIAG_THREADS) void potrf_step_kernel(int N, double* A, int64_t lda, int j0, int32_t* info,
                                                                  double* Wj, pslot_t* msg, unsigned long long msg_tag, int* ctl) {
    extern __shared__ __attribute__((aligned(16))) double S[];
    const int m_below = N - j0 - NB;                        // rows below the diagonal block
    const int chain_wgs = 1 + (m_below + PGROWS - 1) / PGROWS;
    const int nstrips = (j0 > 0) ? (N - j0) / STRIP : 0;
    int* group_cnt = ctl + 1;
    // ---- role A ----
    for (int sidx = blockIdx.x; sidx < nstrips; sidx += gridDim.x) {
        __syncthreads();
        la_strip(S, A, lda, j0, sidx, group_cnt);
    }
    // ---- role B ----
    if ((int)blockIdx.x < chain_wgs) {
        bool ok = true;
        if (nstrips > 0) {
            const int g0 = (blockIdx.x == 0) ? 0 : (int)blockIdx.x + 1;
            const int g1 = (blockIdx.x == 0) ? NB / PGROWS : g0 + 1;
            ok = la_wait_groups(group_cnt, g0, g1, nstrips, info);
        }
        __syncthreads();
        double* Ajj = A + (int64_t)j0 * lda + j0;
        if (blockIdx.x == 0) {
            if (ok) {
                diag_block(NB, Ajj, lda, info, j0, Wj, S, false, msg, msg_tag);
            } else {
                // release the panel workgroups (zero messages): the matrix is reported as failed through info
                for (int k = 0; k < NJB; ++k) {
                    pslot_fwd_t x;
                    x[0] = 0.0;
                    x[1] = __longlong_as_double((long long)(msg_tag + k));
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(msg + (size_t)k * 2 * JB * JB + threadIdx.x), "v"(x) : "memory");
                }
            }
        } else {
            const int r0 = ((int)blockIdx.x - 1) * PGROWS;
            double* P = Ajj + (int64_t)(NB + r0) * lda;
            panel_rows_prog(S, P, lda, min(PGROWS, m_below - r0), Ajj, msg, msg_tag);
        }
    }
    // ---- role C ----
    if (j0 > 0 && m_below > 0) la_trailing(S, A, lda, j0 - NB, j0 + NB, m_below / NB, &ctl[0]);
}

// Batched inversion of the NB x NB diagonal blocks of the n x n lower-triangular L:
// block b -> Winv + w_block_offset(b), ld = LW.
__global__ __launch_bounds__(DIAG_THREADS) void trtri_diag_kernel(int n, const double* L, int64_t ldl,
                                                                  double* Winv) {
    extern __shared__ __attribute__((aligned(16))) double S[];
    double* Wd = S + S_ELEMS;
    const int b = blockIdx.x;
    const int off = b * NB;
    const int nb = min(NB, n - off);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    load_lower_block(S, L + (int64_t)off * ldl + off, ldl, nb, tid);
    __syncthreads();
    invert_diag16(S, Wd, wave, lane);  // 8 waves <-> 8 sub-blocks (identity padding inverts to identity)
    __syncthreads();
    block_trtri(S, Wd, wave, lane & 15, lane >> 4);
    store_inverse(S, Wd, nb, Winv + w_block_offset(b), tid);
}

// dst = lower triangle of src (incl. diagonal), strict upper part = 0
__global__ void tril_copy_kernel(int64_t n, const double* src, int64_t lds, double* dst, int64_t ldd) {
    for (int64_t r = blockIdx.y; r < n; r += gridDim.y)
        for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (int64_t)gridDim.x * blockDim.x)
            dst[r * ldd + c] = (c <= r) ? src[r * lds + c] : 0.0;
}

int set_big_lds(const void* fn) {
    NPW_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DIAG_LDS_BYTES));
    return NPW_OK;
}

inline int64_t split(int64_t n) {  // first-half size: multiple of NB, roughly n/2
    const int64_t blocks = ceil_div(n, NB);
    return (blocks / 2) * NB;
}

inline size_t winv_group_elems(int64_t n) { return (size_t)ceil_div(n, LW) * LW * LW; }
// Behind the groups and their scratch: M = the strictly lower LW-blocks of L, each block row g premultiplied with its group's
// inverse, M[g, h] = inv(L_gg) L[g, h] (h < g), rows LW .. n, columns 0 .. n - LW, ld = n - LW -- what the fused solve
// (trsm_fused) multiplies with instead of L.  Only for factors made of whole groups (n a multiple of LW, at least two).
// OFF by default ($NPW_TRSM_FUSED=1 turns it on): measured in round 5 (profiles/r05_step_level_experiments.md), the one launch
// takes 338 us against 4 x 93 for the four group products it replaces -- 1024 tiles of equal length classes finish, and write
// their 128 KiB each, at the same moments -- and M costs 0.30 ms per factor, which the 3 + 2 + 1 solves per factor of the
// 16384^2 problem never earn back (27.19 against 26.5 ms per step).
inline bool trsm_fused_enabled() {
    static const bool on = [] {
        const char* e = getenv("NPW_TRSM_FUSED");
        return e != nullptr && atoi(e) != 0;
    }();
    return on;
}
inline bool winv_has_m(int64_t n) { return trsm_fused_enabled() && n % LW == 0 && n >= 2 * LW; }
inline size_t winv_m_offset(int64_t n) { return winv_group_elems(n) + (size_t)ceil_div(n, LW) * (LW / 2) * (LW / 2); }
inline size_t winv_bytes(int64_t n) {
    return (winv_m_offset(n) + (winv_has_m(n) ? (size_t)(n - LW) * (n - LW) : 0)) * sizeof(double);
}

// Fill the off-diagonal part of every LW x LW group of Winv by doubling steps (batched GEMMs):
//   inv [L11 0; L21 L22] = [X11 0; -X22 L21 X11, X22]     NB -> 2 NB -> 4 NB (= HW) -> LW
// Zeroes Winv's groups first?  No: the caller memsets before the diagonal kernels run.
int complete_groups(int64_t n, const double* L, int64_t ldl, double* Winv, hipStream_t s) {
    double* T = Winv + winv_group_elems(n);
    const int64_t tstride = (LW / 2) * (LW / 2);
    for (int half = NB; half < LW; half *= 2) {
        const int pairs = LW / (2 * half);  // pairs per group
        // every pair that lies inside the matrix: a ragged last group takes part in the levels it is wide enough for
        // (a 512-wide tail still gets its 512 x 512 inverse), problem z = (group z / pairs, pair z % pairs)
        const int total = (int)(n / (2 * half));
        if (total == 0) break;
        GemmOpts o;
        o.batch = total;
        o.batch_inner = pairs;
        // T = L21 * X11      (half x half each, problem (g, h))
        o.batch_a = (int64_t)LW * (ldl + 1);
        o.batch2_a = (int64_t)2 * half * (ldl + 1);
        o.batch_b = (int64_t)LW * LW;
        o.batch2_b = (int64_t)2 * half * (LW + 1);
        o.batch_d = tstride;
        o.batch2_d = (int64_t)half * half;
        int rc = gemm<double>('N', 'N', half, half, half, 1.0, L + (int64_t)half * ldl, ldl, Winv, LW, 0.0, nullptr, 0, T,
                              half, o, s);
        if (rc) return rc;
        // X21 = -X22 * T
        GemmOpts q;
        q.batch = o.batch;
        q.batch_inner = pairs;
        q.batch_a = (int64_t)LW * LW;
        q.batch2_a = (int64_t)2 * half * (LW + 1);
        q.batch_b = tstride;
        q.batch2_b = (int64_t)half * half;
        q.batch_d = (int64_t)LW * LW;
        q.batch2_d = (int64_t)2 * half * (LW + 1);
        rc = gemm<double>('N', 'N', half, half, half, -1.0, Winv + (int64_t)half * (LW + 1), LW, T, half, 0.0, nullptr, 0,
                          Winv + (int64_t)half * LW, LW, q, s);
        if (rc) return rc;
    }
    if (winv_has_m(n)) {
        // M[g, 0 : g LW] = inv(L_gg) L[g, 0 : g LW], block row by block row (inv(L_gg) is lower triangular: row tile m0 stops
        // at k = m0 + BM).  6.4e9 flop for a 4096^2 factor, once per factor; every solve with it then runs its four group
        // products as ONE balanced launch (trsm_fused).
        // One launch: blockdiag(W_1 .. W_{G-1}) times the block lower triangle of L[LW :, 0 : n - LW] (a_blockdiag; three
        // launches of 64 / 128 / 192 tiles, one per block row, each under-filled the chip: 0.28 ms instead of 0.1).
        double* M = Winv + winv_m_offset(n);
        const int64_t ldm = n - LW;
        GemmOpts o;
        o.a_blockdiag = LW;
        o.force_big = true;
        int rc = gemm<double>('N', 'N', n - LW, n - LW, n - LW, 1.0, Winv + (size_t)LW * LW, LW, L + LW * ldl, ldl, 0.0, nullptr, 0, M,
                              ldm, o, s);
        if (rc) return rc;
    }
    return NPW_OK;
}

// State of one triangular solve  X * L^T = B  (X, B: m x n).
//   Winv : inverses of L's NB x NB diagonal blocks (block index = absolute column offset / NB)
//   T    : m x n scratch (ld = ldt).  Column blocks are *updated* in T and *solved* from T into X, so
//          every GEMM is out of place and free to use the chip-filling tilings.  A column block that has
//          not been updated yet still lives in B (only the very first leaf reads B).
struct TrsmCtx {
    int64_t m;
    const double* L;
    int64_t ldl;
    const double* B;
    int64_t ldb;
    double* X;
    int64_t ldx;
    double* T;
    int64_t ldt;
    const double* Winv;
    int64_t c0;  // absolute column offset of X's / B's / T's column 0 inside L (potrf panels)
    hipStream_t s;
    bool groups_ready = false;  // Winv's full LW groups are complete inverses (complete_groups ran)
    const double* M = nullptr;  // ... and behind them the premultiplied strictly lower blocks of L (winv_has_m), ld = ldm
    int64_t ldm = 0;
    // several right-hand sides that share L (the trsm tasks of one block column of the Cholesky DAG) as ONE solve: every
    // GEMM of the recursion runs as a batch with blockIdx.z = right-hand side; B, X, T above are those of problem 0 and
    // problem z's are dB[z], dX[z], dT[z] elements further (separate allocations: no constant stride)
    int count = 1;
    const int64_t* dB = nullptr;
    const int64_t* dX = nullptr;
    const int64_t* dT = nullptr;
    // optional device flag (per problem: skip + dskip[z] int32 further): set => the right-hand side counts as zero and
    // every product of the solve is skipped, which leaves X = 0 exactly (kernels.trsm's allclose(y, 0) short-circuit)
    const int32_t* skip = nullptr;
    const int64_t* dskip = nullptr;
};

// batch options of one GEMM of the recursion: which of B / X / T each operand walks over (L and Winv are shared)
inline GemmOpts trsm_batch(const TrsmCtx& c, const int64_t* da, const int64_t* dc, const int64_t* dd) {
    GemmOpts o;
    o.skip0 = c.skip;
    if (c.count > 1) {
        o.delta_skip0 = c.skip ? c.dskip : nullptr;
        o.batch = c.count;
        o.delta_a = da;
        o.delta_c = dc;
        o.delta_d = dd;
    }
    return o;
}

// solve the column range [coff, coff + n) (relative to the panel); `touched`: its current values are in T
int trsm_rec(const TrsmCtx& c, int64_t coff, int64_t n, bool touched) {
    const int64_t col = c.c0 + coff;  // absolute column inside L
    // the column block still holds its right-hand side in X itself (first block of an in-place solve): NB-wide leaves only
    const bool inplace0 = !touched && (const void*)c.B == (const void*)c.X;
    const double* src = touched ? c.T + coff : c.B + coff;   // current values of the block
    const int64_t lds = touched ? c.ldt : c.ldb;
    const int64_t* dsrc = touched ? c.dT : c.dB;
    if (c.groups_ready && !inplace0 && n == LW && col % LW == 0) {
        // A whole group: X = src * inv(L_group)^T with the group's explicit inverse W = [W11 0; W21 W22],
        // W21 = -W22 L21 W11 -- ONE product instead of leaf, update, leaf: the same flops (W is lower triangular: column
        // tile n0 stops at k = n0 + BN) in one launch.  On 64 x 64 tiles: their k limits follow the diagonal twice as
        // closely as the 128 x 128 tiling's, whose unequal tiles made this form slower than the three launches
        // (per right-hand side: three launches 1.21 ms, two 1.17, this 1.10).
        const double* Wg = c.Winv + (size_t)(col / LW) * LW * LW;
        GemmOpts g = trsm_batch(c, dsrc, nullptr, c.dX);
        g.b_lower_tri = true;
        g.force_small = true;
        return gemm<double>('N', 'T', c.m, LW, LW, 1.0, src, lds, Wg, LW, 0.0, nullptr, 0, c.X + coff, c.ldx, g, c.s);
    }
    const bool half_leaf = c.groups_ready && !inplace0 && n == HW && col % HW == 0;
    if (n <= NB || half_leaf) {
        const double* Wb = half_leaf ? c.Winv + (size_t)(col / LW) * LW * LW + ((col % LW) ? (size_t)HW * (LW + 1) : 0)
                                     : c.Winv + w_block_offset(col / NB);
        double* Xj = c.X + coff;
        GemmOpts o = trsm_batch(c, dsrc, nullptr, c.dX);
        o.b_lower_tri = half_leaf;  // inv(L_half) is lower triangular: column tile n0 stops at k = n0 + BN
        if (!inplace0) return gemm<double>('N', 'T', c.m, n, n, 1.0, src, lds, Wb, LW, 0.0, nullptr, 0, Xj, c.ldx, o, c.s);
        NPW_REQUIRE(c.count == 1, "trsm: in-place solves are not batched");
        o.inplace_a = true;  // in-place solve of the first block of an in-place panel (row-panel tiling)
        return gemm<double>('N', 'T', c.m, n, n, 1.0, Xj, c.ldx, Wb, LW, 0.0, nullptr, 0, Xj, c.ldx, o, c.s);
    }
    const int64_t n1 = split(n), n2 = n - n1;
    int rc = trsm_rec(c, coff, n1, touched);
    if (rc) return rc;
    // T2 = (T2 | B2) - X1 * L21^T,  L21 = L[c0+coff+n1 : c0+coff+n, c0+coff : c0+coff+n1]
    const double* L21 = c.L + (c.c0 + coff + n1) * c.ldl + (c.c0 + coff);
    const double* C2 = touched ? c.T + coff + n1 : c.B + coff + n1;
    const int64_t ldc = touched ? c.ldt : c.ldb;
    rc = gemm<double>('N', 'T', c.m, n2, n1, -1.0, c.X + coff, c.ldx, L21, c.ldl, 1.0, C2, ldc, c.T + coff + n1, c.ldt,
                      trsm_batch(c, c.dX, touched ? c.dT : c.dB, c.dT), c.s);
    if (rc) return rc;
    return trsm_rec(c, coff + n1, n2, true);
}

// The solve with a factor made of whole groups (n = G LW, G >= 2), out of place:  X = B inv(L)^T  as
//     X   = B blockdiag(W_0 .. W_{G-1})^T                       W_g = inv(L_gg): ONE launch over all groups
//     X_g -= sum_{h < g} X_h M[g, h]^T  (recursively, in place)   M[g, h] = W_g L[g, h], kept with the factor
// instead of  X_g = (B_g - sum_h X_h L[g, h]^T) W_g^T  group after group.  The same flops and the same updates (with M in the
// place of L), but the G triangular products no longer depend on one another: they were G launches of 4.6e9 flop each on
// 64 x 64 tiles whose k ranges run from 64 to 1024 -- 42 - 45 TFLOP/s, a third of a 4096-wide solve's time for a quarter of
// its flops -- and are one launch of 1024 full 128 x 128 tiles handed out longest first on the pinned
// k loop.  4096-wide: 7 launches -> 4.
int trsm_fused_updates(const TrsmCtx& c, int64_t coff, int64_t n) {
    if (n <= LW) return NPW_OK;
    const int64_t n1 = ((n / LW) / 2) * LW, n2 = n - n1;   // whole groups on both sides
    int rc = trsm_fused_updates(c, coff, n1);
    if (rc) return rc;
    const double* M21 = c.M + (coff + n1 - LW) * c.ldm + coff;
    double* X2 = c.X + coff + n1;
    rc = gemm<double>('N', 'T', c.m, n2, n1, -1.0, c.X + coff, c.ldx, M21, c.ldm, 1.0, X2, c.ldx, X2, c.ldx,
                      trsm_batch(c, c.dX, c.dX, c.dX), c.s);
    if (rc) return rc;
    return trsm_fused_updates(c, coff + n1, n2);
}

inline bool trsm_can_fuse(const TrsmCtx& c, int64_t n) {
    return trsm_fused_enabled() && c.groups_ready && c.M != nullptr && c.c0 == 0 && winv_has_m(n) && c.m % 128 == 0 && (const void*)c.B != (const void*)c.X;
}

int trsm_fused(const TrsmCtx& c, int64_t n) {
    GemmOpts g = trsm_batch(c, c.dB, nullptr, c.dX);
    g.b_blockdiag = LW;
    g.force_big = true;
    int rc = gemm<double>('N', 'T', c.m, n, n, 1.0, c.B, c.ldb, c.Winv, LW, 0.0, nullptr, 0, c.X, c.ldx, g, c.s);
    if (rc) return rc;
    return trsm_fused_updates(c, 0, n);
}

// Compute units a launch on `s` can be resident on: the device's, or fewer for a stream created with a CU mask
// (npw_stream_create_masked -- the executor's chain stream, which runs a tile's factorisation beside the trailing
// updates of the previous step).  The panel chain's workgroups spin on messages from one another, so every launch of
// the factorisation must fit the stream's CUs at one workgroup (150 KiB of LDS) per CU.
inline int64_t potrf_resident_wgs(int64_t n) { return n <= NB ? 1 : 1 + ceil_div(n - NB, (int64_t)PGROWS); }

// Right-looking blocked Cholesky with NB-wide panels: three launches per block column
//   diag block (factor + invert, one workgroup) -> panel  P <- P inv(L_jj)^T  (in place, one GEMM) ->
//   trailing update  A22 -= P P^T  (lower tiles only, one GEMM with K = NB).
// Compared with the recursive form this trades GEMM shape (K = NB updates stream the trailing matrix
// once per block column: sum ~ n^3 / (3 NB) * 16 B, 1.4 GB for n = 4096) for a third of the launches;
// a tile is latency-bound on the chain of diagonal blocks, not on flops, so fewer, wider launches win.
int potrf_right(int64_t n, double* A, int64_t lda, int32_t* info, double* Winv, hipStream_t s) {
    // messages of the fused panel kernel (NJB x 512 tagged slots): in the scratch behind the inverse groups, free until
    // complete_groups runs.  The tags are unique per call (process-wide counter seeded from the clock), block column
    // and step, so a recycled workspace can never hold a slot that looks current.
    pslot_t* msg = reinterpret_cast<pslot_t*>(Winv + winv_group_elems(n));
    static std::atomic<unsigned long long> call_counter{
        (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() & 0x3fffffffULL};
    const unsigned long long call_tag = (call_counter.fetch_add(1) + 1) << 20;
    NPW_REQUIRE(potrf_resident_wgs(n) <= resident_cu_count(s), "potrf: the panel chain of a %lld^2 tile needs %lld resident workgroups, the stream offers %d CUs",
                (long long)n, (long long)potrf_resident_wgs(n), resident_cu_count(s));
    for (int64_t j0 = 0; j0 < n; j0 += NB) {
        const int64_t nb = (n - j0 < NB) ? n - j0 : NB;
        double* Ajj = A + j0 * lda + j0;
        double* Wj = Winv + w_block_offset(j0 / NB);
        const int64_t m = n - j0 - nb;
        if (m == 0 || nb < NB) {  // last block column (or a ragged one): nothing below / generic path
            hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(DIAG_THREADS), DIAG_LDS_BYTES, s, (int)nb, Ajj, lda, info,
                               (int)j0, Wj);
            NPW_LAUNCH_CHECK();
            if (m == 0) break;
        } else {
            const unsigned wgs = 1 + (unsigned)ceil_div(m, PGROWS);
            hipLaunchKernelGGL(potrf_panel_kernel, dim3(wgs), dim3(DIAG_THREADS), DIAG_LDS_BYTES, s, (int)nb, Ajj, lda, info,
                               (int)j0, Wj, (int)m, msg, call_tag + (unsigned long long)(j0 / NB) * NJB + 1);
            NPW_LAUNCH_CHECK();
        }
        double* P = A + (j0 + nb) * lda + j0;
        int rc = NPW_OK;
        if (nb < NB) {
            GemmOpts o;
            o.inplace_a = true;
            rc = gemm<double>('N', 'T', m, nb, nb, 1.0, P, lda, Wj, LW, 0.0, nullptr, 0, P, lda, o, s);
            if (rc) return rc;
        }
        double* A22 = A + (j0 + nb) * lda + (j0 + nb);
        GemmOpts u;
        u.lower_only = true;
        rc = gemm<double>('N', 'T', m, m, nb, -1.0, P, lda, P, lda, 1.0, A22, lda, A22, lda, u, s);
        if (rc) return rc;
    }
    if (n > NB) {
        // the fused launches left the block inverses out: all of them now, from the finished factor
        hipLaunchKernelGGL(trtri_diag_kernel, dim3((unsigned)ceil_div(n, NB)), dim3(DIAG_THREADS), DIAG_LDS_BYTES, s, (int)n, A, lda,
                           Winv);
        NPW_LAUNCH_CHECK();
    }
    return NPW_OK;
}

// Right-looking factorisation with look-ahead: one potrf_step_kernel launch per block column (see above).
// n must be a multiple of NB with at most 256 chain workgroups.  ctl lives behind the messages in Winv's scratch.
bool lookahead_applies(int64_t n) {
    static const bool off = getenv("NPW_POTRF_NO_LOOKAHEAD") != nullptr;
    return !off && n % NB == 0 && n >= 2 * NB && n <= 8192;
}
constexpr int CTL_INTS = 1 + 8192 / PGROWS + 8;   // per block column: ticket + group counters

int potrf_lookahead(int64_t n, double* A, int64_t lda, int32_t* info, double* Winv, hipStream_t s) {
    pslot_t* msg = reinterpret_cast<pslot_t*>(Winv + winv_group_elems(n));
    int* ctl = reinterpret_cast<int*>(msg + (size_t)NJB * MSG_SLOTS);
    const int64_t steps = n / NB;
    NPW_HIP_CHECK(hipMemsetAsync(ctl, 0, (size_t)steps * CTL_INTS * sizeof(int), s));
    static std::atomic<unsigned long long> call_counter{
        (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() & 0x3fffffffULL};
    const unsigned long long call_tag = (call_counter.fetch_add(1) + 1) << 20;
    const int num_cus = resident_cu_count(s);   // (the stream's CUs, minus the ones left to RCCL while a communicator is live)
    for (int64_t jb = 0; jb < steps; ++jb) {
        const int64_t j0 = jb * NB;
        const int64_t m = n - j0 - NB;
        const unsigned chain = 1 + (unsigned)ceil_div(m, PGROWS);
        // every role of a launch must be resident at once: never more workgroups than CUs (each needs a whole CU's LDS)
        unsigned wgs = chain;
        if (jb > 0) {
            const unsigned strips = (unsigned)((n - j0) / STRIP);
            const int64_t t = m / NB;
            const unsigned tiles = (unsigned)(t * (t + 1) / 2);
            wgs = std::max(wgs, std::min<unsigned>(std::max(strips, chain + tiles), (unsigned)num_cus));
        }
        NPW_REQUIRE(chain <= (unsigned)num_cus, "potrf: block column needs %u resident workgroups, the stream has %d CUs", chain, num_cus);
        hipLaunchKernelGGL(potrf_step_kernel, dim3(wgs), dim3(DIAG_THREADS), DIAG_LDS_BYTES, s, (int)n, A, lda, (int)j0, info,
                           Winv + w_block_offset(jb), msg, call_tag + (unsigned long long)jb * NJB + 1, ctl + jb * CTL_INTS);
        NPW_LAUNCH_CHECK();
    }
    // the last panel's trailing update never ran as role C of a following launch: the last block column has nothing
    // right of it, so nothing is left -- but block column steps-1 still needs ITS strips, which its own launch did.
    hipLaunchKernelGGL(trtri_diag_kernel, dim3((unsigned)steps), dim3(DIAG_THREADS), DIAG_LDS_BYTES, s, (int)n, A, lda, Winv);
    NPW_LAUNCH_CHECK();
    return NPW_OK;
}

int ensure_big_lds() {
    static thread_local bool attr = false;
    if (!attr) {
        int rc = set_big_lds(reinterpret_cast<const void*>(trtri_diag_kernel));
        if (rc) return rc;
        rc = set_big_lds(reinterpret_cast<const void*>(potrf_diag_kernel));
        if (rc) return rc;
        rc = set_big_lds(reinterpret_cast<const void*>(potrf_panel_kernel));
        if (rc) return rc;
        rc = set_big_lds(reinterpret_cast<const void*>(potrf_step_kernel));
        if (rc) return rc;
        attr = true;
    }
    return NPW_OK;
}

}  // namespace
}  // namespace npw

using namespace npw;

extern "C" {

size_t npw_dtrtri_diag_bytes(int64_t n) { return n <= 0 ? 0 : winv_bytes(n); }

int npw_dtrtri_diag(int64_t n, const double* L, int64_t ldl, double* Winv, npw_stream_t stream) {
    NPW_REQUIRE(n >= 0, "npw_dtrtri_diag: negative dimension");
    if (n == 0) return NPW_OK;
    NPW_REQUIRE(L && Winv && ldl >= n, "npw_dtrtri_diag: bad arguments");
    int rc = ensure_big_lds();
    if (rc) return rc;
    const int nblk = (int)ceil_div(n, NB);
    hipStream_t s = as_stream(stream);
    NPW_HIP_CHECK(hipMemsetAsync(Winv, 0, winv_group_elems(n) * sizeof(double), s));
    hipLaunchKernelGGL(trtri_diag_kernel, dim3(nblk), dim3(DIAG_THREADS), DIAG_LDS_BYTES, s, (int)n, L, ldl, Winv);
    NPW_LAUNCH_CHECK();
    return complete_groups(n, L, ldl, Winv, s);
}

size_t npw_dtrsm_rltn_inv_workspace_bytes(int64_t m, int64_t n) {
    if (m <= 0 || n <= 0) return 0;
    return (size_t)m * n * sizeof(double);
}

int npw_dtrsm_rltn_inv(int64_t m, int64_t n, const double* L, int64_t ldl, const double* Winv, const double* B,
                       int64_t ldb, double* X, int64_t ldx, const int32_t* skip_y, void* workspace, npw_stream_t stream) {
    NPW_REQUIRE(m >= 0 && n >= 0, "npw_dtrsm_rltn_inv: negative dimension");
    if (m == 0 || n == 0) return NPW_OK;
    NPW_REQUIRE(L && Winv && B && X && workspace, "npw_dtrsm_rltn_inv: NULL argument");
    NPW_REQUIRE(ldl >= n && ldb >= n && ldx >= n, "npw_dtrsm_rltn_inv: leading dimension too small");
    NPW_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "npw_dtrsm_rltn_inv: workspace not 16B aligned");
    if (X == B) {
        NPW_REQUIRE(ldx == ldb, "npw_dtrsm_rltn_inv: X == B needs ldx == ldb");
    } else {
        NPW_REQUIRE(X + (m - 1) * ldx + n <= B || B + (m - 1) * ldb + n <= X,
                    "npw_dtrsm_rltn_inv: X and B overlap without being identical");
    }
    TrsmCtx c{m, L, ldl, B, ldb, X, ldx, static_cast<double*>(workspace), n, Winv, 0, as_stream(stream)};
    c.groups_ready = true;  // Winv comes from npw_dtrtri_diag or npw_dpotrf_lower, which both complete the groups
    c.skip = skip_y;
    if (winv_has_m(n)) {
        c.M = Winv + winv_m_offset(n);
        c.ldm = n - LW;
    }
    if (trsm_can_fuse(c, n)) return trsm_fused(c, n);
    return trsm_rec(c, 0, n, false);
}

int npw_dtrsm_rltn_inv_batched(int count, int64_t m, int64_t n, const double* L, int64_t ldl, const double* Winv,
                               const double* const* B, int64_t ldb, double* const* X, int64_t ldx,
                               const int32_t* const* skip_y, void* workspace, npw_stream_t stream) {
    NPW_REQUIRE(count >= 0 && m >= 0 && n >= 0, "npw_dtrsm_rltn_inv_batched: negative argument");
    if (count == 0 || m == 0 || n == 0) return NPW_OK;
    NPW_REQUIRE(count <= 16, "npw_dtrsm_rltn_inv_batched: at most 16 right-hand sides per call");
    NPW_REQUIRE(L && Winv && B && X && workspace, "npw_dtrsm_rltn_inv_batched: NULL argument");
    NPW_REQUIRE(ldl >= n && ldb >= n && ldx >= n, "npw_dtrsm_rltn_inv_batched: leading dimension too small");
    NPW_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "npw_dtrsm_rltn_inv_batched: workspace not 16B aligned");
    int64_t dB[16], dX[16], dT[16], dS[16];
    for (int z = 0; z < count; ++z) {
        dS[z] = 0;
        if (skip_y) {
            NPW_REQUIRE(skip_y[z] != nullptr, "npw_dtrsm_rltn_inv_batched: NULL skip flag (problem %d)", z);
            dS[z] = skip_y[z] - skip_y[0];
        }
        NPW_REQUIRE(B[z] && X[z] && (const void*)B[z] != (const void*)X[z], "npw_dtrsm_rltn_inv_batched: B[%d] / X[%d] NULL or aliased", z, z);
        NPW_REQUIRE(((reinterpret_cast<uintptr_t>(B[z]) | reinterpret_cast<uintptr_t>(X[z])) & 15) == 0,
                    "npw_dtrsm_rltn_inv_batched: tiles must be 16-byte aligned");
        dB[z] = B[z] - B[0];
        dX[z] = X[z] - X[0];
        dT[z] = (int64_t)z * m * n;
    }
    TrsmCtx c{m, L, ldl, B[0], ldb, X[0], ldx, static_cast<double*>(workspace), n, Winv, 0, as_stream(stream)};
    c.groups_ready = true;
    c.count = count;
    c.dB = dB;
    c.dX = dX;
    c.dT = dT;
    if (skip_y) {
        c.skip = skip_y[0];
        c.dskip = dS;
    }
    if (winv_has_m(n)) {
        c.M = Winv + winv_m_offset(n);
        c.ldm = n - LW;
    }
    if (trsm_can_fuse(c, n)) return trsm_fused(c, n);
    return trsm_rec(c, 0, n, false);
}

size_t npw_dtrsm_rltn_workspace_bytes(int64_t m, int64_t n) {
    if (m <= 0 || n <= 0) return 0;
    return winv_bytes(n) + npw_dtrsm_rltn_inv_workspace_bytes(m, n);
}

int npw_dtrsm_rltn(int64_t m, int64_t n, const double* L, int64_t ldl, const double* B,
                   int64_t ldb, double* X, int64_t ldx, void* workspace, npw_stream_t stream) {
    NPW_REQUIRE(m >= 0 && n >= 0, "npw_dtrsm_rltn: negative dimension");
    if (m == 0 || n == 0) return NPW_OK;
    NPW_REQUIRE(L && B && X && workspace, "npw_dtrsm_rltn: NULL argument");
    NPW_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "npw_dtrsm_rltn: workspace not 16B aligned");
    double* Winv = static_cast<double*>(workspace);
    int rc = npw_dtrtri_diag(n, L, ldl, Winv, stream);
    if (rc) return rc;
    return npw_dtrsm_rltn_inv(m, n, L, ldl, Winv, B, ldb, X, ldx, nullptr, static_cast<char*>(workspace) + winv_bytes(n), stream);
}

int npw_dpotrf_lower_resident_cus(int64_t n) { return n <= 0 ? 0 : (int)potrf_resident_wgs(n); }

size_t npw_dpotrf_lower_workspace_bytes(int64_t n) {
    if (n <= 0) return 0;
    return winv_bytes(n);  // the block inverses (kept with the factor for its trsm consumers) + their scratch
}

static int potrf_lower_impl(int64_t n, const double* A, int64_t lda, double* Lout, int64_t ldl, int32_t* info_dev,
                            void* workspace, npw_stream_t stream, bool complete);

int npw_dpotrf_lower(int64_t n, const double* A, int64_t lda, double* Lout, int64_t ldl,
                     int32_t* info_dev, void* workspace, npw_stream_t stream) {
    return potrf_lower_impl(n, A, lda, Lout, ldl, info_dev, workspace, stream, true);
}

int npw_dpotrf_lower_blocks(int64_t n, const double* A, int64_t lda, double* Lout, int64_t ldl,
                            int32_t* info_dev, void* workspace, npw_stream_t stream) {
    return potrf_lower_impl(n, A, lda, Lout, ldl, info_dev, workspace, stream, false);
}

int npw_dtrtri_complete(int64_t n, const double* L, int64_t ldl, double* Winv, npw_stream_t stream) {
    NPW_REQUIRE(n >= 0, "npw_dtrtri_complete: negative dimension");
    if (n == 0) return NPW_OK;
    NPW_REQUIRE(L && Winv && ldl >= n, "npw_dtrtri_complete: bad argument");
    return complete_groups(n, L, ldl, Winv, as_stream(stream));
}

static int potrf_lower_impl(int64_t n, const double* A, int64_t lda, double* Lout, int64_t ldl, int32_t* info_dev,
                            void* workspace, npw_stream_t stream, bool complete) {
    NPW_REQUIRE(n >= 0, "npw_dpotrf_lower: negative dimension");
    NPW_REQUIRE(info_dev != nullptr, "npw_dpotrf_lower: info is NULL");
    hipStream_t s = as_stream(stream);
    NPW_HIP_CHECK(hipMemsetAsync(info_dev, 0, sizeof(int32_t), s));
    if (n == 0) return NPW_OK;
    NPW_REQUIRE(A && Lout && workspace, "npw_dpotrf_lower: NULL argument");
    NPW_REQUIRE(lda >= n && ldl >= n, "npw_dpotrf_lower: leading dimension too small");
    NPW_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "npw_dpotrf_lower: workspace not 16B aligned");
    int rc = ensure_big_lds();
    if (rc) return rc;
    if (Lout == A) {
        NPW_REQUIRE(lda == ldl, "npw_dpotrf_lower: Lout == A needs lda == ldl");
        rc = npw_dtri_keep('L', 0, n, n, Lout, ldl, stream);
        if (rc) return rc;
    } else {
        const unsigned gx = (unsigned)(ceil_div(n, 256) > 16 ? 16 : ceil_div(n, 256));
        const unsigned gy = (unsigned)(n > 256 ? 256 : n);
        hipLaunchKernelGGL(tril_copy_kernel, dim3(gx, gy), dim3(256), 0, s, n, A, lda, Lout, ldl);
        NPW_LAUNCH_CHECK();
    }
    double* Winv = static_cast<double*>(workspace);
    NPW_HIP_CHECK(hipMemsetAsync(Winv, 0, winv_group_elems(n) * sizeof(double), s));
    rc = lookahead_applies(n) ? potrf_lookahead(n, Lout, ldl, info_dev, Winv, s) : potrf_right(n, Lout, ldl, info_dev, Winv, s);
    if (rc) return rc;
    return complete ? complete_groups(n, Lout, ldl, Winv, s) : NPW_OK;  // the factor's trsm consumers use LW-wide leaves
}

#ifdef NPW_DIAG_STAMPS
// one fused block-column launch (progressive mode) on a (128 + m_below) x 128 panel: stamps 0..18 from workgroup 0,
// stamp 24 = end of the last panel workgroup
int npw_debug_fused(double* A, int64_t lda, int m_below, int32_t* info, double* Winv, void* msg, unsigned long long tag,
                    long long* stamps) {
    ensure_big_lds();
    hipMemcpyToSymbol(HIP_SYMBOL(g_diag_stamps), &stamps, sizeof(stamps));
    const unsigned wgs = 1 + (unsigned)ceil_div(m_below, PGROWS);
    hipLaunchKernelGGL(potrf_panel_kernel, dim3(wgs), dim3(DIAG_THREADS), DIAG_LDS_BYTES, 0, 128, A, lda, info, 0, Winv, m_below,
                       static_cast<pslot_t*>(msg), tag);
    return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
int npw_debug_diag(double* A, int64_t lda, int32_t* info, double* Winv, long long* stamps) {
    ensure_big_lds();
    hipMemcpyToSymbol(HIP_SYMBOL(g_diag_stamps), &stamps, sizeof(stamps));
    hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(DIAG_THREADS), DIAG_LDS_BYTES, 0, 128, A, lda, info, 0, Winv);
    return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
#endif

}  // extern "C"
