// factor.hip -- tile Cholesky (potrf) and triangular solve (trsm) for gfx950.
//
//   npw_dpotrf_lower  replaces kernels.chol (reference numpywren/kernels.py:225-226)
//   npw_dtrsm_rltn    replaces kernels.trsm (reference numpywren/kernels.py:254-257)
//
// Both are recursive blocked algorithms whose flops live in the MFMA GEMM of gemm.hip:
//
//   potrf(A):   A11 = potrf(A11);  A21 = A21 * L11^-T (trsm);  A22 -= A21 A21^T (lower tiles
//               only);  A22 = potrf(A22)                         -- leaves are NB x NB blocks
//   trsm(X,L):  X1 = trsm(X1, L11);  X2 -= X1 * L21^T;  X2 = trsm(X2, L22)
//               leaf:  X_j = X_j * inv(L_jj)^T   (in-place row-panel GEMM)
//
// The NB x NB (128) diagonal blocks are handled by one workgroup each, entirely in LDS, as a
// blocked algorithm over 16 x 16 sub-blocks: the diagonal sub-block is factored and inverted
// by one wave in registers (pivots broadcast with v_readlane), the panel / trailing /
// inverse sub-block products run on v_mfma_f64_16x16x4_f64.  Multiplying by the explicit
// inverse of a small diagonal block instead of substituting is the standard GPU trsm
// formulation (the error grows with cond(L_jj) of the 128-wide block, not of the tile).
#include "npw_internal.h"

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

typedef double d4_t __attribute__((ext_vector_type(4)));

__device__ inline double readlane_d(double x, int src_lane) {  // src_lane must be wave-uniform
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_readlane(lo, src_lane);
    hi = __builtin_amdgcn_readlane(hi, src_lane);
    return __hiloint2double(hi, lo);
}

// Cholesky of a 16x16 block held one row per lane (lane & 15 owns row li as a[0..15]); fully
// unrolled, pivots and multipliers travel through v_readlane (no LDS, no barrier).
// Returns false (wave-uniform) if a pivot is not positive; *bad_col gets its index.
// 1/sqrt(d) and sqrt(d) to ~1 ulp from the hardware v_rsq_f64 estimate + two Newton steps: the
// 128 pivots of a diagonal block form one serial dependency chain, so the sqrt + divide sequences
// of the naive formulation (~300 cycles per pivot) are what the kernel waits for; this is ~5x shorter
// and turns the column scaling into a multiplication.
__device__ inline void rsqrt_sqrt(double d, double& rinv, double& root) {
    double y = __builtin_amdgcn_rsq(d);
    y = y * fma(-0.5 * d * y, y, 1.5);
    y = y * fma(-0.5 * d * y, y, 1.5);
    double l = d * y;
    l = fma(0.5 * y, fma(-l, l, d), l);  // Heron correction: sqrt(d)
    rinv = y * fma(-l, y, 2.0);          // 1 / l
    root = l;
}

// Cholesky of a 16 x 16 block held one row per lane (lane & 15 owns row li as a[0..15]), fully unrolled;
// pivots and multipliers travel through v_readlane (no LDS, no barrier).  rdiag[j] receives 1 / l_jj.
// Returns false (wave-uniform) if a pivot is not positive; *bad_col gets its index.
__device__ inline bool chol16(double (&a)[JB], double (&rdiag)[JB], int li, int* bad_col) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const double d = readlane_d(a[j], j);
        if (!(d > 0.0) && ok) {
            ok = false;
            *bad_col = j;
        }
        double rinv, root;
        rsqrt_sqrt(d, rinv, root);
        rdiag[j] = rinv;
        a[j] = (li == j) ? root : a[j] * rinv;  // rows below the pivot become l_ij (rows above: don't care)
#pragma unroll
        for (int k = j + 1; k < JB; ++k) {
            const double lkj = readlane_d(a[j], k);
            a[k] = fma(-a[j], lkj, a[k]);
        }
    }
    return ok;
}

// inverse of the lower-triangular 16 x 16 block whose row r lives in lane r (a[0..15]); rdiag[r] = 1 / a_rr.
// Lane c (= lane & 15) produces column c of the inverse in w[0..15].  Two partial sums halve the FMA chain.
__device__ inline void trtri16(const double (&a)[JB], const double (&rdiag)[JB], int c, double (&w)[JB]) {
#pragma unroll
    for (int r = 0; r < JB; ++r) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < r; ++k) {
            const double lrk = readlane_d(a[k], r);
            if (k & 1)
                s1 = fma(lrk, w[k], s1);
            else
                s0 = fma(lrk, w[k], s0);
        }
        w[r] = (r < c) ? 0.0 : ((r == c ? 1.0 : 0.0) - (s0 + s1)) * rdiag[r];
    }
}

__device__ inline void recip16(const double (&a)[JB], double (&rdiag)[JB]) {
#pragma unroll
    for (int r = 0; r < JB; ++r) rdiag[r] = 1.0 / readlane_d(a[r], r);
}

// Given L (lower, in S) and the inverses of its 16x16 diagonal sub-blocks (in Wd), overwrite
// the strictly-lower sub-blocks of S with those of X = inv(L), block row by block row:
//   X_ij = -X_ii * sum_{k=j}^{i-1} L_ik X_kj   (row i of L is dead once row i of X is known)
__device__ inline void block_trtri(double* S, const double* Wd, int nbk, int wave, int nwaves, int li,
                                   int lg) {
    for (int i = 1; i < nbk; ++i) {
        d4_t res = {0, 0, 0, 0};
        const int j = wave;  // i <= 7 < nwaves: at most one sub-block per wave and block row
        const bool active = (j < i);
        if (active) {
            d4_t acc = {0, 0, 0, 0};
            for (int k = j; k < i; ++k) {
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const double a = S[(i * JB + li) * SLD + k * JB + 4 * st + lg];
                    const double b = (k == j) ? Wd[(j * JB + 4 * st + lg) * WLD + li]
                                              : S[(k * JB + 4 * st + lg) * SLD + j * JB + li];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                }
            }
            // the D layout (row = lg + 4r) is exactly the B-operand layout of step r
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const double a = -Wd[(i * JB + li) * WLD + 4 * st + lg];
                res = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[st], res, 0, 0, 0);
            }
        }
        __syncthreads();  // every wave has finished reading block row i of L
        if (active) {
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(i * JB + lg + 4 * r) * SLD + j * JB + li] = res[r];
        }
        __syncthreads();
    }
    (void)nwaves;
}

// write inv(L) (diagonal sub-blocks from Wd, strictly-lower ones from S) row-major, ld = NB
__device__ inline void store_inverse(const double* S, const double* Wd, int n, double* Winv) {
    for (int idx = threadIdx.x; idx < NB * NB; idx += blockDim.x) {
        const int r = idx / NB, c = idx - r * NB;
        double v = 0.0;
        if (r < n && c <= r) {
            const int rb = r / JB, cb = c / JB;
            v = (rb == cb) ? Wd[(rb * JB + (r - rb * JB)) * WLD + (c - cb * JB)] : S[r * SLD + c];
        }
        Winv[idx] = v;
    }
}

// Factor the n x n (n <= NB) diagonal block A (lower triangle used) in place: on exit the
// lower triangle holds L, the strict upper triangle is zero; Winv receives inv(L).
// One workgroup of 8 waves, everything in LDS; sub-block products on the fp64 MFMA.
__global__ __launch_bounds__(DIAG_THREADS) void potrf_diag_kernel(int n, double* A, int64_t lda,
                                                                  int32_t* info, int base, double* Winv) {
    extern __shared__ __attribute__((aligned(16))) double S[];
    double* Wd = S + S_ELEMS;
    int* flag = reinterpret_cast<int*>(Wd + W_ELEMS);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    constexpr int NWAVES = DIAG_THREADS / 64;
    const int nbk = (n + JB - 1) / JB;
    const int npad = nbk * JB;

    // lower triangle of A, padded to a multiple of 16 with an identity diagonal
    for (int idx = tid; idx < npad * npad; idx += DIAG_THREADS) {
        const int r = idx / npad, c = idx - r * npad;
        double v = (r == c) ? 1.0 : 0.0;
        if (r < n && c < n) v = (c <= r) ? A[(int64_t)r * lda + c] : 0.0;
        S[r * SLD + c] = v;
    }
    if (tid == 0) *flag = (*info != 0) ? 1 : 0;
    __syncthreads();
    bool failed = (*flag != 0);  // an earlier block of the same matrix already failed

    for (int jb = 0; jb < nbk && !failed; ++jb) {
        // (1) diagonal sub-block: factor + invert, one wave, registers only
        if (wave == 0) {
            double a[JB], w[JB], rdiag[JB];
#pragma unroll
            for (int k = 0; k < JB; ++k) a[k] = S[(jb * JB + li) * SLD + jb * JB + k];
            int bad_col = 0;
            const bool ok = chol16(a, rdiag, li, &bad_col);
            if (!ok) {
                if (lane == 0) {
                    atomicCAS(info, 0, base + jb * JB + bad_col + 1);
                    *flag = 1;
                }
            } else {
                trtri16(a, rdiag, li, w);
                if (lane < JB) {
#pragma unroll
                    for (int k = 0; k < JB; ++k) {
                        S[(jb * JB + li) * SLD + jb * JB + k] = (k <= li) ? a[k] : 0.0;
                        Wd[(jb * JB + k) * WLD + li] = w[k];  // W[r=k][c=li]
                    }
                }
            }
        }
        __syncthreads();
        if (*flag != 0) {
            failed = true;
            break;
        }
        // (2) panel: L_ib = A_ib * inv(L_jj)^T for the sub-blocks below
        {
            const int ib = jb + 1 + wave;
            if (ib < nbk) {
                double av[4], bv[4];
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    av[st] = S[(ib * JB + li) * SLD + jb * JB + 4 * st + lg];
                    bv[st] = Wd[(jb * JB + li) * WLD + 4 * st + lg];  // B[k][c] = W[c][k]
                }
                d4_t acc = {0, 0, 0, 0};
#pragma unroll
                for (int st = 0; st < 4; ++st) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[st], bv[st], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) S[(ib * JB + lg + 4 * r) * SLD + jb * JB + li] = acc[r];
            }
        }
        __syncthreads();
        // (3) trailing update: A_ik -= L_i L_k^T for jb < kb <= ib
        {
            const int t = nbk - jb - 1;
            const int npairs = t * (t + 1) / 2;
            for (int pr = wave; pr < npairs; pr += NWAVES) {
                // unrank pr -> (ii >= kk) in the t x t lower triangle
                int ii = (int)((sqrtf(8.0f * pr + 1.0f) - 1.0f) * 0.5f);
                while ((ii + 1) * (ii + 2) / 2 <= pr) ++ii;
                while (ii * (ii + 1) / 2 > pr) --ii;
                const int kk = pr - ii * (ii + 1) / 2;
                const int ib = jb + 1 + ii, kb = jb + 1 + kk;
                d4_t acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = S[(ib * JB + lg + 4 * r) * SLD + kb * JB + li];
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const double a = -S[(ib * JB + li) * SLD + jb * JB + 4 * st + lg];
                    const double b = S[(kb * JB + li) * SLD + jb * JB + 4 * st + lg];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) S[(ib * JB + lg + 4 * r) * SLD + kb * JB + li] = acc[r];
            }
        }
        __syncthreads();
    }

    // L back to global: lower triangle, zeros above
    for (int idx = tid; idx < n * n; idx += DIAG_THREADS) {
        const int r = idx / n, c = idx - r * n;
        A[(int64_t)r * lda + c] = (c <= r) ? S[r * SLD + c] : 0.0;
    }
    if (failed) {
        for (int idx = tid; idx < NB * NB; idx += DIAG_THREADS) Winv[idx] = 0.0;
        return;
    }
    __syncthreads();
    block_trtri(S, Wd, nbk, wave, NWAVES, li, lg);
    store_inverse(S, Wd, n, Winv);
}

// Batched inversion of the NB x NB diagonal blocks of the n x n lower-triangular L:
// block b -> Winv + b * NB * NB.
__global__ __launch_bounds__(DIAG_THREADS) void trtri_diag_kernel(int n, const double* L, int64_t ldl,
                                                                  double* Winv) {
    extern __shared__ __attribute__((aligned(16))) double S[];
    double* Wd = S + S_ELEMS;
    const int b = blockIdx.x;
    const int off = b * NB;
    const int nb = min(NB, n - off);
    const double* Lb = L + (int64_t)off * ldl + off;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    constexpr int NWAVES = DIAG_THREADS / 64;
    const int nbk = (nb + JB - 1) / JB;
    const int npad = nbk * JB;
    for (int idx = tid; idx < npad * npad; idx += DIAG_THREADS) {
        const int r = idx / npad, c = idx - r * npad;
        double v = (r == c) ? 1.0 : 0.0;
        if (r < nb && c < nb) v = (c <= r) ? Lb[(int64_t)r * ldl + c] : 0.0;
        S[r * SLD + c] = v;
    }
    __syncthreads();
    for (int jb = wave; jb < nbk; jb += NWAVES) {
        double a[JB], w[JB], rdiag[JB];
#pragma unroll
        for (int k = 0; k < JB; ++k) a[k] = S[(jb * JB + li) * SLD + jb * JB + k];
        recip16(a, rdiag);
        trtri16(a, rdiag, li, w);
        if (lane < JB) {
#pragma unroll
            for (int k = 0; k < JB; ++k) Wd[(jb * JB + k) * WLD + li] = w[k];
        }
    }
    __syncthreads();
    block_trtri(S, Wd, nbk, wave, NWAVES, li, lg);
    store_inverse(S, Wd, nb, Winv + (size_t)b * NB * NB);
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

inline size_t winv_bytes(int64_t n) { return (size_t)ceil_div(n, NB) * NB * NB * sizeof(double); }

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
};

// solve the column range [coff, coff + n) (relative to the panel); `touched`: its current values are in T
int trsm_rec(const TrsmCtx& c, int64_t coff, int64_t n, bool touched) {
    if (n <= NB) {
        const double* Wb = c.Winv + (size_t)((c.c0 + coff) / NB) * NB * NB;
        double* Xj = c.X + coff;
        if (touched)
            return gemm<double>('N', 'T', c.m, n, n, 1.0, c.T + coff, c.ldt, Wb, NB, 0.0, nullptr, 0, Xj, c.ldx,
                                GemmOpts(), c.s);
        if ((const void*)c.B != (const void*)c.X)
            return gemm<double>('N', 'T', c.m, n, n, 1.0, c.B + coff, c.ldb, Wb, NB, 0.0, nullptr, 0, Xj, c.ldx,
                                GemmOpts(), c.s);
        GemmOpts o;  // in-place solve of the first block of an in-place panel (row-panel tiling)
        o.inplace_a = true;
        return gemm<double>('N', 'T', c.m, n, n, 1.0, Xj, c.ldx, Wb, NB, 0.0, nullptr, 0, Xj, c.ldx, o, c.s);
    }
    const int64_t n1 = split(n), n2 = n - n1;
    int rc = trsm_rec(c, coff, n1, touched);
    if (rc) return rc;
    // T2 = (T2 | B2) - X1 * L21^T,  L21 = L[c0+coff+n1 : c0+coff+n, c0+coff : c0+coff+n1]
    const double* L21 = c.L + (c.c0 + coff + n1) * c.ldl + (c.c0 + coff);
    const double* C2 = touched ? c.T + coff + n1 : c.B + coff + n1;
    const int64_t ldc = touched ? c.ldt : c.ldb;
    rc = gemm<double>('N', 'T', c.m, n2, n1, -1.0, c.X + coff, c.ldx, L21, c.ldl, 1.0, C2, ldc, c.T + coff + n1, c.ldt,
                      GemmOpts(), c.s);
    if (rc) return rc;
    return trsm_rec(c, coff + n1, n2, true);
}

// in-place Cholesky of the trailing n x n block of A starting at (off, off)
int potrf_rec(int64_t n, int64_t off, double* A, int64_t lda, int32_t* info, double* Winv, double* T, hipStream_t s) {
    double* Ab = A + off * lda + off;
    if (n <= NB) {
        hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(DIAG_THREADS), DIAG_LDS_BYTES, s, (int)n, Ab, lda, info,
                           (int)off, Winv + (size_t)(off / NB) * NB * NB);
        NPW_LAUNCH_CHECK();
        return NPW_OK;
    }
    const int64_t n1 = split(n), n2 = n - n1;
    int rc = potrf_rec(n1, off, A, lda, info, Winv, T, s);
    if (rc) return rc;
    // A21 <- A21 * L11^-T : rows [off+n1, off+n), columns [off, off+n1), in place
    double* A21 = A + (off + n1) * lda + off;
    TrsmCtx c{n2, A, lda, A21, lda, A21, lda, T, n1, Winv, off, s};
    rc = trsm_rec(c, 0, n1, false);
    if (rc) return rc;
    // A22 -= A21 A21^T (only tiles touching the lower triangle)
    double* A22 = A + (off + n1) * lda + (off + n1);
    GemmOpts o;
    o.lower_only = true;
    rc = gemm<double>('N', 'T', n2, n2, n1, -1.0, A21, lda, A21, lda, 1.0, A22, lda, A22, lda, o, s);
    if (rc) return rc;
    return potrf_rec(n2, off + n1, A, lda, info, Winv, T, s);
}

int ensure_big_lds() {
    static thread_local bool attr = false;
    if (!attr) {
        int rc = set_big_lds(reinterpret_cast<const void*>(trtri_diag_kernel));
        if (rc) return rc;
        rc = set_big_lds(reinterpret_cast<const void*>(potrf_diag_kernel));
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
    hipLaunchKernelGGL(trtri_diag_kernel, dim3(nblk), dim3(DIAG_THREADS), DIAG_LDS_BYTES, as_stream(stream), (int)n, L,
                       ldl, Winv);
    NPW_LAUNCH_CHECK();
    return NPW_OK;
}

size_t npw_dtrsm_rltn_inv_workspace_bytes(int64_t m, int64_t n) {
    if (m <= 0 || n <= 0) return 0;
    return (size_t)m * n * sizeof(double);
}

int npw_dtrsm_rltn_inv(int64_t m, int64_t n, const double* L, int64_t ldl, const double* Winv, const double* B,
                       int64_t ldb, double* X, int64_t ldx, void* workspace, npw_stream_t stream) {
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
    return npw_dtrsm_rltn_inv(m, n, L, ldl, Winv, B, ldb, X, ldx, static_cast<char*>(workspace) + winv_bytes(n), stream);
}

size_t npw_dpotrf_lower_workspace_bytes(int64_t n) {
    if (n <= 0) return 0;
    const int64_t h = n - split(n);  // largest panel: (n - n1) x n1
    return winv_bytes(n) + (size_t)h * (size_t)(n > NB ? split(n) : 0) * sizeof(double);
}

int npw_dpotrf_lower(int64_t n, const double* A, int64_t lda, double* Lout, int64_t ldl,
                     int32_t* info_dev, void* workspace, npw_stream_t stream) {
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
    double* T = reinterpret_cast<double*>(static_cast<char*>(workspace) + winv_bytes(n));
    return potrf_rec(n, 0, Lout, ldl, info_dev, Winv, T, s);
}

}  // extern "C"
