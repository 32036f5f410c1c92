// bidiag.hip -- reduction of a dense square block to upper bidiagonal form, B = Q^T A P (d, e only).
//
//   npw_dgebd2   replaces the arithmetic of kernels.banded_to_bidiagonal (reference numpywren/kernels.py:43-65).
//
// The reference packs a list of s x s blocks -- block i sits at A[i s : (i+1) s, i s : (i+1) s], nothing else is stored --
// into LAPACK band storage with kl = ku = s - 1 and calls DGBBRD(vect = 'N'), which returns the diagonal d and the
// superdiagonal e of the bidiagonal form.  A matrix of that shape is block diagonal, so every block is reduced on its
// own (the entries of e that couple two blocks are exactly zero), and with P e_1 = e_1 the bidiagonal form of a block
// is unique up to the signs of d_i, e_i (implicit-Q theorem; non-degenerate case).  DGBBRD reaches it with plane
// rotations inside the band; here each block is reduced by the Golub-Kahan sequence of Householder reflections from
// the left and the right (LAPACK DGEBD2's order and DLARFG's sign convention), which needs no band structure and maps
// onto streaming kernels: per column one reflector generation, one matrix-vector product and one rank-1 update from
// each side.  Every pass is HBM-bound (8 n^3 / 3 flop against 32 n^3 / 3 bytes); nothing on the LambdaPACK programs'
// paths calls this kernel (alg_wrappers never does), so it is built for parity of the surface, not for speed.
#include "npw_internal.h"

namespace npw {
namespace {

constexpr int BT = 256;

__device__ inline double block_sum(double v, double* sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w];
    return s;
}

// DLARFG on x[0 .. len) (stride incx): beta = -sign(alpha) ||x||, tau = (beta - alpha) / beta, v = x / (alpha - beta)
// with v[0] = 1 written in place (nobody needs the overwritten alpha: vect = 'N').  *out = beta (the d or e entry),
// *tau_out = tau (0 for a vector that is already a multiple of e_1).  Also clears `w` (wlen doubles) for the
// atomically accumulated matrix-vector product that follows.  One workgroup.
__global__ __launch_bounds__(BT) void bd_house_kernel(int64_t len, double* x, int64_t incx, double* out, double* tau_out,
                                                      double* w, int64_t wlen) {
    __shared__ double sh[BT / 64];
    const double alpha = x[0];   // (read before anybody can have replaced it by the reflector's leading 1)
    for (int64_t i = threadIdx.x; i < wlen; i += BT) w[i] = 0.0;
    double ss = 0.0;
    for (int64_t i = 1 + threadIdx.x; i < len; i += BT) {
        const double v = x[i * incx];
        ss += v * v;
    }
    const double xnorm2 = block_sum(ss, sh);
    if (xnorm2 == 0.0) {   // H = I
        if (threadIdx.x == 0) {
            *out = alpha;
            *tau_out = 0.0;
            x[0] = 1.0;
        }
        return;
    }
    const double nrm = sqrt(alpha * alpha + xnorm2);
    const double beta = alpha >= 0.0 ? -nrm : nrm;
    const double scale = 1.0 / (alpha - beta);
    for (int64_t i = 1 + threadIdx.x; i < len; i += BT) x[i * incx] *= scale;
    if (threadIdx.x == 0) {
        *out = beta;
        *tau_out = (beta - alpha) / beta;
        x[0] = 1.0;
    }
}

// w[j] += sum_{i in this block's row chunk} A[i][j] v[i]     (A: rows x cols, row-major; v stride incv)
__global__ __launch_bounds__(BT) void bd_gemv_t_kernel(int64_t rows, int64_t cols, const double* A, int64_t lda, const double* v,
                                                       int64_t incv, double* w, int64_t rows_per_block) {
    const int64_t j = (int64_t)blockIdx.x * BT + threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
    if (j >= cols) return;
    double s = 0.0;
    for (int64_t i = r0; i < r1; ++i) s = fma(A[i * lda + j], v[i * incv], s);
    atomicAdd(&w[j], s);
}

// w[i] = sum_j A[i][j] u[j]       (one wave per row, coalesced along the row)
__global__ __launch_bounds__(BT) void bd_gemv_n_kernel(int64_t rows, int64_t cols, const double* A, int64_t lda, const double* u,
                                                       double* w) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (BT / 64) + (threadIdx.x >> 6);
    if (i >= rows) return;
    double s = 0.0;
    for (int64_t j = lane; j < cols; j += 64) s = fma(A[i * lda + j], u[j], s);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) w[i] = s;
}

// A[i][j] -= tau * a[i] * b[j]      (a stride inca, b stride incb)
__global__ __launch_bounds__(BT) void bd_ger_kernel(int64_t rows, int64_t cols, double* A, int64_t lda, const double* a,
                                                    int64_t inca, const double* b, int64_t incb, const double* tau) {
    const double t = *tau;
    if (t == 0.0) return;
    for (int64_t i = blockIdx.y; i < rows; i += gridDim.y) {
        const double ta = t * a[i * inca];
        for (int64_t j = (int64_t)blockIdx.x * BT + threadIdx.x; j < cols; j += (int64_t)gridDim.x * BT)
            A[i * lda + j] = fma(-ta, b[j * incb], A[i * lda + j]);
    }
}

}  // namespace
}  // namespace npw

using namespace npw;

extern "C" {

size_t npw_dgebd2_workspace_bytes(int64_t n) { return n <= 0 ? 0 : (size_t)(n + 8) * sizeof(double); }

int npw_dgebd2(int64_t n, double* A, int64_t lda, double* d, double* e, void* workspace, npw_stream_t stream) {
    NPW_REQUIRE(n >= 0, "npw_dgebd2: negative dimension");
    if (n == 0) return NPW_OK;
    NPW_REQUIRE(A && d && workspace && (e || n == 1), "npw_dgebd2: NULL argument");
    NPW_REQUIRE(lda >= n, "npw_dgebd2: leading dimension too small");
    hipStream_t s = as_stream(stream);
    double* w = static_cast<double*>(workspace);
    double* tau = w + n;   // two scalars
    for (int64_t k = 0; k < n; ++k) {
        const int64_t mr = n - k, nc = n - k - 1;   // rows of the column reflector, columns right of column k
        double* Akk = A + k * lda + k;
        // H_k from A[k:, k]: d[k] = beta
        hipLaunchKernelGGL(bd_house_kernel, dim3(1), dim3(BT), 0, s, mr, Akk, lda, d + k, tau, w, nc);
        NPW_LAUNCH_CHECK();
        if (nc == 0) break;
        // A[k:, k+1:] <- H_k A[k:, k+1:]:  w = A^T v,  A -= tau v w^T
        const int64_t rpb = 128;
        hipLaunchKernelGGL(bd_gemv_t_kernel, dim3((unsigned)ceil_div(nc, BT), (unsigned)ceil_div(mr, rpb)), dim3(BT), 0, s, mr, nc,
                           Akk + 1, lda, Akk, lda, w, rpb);
        NPW_LAUNCH_CHECK();
        unsigned gx = (unsigned)(ceil_div(nc, BT) > 16 ? 16 : ceil_div(nc, BT));
        unsigned gy = (unsigned)(mr > 1024 ? 1024 : mr);
        hipLaunchKernelGGL(bd_ger_kernel, dim3(gx, gy), dim3(BT), 0, s, mr, nc, Akk + 1, lda, Akk, lda, w, (int64_t)1, tau);
        NPW_LAUNCH_CHECK();
        // G_k from A[k, k+1:]: e[k] = beta
        double* Ak1 = Akk + 1;
        hipLaunchKernelGGL(bd_house_kernel, dim3(1), dim3(BT), 0, s, nc, Ak1, (int64_t)1, e + k, tau + 1, w, (int64_t)0);
        NPW_LAUNCH_CHECK();
        const int64_t mr2 = n - k - 1;
        if (mr2 > 0) {
            // A[k+1:, k+1:] <- A[k+1:, k+1:] G_k:  w = A u,  A -= tau w u^T
            double* A11 = A + (k + 1) * lda + (k + 1);
            hipLaunchKernelGGL(bd_gemv_n_kernel, dim3((unsigned)ceil_div(mr2, BT / 64)), dim3(BT), 0, s, mr2, nc, A11, lda, Ak1, w);
            NPW_LAUNCH_CHECK();
            gy = (unsigned)(mr2 > 1024 ? 1024 : mr2);
            hipLaunchKernelGGL(bd_ger_kernel, dim3(gx, gy), dim3(BT), 0, s, mr2, nc, A11, lda, w, (int64_t)1, Ak1, (int64_t)1, tau + 1);
            NPW_LAUNCH_CHECK();
        }
    }
    return NPW_OK;
}

}  // extern "C"
