// elementwise.hip -- HBM-bound tile helpers: n-ary add, diagonal shift, zero test,
// transpose, triangle masks, dtype conversion, synthetic fills, sum of squares.
//
// All of them stream each byte once with 16-byte accesses where the row alignment allows
// it; the grid is capped at a few blocks per CU and strides over the rest (HBM roofline:
// bytes_in + bytes_out at ~6.3 TB/s achievable).
#include "npw_internal.h"

namespace npw {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 256 * 8;

inline int grid_for(int64_t work_items) {
    int64_t b = ceil_div(work_items, kThreads);
    if (b < 1) b = 1;
    if (b > kMaxBlocks) b = kMaxBlocks;
    return (int)b;
}

// 2-D launch shape for row-major rows x cols sweeps: x covers the columns (coalesced), y strides over
// rows -- no per-element integer division (a 64-bit divide costs far more than the 8-byte access).
inline dim3 grid2d(int64_t rows, int64_t cols) {
    int64_t gx = ceil_div(cols, kThreads);
    if (gx > 64) gx = 64;
    int64_t gy = kMaxBlocks * 2 / gx;
    if (gy > rows) gy = rows;
    if (gy < 1) gy = 1;
    if (gy > 65535) gy = 65535;
    return dim3((unsigned)gx, (unsigned)gy);
}

#define NPW_FOR_2D(r, c, rows, cols)                                                       \
    for (int64_t r = blockIdx.y; r < (rows); r += gridDim.y)                                \
        for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < (cols);        \
             c += (int64_t)gridDim.x * blockDim.x)

typedef double d2_t __attribute__((ext_vector_type(2)));

// ---- add_n -------------------------------------------------------------------------------
constexpr int kMaxAdd = 8;
struct AddArgs {
    const void* in[kMaxAdd];
    int64_t ld[kMaxAdd];
    int32_t is_f32[kMaxAdd];
    int count;
};

__global__ void add_n_kernel(AddArgs a, int64_t rows, int64_t cols, double* out, int64_t ld_out,
                             bool accumulate) {
    NPW_FOR_2D(r, c, rows, cols) {
        // the reference accumulates left to right into zeros: ((0 + a0) + a1) + ...
        double s = accumulate ? out[r * ld_out + c] : 0.0;
#pragma unroll
        for (int i = 0; i < kMaxAdd; ++i) {
            if (i < a.count) {
                const int64_t off = r * a.ld[i] + c;
                s += a.is_f32[i] ? (double)reinterpret_cast<const float*>(a.in[i])[off]
                                 : reinterpret_cast<const double*>(a.in[i])[off];
            }
        }
        out[r * ld_out + c] = s;
    }
}

// vectorised flavour: all operands fp64, contiguous rows (ld == cols), cols even, 16B aligned
__global__ void add_n_vec_kernel(AddArgs a, int64_t total2, d2_t* out, bool accumulate) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total2;
         idx += (int64_t)gridDim.x * blockDim.x) {
        d2_t s = accumulate ? out[idx] : d2_t{0.0, 0.0};
#pragma unroll
        for (int i = 0; i < kMaxAdd; ++i)
            if (i < a.count) s += reinterpret_cast<const d2_t*>(a.in[i])[idx];
        out[idx] = s;
    }
}

__global__ void add_diag_kernel(double* A, int64_t n, int64_t lda, double lambda) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        A[i * lda + i] += lambda;
}

__global__ void init_flag_kernel(int32_t* flag, int32_t v) { *flag = v; }

__global__ void is_zero_kernel(const double* A, int64_t rows, int64_t cols, int64_t lda, double atol,
                               int32_t* flag) {
    // The answer for a dense tile is known after the first non-zero element, so the sweep is built to stop:
    // a small grid (<= 512 workgroups), every workgroup polls the flag (relaxed, L2) before each batch of
    // four rows and leaves as soon as anyone has cleared it; a workgroup that finds a non-zero clears the flag
    // with ONE store, and only if it still reads 1.  (Same-address stores serialise in the L2 channel just
    // like atomics: one store per wave of a 4096-workgroup grid cost 0.28 ms per dense 4096^2 tile.)
    constexpr int RB = 4;
    for (int64_t r0 = (int64_t)blockIdx.y * RB; r0 < rows; r0 += (int64_t)gridDim.y * RB) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
        bool bad = false;
        for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += (int64_t)gridDim.x * blockDim.x) {
            double v[RB];
#pragma unroll
            for (int i = 0; i < RB; ++i) v[i] = (r0 + i < rows) ? A[(r0 + i) * lda + c] : 0.0;
            // np.allclose(v, 0): |v - 0| <= atol + rtol*|0|, and non-finite values never match
#pragma unroll
            for (int i = 0; i < RB; ++i)
                if (!(fabs(v[i]) <= atol)) bad = true;
        }
        if (__syncthreads_or(bad)) {
            if (threadIdx.x == 0 && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
                __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
}

struct ZeroBatch {
    const double* A[16];
};

// the same sweep for up to 16 tiles of one shape in one launch (blockIdx.z = tile); flags[z] must be non-zero on entry
__global__ void is_zero_batched_kernel(ZeroBatch b, int64_t rows, int64_t cols, int64_t lda, double atol, int32_t* flags) {
    constexpr int RB = 4;
    const double* A = b.A[blockIdx.z];
    int32_t* flag = flags + blockIdx.z;
    for (int64_t r0 = (int64_t)blockIdx.y * RB; r0 < rows; r0 += (int64_t)gridDim.y * RB) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
        bool bad = false;
        for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += (int64_t)gridDim.x * blockDim.x) {
            double v[RB];
#pragma unroll
            for (int i = 0; i < RB; ++i) v[i] = (r0 + i < rows) ? A[(r0 + i) * lda + c] : 0.0;
#pragma unroll
            for (int i = 0; i < RB; ++i)
                if (!(fabs(v[i]) <= atol)) bad = true;
        }
        if (__syncthreads_or(bad)) {
            if (threadIdx.x == 0 && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
                __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
}

__global__ void zero_if_kernel(double* A, int64_t rows, int64_t cols, int64_t lda, const int32_t* flag) {
    if (*flag == 0) return;
    NPW_FOR_2D(r, c, rows, cols) {
        A[r * lda + c] = 0.0;
    }
}

__global__ void axpby_kernel(int64_t rows, int64_t cols, double alpha, const double* X, int64_t ldx,
                             double beta, const double* Y, int64_t ldy, double* D, int64_t ldd) {
    NPW_FOR_2D(r, c, rows, cols) {
        D[r * ldd + c] = alpha * X[r * ldx + c] + beta * Y[r * ldy + c];
    }
}

__global__ void mul_kernel(int64_t rows, int64_t cols, const double* X, int64_t ldx, const double* Y, int64_t ldy, double* D,
                           int64_t ldd) {
    NPW_FOR_2D(r, c, rows, cols) { D[r * ldd + c] = X[r * ldx + c] * Y[r * ldy + c]; }
}

// B[r][c] = A[fr ? rows - 1 - r : r][fc ? cols - 1 - c : c]
__global__ void flip_kernel(int64_t rows, int64_t cols, const double* A, int64_t lda, double* B, int64_t ldb, int fr, int fc) {
    NPW_FOR_2D(r, c, rows, cols) { B[r * ldb + c] = A[(fr ? rows - 1 - r : r) * lda + (fc ? cols - 1 - c : c)]; }
}

template <typename T>
__global__ void transpose_kernel(int64_t rows, int64_t cols, const T* A, int64_t lda, T* B, int64_t ldb) {
    __shared__ T tile[32][33];
    const int64_t tiles_c = (cols + 31) / 32, tiles_r = (rows + 31) / 32;
    for (int64_t t = blockIdx.x; t < tiles_r * tiles_c; t += gridDim.x) {
        const int64_t tr = t / tiles_c, tc = t - tr * tiles_c;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
        for (int j = ty; j < 32; j += 8) {
            const int64_t r = tr * 32 + j, c = tc * 32 + tx;
            if (r < rows && c < cols) tile[j][tx] = A[r * lda + c];
        }
        __syncthreads();
        for (int j = ty; j < 32; j += 8) {
            const int64_t r = tc * 32 + j, c = tr * 32 + tx;  // B is cols x rows
            if (r < cols && c < rows) B[r * ldb + c] = tile[tx][j];
        }
        __syncthreads();
    }
}

__global__ void tri_keep_kernel(bool lower, bool unit, int64_t rows, int64_t cols, double* A,
                                int64_t lda) {
    NPW_FOR_2D(r, c, rows, cols) {
        if (r == c) {
            if (unit) A[r * lda + c] = 1.0;
        } else if (lower ? (c > r) : (c < r)) {
            A[r * lda + c] = 0.0;
        }
    }
}

// Out (n x n) = 0 except its first nb rows: Out[i][c] = T[(c / nb) * nb + i][c]  -- the nb x nb diagonal blocks of
// the compact-WY factor T side by side, which is LAPACK's blocked (nb x n) T of DGEQRT / DTPQRT.
__global__ void blockdiag_rows_kernel(int64_t n, int64_t nb, const double* T, int64_t ldt, double* Out, int64_t ldo) {
    NPW_FOR_2D(r, c, n, n) {
        double v = 0.0;
        if (r < nb) {
            const int64_t src = (c / nb) * nb + r;
            if (src < n) v = T[src * ldt + c];
        }
        Out[r * ldo + c] = v;
    }
}

template <typename S, typename D>
__global__ void convert_kernel(int64_t rows, int64_t cols, const S* src, int64_t lds, D* dst,
                               int64_t ldd) {
    NPW_FOR_2D(r, c, rows, cols) {
        dst[r * ldd + c] = (D)src[r * lds + c];
    }
}

__global__ void fill_outer_kernel(double* A, int64_t rows, int64_t cols, int64_t lda, const double* x,
                                  int64_t row0, int64_t col0, double lambda) {
    NPW_FOR_2D(r, c, rows, cols) {
        double v = x[row0 + r] * x[col0 + c];
        if (row0 + r == col0 + c) v += lambda;
        A[r * lda + c] = v;
    }
}

__device__ inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// counter-based N(0,1) samples (Box-Muller on two splitmix64 draws): value depends only on
// (seed, global row, global col), so any tiling of the same matrix gives the same numbers.
__global__ void fill_random_kernel(double* A, int64_t rows, int64_t cols, int64_t lda, uint64_t seed,
                                   int64_t row0, int64_t col0) {
    NPW_FOR_2D(r, c, rows, cols) {
        const uint64_t key = splitmix64(seed ^ splitmix64((uint64_t)(row0 + r) * 0x100000001B3ULL + (uint64_t)(col0 + c)));
        const uint64_t k2 = splitmix64(key);
        const double u1 = ((double)(key >> 11) + 1.0) * (1.0 / 9007199254740993.0);
        const double u2 = (double)(k2 >> 11) * (1.0 / 9007199254740992.0);
        A[r * lda + c] = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    }
}

__global__ void sumsq_kernel(const double* A, int64_t rows, int64_t cols, int64_t lda, double* out) {
    double s = 0;
    NPW_FOR_2D(r, c, rows, cols) {
        const double v = A[r * lda + c];
        s += v * v;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __shared__ double part[kThreads / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int i = 0; i < kThreads / 64; ++i) t += part[i];
        atomicAdd(out, t);
    }
}

}  // namespace
}  // namespace npw

using namespace npw;

extern "C" {

int npw_add_n(int count, const void* const* in, const int64_t* ld_in, const int32_t* in_is_f32,
              int64_t rows, int64_t cols, double* out, int64_t ld_out, npw_stream_t stream) {
    NPW_REQUIRE(count >= 0 && rows >= 0 && cols >= 0, "npw_add_n: bad sizes");
    NPW_REQUIRE(out != nullptr || rows * cols == 0, "npw_add_n: out is NULL");
    if (rows == 0 || cols == 0) return NPW_OK;
    hipStream_t s = as_stream(stream);
    if (count == 0) {
        NPW_HIP_CHECK(hipMemset2DAsync(out, ld_out * 8, 0, cols * 8, rows, s));
        return NPW_OK;
    }
    // operands are consumed kMaxAdd at a time, left to right; later passes accumulate into out
    for (int base = 0; base < count; base += kMaxAdd) {
        AddArgs a;
        a.count = (count - base < kMaxAdd) ? count - base : kMaxAdd;
        bool vec = (ld_out == cols) && (cols % 2 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
        for (int i = 0; i < kMaxAdd; ++i) {
            if (i < a.count) {
                a.in[i] = in[base + i];
                a.ld[i] = ld_in[base + i];
                a.is_f32[i] = in_is_f32 ? in_is_f32[base + i] : 0;
                NPW_REQUIRE(a.in[i] != nullptr, "npw_add_n: operand %d is NULL", base + i);
                vec = vec && !a.is_f32[i] && a.ld[i] == cols &&
                      ((reinterpret_cast<uintptr_t>(a.in[i]) & 15) == 0);
            } else {
                a.in[i] = nullptr;
                a.ld[i] = 0;
                a.is_f32[i] = 0;
            }
        }
        if (vec) {
            const int64_t total2 = rows * cols / 2;
            hipLaunchKernelGGL(add_n_vec_kernel, dim3(grid_for(total2)), dim3(kThreads), 0, s, a, total2,
                               reinterpret_cast<d2_t*>(out), base > 0);
        } else {
            hipLaunchKernelGGL(add_n_kernel, grid2d(rows, cols), dim3(kThreads), 0, s, a, rows,
                               cols, out, ld_out, base > 0);
        }
        NPW_LAUNCH_CHECK();
    }
    return NPW_OK;
}

int npw_add_diag(double* A, int64_t rows, int64_t cols, int64_t lda, double lambda,
                 npw_stream_t stream) {
    const int64_t n = rows < cols ? rows : cols;
    if (n <= 0) return NPW_OK;
    NPW_REQUIRE(A != nullptr && lda >= cols, "npw_add_diag: bad arguments");
    hipLaunchKernelGGL(add_diag_kernel, dim3(grid_for(n)), dim3(kThreads), 0, as_stream(stream), A, n,
                       lda, lambda);
    NPW_LAUNCH_CHECK();
    return NPW_OK;
}

int npw_is_zero(const double* A, int64_t rows, int64_t cols, int64_t lda, double atol,
                int32_t* flag_dev, npw_stream_t stream) {
    NPW_REQUIRE(flag_dev != nullptr, "npw_is_zero: flag is NULL");
    NPW_REQUIRE(rows >= 0 && cols >= 0, "npw_is_zero: bad sizes");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(init_flag_kernel, dim3(1), dim3(1), 0, s, flag_dev, 1);
    NPW_LAUNCH_CHECK();
    if (rows * cols == 0) return NPW_OK;
    NPW_REQUIRE(A != nullptr && lda >= cols, "npw_is_zero: bad arguments");
    dim3 grid = grid2d(rows, cols);
    if (grid.y > 32) grid.y = 32;
    hipLaunchKernelGGL(is_zero_kernel, grid, dim3(kThreads), 0, s, A, rows, cols, lda, atol, flag_dev);
    NPW_LAUNCH_CHECK();
    return NPW_OK;
}

int npw_is_zero_batched(int count, const double* const* A, int64_t rows, int64_t cols, int64_t lda, double atol,
                        int32_t* flags_dev, npw_stream_t stream) {
    NPW_REQUIRE(count >= 0 && count <= 16, "npw_is_zero_batched: 0 .. 16 tiles per call");
    NPW_REQUIRE(rows >= 0 && cols >= 0, "npw_is_zero_batched: bad sizes");
    if (count == 0 || rows * cols == 0) return NPW_OK;   // (empty tiles: the preset flags already say "all zero")
    NPW_REQUIRE(A != nullptr && flags_dev != nullptr && lda >= cols, "npw_is_zero_batched: bad arguments");
    ZeroBatch b;
    for (int z = 0; z < 16; ++z) {
        b.A[z] = A[z < count ? z : 0];
        NPW_REQUIRE(b.A[z] != nullptr, "npw_is_zero_batched: NULL tile");
    }
    dim3 grid = grid2d(rows, cols);
    if (grid.y > 32) grid.y = 32;
    grid.z = (unsigned)count;
    hipLaunchKernelGGL(is_zero_batched_kernel, grid, dim3(kThreads), 0, as_stream(stream), b, rows, cols, lda, atol, flags_dev);
    NPW_LAUNCH_CHECK();
    return NPW_OK;
}

int npw_zero_if(double* A, int64_t rows, int64_t cols, int64_t lda, const int32_t* flag_dev,
                npw_stream_t stream) {
    if (rows <= 0 || cols <= 0) return NPW_OK;
    NPW_REQUIRE(A != nullptr && flag_dev != nullptr && lda >= cols, "npw_zero_if: bad arguments");
    hipLaunchKernelGGL(zero_if_kernel, grid2d(rows, cols), dim3(kThreads), 0, as_stream(stream), A,
                       rows, cols, lda, flag_dev);
    NPW_LAUNCH_CHECK();
    return NPW_OK;
}

int npw_daxpby(int64_t rows, int64_t cols, double alpha, const double* X, int64_t ldx, double beta,
               const double* Y, int64_t ldy, double* D, int64_t ldd, npw_stream_t stream) {
    if (rows <= 0 || cols <= 0) return NPW_OK;
    NPW_REQUIRE(X && Y && D && ldx >= cols && ldy >= cols && ldd >= cols, "npw_daxpby: bad arguments");
    hipLaunchKernelGGL(axpby_kernel, grid2d(rows, cols), dim3(kThreads), 0, as_stream(stream), rows,
                       cols, alpha, X, ldx, beta, Y, ldy, D, ldd);
    NPW_LAUNCH_CHECK();
    return NPW_OK;
}

int npw_dmul(int64_t rows, int64_t cols, const double* X, int64_t ldx, const double* Y, int64_t ldy, double* D, int64_t ldd,
             npw_stream_t stream) {
    if (rows <= 0 || cols <= 0) return NPW_OK;
    NPW_REQUIRE(X && Y && D && ldx >= cols && ldy >= cols && ldd >= cols, "npw_dmul: bad arguments");
    hipLaunchKernelGGL(mul_kernel, grid2d(rows, cols), dim3(kThreads), 0, as_stream(stream), rows, cols, X, ldx, Y, ldy, D, ldd);
    NPW_LAUNCH_CHECK();
    return NPW_OK;
}

int npw_dflip(int64_t rows, int64_t cols, const double* A, int64_t lda, double* B, int64_t ldb, int flip_rows, int flip_cols,
              npw_stream_t stream) {
    if (rows <= 0 || cols <= 0) return NPW_OK;
    NPW_REQUIRE(A && B && A != B && lda >= cols && ldb >= cols, "npw_dflip: bad arguments");
    hipLaunchKernelGGL(flip_kernel, grid2d(rows, cols), dim3(kThreads), 0, as_stream(stream), rows, cols, A, lda, B, ldb, flip_rows,
                       flip_cols);
    NPW_LAUNCH_CHECK();
    return NPW_OK;
}

int npw_dtranspose(int64_t rows, int64_t cols, const double* A, int64_t lda, double* B,
                   int64_t ldb, npw_stream_t stream) {
    if (rows <= 0 || cols <= 0) return NPW_OK;
    NPW_REQUIRE(A != nullptr && B != nullptr && lda >= cols && ldb >= rows, "npw_dtranspose: bad arguments");
    int64_t tiles = ceil_div(rows, 32) * ceil_div(cols, 32);
    int grid = (int)(tiles < 65536 ? tiles : 65536);
    hipLaunchKernelGGL(transpose_kernel<double>, dim3(grid), dim3(256), 0, as_stream(stream), rows, cols, A, lda,
                       B, ldb);
    NPW_LAUNCH_CHECK();
    return NPW_OK;
}

int npw_stranspose(int64_t rows, int64_t cols, const float* A, int64_t lda, float* B, int64_t ldb,
                   npw_stream_t stream) {
    if (rows <= 0 || cols <= 0) return NPW_OK;
    NPW_REQUIRE(A != nullptr && B != nullptr && lda >= cols && ldb >= rows, "npw_stranspose: bad arguments");
    int64_t tiles = ceil_div(rows, 32) * ceil_div(cols, 32);
    int grid = (int)(tiles < 65536 ? tiles : 65536);
    hipLaunchKernelGGL(transpose_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), rows, cols, A, lda, B, ldb);
    NPW_LAUNCH_CHECK();
    return NPW_OK;
}

int npw_dtri_keep(char uplo, int unit_diag, int64_t rows, int64_t cols, double* A, int64_t lda,
                  npw_stream_t stream) {
    NPW_REQUIRE(uplo == 'L' || uplo == 'U' || uplo == 'l' || uplo == 'u', "npw_dtri_keep: bad uplo");
    if (rows <= 0 || cols <= 0) return NPW_OK;
    NPW_REQUIRE(A != nullptr && lda >= cols, "npw_dtri_keep: bad arguments");
    hipLaunchKernelGGL(tri_keep_kernel, grid2d(rows, cols), dim3(kThreads), 0, as_stream(stream),
                       uplo == 'L' || uplo == 'l', unit_diag != 0, rows, cols, A, lda);
    NPW_LAUNCH_CHECK();
    return NPW_OK;
}

int npw_dblockdiag_rows(int64_t n, int64_t nb, const double* T, int64_t ldt, double* Out, int64_t ldo,
                        npw_stream_t stream) {
    if (n <= 0) return NPW_OK;
    NPW_REQUIRE(nb > 0 && T != nullptr && Out != nullptr && ldt >= n && ldo >= n && T != Out,
                "npw_dblockdiag_rows: bad arguments");
    hipLaunchKernelGGL(blockdiag_rows_kernel, grid2d(n, n), dim3(kThreads), 0, as_stream(stream), n, nb, T, ldt, Out, ldo);
    NPW_LAUNCH_CHECK();
    return NPW_OK;
}

int npw_convert(int64_t rows, int64_t cols, const void* src, int64_t lds, int src_type, void* dst,
                int64_t ldd, int dst_type, npw_stream_t stream) {
    if (rows <= 0 || cols <= 0) return NPW_OK;
    NPW_REQUIRE(src != nullptr && dst != nullptr && lds >= cols && ldd >= cols, "npw_convert: bad arguments");
    NPW_REQUIRE((src_type == 0 || src_type == 1) && (dst_type == 0 || dst_type == 1), "npw_convert: bad type");
    hipStream_t s = as_stream(stream);
    dim3 g = grid2d(rows, cols), b(kThreads);
    if (src_type == 0 && dst_type == 1)
        hipLaunchKernelGGL((convert_kernel<double, float>), g, b, 0, s, rows, cols, (const double*)src, lds, (float*)dst, ldd);
    else if (src_type == 1 && dst_type == 0)
        hipLaunchKernelGGL((convert_kernel<float, double>), g, b, 0, s, rows, cols, (const float*)src, lds, (double*)dst, ldd);
    else if (src_type == 0)
        hipLaunchKernelGGL((convert_kernel<double, double>), g, b, 0, s, rows, cols, (const double*)src, lds, (double*)dst, ldd);
    else
        hipLaunchKernelGGL((convert_kernel<float, float>), g, b, 0, s, rows, cols, (const float*)src, lds, (float*)dst, ldd);
    NPW_LAUNCH_CHECK();
    return NPW_OK;
}

int npw_fill_outer(double* A, int64_t rows, int64_t cols, int64_t lda, const double* x,
                   int64_t row0, int64_t col0, double lambda, npw_stream_t stream) {
    if (rows <= 0 || cols <= 0) return NPW_OK;
    NPW_REQUIRE(A != nullptr && x != nullptr && lda >= cols, "npw_fill_outer: bad arguments");
    hipLaunchKernelGGL(fill_outer_kernel, grid2d(rows, cols), dim3(kThreads), 0, as_stream(stream),
                       A, rows, cols, lda, x, row0, col0, lambda);
    NPW_LAUNCH_CHECK();
    return NPW_OK;
}

int npw_fill_random(double* A, int64_t rows, int64_t cols, int64_t lda, uint64_t seed,
                    int64_t row0, int64_t col0, npw_stream_t stream) {
    if (rows <= 0 || cols <= 0) return NPW_OK;
    NPW_REQUIRE(A != nullptr && lda >= cols, "npw_fill_random: bad arguments");
    hipLaunchKernelGGL(fill_random_kernel, grid2d(rows, cols), dim3(kThreads), 0, as_stream(stream),
                       A, rows, cols, lda, seed, row0, col0);
    NPW_LAUNCH_CHECK();
    return NPW_OK;
}

int npw_dsumsq(const double* A, int64_t rows, int64_t cols, int64_t lda, double* out_dev,
               npw_stream_t stream) {
    NPW_REQUIRE(out_dev != nullptr, "npw_dsumsq: out is NULL");
    hipStream_t s = as_stream(stream);
    NPW_HIP_CHECK(hipMemsetAsync(out_dev, 0, sizeof(double), s));
    if (rows <= 0 || cols <= 0) return NPW_OK;
    NPW_REQUIRE(A != nullptr && lda >= cols, "npw_dsumsq: bad arguments");
    hipLaunchKernelGGL(sumsq_kernel, grid2d(rows, cols), dim3(kThreads), 0, s, A, rows, cols, lda,
                       out_dev);
    NPW_LAUNCH_CHECK();
    return NPW_OK;
}

}  // extern "C"
