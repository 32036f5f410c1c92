// Internal helpers shared by the libnpw_hip.so translation units (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "npw_hip.h"

namespace npw {

// thread-local error message behind npw_last_error()
char* error_buffer();
int set_error(int code, const char* fmt, ...);

#define NPW_HIP_CHECK(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            (void)hipGetLastError(); /* reported here: must not resurface at a later launch check */ \
            return ::npw::set_error(NPW_ERR_HIP, "%s failed: %s (%s:%d)", #expr,              \
                                    hipGetErrorString(_e), __FILE__, __LINE__);               \
        }                                                                                     \
    } while (0)

#define NPW_LAUNCH_CHECK()                                                                    \
    do {                                                                                      \
        hipError_t _e = hipGetLastError();                                                    \
        if (_e != hipSuccess)                                                                 \
            return ::npw::set_error(NPW_ERR_HIP, "kernel launch failed: %s (%s:%d)",          \
                                    hipGetErrorString(_e), __FILE__, __LINE__);               \
    } while (0)

#define NPW_REQUIRE(cond, ...)                                                                \
    do {                                                                                      \
        if (!(cond)) return ::npw::set_error(NPW_ERR_ARG, __VA_ARGS__);                       \
    } while (0)

static inline hipStream_t as_stream(npw_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- internal (C++) entry points used across translation units -------------------------

struct GemmOpts {
    const int32_t* skip0 = nullptr;  // device flags: either one set => the product is skipped
    const int32_t* skip1 = nullptr;
    bool lower_only = false;  // only output tiles touching the lower triangle are computed/written
    bool inplace_a = false;   // D aliases A (row panel update, n <= 128): forces one tile column
    int tag = 0;              // 1 = tile-level trailing update (separate kernel symbol for profiling); 2 = its symmetric
                              // form (op(A) op(B) = X X^T, strict_lower): every tile also writes its mirror tile
    bool force_big = false;   // take the 128 x 128 tiling whatever the grid size
    bool force_small = false; // take the 64 x 64 tiling whatever the grid size (finer k limits for triangular operands)
    bool wide_n = false;      // op(A) = T, op(B) = N, fp64, n % 256 == 0: 128 x 256 tiles on 8-wave workgroups (an A panel is fetched
                              // once for 256 output columns instead of by two workgroups that drift apart)
    int splitk = 1;           // > 1 with splitk_ws: cut k into this many chunks (skinny outputs, long k)
    void* splitk_ws = nullptr;  // splitk * m * n elements of scratch
    int* splitk_keep = nullptr;  // non-NULL: leave the partial products in splitk_ws ([problem][split][m][n], alpha and
                                 // beta NOT applied), store their number here and skip the reduction launch
    int k_chunk_ = 0;         // internal: k range per blockIdx.y of the partial-product launch
    bool strict_lower = false;  // with lower_only on a square output: skip the diagonal tiles as well
    int batch = 1;            // > 1: blockIdx.z walks `batch` problems, operand b at base + b * batch_x elements
    int64_t batch_a = 0, batch_b = 0, batch_c = 0, batch_d = 0;
    int batch_inner = 0;      // > 0: two-level batch, problem z = (z / inner) * batch_x + (z % inner) * batch2_x
    int64_t batch2_a = 0, batch2_b = 0, batch2_c = 0, batch2_d = 0;
    bool b_lower_tri = false;  // op(B) = W^T with W (n x k) lower triangular: column tile n0 only needs k < n0 + BN
    bool k_from_diag = false;  // op(A)^T, op(B) (k x m, k x n) lower trapezoidal: tile (m0, n0) only needs k >= max(m0, n0)
    bool a_upper_tri = false;  // op(A) (m x k) upper triangular: row tile m0 only needs k >= m0
    // op(B) = W^T with W block diagonal: `b_blockdiag`-wide lower triangular diagonal blocks stored back to back (block g at
    // B + g * b_blockdiag^2, ldb = b_blockdiag).  Column tile n0 of block g sums over k in [g * b_blockdiag, n0 + BN).  One launch
    // over all blocks, 128 x 128 tiles handed out longest first (trsm's group products: factor.hip trsm_fused).
    int b_blockdiag = 0;
    // op(A) = W, the same block diagonal matrix on the left (op(A) = N, lda = a_blockdiag): row tile m0 of block g sums over k in
    // [g * a_blockdiag, m0 + BM), and output tiles right of block column g (n0 >= (g + 1) * a_blockdiag) are not computed at all
    // -- W times the block lower triangle of op(B), one launch (the factor's premultiplied blocks: factor.hip complete_groups).
    int a_blockdiag = 0;
    // tag 2 (the symmetric trailing update) only: the diagonal 128 x 128 blocks ride in the same launch, cut into `diag_split`
    // k chunks whose raw partial products go to diag_ws ([problem][block][chunk][128 x 128]); appended AFTER the tile pairs in
    // block-id order, so that the slots the pairs leave free -- 16 of 512 for a 4096^2 tile -- work through them while the
    // pairs run and the rest fills the launch's tail.  The caller sums the chunks in a fixed order (splitk_reduce).
    void* diag_ws = nullptr;
    int diag_split = 0;
    // Irregular batch: problem z's operand is (pointer of problem 0) + delta_x[z] ELEMENTS instead of z * batch_x
    // (tiles of one batch live in separate allocations).  Arrays of `batch` entries (<= 16) or NULL.
    const int64_t* delta_a = nullptr;
    const int64_t* delta_b = nullptr;
    const int64_t* delta_c = nullptr;
    const int64_t* delta_d = nullptr;
    const int64_t* delta_skip0 = nullptr;   // the same for the skip flags (int32 units)
    const int64_t* delta_skip1 = nullptr;
};

// D = alpha * op(A) op(B) + beta * C
template <typename T>
int gemm(char transA, char transB, int64_t m, int64_t n, int64_t k, T alpha, const T* A,
         int64_t lda, const T* B, int64_t ldb, T beta, const T* C, int64_t ldc, T* D, int64_t ldd,
         const GemmOpts& opts, hipStream_t stream);

// A helper stream + two events attached to a caller stream, for fork/join of independent launches inside one
// C-ABI call (created on first use, one per caller stream and host thread, never destroyed).
struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    hipStream_t stream2 = nullptr;            // a second helper stream (QR: the next block's near updates, off the panel chain)
    hipEvent_t fork2 = nullptr, join2 = nullptr;
    hipStream_t stream3 = nullptr;            // a third one for chip-filling work beside a latency-bound chain (QR: the superblock
    hipEvent_t fork3 = nullptr, join3 = nullptr;   // reflectors' far updates); $NPW_QR_FAR_RESERVE_CUS keeps it off that many CUs
};
int side_stream(hipStream_t main, SideStream** out);

// Compute units a launch on `s` can be resident on: the device's, or fewer for a stream created with a CU mask
// (npw_stream_create_masked: the executor's chain partition).
int stream_cu_count(hipStream_t s);
int device_cu_count();
// What a RESIDENT-GRID kernel (every workgroup of a launch waits for the others: the QR panel kernels, the Cholesky panel
// chain, the fused near update) may count on: the stream's compute units minus the ones set aside for RCCL's transfer
// kernels while a communicator is live in this process (comm.hip: a send / receive kernel parked on a CU waiting for its
// peer holds registers and LDS there for as long as the peer takes, and is not part of the stream's CU mask).
int resident_cu_count(hipStream_t s);
void comm_live_changed(int delta);   // comm.hip: +1 on npw_comm_init, -1 on destroy / abort
int comm_reserved_cus();             // 0 without a live communicator, else $NPW_COMM_RESERVE_CUS (default: see runtime.hip)

}  // namespace npw
