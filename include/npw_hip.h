/*
 * npw_hip.h -- C-ABI of libnpw_hip.so, the MI355X (gfx950 / CDNA4) tile-kernel
 * library underneath numpywren_amd.
 *
 * The reference (Vaishaal/numpywren) has no FFI: its seam is the duck-typed
 * Python kernel interface `compute(*ndarrays, **kwargs)` invoked by
 * RemoteCall (reference numpywren/lambdapack.py:354-384) with the callables of
 * numpywren/kernels.py.  Each entry point below states which reference
 * callable (file:line) it replaces.  INTEGRATION.md shows the ctypes stub a
 * reference maintainer would add to kernels.py to bind them.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  No torch / numpy types.
 *   - every function returns int: 0 = NPW_OK, <0 = error; the message for the
 *     calling thread is available from npw_last_error().
 *   - matrices are ROW-MAJOR (C order, matches NumPy) with an explicit leading
 *     dimension `ld*` counted in ELEMENTS (row stride).
 *   - all device pointers must belong to the current device of the calling
 *     thread (npw_set_device).  All kernels are asynchronous on `stream`
 *     (npw_stream_t == hipStream_t; NULL = the null stream).
 *   - the caller owns every buffer, including workspaces (sizes come from the
 *     *_workspace_bytes queries).  Inputs are never modified unless an output
 *     pointer aliases them, which every kernel explicitly allows or forbids.
 *   - thread-safe: may be called concurrently from several host threads.
 */
#ifndef NPW_HIP_H
#define NPW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NPW_OK 0
#define NPW_ERR_HIP (-1)      /* a HIP runtime call failed                     */
#define NPW_ERR_ARG (-2)      /* invalid argument                              */
#define NPW_ERR_NOT_PD (-3)   /* potrf: matrix not positive definite           */
#define NPW_ERR_UNSUPPORTED (-4)

typedef void* npw_stream_t; /* hipStream_t */
typedef void* npw_event_t;  /* hipEvent_t  */

/* ---- library / device management ------------------------------------------ */
int npw_version(void);

/* Named ranges for profilers (roctx, loaded on first use: librocprofiler-sdk-roctx / libroctx64; without either the calls
 * succeed and do nothing): the executor brackets the kernels it enqueues for one task with a range "<kernel>(<node>)" when
 * executor.roctx_ranges is on, so that `rocprofv3 --marker-trace` shows tasks beside kernels -- the counterpart of the
 * reference's per-instruction start_time / end_time records (numpywren/lambdapack.py:210-211, 361, 379).  Host-side ranges:
 * they mark when a task's work was ENQUEUED.  npw_range_push returns the nesting depth (>= 0) or a negative error. */
int npw_range_push(const char* name);
int npw_range_pop(void);
/* thread-local message of the last failing call on this thread ("" if none) */
const char* npw_last_error(void);
int npw_device_count(int* count);
int npw_set_device(int device);
int npw_get_device(int* device);
/* name: gcnArchName of the device (e.g. "gfx950:sramecc+:xnack-") */
int npw_device_info(int device, char* name, size_t name_len, size_t* total_mem_bytes,
                    int* compute_units, int* clock_khz);
int npw_mem_info(size_t* free_bytes, size_t* total_bytes);
/* "domain:bus:device.function" of the physical device: what tells two ranks that see one device each (per-rank
 * HIP_VISIBLE_DEVICES) whether they sit on the same GPU -- RCCL refuses two ranks of a communicator on one device */
int npw_device_pci_bus_id(int device, char* out, size_t out_len);

/* ---- memory: the HBM tile store + pinned host staging ----------------------
 * Replaces the S3 object GET/PUT of BigMatrix (reference numpywren/matrix.py:
 * 497-533 __s3_key_to_byte_io__/__save_matrix_to_s3__): tiles live in HBM,
 * spill/gather goes through pinned host memory with async copies.            */
int npw_malloc(void** dptr, size_t bytes);
int npw_free(void* dptr);
int npw_host_alloc(void** hptr, size_t bytes); /* pinned */
int npw_host_free(void* hptr);
int npw_memcpy_h2d_async(void* dst, const void* src, size_t bytes, npw_stream_t stream);
int npw_memcpy_d2h_async(void* dst, const void* src, size_t bytes, npw_stream_t stream);
int npw_memcpy_d2d_async(void* dst, const void* src, size_t bytes, npw_stream_t stream);
int npw_memset_async(void* dst, int byte_value, size_t bytes, npw_stream_t stream);
/* strided 2-D copies (rows x row_bytes) for scatter/gather of ragged tiles */
int npw_memcpy2d_h2d_async(void* dst, size_t dpitch, const void* src, size_t spitch,
                           size_t row_bytes, size_t rows, npw_stream_t stream);
int npw_memcpy2d_d2h_async(void* dst, size_t dpitch, const void* src, size_t spitch,
                           size_t row_bytes, size_t rows, npw_stream_t stream);
int npw_memcpy2d_d2d_async(void* dst, size_t dpitch, const void* src, size_t spitch,
                           size_t row_bytes, size_t rows, npw_stream_t stream);

/* ---- streams / events: the local worker pool --------------------------------
 * Replaces the pywren worker fan-out + asyncio read/compute/write pipeline
 * (reference numpywren/job_runner.py:224-370).                                */
int npw_stream_create(npw_stream_t* stream, int high_priority);
/* stream whose kernels may only run on the compute units whose bit is set in cu_mask (bit i of
 * word i/32 = CU i): used to keep a few CUs free of long-running trailing-update workgroups so
 * the latency-bound panel kernels of the critical path always find a slot.               */
int npw_stream_create_masked(npw_stream_t* stream, const uint32_t* cu_mask, int words);
/* Also retires what the library keeps per stream (the helper streams of the factorisations, the cached CU count): a later
 * stream that is handed the same handle starts clean.  The stream must be idle.  A CU-MASKED stream is not handed back to the
 * HIP runtime but parked (synchronised, with its helpers) and given out again by the next npw_stream_create_masked with the
 * same mask: create / destroy cycles of masked streams hang inside the HIP runtime of ROCm 7.2 now and then (observed about
 * every tenth cycle, with nothing but one GEMM on the stream in between), so the library never runs one. */
int npw_stream_destroy(npw_stream_t stream);
/* compute_units: the compute units `stream` may run on (its CU mask; the whole device for a plain stream).
 * resident_units: what the resident-grid kernels (npw_dgeqrt_batched, npw_dtpqrt_batched, npw_dpotrf_lower: every workgroup
 * of a launch waits for the others) size their launches to and check against -- compute_units minus the ones left to RCCL's
 * transfer kernels while a communicator of npw_comm_init is live in this process ($NPW_COMM_RESERVE_CUS, default 64 -- RCCL's channel limit: a
 * send / receive workgroup parked on a CU waiting for its peer is not part of any stream's mask).  A caller that splits a
 * batch so that it fits asks here, with the same number the library will check.  Either pointer may be NULL. */
int npw_stream_cu_count(npw_stream_t stream, int* compute_units, int* resident_units);
int npw_stream_synchronize(npw_stream_t stream);
int npw_stream_query(npw_stream_t stream, int* done);
int npw_device_synchronize(void);
int npw_event_create(npw_event_t* event, int timing);
int npw_event_destroy(npw_event_t event);
int npw_event_record(npw_event_t event, npw_stream_t stream);
int npw_event_synchronize(npw_event_t event);
int npw_event_query(npw_event_t event, int* done);
int npw_stream_wait_event(npw_stream_t stream, npw_event_t event);
int npw_event_elapsed_ms(npw_event_t start, npw_event_t stop, float* ms);

/* ---- tile kernels ------------------------------------------------------------ */

/* D = alpha * op(A) * op(B) + beta * C      (fp64, MFMA v_mfma_f64_16x16x4_f64)
 *   op(X) = X if trans == 'N', X^T if trans == 'T'.
 *   op(A) is m x k, op(B) is k x n, C and D are m x n.  C may be NULL iff beta == 0.
 *   D may alias C exactly (same pointer and ld); D must not overlap A or B.
 *   skip_flag (device int32, may be NULL): when non-NULL and *skip_flag != 0 at kernel
 *   run time the product is skipped, i.e. D = beta * C (used for the reference's
 *   allclose(x, 0) short-circuits without a host round trip).
 * Replaces kernels.gemm (reference numpywren/kernels.py:239-244: A.dot(B) with
 * transpose_A / transpose_B kwargs).                                           */
int npw_dgemm(char transA, char transB, int64_t m, int64_t n, int64_t k, double alpha,
              const double* A, int64_t lda, const double* B, int64_t ldb, double beta,
              const double* C, int64_t ldc, double* D, int64_t ldd, const int32_t* skip_flag,
              npw_stream_t stream);
/* `count` (<= 16) independent products D[z] = op(A[z]) * op(B[z]) of ONE shape, transposition and leading dimensions as ONE
 * launch (blockIdx.z = problem): the ready `gemm` tasks of the GEMM program (reference algs.py:251-266: M N K independent
 * starters, one RemoteCall of kernels.gemm each, lambdapack.py:360-380) handed over together by the executor -- workgroups
 * flow from one problem's tiles into the next one's, the chip drains once per batch instead of once per product.  Arrays of
 * `count` device pointers; every tile 16-byte aligned.  Each problem is computed exactly as npw_dgemm / npw_sgemm computes it
 * (same tiling, same order of products: the same bits). */
int npw_dgemm_batched(int count, char transA, char transB, int64_t m, int64_t n, int64_t k, const double* const* A, int64_t lda,
                      const double* const* B, int64_t ldb, double* const* D, int64_t ldd, npw_stream_t stream);
int npw_sgemm_batched(int count, char transA, char transB, int64_t m, int64_t n, int64_t k, const float* const* A, int64_t lda,
                      const float* const* B, int64_t ldb, float* const* D, int64_t ldd, npw_stream_t stream);
/* fp32 flavour on v_mfma_f32_16x16x4_f32 (BASELINE config 5). */
int npw_sgemm(char transA, char transB, int64_t m, int64_t n, int64_t k, float alpha,
              const float* A, int64_t lda, const float* B, int64_t ldb, float beta, const float* C,
              int64_t ldc, float* D, int64_t ldd, const int32_t* skip_flag, npw_stream_t stream);

/* D = S - X * Y^T   (S,D: m x n; X: m x k; Y: n x k).  D may alias S.
 * skip_x / skip_y: optional device flags (see npw_is_zero): if either is set D = S.
 * X == Y (same pointer and ld, m == n >= 1024 a multiple of 128: the diagonal tiles of the trailing
 * matrix): X X^T is symmetric bit for bit, so only the 128 x 128 tiles below the diagonal are multiplied
 * and each of them also writes its mirror tile (S is read at both positions: the result is the full
 * S - X X^T for any S, symmetric or not).
 * workspace: npw_dgemm_nt_sub_workspace_bytes(m, n, k) bytes (0 unless the symmetric path applies), may
 * be NULL: with it the diagonal blocks of the symmetric path ride in the pair launch as short k chunks (the slots the
 * pairs leave free work through them, the rest fills the launch's tail) and are summed in a fixed order by one small
 * launch; without it they are a second, k-split launch of their own.
 * Replaces kernels.syrk (reference numpywren/kernels.py:212-215) -- the Cholesky
 * trailing update, the north-star kernel.                                      */
size_t npw_dgemm_nt_sub_workspace_bytes(int64_t m, int64_t n, int64_t k);
int npw_dgemm_nt_sub(int64_t m, int64_t n, int64_t k, const double* S, int64_t lds,
                     const double* X, int64_t ldx, const double* Y, int64_t ldy, double* D,
                     int64_t ldd, const int32_t* skip_x, const int32_t* skip_y, void* workspace,
                     npw_stream_t stream);
/* `count` (<= 16) independent updates D[z] = S[z] - X[z] * Y[z]^T of one shape as ONE launch: the ready trailing
 * updates of a block column of the Cholesky DAG (one RemoteCall of kernels.syrk each, reference lambdapack.py:360-380)
 * handed over together by the executor.  Arrays of `count` device pointers (16-byte aligned tiles; D[z] may alias
 * S[z]); skip_x / skip_y: arrays of per-problem flags or both NULL.  Every problem is computed exactly as
 * npw_dgemm_nt_sub computes it; the symmetric route (X[z] == Y[z]) is taken when ALL problems qualify -- callers batch
 * diagonal and off-diagonal tiles separately.  workspace: npw_dgemm_nt_sub_batched_workspace_bytes(count, m, n, k) bytes
 * (= `count` x the single call's size since version 110: every problem's diagonal blocks are in flight in the one launch)
 * or NULL.  A caller that sized the buffer by the pre-110 rule (ONE problem's worth) must pass NULL instead. */
size_t npw_dgemm_nt_sub_batched_workspace_bytes(int count, int64_t m, int64_t n, int64_t k);
int npw_dgemm_nt_sub_batched(int count, int64_t m, int64_t n, int64_t k, const double* const* S, int64_t lds,
                             const double* const* X, int64_t ldx, const double* const* Y, int64_t ldy,
                             double* const* D, int64_t ldd, const int32_t* const* skip_x,
                             const int32_t* const* skip_y, void* workspace, npw_stream_t stream);

/* Solve X * L^T = B for X, L lower triangular n x n (non-unit), B and X m x n.
 * Only the lower triangle of L is read.  X may alias B.
 * workspace: npw_dtrsm_rltn_workspace_bytes(m, n) bytes of device memory.
 * Replaces kernels.trsm (reference numpywren/kernels.py:254-257:
 * scipy.linalg.blas.dtrsm(1.0, x.T, y, lower=False, side=1) with x = L).      */
size_t npw_dtrsm_rltn_workspace_bytes(int64_t m, int64_t n);
int npw_dtrsm_rltn(int64_t m, int64_t n, const double* L, int64_t ldl, const double* B,
                   int64_t ldb, double* X, int64_t ldx, void* workspace, npw_stream_t stream);

/* The same solve split in two so that the many trsm tasks of one Cholesky step, which share
 * the same L (reference numpywren/algs.py:243-246: O[j,i] = trsm(O[i,i], S[i,j,i]) for all j),
 * invert its diagonal blocks once:
 *   npw_dtrtri_diag      Winv <- inverses of the diagonal blocks of the n x n lower triangular L
 *                        (1024 x 1024 diagonal groups, 512 / 128 wide ones next to a ragged end).  The
 *                        layout is private to the library: callers only size it with
 *                        npw_dtrtri_diag_bytes(n) and hand it to npw_dtrsm_rltn_inv.
 *   npw_dtrsm_rltn_inv   X = B * L^-T using those inverses; workspace:
 *                        npw_dtrsm_rltn_inv_workspace_bytes(m, n) bytes.  skip_y: optional device flag (see
 *                        npw_is_zero): if set, X = 0 exactly -- kernels.trsm's `if np.allclose(y, 0): return zeros`
 *                        (reference kernels.py:255-256) without a launch of its own.
 * npw_dpotrf_lower leaves exactly such a Winv in the first npw_dtrtri_diag_bytes(n) bytes of
 * its workspace, so a factor and its block inverses can be handed on together.            */
size_t npw_dtrtri_diag_bytes(int64_t n);
int npw_dtrtri_diag(int64_t n, const double* L, int64_t ldl, double* Winv, npw_stream_t stream);
size_t npw_dtrsm_rltn_inv_workspace_bytes(int64_t m, int64_t n);
int npw_dtrsm_rltn_inv(int64_t m, int64_t n, const double* L, int64_t ldl, const double* Winv,
                       const double* B, int64_t ldb, double* X, int64_t ldx, const int32_t* skip_y,
                       void* workspace, npw_stream_t stream);
/* The same solve for `count` (<= 16) right-hand sides that share L -- the trsm tasks of one block column of the Cholesky
 * DAG (reference algs.py:236-249, statements 1 and 4: O[j, i] = trsm(O[i, i], S[i, j, i]) for every j > i) -- as ONE
 * sequence of batched launches: B[z], X[z] are m x n tiles in separate allocations (16-byte aligned, never aliased);
 * workspace: count * npw_dtrsm_rltn_inv_workspace_bytes(m, n); skip_y: `count` flags or NULL.  Same numbers as count
 * separate calls.                                                                                                     */
int npw_dtrsm_rltn_inv_batched(int count, int64_t m, int64_t n, const double* L, int64_t ldl, const double* Winv,
                               const double* const* B, int64_t ldb, double* const* X, int64_t ldx,
                               const int32_t* const* skip_y, void* workspace, npw_stream_t stream);

/* Cholesky factor of the n x n SPD matrix A (only its lower triangle is read):
 * Lout = lower triangular L with A = L L^T, strictly-upper part of Lout set to 0.
 * Lout may alias A.  info_dev: device int32, set to 0 on success or to the 1-based
 * index of the first non-positive pivot (then Lout is unspecified), like LAPACK.
 * Replaces kernels.chol (reference numpywren/kernels.py:225-226 np.linalg.cholesky).*/
size_t npw_dpotrf_lower_workspace_bytes(int64_t n);
int npw_dpotrf_lower(int64_t n, const double* A, int64_t lda, double* Lout, int64_t ldl,
                     int32_t* info_dev, void* workspace, npw_stream_t stream);
/* The same factorisation, leaving only the inverses of the 128 x 128 diagonal blocks in `workspace`: a factor nobody
 * solves with (the last diagonal tile of a matrix) never pays for the rest, and one that is produced beside other work
 * on a few CUs leaves it to its first consumer.  npw_dtrtri_complete(n, Lout, ldl, workspace, stream) then turns the
 * cache into what npw_dpotrf_lower leaves / npw_dtrsm_rltn_inv[_batched] expects (idempotent: it recomputes the same
 * values).  Reference: kernels.chol / kernels.trsm, numpywren/kernels.py:225-226, 254-257. */
int npw_dpotrf_lower_blocks(int64_t n, const double* A, int64_t lda, double* Lout, int64_t ldl,
                            int32_t* info_dev, void* workspace, npw_stream_t stream);
int npw_dtrtri_complete(int64_t n, const double* L, int64_t ldl, double* Winv, npw_stream_t stream);

/* Compute units a stream must offer for npw_dpotrf_lower(n): the workgroups of the panel chain wait for one another,
 * one per CU.  A stream from npw_stream_create_masked with fewer CUs makes npw_dpotrf_lower fail (NPW_ERR_ARG), never
 * hang.  (The executor asks before it moves a chol task onto its masked chain stream, job_runner.py.) */
int npw_dpotrf_lower_resident_cus(int64_t n);

/* Householder QR with compact-WY T of the m x n matrix A, LAPACK
 * DGEQRT3 conventions (H_j = I - tau_j v_j v_j^T, beta = -sign(alpha)*norm):
 *   V (m x n, ldv): unit-lower-trapezoidal Householder vectors (diag = 1, upper = 0)
 *   T (n x n, ldt): upper triangular, Q = I - V T V^T      (strictly-lower = 0)
 *   R (n x n, ldr): upper triangular factor                 (strictly-lower = 0)
 * A is not modified.  Replaces kernels.qr_factor -> fast_qr (reference
 * numpywren/kernels.py:86-105,127-130; f2py dgeqrt3 + post-processing).
 * m < n (the reference's slow_qr, kernels.py:67-84: DGEQRF + DLARFT): k = m reflectors
 * from the leading m x m block; V is m x m, T m x m and R the m x n upper trapezoid
 * [R1 | Q^T A2] (ldv, ldt >= m; ldr >= n).
 * T == NULL (m >= n; here and in the two batched forms): an explicit "R only" request -- the
 * n x n compact-WY factor is neither returned nor assembled beyond the diagonal blocks the
 * factorisation itself applies; V and R are complete.  The reference always returns T
 * (kernels.py:104-105); the executor asks for this form only for tiles nobody reads, under
 * its `drop_unread_outputs` option.                                                      */
size_t npw_dgeqrt_workspace_bytes(int64_t m, int64_t n);
int npw_dgeqrt(int64_t m, int64_t n, const double* A, int64_t lda, double* V, int64_t ldv,
               double* T, int64_t ldt, double* R, int64_t ldr, void* workspace,
               npw_stream_t stream);

/* `count` independent QR factorisations of equal shape (m >= n) in lock step: one sequence of
 * launches serves the whole batch (the panel kernel runs count x slabs workgroups, every GEMM
 * is a strided batch), so the latency-bound panel chain is paid once per batch instead of once
 * per matrix -- the leaves and the tree levels of TSQR (reference algs.py:30-36) are such
 * batches.  A: HOST array of `count` device pointers (ld lda each).  Matrix z of the outputs
 * lives at V + z * stride_v (elements), T + z * stride_t, R + z * stride_r; each result is
 * what npw_dgeqrt returns for A[z].  workspace: npw_dgeqrt_batched_workspace_bytes bytes.    */
size_t npw_dgeqrt_batched_workspace_bytes(int count, int64_t m, int64_t n);
int npw_dgeqrt_batched(int count, int64_t m, int64_t n, const double* const* A, int64_t lda,
                       double* V, int64_t ldv, int64_t stride_v, double* T, int64_t ldt,
                       int64_t stride_t, double* R, int64_t ldr, int64_t stride_r,
                       void* workspace, npw_stream_t stream);

/* QR of two stacked n x n UPPER TRIANGULAR blocks [A1[z]; A2[z]] (what a TSQR tree node factors:
 * the R factors of its children; LAPACK's DTPQRT with l = n), `count` of them in lock step.
 * The strictly lower parts of A1, A2 must be zero (they are not read as zero, they are assumed
 * zero).  Outputs exactly as npw_dgeqrt(2n, n, [A1; A2]) gives them -- V (2n x n) = [I; V2] with
 * V2 upper triangular, T (n x n), R (n x n) -- for about a third of the work: reflector j only
 * involves pivot row j and the first j + 1 rows of the lower block.
 * A1, A2: HOST arrays of `count` device pointers.                                            */
size_t npw_dtpqrt_batched_workspace_bytes(int count, int64_t n);
int npw_dtpqrt_batched(int count, int64_t n, const double* const* A1, const double* const* A2,
                       int64_t lda, double* V, int64_t ldv, int64_t stride_v, double* T, int64_t ldt,
                       int64_t stride_t, double* R, int64_t ldr, int64_t stride_r, void* workspace,
                       npw_stream_t stream);

/* The QR panel kernel's workgroups hand partial sums to one another through tagged slots and wait for them with a BOUNDED
 * spin (a lost hand-off must not hang the GPU).  A wait that expires leaves that call's V / T / R undefined; this returns how
 * many lanes' waits have expired since the last reset (0 in every correct run) so that the caller can fail loudly -- the
 * executor raises when a run with qr_factor / lq_factor tasks settles with a non-zero count.  Blocks until the copy of the
 * 4-byte counter is done (not a stream operation); reset != 0 clears it. */
int npw_dgeqrt_handoff_timeouts(int* count, int reset);

/* out = sum_i in[i]   (count operands of rows x cols each; fp64 accumulate/output).
 * in_is_f32[i] != 0 marks a float32 operand (the reference's add_matrices always
 * produces float64: np.zeros(args[0].shape) += a).  in pointers are HOST arrays of
 * device pointers / lds.  out may alias any in[i] with the same ld.
 * Replaces kernels.add_matrices (reference numpywren/kernels.py:16-20).        */
int npw_add_n(int count, const void* const* in, const int64_t* ld_in, const int32_t* in_is_f32,
              int64_t rows, int64_t cols, double* out, int64_t ld_out, npw_stream_t stream);

/* A[i,i] += lambda for i < min(rows, cols).  Replaces the `lambdav` diagonal shift
 * applied on every read of a diagonal tile (reference numpywren/matrix.py:307-309).*/
int npw_add_diag(double* A, int64_t rows, int64_t cols, int64_t lda, double lambda,
                 npw_stream_t stream);

/* *flag_dev = 1 if all |A[i,j]| <= atol (np.allclose(A, 0) with the default
 * atol = 1e-8; NaN/Inf => 0), else 0.  flag_dev: device int32.
 * Replaces the np.allclose(x, 0) tests in kernels.syrk / kernels.trsm
 * (reference numpywren/kernels.py:213,255) and RemoteWrite's sparse-write test
 * (reference numpywren/lambdapack.py:311).                                     */
int npw_is_zero(const double* A, int64_t rows, int64_t cols, int64_t lda, double atol,
                int32_t* flag_dev, npw_stream_t stream);

/* The same test for `count` (<= 16) tiles of one shape in ONE launch: flags_dev[z] for tile A[z].  The flags must be
 * NON-ZERO on entry (the caller presets a pool of them once, e.g. npw_memset_async(flags, 1, bytes)); the kernel only
 * ever clears one.  Any non-zero value reads as "all close to zero" in every consumer of a flag. */
int npw_is_zero_batched(int count, const double* const* A, int64_t rows, int64_t cols, int64_t lda, double atol,
                        int32_t* flags_dev, npw_stream_t stream);

/* A[:] = 0 iff *flag_dev != 0 (device-side select, no host round trip): implements
 * kernels.trsm's `if np.allclose(y, 0): return np.zeros(...)` (reference
 * numpywren/kernels.py:255-256) on the asynchronous path.                     */
int npw_zero_if(double* A, int64_t rows, int64_t cols, int64_t lda, const int32_t* flag_dev,
                npw_stream_t stream);

/* D = alpha * X + beta * Y (rows x cols, fp64).  D may alias X or Y exactly.
 * The elementwise part of qr_leaf / lq_leaf / *_trailing_update (S0 - W, S1 - V W;
 * reference numpywren/kernels.py:154-164,181-208).                            */
int npw_daxpby(int64_t rows, int64_t cols, double alpha, const double* X, int64_t ldx, double beta,
               const double* Y, int64_t ldy, double* D, int64_t ldd, npw_stream_t stream);

/* D = X * Y elementwise (rows x cols, fp64).  D may alias X or Y exactly.  kernels.mul for two tiles
 * (reference numpywren/kernels.py:233-234: `x * y`).                          */
int npw_dmul(int64_t rows, int64_t cols, const double* X, int64_t ldx, const double* Y, int64_t ldy, double* D, int64_t ldd,
             npw_stream_t stream);

/* B = A with its rows and / or its columns in reverse order (B[r][c] = A[rows - 1 - r][cols - 1 - c] for both).  No
 * aliasing.  Reversing both orders turns an upper triangle into a lower one: the three forms of kernels.trsm other than
 * the default (`lower=True`, `right=False`: reference numpywren/kernels.py:254-257 passes them to DTRSM) and
 * kernels.trsm_sub (kernels.py:178-179, solve_triangular of the upper triangle) run on the one solve the library
 * has, X L^T = Y, through this and npw_dtranspose.                             */
int npw_dflip(int64_t rows, int64_t cols, const double* A, int64_t lda, double* B, int64_t ldb, int flip_rows, int flip_cols,
              npw_stream_t stream);

/* B = A^T (A rows x cols, B cols x rows).  No aliasing.  Replaces the per-tile
 * `.T` of BigMatrixView (reference numpywren/matrix.py:646-647,658-659).      */
int npw_dtranspose(int64_t rows, int64_t cols, const double* A, int64_t lda, double* B,
                   int64_t ldb, npw_stream_t stream);
/* the same for float32 tiles (the GEMM program's inputs, reference algs.py:251-266: a B tile that several products
 * read is transposed once, so that those products run in the k-contiguous NT form)                     */
int npw_stranspose(int64_t rows, int64_t cols, const float* A, int64_t lda, float* B, int64_t ldb,
                   npw_stream_t stream);

/* keep the lower (uplo='L') or upper ('U') triangle incl. diagonal of the rows x cols
 * matrix, zero the rest; unit_diag != 0 forces the diagonal to 1.  In place.   */
int npw_dtri_keep(char uplo, int unit_diag, int64_t rows, int64_t cols, double* A, int64_t lda,
                  npw_stream_t stream);

/* Out (n x n) <- zeros, its first nb rows <- the nb x nb diagonal blocks of the upper triangular
 * compact-WY factor T laid side by side: LAPACK's blocked "NB-by-N" T.  Used by
 * kernels.qr_factor_triangular, whose reference (numpywren/kernels.py:107-119) hands DTPQRT an n x n
 * array for T with nb = min(n, 32) and returns that array as is.  No aliasing.            */
int npw_dblockdiag_rows(int64_t n, int64_t nb, const double* T, int64_t ldt, double* Out, int64_t ldo,
                        npw_stream_t stream);

/* dst[i,j] = (dst_type) src[i,j]; types: 0 = f64, 1 = f32 */
int npw_convert(int64_t rows, int64_t cols, const void* src, int64_t lds, int src_type, void* dst,
                int64_t ldd, int dst_type, npw_stream_t stream);

/* Fill a rows x cols fp64 tile with the deterministic synthetic generators used by
 * bench.py / the experiments (never on the parity path):
 *   kind 0: A[i,j] = x[row0+i] * x[col0+j]  (+ lambda on the global diagonal), x a
 *           device vector -- the reference generator X X^T + lambda I with X = N x 1
 *           (reference experiments/cholesky_experiment.py:78-92).
 *   kind 1: counter-based standard-normal-ish noise seeded by (seed,row0+i,col0+j). */
int npw_fill_outer(double* A, int64_t rows, int64_t cols, int64_t lda, const double* x,
                   int64_t row0, int64_t col0, double lambda, npw_stream_t stream);
int npw_fill_random(double* A, int64_t rows, int64_t cols, int64_t lda, uint64_t seed,
                    int64_t row0, int64_t col0, npw_stream_t stream);

/* sum of squares of a rows x cols matrix into *out_dev (device double, overwritten):
 * used by residual checks at full size.                                        */
int npw_dsumsq(const double* A, int64_t rows, int64_t cols, int64_t lda, double* out_dev,
               npw_stream_t stream);

/* Reduction of the n x n block A (overwritten) to upper bidiagonal form B = Q^T A P by Householder
 * reflections from both sides (LAPACK DGEBD2 order, DLARFG signs): d[0..n) = diagonal, e[0..n-1) =
 * superdiagonal of B; Q and P are not formed.  The arithmetic of kernels.banded_to_bidiagonal
 * (reference numpywren/kernels.py:43-65: DGBBRD with vect = 'N' on a band-packed list of s x s
 * diagonal blocks -- a block-diagonal matrix, so every block is reduced on its own and, P keeping e_1,
 * its bidiagonal form is unique up to the signs of d_i, e_i).  workspace: npw_dgebd2_workspace_bytes(n). */
size_t npw_dgebd2_workspace_bytes(int64_t n);
int npw_dgebd2(int64_t n, double* A, int64_t lda, double* d, double* e, void* workspace, npw_stream_t stream);

/* ---- tile transport between the GPUs of one node: RCCL over xGMI --------------------
 * Replaces the reference's only way to move a tile from one worker to another: an S3 PUT
 * by the producer and an S3 GET by every consumer (reference numpywren/matrix.py:508
 * `get_object`, :527 `put_object`; RemoteRead / RemoteWrite, lambdapack.py:96-186).  One
 * process per GPU; a tile stays in the HBM of its producer and is pushed to the GPUs that
 * own a consumer task.  librccl is loaded on the first npw_comm_* call (dlopen), never by
 * a single-GPU process.
 *
 *   npw_comm_unique_id  rank 0 creates the rendezvous id (ncclGetUniqueId) and hands its
 *                       NPW_COMM_ID_BYTES bytes to the other ranks over any side channel
 *                       (dist.py: the gloo control group; a file or a socket do as well).
 *   npw_comm_init       ncclCommInitRank on the calling thread's current device + the
 *                       rank's transport stream (high priority).  Collective over ranks.
 *   npw_send_tile /     one tile = `bytes` raw bytes of device memory.  Asynchronous on
 *   npw_recv_tile       `stream` (NULL = the communicator's transport stream): order them
 *                       behind the producer / ahead of the consumers with events.  Every
 *                       pair of ranks must issue its transfers in the same order.
 *   npw_bcast_tile      the panel broadcast: root -> members as k grouped sends on k xGMI
 *                       links (one fused launch); on a member, the matching receive into
 *                       `tile`; on other ranks a no-op.
 *   npw_comm_group_start / _end   bracket several transfers into ONE launch whose sends and
 *                       receives progress side by side on their links: dist.py's prologue
 *                       (the GEMM program's A / B panel pushes: SUMMA's traffic as one
 *                       all-to-all-v) and the outputs of one batched group of tasks.  Every
 *                       rank opens and closes its groups at the same points of the sequence.
 *   npw_comm_abort      a rank that fails in the middle of an exchange drops what it has posted
 *                       (ncclCommAbort) instead of launching part of a group; afterwards only
 *                       npw_comm_destroy is valid on the handle.
 * (Control values -- timings, failure flags, the collective time limit -- travel over the
 *  host-side control group, not through this library.)                                  */
#define NPW_COMM_ID_BYTES 128
typedef void* npw_comm_t;
int npw_comm_unique_id(void* id_out, size_t id_bytes);
int npw_comm_init(npw_comm_t* comm, int rank, int world, const void* unique_id);
int npw_comm_destroy(npw_comm_t comm);
int npw_comm_abort(npw_comm_t comm);
/* rank / world: what RCCL itself reports for the live communicator (ncclCommUserRank / ncclCommCount) -- the number of
 * ranks that really joined, which bench.py puts on every N > 1 line as config.rccl_nranks */
int npw_comm_info(npw_comm_t comm, int* rank, int* world, npw_stream_t* transport_stream);
int npw_comm_group_start(npw_comm_t comm);
int npw_comm_group_end(npw_comm_t comm);
int npw_send_tile(npw_comm_t comm, const void* tile, size_t bytes, int dst, npw_stream_t stream);
int npw_recv_tile(npw_comm_t comm, void* tile, size_t bytes, int src, npw_stream_t stream);
int npw_bcast_tile(npw_comm_t comm, void* tile, size_t bytes, int root, const int* members, int nmembers,
                   npw_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NPW_HIP_H */
