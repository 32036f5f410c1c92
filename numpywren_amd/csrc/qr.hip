// qr.hip -- Householder QR with compact-WY T for one tile (or a stack of tiles).
//
//   npw_dgeqrt replaces kernels.qr_factor -> fast_qr (reference numpywren/kernels.py:86-105,
//   127-130): LAPACK DGEQRT3 through f2py, then V = unit-lower-trapezoid of the factored
//   matrix, T = n x n upper-triangular compact-WY factor (Q = I - V T V^T), R = leading n x n
//   upper triangle.  Householder reflectors with the LAPACK DLARFG sign convention
//   (beta = -sign(alpha) * ||x||) are unique, so V, T and R agree with LAPACK's up to
//   rounding although the blocking differs from DGEQRT3's recursion.
//
// Structure (blocked right-looking, panel width PB = 32):
//   for each panel:  qr_panel_kernel (one workgroup, the panel kept column-contiguous in a
//                    workspace so the norm / dot-product sweeps are coalesced) produces the
//                    panel's reflectors, its PB x PB T block and its R block;
//                    trailing columns:  W2 -= V_p * (T_p^T * (V_p^T * W2))   -- three MFMA GEMMs
//   T off-diagonal blocks bottom-up:  T12 = -T1 * (V1^T V2) * T2 with V^T V from one big GEMM.
#include "npw_internal.h"

namespace npw {
namespace {

constexpr int PB = 32;
constexpr int PANEL_THREADS = 1024;
constexpr int PANEL_WAVES = PANEL_THREADS / 64;

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Factor the mp x pb panel starting at W (row-major, ld ldw).
//   Pt  : workspace, pb x ldp (column c of the panel stored contiguously at Pt + c*ldp)
//   out : W panel <- V form (unit diagonal, zeros above); Rjj <- pb x pb upper triangle;
//         Tjj <- pb x pb upper triangular T of the panel
__global__ __launch_bounds__(PANEL_THREADS) void qr_panel_kernel(int mp, int pb, double* W, int64_t ldw,
                                                                 double* Pt, int64_t ldp, double* Tjj,
                                                                 int64_t ldt, double* Rjj, int64_t ldr) {
    __shared__ double red[PANEL_WAVES][PB + 1];
    __shared__ double dots[PB];
    __shared__ double Ts[PB][PB + 1];
    __shared__ double s_tau, s_scale;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int idx = tid; idx < mp * pb; idx += PANEL_THREADS) {
        const int r = idx / pb, c = idx - r * pb;
        Pt[(int64_t)c * ldp + r] = W[(int64_t)r * ldw + c];
    }
    for (int idx = tid; idx < PB * (PB + 1); idx += PANEL_THREADS) (&Ts[0][0])[idx] = 0.0;
    __syncthreads();

    for (int c = 0; c < pb; ++c) {
        double* vc = Pt + (int64_t)c * ldp;
        // ---- DLARFG: norm of the column below the diagonal ---------------------------------
        double ss = 0.0;
        for (int r = c + 1 + tid; r < mp; r += PANEL_THREADS) {
            const double x = vc[r];
            ss = fma(x, x, ss);
        }
        ss = wave_sum(ss);
        if (lane == 0) red[wave][0] = ss;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int w = 0; w < PANEL_WAVES; ++w) tot += red[w][0];
            const double alpha = vc[c];
            double tau = 0.0, scale = 0.0, beta = alpha;
            if (tot != 0.0) {
                const double nrm = sqrt(fma(alpha, alpha, tot));
                beta = (alpha >= 0.0) ? -nrm : nrm;
                tau = (beta - alpha) / beta;
                scale = 1.0 / (alpha - beta);
            }
            vc[c] = beta;  // R diagonal entry
            s_tau = tau;
            s_scale = scale;
        }
        __syncthreads();
        const double tau = s_tau, scale = s_scale;

        // ---- scale v and form d_k = v^T P[:,k] for every other column k -----------------------
        double acc[PB];
#pragma unroll
        for (int k = 0; k < PB; ++k) acc[k] = 0.0;
        for (int r = c + tid; r < mp; r += PANEL_THREADS) {
            double v;
            if (r == c) {
                v = 1.0;
            } else {
                v = vc[r] * scale;
                vc[r] = v;
            }
#pragma unroll
            for (int k = 0; k < PB; ++k)
                if (k < pb) acc[k] = fma(v, Pt[(int64_t)k * ldp + r], acc[k]);
        }
#pragma unroll
        for (int k = 0; k < PB; ++k) {
            const double s = wave_sum(acc[k]);
            if (lane == 0) red[wave][k] = s;
        }
        __syncthreads();
        if (tid < pb) {
            double tot = 0.0;
            for (int w = 0; w < PANEL_WAVES; ++w) tot += red[w][tid];
            dots[tid] = tot;
        }
        __syncthreads();

        // ---- apply H_c to the columns to the right: P[:,k] -= tau * d_k * v --------------------
        for (int r = c + tid; r < mp; r += PANEL_THREADS) {
            const double v = (r == c) ? 1.0 : vc[r];
            for (int k = c + 1; k < pb; ++k) {
                double* pk = Pt + (int64_t)k * ldp;
                pk[r] = fma(-tau * dots[k], v, pk[r]);
            }
        }
        // ---- DLARFT column: T[0:c, c] = -tau * T[0:c, 0:c] * (V[:, 0:c]^T v_c),  T[c][c] = tau -----
        if (tid < c) {
            double s = 0.0;
            for (int q = tid; q < c; ++q) s = fma(Ts[tid][q], dots[q], s);
            Ts[tid][c] = -tau * s;
        } else if (tid == c) {
            Ts[c][c] = tau;
        }
        __syncthreads();
    }

    // ---- write back ------------------------------------------------------------------------------
    for (int idx = tid; idx < mp * pb; idx += PANEL_THREADS) {
        const int r = idx / pb, c = idx - r * pb;
        const double p = Pt[(int64_t)c * ldp + r];
        W[(int64_t)r * ldw + c] = (r > c) ? p : (r == c ? 1.0 : 0.0);
        if (r < pb) Rjj[(int64_t)r * ldr + c] = (r <= c) ? p : 0.0;
    }
    for (int idx = tid; idx < pb * pb; idx += PANEL_THREADS) {
        const int r = idx / pb, c = idx - r * pb;
        Tjj[(int64_t)r * ldt + c] = Ts[r][c];
    }
}

// ------------------------------------------------------------------------------------------------
// Panel factorisation on many CUs (tall panels).  One launch per column; workgroup g owns the
// 256-row slab g of the column-contiguous panel copy Pt, one row per thread.  Because the Householder
// scaling is linear, ONE fused reduction per column suffices: with x = column c below the diagonal the
// raw sums q_k = sum_{r>c} x[r] * P[r][k] (k = 0..pb-1; q_c = ||x||^2) give beta, tau, the scale of v
// and every d_k = v^T P[:,k] = P[c][k] + scale * q_k.  Step kernel c therefore
//   1. reduces the slab partials of column c written by the previous launch,
//   2. scales column c into v and applies H_c to its slab (columns > c),
//   3. accumulates, on the updated slab, the raw sums for column c+1 and writes its partials,
//   4. (workgroup 0) appends column c of the panel's T block (DLARFT recurrence).
// Row c of the panel is handed from launch to launch through `rowbuf` so that no workgroup reads an
// element another workgroup is updating in the same launch.
// ------------------------------------------------------------------------------------------------
constexpr int SLAB = 256;

__device__ inline void slab_reduce(double (&acc)[PB], int pb, double* out /* [PB] in global */) {
    __shared__ double red[SLAB / 64][PB + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < PB; ++k) {
        const double s = wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < pb) {
        double t = 0.0;
        for (int w = 0; w < SLAB / 64; ++w) t += red[w][threadIdx.x];
        out[threadIdx.x] = t;
    }
}

__global__ __launch_bounds__(SLAB) void qr_panel2_prep(int mp, int pb, const double* W, int64_t ldw, double* Pt,
                                                       int64_t ldp, double* part, double* rowbuf, double* Tg) {
    const int r = blockIdx.x * SLAB + threadIdx.x;
    double acc[PB];
#pragma unroll
    for (int k = 0; k < PB; ++k) acc[k] = 0.0;
    if (r < mp) {
        double pk[PB];
#pragma unroll
        for (int k = 0; k < PB; ++k) pk[k] = (k < pb) ? W[(int64_t)r * ldw + k] : 0.0;
#pragma unroll
        for (int k = 0; k < PB; ++k)
            if (k < pb) Pt[(int64_t)k * ldp + r] = pk[k];
        if (r == 0) {
#pragma unroll
            for (int k = 0; k < PB; ++k) rowbuf[k] = pk[k];
        }
        const double x = (r > 0) ? pk[0] : 0.0;
#pragma unroll
        for (int k = 0; k < PB; ++k) acc[k] = x * pk[k];
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < PB * PB; i += SLAB) Tg[i] = 0.0;
    slab_reduce(acc, pb, part + (size_t)blockIdx.x * PB);
}

__global__ __launch_bounds__(SLAB) void qr_panel2_step(int mp, int pb, int c, double* Pt, int64_t ldp,
                                                       const double* part_in, double* part_out, int G,
                                                       const double* row_in, double* row_out, double* Tg) {
    __shared__ double q[PB], d[PB], hh[3];
    const int tid = threadIdx.x;
    {
        // 8 lanes per column sum the slab partials with independent loads in flight, then fold
        const int k = tid >> 3, sub = tid & 7;
        double t = 0.0;
        if (k < pb)
            for (int g = sub; g < G; g += 8) t += part_in[(size_t)g * PB + k];
        t += __shfl_down(t, 4, 8);
        t += __shfl_down(t, 2, 8);
        t += __shfl_down(t, 1, 8);
        if (sub == 0 && k < pb) q[k] = t;
    }
    __syncthreads();
    if (tid == 0) {
        const double alpha = row_in[c], ss = q[c];
        double tau = 0.0, scale = 0.0, beta = alpha;
        if (ss != 0.0) {
            const double nrm = sqrt(fma(alpha, alpha, ss));
            beta = (alpha >= 0.0) ? -nrm : nrm;
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        hh[0] = tau;
        hh[1] = scale;
        hh[2] = beta;
    }
    __syncthreads();
    const double tau = hh[0], scale = hh[1], beta = hh[2];
    if (tid < pb) d[tid] = row_in[tid] + scale * q[tid];
    __syncthreads();

    const int r = blockIdx.x * SLAB + tid;
    double acc[PB];
#pragma unroll
    for (int k = 0; k < PB; ++k) acc[k] = 0.0;
    if (r < mp && r >= c) {
        double pk[PB];
        const double x = Pt[(int64_t)c * ldp + r];
        const double v = (r == c) ? 1.0 : x * scale;
        Pt[(int64_t)c * ldp + r] = (r == c) ? beta : v;
#pragma unroll
        for (int k = 0; k < PB; ++k) {
            if (k < pb) {
                if (k > c) {
                    double p = Pt[(int64_t)k * ldp + r];
                    p = fma(-tau * d[k], v, p);
                    Pt[(int64_t)k * ldp + r] = p;
                    pk[k] = p;
                } else if (k == c) {
                    pk[k] = v;
                } else {
                    pk[k] = Pt[(int64_t)k * ldp + r];
                }
            } else {
                pk[k] = 0.0;
            }
        }
        if (c + 1 < pb) {
            if (r == c + 1) {
#pragma unroll
                for (int k = 0; k < PB; ++k) row_out[k] = pk[k];
            }
            double xn = 0.0;
#pragma unroll
            for (int k = 0; k < PB; ++k)
                if (k == c + 1) xn = pk[k];
            if (r <= c + 1) xn = 0.0;
#pragma unroll
            for (int k = 0; k < PB; ++k) acc[k] = xn * pk[k];
        }
    }
    if (blockIdx.x == 0) {
        // DLARFT: T[0:c, c] = -tau * T[0:c, 0:c] * z,  z_k = d_k (k < c);  T[c][c] = tau
        if (tid < c) {
            double sacc = 0.0;
            for (int j = tid; j < c; ++j) sacc = fma(Tg[tid * PB + j], d[j], sacc);
            Tg[tid * PB + c] = -tau * sacc;
        } else if (tid == c) {
            Tg[c * PB + c] = tau;
        }
    }
    if (c + 1 < pb) slab_reduce(acc, pb, part_out + (size_t)blockIdx.x * PB);
}

__global__ __launch_bounds__(SLAB) void qr_panel2_finish(int mp, int pb, const double* Pt, int64_t ldp, double* W,
                                                         int64_t ldw, double* Tjj, int64_t ldt, double* Rjj,
                                                         int64_t ldr, const double* Tg) {
    const int r = blockIdx.x * SLAB + threadIdx.x;
    if (r < mp) {
#pragma unroll
        for (int k = 0; k < PB; ++k) {
            if (k < pb) {
                const double p = Pt[(int64_t)k * ldp + r];
                W[(int64_t)r * ldw + k] = (r > k) ? p : (r == k ? 1.0 : 0.0);
                if (r < pb) Rjj[(int64_t)r * ldr + k] = (r <= k) ? p : 0.0;
            }
        }
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < pb * pb; i += SLAB) {
            const int a = i / pb, b = i - a * pb;
            Tjj[(int64_t)a * ldt + b] = Tg[a * PB + b];
        }
}

struct QrWorkspace {
    double* Pt;   // PB x m
    double* X1;   // PB x n
    double* X2;   // PB x n
    double* G;    // n x n   (V^T V)
    double* Tmp;  // (n/2 rounded up) x n
    double* Part;    // 2 x slabs x PB   partial sums of the multi-workgroup panel
    double* RowBuf;  // 2 x PB
    double* Tg;      // PB x PB
};

inline size_t align2(size_t x) { return (x + 1) & ~(size_t)1; }

QrWorkspace carve(void* ws, int64_t m, int64_t n) {
    QrWorkspace q;
    double* p = static_cast<double*>(ws);
    q.Pt = p;
    p += align2((size_t)PB * m);
    q.X1 = p;
    p += align2((size_t)PB * n);
    q.X2 = p;
    p += align2((size_t)PB * n);
    q.G = p;
    p += align2((size_t)n * n);
    q.Tmp = p;
    p += align2((size_t)((n + 1) / 2 + PB) * n);
    q.Part = p;
    p += align2((size_t)2 * ceil_div(m, SLAB) * PB);
    q.RowBuf = p;
    p += 2 * PB;
    q.Tg = p;
    return q;
}

// T[lo:hi, lo:hi] is built from panel blocks by merging halves:
//   T12 = -T1 * G[lo:mid, mid:hi] * T2        (T1, T2 upper triangular, already final)
int merge_t(int64_t lo, int64_t hi, double* T, int64_t ldt, const double* G, int64_t ldg, double* Tmp,
            hipStream_t s) {
    const int64_t npanels = ceil_div(hi - lo, PB);
    if (npanels <= 1) return NPW_OK;
    const int64_t mid = lo + (npanels / 2) * PB;
    int rc = merge_t(lo, mid, T, ldt, G, ldg, Tmp, s);
    if (rc) return rc;
    rc = merge_t(mid, hi, T, ldt, G, ldg, Tmp, s);
    if (rc) return rc;
    const int64_t w1 = mid - lo, w2 = hi - mid;
    // Tmp (w1 x w2) = G12 * T2
    rc = gemm<double>('N', 'N', w1, w2, w2, 1.0, G + lo * ldg + mid, ldg, T + mid * ldt + mid, ldt, 0.0, nullptr,
                      0, Tmp, w2, GemmOpts(), s);
    if (rc) return rc;
    // T12 = -T1 * Tmp
    return gemm<double>('N', 'N', w1, w2, w1, -1.0, T + lo * ldt + lo, ldt, Tmp, w2, 0.0, nullptr, 0,
                        T + lo * ldt + mid, ldt, GemmOpts(), s);
}

}  // namespace
}  // namespace npw

using namespace npw;

extern "C" {

size_t npw_dgeqrt_workspace_bytes(int64_t m, int64_t n) {
    if (m <= 0 || n <= 0) return 0;
    const size_t doubles = align2((size_t)PB * m) + 2 * align2((size_t)PB * n) + align2((size_t)n * n) +
                           align2((size_t)((n + 1) / 2 + PB) * n) + align2((size_t)2 * ceil_div(m, SLAB) * PB) +
                           2 * PB + PB * PB;
    return doubles * sizeof(double);
}

int npw_dgeqrt(int64_t m, int64_t n, const double* A, int64_t lda, double* V, int64_t ldv,
               double* T, int64_t ldt, double* R, int64_t ldr, void* workspace,
               npw_stream_t stream) {
    NPW_REQUIRE(m >= 0 && n >= 0, "npw_dgeqrt: negative dimension");
    if (n == 0) return NPW_OK;
    if (m < n) return set_error(NPW_ERR_UNSUPPORTED, "npw_dgeqrt: m (%lld) < n (%lld) is not supported",
                                (long long)m, (long long)n);
    NPW_REQUIRE(A && V && T && R && workspace, "npw_dgeqrt: NULL argument");
    NPW_REQUIRE(lda >= n && ldv >= n && ldt >= n && ldr >= n, "npw_dgeqrt: leading dimension too small");
    NPW_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "npw_dgeqrt: workspace not 16B aligned");
    NPW_REQUIRE((const void*)A != (const void*)V, "npw_dgeqrt: V must not alias A");
    hipStream_t s = as_stream(stream);
    const QrWorkspace q = carve(workspace, m, n);

    // working copy: the factorisation runs in place inside V
    NPW_HIP_CHECK(hipMemcpy2DAsync(V, ldv * 8, A, lda * 8, n * 8, m, hipMemcpyDeviceToDevice, s));
    NPW_HIP_CHECK(hipMemset2DAsync(T, ldt * 8, 0, n * 8, n, s));
    NPW_HIP_CHECK(hipMemset2DAsync(R, ldr * 8, 0, n * 8, n, s));

    for (int64_t j0 = 0; j0 < n; j0 += PB) {
        const int64_t pb = (n - j0 < PB) ? n - j0 : PB;
        const int64_t mp = m - j0;
        double* Wp = V + j0 * ldv + j0;
        if (mp <= 1024) {
            hipLaunchKernelGGL(qr_panel_kernel, dim3(1), dim3(PANEL_THREADS), 0, s, (int)mp, (int)pb, Wp, ldv, q.Pt,
                               mp, T + j0 * ldt + j0, ldt, R + j0 * ldr + j0, ldr);
            NPW_LAUNCH_CHECK();
        } else {
            const int G = (int)ceil_div(mp, SLAB);
            double* part[2] = {q.Part, q.Part + (size_t)G * PB};
            double* rowb[2] = {q.RowBuf, q.RowBuf + PB};
            hipLaunchKernelGGL(qr_panel2_prep, dim3(G), dim3(SLAB), 0, s, (int)mp, (int)pb, Wp, ldv, q.Pt, mp, part[0],
                               rowb[0], q.Tg);
            for (int c = 0; c < (int)pb; ++c)
                hipLaunchKernelGGL(qr_panel2_step, dim3(G), dim3(SLAB), 0, s, (int)mp, (int)pb, c, q.Pt, mp,
                                   part[c & 1], part[(c + 1) & 1], G, rowb[c & 1], rowb[(c + 1) & 1], q.Tg);
            hipLaunchKernelGGL(qr_panel2_finish, dim3(G), dim3(SLAB), 0, s, (int)mp, (int)pb, q.Pt, mp, Wp, ldv,
                               T + j0 * ldt + j0, ldt, R + j0 * ldr + j0, ldr, q.Tg);
            NPW_LAUNCH_CHECK();
        }
        const int64_t n2 = n - j0 - pb;
        if (n2 > 0) {
            double* W2 = Wp + pb;
            // X1 = V_p^T W2 is 32 x n2 with a contraction over all mp rows: split k so that the launch has
            // a few hundred workgroups instead of n2/64 (the V^T V buffer is free until the panels are done)
            GemmOpts sk;
            int64_t want = 512 / (ceil_div(n2, 64) > 0 ? ceil_div(n2, 64) : 1);
            if (want > mp / 256) want = mp / 256;
            if (want > 32) want = 32;
            if (want > 1 && (size_t)want * pb * n2 <= (size_t)n * n) {
                sk.splitk = (int)want;
                sk.splitk_ws = q.G;
            }
            int rc = gemm<double>('T', 'N', pb, n2, mp, 1.0, Wp, ldv, W2, ldv, 0.0, nullptr, 0, q.X1, n2, sk, s);
            if (rc) return rc;
            rc = gemm<double>('T', 'N', pb, n2, pb, 1.0, T + j0 * ldt + j0, ldt, q.X1, n2, 0.0, nullptr, 0, q.X2,
                              n2, GemmOpts(), s);
            if (rc) return rc;
            rc = gemm<double>('N', 'N', mp, n2, pb, -1.0, Wp, ldv, q.X2, n2, 1.0, W2, ldv, W2, ldv, GemmOpts(), s);
            if (rc) return rc;
            // rows j0 .. j0+pb of the updated trailing block are final rows of R; V is zero there
            NPW_HIP_CHECK(hipMemcpy2DAsync(R + j0 * ldr + j0 + pb, ldr * 8, W2, ldv * 8, n2 * 8, pb,
                                           hipMemcpyDeviceToDevice, s));
            NPW_HIP_CHECK(hipMemset2DAsync(W2, ldv * 8, 0, n2 * 8, pb, s));
        }
    }
    if (n > PB) {
        // G = V^T V, then the off-diagonal blocks of T bottom-up
        int rc = gemm<double>('T', 'N', n, n, m, 1.0, V, ldv, V, ldv, 0.0, nullptr, 0, q.G, n, GemmOpts(), s);
        if (rc) return rc;
        rc = merge_t(0, n, T, ldt, q.G, n, q.Tmp, s);
        if (rc) return rc;
    }
    return NPW_OK;
}

}  // extern "C"
