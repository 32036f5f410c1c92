// devcheck -- developer harness (not part of the product): checks libnpw_hip GEMM variants
// against a naive host loop on small shapes and times the 4096^3 trailing update + an MFMA
// issue-rate microbenchmark.  Build: make -C numpywren_amd/csrc devcheck
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "npw_hip.h"

#define CK(x)                                                              \
    do {                                                                   \
        int _r = (x);                                                      \
        if (_r != 0) {                                                     \
            printf("FAIL %s -> %d: %s\n", #x, _r, npw_last_error());       \
            exit(1);                                                       \
        }                                                                  \
    } while (0)

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));

__global__ void mfma_f64_peak(double* out, int iters) {
    d4_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = {0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void mfma_f32_peak(float* out, int iters) {
    f4_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = {0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double urand() { return (double)rand() / RAND_MAX * 2.0 - 1.0; }

template <typename T>
static double check_gemm(char ta, char tb, int m, int n, int k, int pad) {
    const int a_rows = (ta == 'N') ? m : k, a_cols = (ta == 'N') ? k : m;
    const int b_rows = (tb == 'N') ? k : n, b_cols = (tb == 'N') ? n : k;
    const int lda = a_cols + pad, ldb = b_cols + pad, ldc = n + pad, ldd = n + pad;
    std::vector<T> A((size_t)a_rows * lda), B((size_t)b_rows * ldb), C((size_t)m * ldc), D((size_t)m * ldd, 0);
    for (auto& x : A) x = (T)urand();
    for (auto& x : B) x = (T)urand();
    for (auto& x : C) x = (T)urand();
    T *dA, *dB, *dC, *dD;
    CK(npw_malloc((void**)&dA, A.size() * sizeof(T) + 16));
    CK(npw_malloc((void**)&dB, B.size() * sizeof(T) + 16));
    CK(npw_malloc((void**)&dC, C.size() * sizeof(T) + 16));
    CK(npw_malloc((void**)&dD, D.size() * sizeof(T) + 16));
    CK(npw_memcpy_h2d_async(dA, A.data(), A.size() * sizeof(T), 0));
    CK(npw_memcpy_h2d_async(dB, B.data(), B.size() * sizeof(T), 0));
    CK(npw_memcpy_h2d_async(dC, C.data(), C.size() * sizeof(T), 0));
    CK(npw_memset_async(dD, 0, D.size() * sizeof(T), 0));
    const T alpha = (T)-1.0, beta = (T)1.0;
    if (sizeof(T) == 8)
        CK(npw_dgemm(ta, tb, m, n, k, alpha, (double*)dA, lda, (double*)dB, ldb, beta, (double*)dC, ldc,
                     (double*)dD, ldd, nullptr, 0));
    else
        CK(npw_sgemm(ta, tb, m, n, k, alpha, (float*)dA, lda, (float*)dB, ldb, beta, (float*)dC, ldc,
                     (float*)dD, ldd, nullptr, 0));
    CK(npw_memcpy_d2h_async(D.data(), dD, D.size() * sizeof(T), 0));
    CK(npw_device_synchronize());
    double maxerr = 0;
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0;
            for (int kk = 0; kk < k; ++kk) {
                double a = (ta == 'N') ? A[(size_t)i * lda + kk] : A[(size_t)kk * lda + i];
                double b = (tb == 'N') ? B[(size_t)kk * ldb + j] : B[(size_t)j * ldb + kk];
                s += a * b;
            }
            double ref = (double)beta * C[(size_t)i * ldc + j] + (double)alpha * s;
            maxerr = std::fmax(maxerr, std::fabs(ref - (double)D[(size_t)i * ldd + j]));
        }
    // padding columns of D must be untouched
    for (int i = 0; i < m; ++i)
        for (int j = n; j < ldd; ++j)
            if (D[(size_t)i * ldd + j] != 0) maxerr = 1e30;
    npw_free(dA);
    npw_free(dB);
    npw_free(dC);
    npw_free(dD);
    return maxerr;
}

// host reference Cholesky (lower), returns false if not PD
static bool host_chol(int n, std::vector<double>& a) {
    for (int j = 0; j < n; ++j) {
        double d = a[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= a[(size_t)j * n + k] * a[(size_t)j * n + k];
        if (!(d > 0)) return false;
        d = std::sqrt(d);
        a[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = a[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= a[(size_t)i * n + k] * a[(size_t)j * n + k];
            a[(size_t)i * n + j] = s / d;
        }
        for (int k = j + 1; k < n; ++k) a[(size_t)j * n + k] = 0;
    }
    return true;
}

static int check_factor(int n, int m) {
    // SPD matrix A = G G^T + n I
    std::vector<double> G((size_t)n * n), A((size_t)n * n), L((size_t)n * n), B((size_t)m * n), X((size_t)m * n);
    for (auto& x : G) x = urand();
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = 0;
            for (int k = 0; k < n; ++k) s += G[(size_t)i * n + k] * G[(size_t)j * n + k];
            if (i == j) s += n;
            A[(size_t)i * n + j] = A[(size_t)j * n + i] = s;
        }
    for (auto& x : B) x = urand();
    double *dA, *dL, *dB, *dX;
    void *wsp, *wst;
    int32_t* dinfo;
    CK(npw_malloc((void**)&dA, A.size() * 8));
    CK(npw_malloc((void**)&dL, A.size() * 8));
    CK(npw_malloc((void**)&dB, B.size() * 8));
    CK(npw_malloc((void**)&dX, B.size() * 8));
    CK(npw_malloc(&wsp, npw_dpotrf_lower_workspace_bytes(n)));
    CK(npw_malloc(&wst, npw_dtrsm_rltn_workspace_bytes(m, n)));
    CK(npw_malloc((void**)&dinfo, 4));
    CK(npw_memcpy_h2d_async(dA, A.data(), A.size() * 8, 0));
    CK(npw_memcpy_h2d_async(dB, B.data(), B.size() * 8, 0));
    CK(npw_dpotrf_lower(n, dA, n, dL, n, dinfo, wsp, 0));
    CK(npw_dtrsm_rltn(m, n, dL, n, dB, n, dX, n, wst, 0));
    int32_t info = -1;
    CK(npw_memcpy_d2h_async(L.data(), dL, L.size() * 8, 0));
    CK(npw_memcpy_d2h_async(X.data(), dX, X.size() * 8, 0));
    CK(npw_memcpy_d2h_async(&info, dinfo, 4, 0));
    CK(npw_device_synchronize());
    std::vector<double> Lref = A;
    bool pd = host_chol(n, Lref);
    double el = 0, lmax = 0;
    for (size_t i = 0; i < L.size(); ++i) {
        el = std::fmax(el, std::fabs(L[i] - Lref[i]));
        lmax = std::fmax(lmax, std::fabs(Lref[i]));
    }
    // trsm residual: X L^T = B
    double et = 0;
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0;
            for (int k = 0; k <= j; ++k) s += X[(size_t)i * n + k] * Lref[(size_t)j * n + k];
            et = std::fmax(et, std::fabs(s - B[(size_t)i * n + j]));
        }
    bool ok = pd && info == 0 && el < 1e-12 * lmax * n && et < 1e-11 * n;
    printf("factor n=%d m=%d: info=%d |L-Lref|max=%.3e (|L|max %.2e)  trsm resid=%.3e %s\n", n, m, info, el, lmax, et,
           ok ? "ok" : "BAD");
    // non-PD detection: flip a diagonal entry
    A[(size_t)(n / 2) * n + n / 2] = -1.0;
    CK(npw_memcpy_h2d_async(dA, A.data(), A.size() * 8, 0));
    CK(npw_dpotrf_lower(n, dA, n, dL, n, dinfo, wsp, 0));
    CK(npw_memcpy_d2h_async(&info, dinfo, 4, 0));
    CK(npw_device_synchronize());
    if (info != n / 2 + 1) {
        printf("  non-PD detection BAD: info=%d expected %d\n", info, n / 2 + 1);
        ok = false;
    }
    npw_free(dA); npw_free(dL); npw_free(dB); npw_free(dX); npw_free(wsp); npw_free(wst); npw_free(dinfo);
    return ok ? 0 : 1;
}

// host Householder QR (LAPACK dgeqr2 + dlarft conventions), row-major
static void host_qr(int m, int n, const std::vector<double>& A, std::vector<double>& V, std::vector<double>& T,
                    std::vector<double>& R) {
    std::vector<double> W = A;
    std::vector<double> tau(n);
    for (int c = 0; c < n; ++c) {
        double ss = 0;
        for (int r = c + 1; r < m; ++r) ss += W[(size_t)r * n + c] * W[(size_t)r * n + c];
        double alpha = W[(size_t)c * n + c], beta = alpha, t = 0, scale = 0;
        if (ss != 0) {
            double nrm = std::sqrt(alpha * alpha + ss);
            beta = alpha >= 0 ? -nrm : nrm;
            t = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        for (int r = c + 1; r < m; ++r) W[(size_t)r * n + c] *= scale;
        W[(size_t)c * n + c] = beta;
        tau[c] = t;
        for (int k = c + 1; k < n; ++k) {
            double d = W[(size_t)c * n + k];
            for (int r = c + 1; r < m; ++r) d += W[(size_t)r * n + c] * W[(size_t)r * n + k];
            W[(size_t)c * n + k] -= t * d;
            for (int r = c + 1; r < m; ++r) W[(size_t)r * n + k] -= t * d * W[(size_t)r * n + c];
        }
    }
    V.assign((size_t)m * n, 0);
    R.assign((size_t)n * n, 0);
    T.assign((size_t)n * n, 0);
    for (int r = 0; r < m; ++r)
        for (int c = 0; c < n; ++c) {
            if (r > c) V[(size_t)r * n + c] = W[(size_t)r * n + c];
            else if (r == c) V[(size_t)r * n + c] = 1;
            if (r <= c && r < n) R[(size_t)r * n + c] = W[(size_t)r * n + c];
        }
    for (int c = 0; c < n; ++c) {
        std::vector<double> z(c);
        for (int k = 0; k < c; ++k) {
            double d = 0;
            for (int r = 0; r < m; ++r) d += V[(size_t)r * n + k] * V[(size_t)r * n + c];
            z[k] = d;
        }
        for (int t = 0; t < c; ++t) {
            double sacc = 0;
            for (int q = t; q < c; ++q) sacc += T[(size_t)t * n + q] * z[q];
            T[(size_t)t * n + c] = -tau[c] * sacc;
        }
        T[(size_t)c * n + c] = tau[c];
    }
}

static int check_qr(int m, int n) {
    std::vector<double> A((size_t)m * n), V, T, R, Vd((size_t)m * n), Td((size_t)n * n), Rd((size_t)n * n);
    for (auto& x : A) x = urand();
    host_qr(m, n, A, V, T, R);
    double *dA, *dV, *dT, *dR;
    void* ws;
    CK(npw_malloc((void**)&dA, A.size() * 8));
    CK(npw_malloc((void**)&dV, A.size() * 8));
    CK(npw_malloc((void**)&dT, Td.size() * 8));
    CK(npw_malloc((void**)&dR, Rd.size() * 8));
    CK(npw_malloc(&ws, npw_dgeqrt_workspace_bytes(m, n)));
    CK(npw_memcpy_h2d_async(dA, A.data(), A.size() * 8, 0));
    CK(npw_dgeqrt(m, n, dA, n, dV, n, dT, n, dR, n, ws, 0));
    CK(npw_memcpy_d2h_async(Vd.data(), dV, Vd.size() * 8, 0));
    CK(npw_memcpy_d2h_async(Td.data(), dT, Td.size() * 8, 0));
    CK(npw_memcpy_d2h_async(Rd.data(), dR, Rd.size() * 8, 0));
    CK(npw_device_synchronize());
    double ev = 0, et = 0, er = 0;
    for (size_t i = 0; i < V.size(); ++i) ev = std::fmax(ev, std::fabs(V[i] - Vd[i]));
    for (size_t i = 0; i < T.size(); ++i) et = std::fmax(et, std::fabs(T[i] - Td[i]));
    for (size_t i = 0; i < R.size(); ++i) er = std::fmax(er, std::fabs(R[i] - Rd[i]));
    const double tol = 1e-12 * (m + n);
    bool ok = ev < tol && et < tol && er < tol * std::sqrt((double)m);
    printf("geqrt m=%d n=%d: |dV|=%.3e |dT|=%.3e |dR|=%.3e %s\n", m, n, ev, et, er, ok ? "ok" : "BAD");
    npw_free(dA); npw_free(dV); npw_free(dT); npw_free(dR); npw_free(ws);
    return ok ? 0 : 1;
}

int main(int argc, char** argv) {
    int ndev = 0;
    CK(npw_device_count(&ndev));
    char name[128];
    size_t mem;
    int cus, khz;
    CK(npw_device_info(0, name, sizeof(name), &mem, &cus, &khz));
    printf("device0: %s  mem=%.1f GiB  CUs=%d  clock=%d kHz  (ndev=%d)\n", name, mem / 1073741824.0, cus, khz, ndev);

    // ---- correctness on assorted shapes ------------------------------------------------
    int bad = 0;
    const char tr[2] = {'N', 'T'};
    struct Shape { int m, n, k, pad; };
    Shape shapes[] = {{8, 8, 8, 0},     {7, 5, 3, 0},     {64, 64, 16, 0},   {128, 128, 32, 0},
                      {100, 37, 19, 1}, {256, 192, 48, 0}, {130, 257, 33, 3}, {2048, 2048, 64, 0},
                      {2048, 1920, 40, 0}, {1, 1, 1, 0}, {16, 16, 0, 0}};
    for (auto sh : shapes)
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) {
                double e64 = check_gemm<double>(tr[a], tr[b], sh.m, sh.n, sh.k, sh.pad);
                double e32 = check_gemm<float>(tr[a], tr[b], sh.m, sh.n, sh.k, sh.pad);
                bool ok = e64 < 1e-12 * (sh.k + 1) && e32 < 2e-6 * (sh.k + 1);
                if (!ok) ++bad;
                printf("gemm %c%c m=%d n=%d k=%d pad=%d  err64=%.3e err32=%.3e %s\n", tr[a], tr[b], sh.m,
                       sh.n, sh.k, sh.pad, e64, e32, ok ? "ok" : "BAD");
            }
    {
        int ns[] = {1, 8, 37, 128, 129, 200, 256, 300, 640, 1024};
        for (int n : ns) bad += check_factor(n, n);
        bad += check_factor(200, 77);
        bad += check_factor(513, 1000);
        int qs[][2] = {{1, 1}, {8, 8}, {16, 8}, {64, 32}, {100, 37}, {64, 64}, {256, 128}, {300, 300}, {512, 256}, {1024, 200}, {2048, 96}, {1500, 70}, {3000, 33}, {1300, 300}};
        for (auto& q : qs) bad += check_qr(q[0], q[1]);
    }
    printf("correctness: %s (%d bad)\n", bad ? "FAILED" : "PASSED", bad);

    // ---- MFMA issue-rate microbenchmarks ---------------------------------------------------
    {
        double* out;
        CK(npw_malloc((void**)&out, 256 * 8 * 256 * 8));
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        const int iters = 20000;
        for (int waves = 1; waves <= 2; ++waves) {
            int blocks = 256 * waves;  // 256 threads = 1 wave / SIMD per block
            hipLaunchKernelGGL(mfma_f64_peak, dim3(blocks), dim3(256), 0, 0, out, 100);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(mfma_f64_peak, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double flops = (double)blocks * 4 * iters * 8 * 2.0 * 16 * 16 * 4;
            printf("mfma_f64_16x16x4 peak (%d waves/SIMD): %.2f TFLOP/s  (%.3f ms)\n", waves, flops / ms / 1e9, ms);
            hipLaunchKernelGGL(mfma_f32_peak, dim3(blocks), dim3(256), 0, 0, (float*)out, 100);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(mfma_f32_peak, dim3(blocks), dim3(256), 0, 0, (float*)out, iters);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("mfma_f32_16x16x4 peak (%d waves/SIMD): %.2f TFLOP/s  (%.3f ms)\n", waves, flops / ms / 1e9, ms);
        }
        npw_free(out);
    }

    // ---- 4096^3 timings ------------------------------------------------------------------
    {
        const int n = (argc > 1) ? atoi(argv[1]) : 4096;
        size_t bytes = (size_t)n * n * 8;
        double *S, *X, *Y, *D;
        CK(npw_malloc((void**)&S, bytes));
        CK(npw_malloc((void**)&X, bytes));
        CK(npw_malloc((void**)&Y, bytes));
        CK(npw_malloc((void**)&D, bytes));
        CK(npw_fill_random(S, n, n, n, 1, 0, 0, 0));
        CK(npw_fill_random(X, n, n, n, 2, 0, 0, 0));
        CK(npw_fill_random(Y, n, n, n, 3, 0, 0, 0));
        npw_event_t e0, e1;
        CK(npw_event_create(&e0, 1));
        CK(npw_event_create(&e1, 1));
        const char* names[4] = {"NN", "NT(syrk)", "TN", "TT"};
        for (int v = 0; v < 4; ++v) {
            char ta = (v & 2) ? 'T' : 'N', tb = (v & 1) ? 'T' : 'N';
            for (int w = 0; w < 2; ++w)
                CK(npw_dgemm(ta, tb, n, n, n, -1.0, X, n, Y, n, 1.0, S, n, D, n, nullptr, 0));
            const int reps = 10;
            CK(npw_event_record(e0, 0));
            for (int r = 0; r < reps; ++r)
                CK(npw_dgemm(ta, tb, n, n, n, -1.0, X, n, Y, n, 1.0, S, n, D, n, nullptr, 0));
            CK(npw_event_record(e1, 0));
            CK(npw_event_synchronize(e1));
            float ms;
            CK(npw_event_elapsed_ms(e0, e1, &ms));
            ms /= reps;
            printf("dgemm %s n=%d: %.3f ms  %.2f TFLOP/s\n", names[v], n, ms, 2.0 * n * n * (double)n / ms / 1e9);
        }
        if (n >= 4096) {   // latency table for the small products potrf/trsm recursion issues
            const int shapes[][3] = {{128, 128, 128}, {256, 256, 128}, {256, 256, 256}, {512, 512, 512}, {1024, 1024, 1024},
                                     {2048, 128, 128}, {4096, 128, 128}, {4096, 256, 256}, {4096, 512, 512},
                                     {4096, 1024, 1024}, {2048, 2048, 2048}, {4096, 2048, 2048},
                                     {3968, 3968, 128}, {3968, 3968, 256}, {3968, 3968, 512}, {2048, 2048, 128}, {1024, 1024, 128}};
            for (auto& sh : shapes) {
                int m = sh[0], nn = sh[1], k = sh[2];
                for (int v = 0; v < 2; ++v) {
                    char tb = v ? 'T' : 'N';
                    for (int w = 0; w < 3; ++w)
                        CK(npw_dgemm('N', tb, m, nn, k, -1.0, X, n, Y, n, 1.0, S, n, D, n, nullptr, 0));
                    const int reps = 50;
                    CK(npw_event_record(e0, 0));
                    for (int r = 0; r < reps; ++r)
                        CK(npw_dgemm('N', tb, m, nn, k, -1.0, X, n, Y, n, 1.0, S, n, D, n, nullptr, 0));
                    CK(npw_event_record(e1, 0));
                    CK(npw_event_synchronize(e1));
                    float ms;
                    CK(npw_event_elapsed_ms(e0, e1, &ms));
                    ms /= reps;
                    printf("dgemm N%c %4dx%4dx%4d: %8.2f us  %6.2f TFLOP/s\n", tb, m, nn, k, ms * 1e3,
                           2.0 * m * nn * (double)k / ms / 1e9);
                }
            }
        }
        {   // potrf / trsm at tile size: A = S S^T/n + n*I built on device
            CK(npw_dgemm('N', 'T', n, n, n, 1.0 / n, S, n, S, n, 0.0, nullptr, n, D, n, nullptr, 0));
            CK(npw_add_diag(D, n, n, n, (double)n, 0));
            void *wsp, *wst;
            int32_t* dinfo;
            CK(npw_malloc(&wsp, npw_dpotrf_lower_workspace_bytes(n)));
            CK(npw_malloc(&wst, npw_dtrsm_rltn_workspace_bytes(n, n)));
            CK(npw_malloc((void**)&dinfo, 4));
            for (int w = 0; w < 2; ++w) CK(npw_dpotrf_lower(n, D, n, X, n, dinfo, wsp, 0));
            const int reps = 5;
            CK(npw_event_record(e0, 0));
            for (int r = 0; r < reps; ++r) CK(npw_dpotrf_lower(n, D, n, X, n, dinfo, wsp, 0));
            CK(npw_event_record(e1, 0));
            CK(npw_event_synchronize(e1));
            float ms;
            CK(npw_event_elapsed_ms(e0, e1, &ms));
            ms /= reps;
            int32_t info;
            CK(npw_memcpy_d2h_async(&info, dinfo, 4, 0));
            CK(npw_device_synchronize());
            printf("dpotrf n=%d: %.3f ms  %.2f TFLOP/s (n^3/3)  info=%d\n", n, ms, n * (double)n * n / 3 / ms / 1e9, info);
            for (int w = 0; w < 2; ++w) CK(npw_dtrsm_rltn(n, n, X, n, S, n, Y, n, wst, 0));
            CK(npw_event_record(e0, 0));
            for (int r = 0; r < reps; ++r) CK(npw_dtrsm_rltn(n, n, X, n, S, n, Y, n, wst, 0));
            CK(npw_event_record(e1, 0));
            CK(npw_event_synchronize(e1));
            CK(npw_event_elapsed_ms(e0, e1, &ms));
            ms /= reps;
            printf("dtrsm n=%d: %.3f ms  %.2f TFLOP/s (n^3)\n", n, ms, n * (double)n * n / ms / 1e9);
            // residual check at size: ||Y L^T - S||_F / ||S||_F
            double* dsum;
            CK(npw_malloc((void**)&dsum, 16));
            CK(npw_dgemm('N', 'T', n, n, n, 1.0, Y, n, X, n, -1.0, S, n, D, n, nullptr, 0));
            CK(npw_dsumsq(D, n, n, n, dsum, 0));
            CK(npw_dsumsq(S, n, n, n, dsum + 1, 0));
            double hs[2];
            CK(npw_memcpy_d2h_async(hs, dsum, 16, 0));
            CK(npw_device_synchronize());
            printf("trsm residual at n=%d: %.3e\n", n, std::sqrt(hs[0] / hs[1]));
            CK(npw_fill_random(Y, n, n, n, 3, 0, 0, 0));
        }
        {   // QR of one tile and of a stacked pair
            double *V2, *T2;
            void* ws;
            CK(npw_malloc((void**)&V2, 2 * bytes));
            CK(npw_malloc((void**)&T2, bytes));
            CK(npw_malloc(&ws, npw_dgeqrt_workspace_bytes(2 * n, n)));
            for (int mm = n; mm <= 2 * n; mm += n) {
                // input: rows of S (and Y for the stacked case) -- use X as R output
                double* In = V2;  // reuse: copy S into a 2n x n buffer D2
                double* D2;
                CK(npw_malloc((void**)&D2, 2 * bytes));
                CK(npw_memcpy_d2d_async(D2, S, bytes, 0));
                CK(npw_memcpy_d2d_async(D2 + (size_t)n * n, Y, bytes, 0));
                (void)In;
                CK(npw_dgeqrt(mm, n, D2, n, V2, n, T2, n, X, n, ws, 0));
                CK(npw_event_record(e0, 0));
                CK(npw_dgeqrt(mm, n, D2, n, V2, n, T2, n, X, n, ws, 0));
                CK(npw_event_record(e1, 0));
                CK(npw_event_synchronize(e1));
                float ms;
                CK(npw_event_elapsed_ms(e0, e1, &ms));
                printf("dgeqrt m=%d n=%d: %.3f ms  %.2f TFLOP/s (2mn^2-2n^3/3)\n", mm, n, ms,
                       (2.0 * mm * n * (double)n - 2.0 * n * (double)n * n / 3) / ms / 1e9);
                npw_free(D2);
            }
            npw_free(V2); npw_free(T2); npw_free(ws);
            CK(npw_fill_random(X, n, n, n, 2, 0, 0, 0));
        }
        float *Xf = (float*)X, *Yf = (float*)Y, *Df = (float*)D;
        CK(npw_convert(n, n, S, n, 0, Xf, n, 1, 0));
        CK(npw_convert(n, n, S, n, 0, Yf, n, 1, 0));
        for (int v = 0; v < 4; ++v) {
            char ta = (v & 2) ? 'T' : 'N', tb = (v & 1) ? 'T' : 'N';
            for (int w = 0; w < 2; ++w)
                CK(npw_sgemm(ta, tb, n, n, n, 1.0f, Xf, n, Yf, n, 0.0f, nullptr, n, Df, n, nullptr, 0));
            const int reps = 10;
            CK(npw_event_record(e0, 0));
            for (int r = 0; r < reps; ++r)
                CK(npw_sgemm(ta, tb, n, n, n, 1.0f, Xf, n, Yf, n, 0.0f, nullptr, n, Df, n, nullptr, 0));
            CK(npw_event_record(e1, 0));
            CK(npw_event_synchronize(e1));
            float ms;
            CK(npw_event_elapsed_ms(e0, e1, &ms));
            ms /= reps;
            printf("sgemm %s n=%d: %.3f ms  %.2f TFLOP/s\n", names[v], n, ms, 2.0 * n * n * (double)n / ms / 1e9);
        }
    }
    return bad ? 1 : 0;
}
