// Developer aid: phase timing of potrf_diag_kernel (build factor.hip with -DNPW_DIAG_STAMPS, see tools/diag_stamps.sh)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
extern "C" int npw_debug_diag(double* A, int64_t lda, int32_t* info, double* Winv, long long* stamps);
extern "C" int npw_debug_fused(double* A, int64_t lda, int m_below, int32_t* info, double* Winv, void* msg, unsigned long long tag,
                               long long* stamps);
int main() {
    const int n = 128, lda = 4096;
    std::vector<double> h((size_t)n * lda, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) h[(size_t)i * lda + j] = (i == j ? 200.0 : 0.0) + 1.0 / (1 + abs(i - j));
    double *A, *W;
    int32_t* info;
    long long* st;
    (void)hipMalloc(&A, h.size() * 8);
    (void)hipMalloc(&W, 128 * 128 * 8);
    (void)hipMalloc(&info, 4);
    (void)hipMalloc(&st, 32 * 8);
    (void)hipMemset(info, 0, 4);
    for (int it = 0; it < 3; ++it) {
        (void)hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        npw_debug_diag(A, lda, info, W, st);
        long long t[32];
        (void)hipMemcpy(t, st, 32 * 8, hipMemcpyDeviceToHost);
        printf("load %lld | P(0) %lld |", t[1] - t[0], t[2] - t[1]);
        for (int jb = 0; jb < 8; ++jb) printf(" [Ucol %lld P+Urest %lld]", t[3 + 2 * jb] - t[2 + 2 * jb], t[(jb < 7 ? 4 + 2 * jb : 18)] - t[3 + 2 * jb]);
        printf(" | storeL %lld | trtri %lld | storeInv %lld | total %lld (x10 ns)\n", t[19] - t[18], t[20] - t[19], t[21] - t[20], t[21] - t[0]);
        printf("  clock64 delta %lld over %lld x10ns -> %.0f MHz\n", t[23] - t[22], t[21] - t[0], (double)(t[23] - t[22]) / ((t[21] - t[0]) * 0.01));
    }
    {   // the fused block-column launch: diagonal block + 3968 panel rows following it by substitution
        const int m = 3968, ld2 = 4096;
        std::vector<double> hp((size_t)(n + m) * ld2, 0.0);
        for (int i = 0; i < n + m; ++i)
            for (int j = 0; j < n; ++j) hp[(size_t)i * ld2 + j] = (i == j ? 200.0 : 0.0) + 1.0 / (1 + abs(i - j));
        double* Ap;
        void* msg;
        (void)hipMalloc(&Ap, hp.size() * 8);
        (void)hipMalloc(&msg, 8 * 512 * 16);
        for (int it = 0; it < 3; ++it) {
            (void)hipMemcpy(Ap, hp.data(), hp.size() * 8, hipMemcpyHostToDevice);
            (void)hipMemset(st, 0, 32 * 8);
            npw_debug_fused(Ap, ld2, m, info, W, msg, 1000ull * (it + 1), st);
            long long t[32];
            (void)hipMemcpy(t, st, 32 * 8, hipMemcpyDeviceToHost);
            printf("fused: load %lld | P(0) %lld |", t[1] - t[0], t[2] - t[1]);
            for (int jb = 0; jb < 8; ++jb) printf(" [%lld %lld]", t[3 + 2 * jb] - t[2 + 2 * jb], t[(jb < 7 ? 4 + 2 * jb : 18)] - t[3 + 2 * jb]);
            printf(" | diag block done at %lld | last panel workgroup done at %lld (x10 ns)\n", t[18] - t[0], t[24] - t[0]);
        }
    }
    return 0;
}
