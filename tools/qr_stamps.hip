#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "npw_hip.h"
extern "C" int npw_debug_qr_stamps(long long* out, int reset);
int main() {
    for (int m : {256, 4096, 8192}) {
        const int n = 32;
        std::vector<double> h((size_t)m * n);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
        double *A, *V, *T, *R; void* ws;
        hipMalloc(&A, h.size() * 8); hipMalloc(&V, h.size() * 8); hipMalloc(&T, n * n * 8); hipMalloc(&R, n * n * 8);
        hipMalloc(&ws, npw_dgeqrt_workspace_bytes(m, n));
        hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        for (int i = 0; i < 3; ++i) npw_dgeqrt(m, n, A, n, V, n, T, n, R, n, ws, nullptr);
        hipDeviceSynchronize();
        npw_debug_qr_stamps(nullptr, 1);
        const int reps = 50;
        for (int i = 0; i < reps; ++i) npw_dgeqrt(m, n, A, n, V, n, T, n, R, n, ws, nullptr);
        hipDeviceSynchronize();
        long long st[8];
        npw_debug_qr_stamps(st, 0);
        const double per = 10.0 / 1000.0 / reps / 32;  // 10 ns units -> us per column
        printf("m=%5d: per column  hand-off wait %.2f us | scalar %.2f us | row update + T %.2f us | publish %.2f us | sum %.2f us\n", m,
               st[0] * per, st[1] * per, st[2] * per, st[3] * per, (st[0] + st[1] + st[2] + st[3]) * per);
    }
    return 0;
}
