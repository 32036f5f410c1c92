// Developer aid: two workgroups bounce a 16-byte {value, tag} slot: round-trip latency by cache-scope bits and by
// placement (same / different XCD).  hipcc --offload-arch=gfx950 -O2 tools/slot_pingpong.hip -o slot_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double slot_t __attribute__((ext_vector_type(2)));
template <int MODE> __device__ inline void st(slot_t* p, slot_t x) {
    if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
    if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(x) : "memory");
    if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(x) : "memory");
    if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off nt sc1" ::"v"(p), "v"(x) : "memory");
}
template <int MODE> __device__ inline slot_t ld(const slot_t* p) {
    slot_t x;
    if (MODE == 0) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(p) : "memory");
    if (MODE == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(p) : "memory");
    if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(p) : "memory");
    if (MODE == 3) asm volatile("global_load_dwordx4 %0, %1, off nt sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(p) : "memory");
    return x;
}
template <int MODE> __global__ void pp(slot_t* a, slot_t* b, int partner, int iters, long long* out, int* xcc) {
    if (threadIdx.x != 0) return;
    if (blockIdx.x != 0 && blockIdx.x != partner) return;
    int id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    const bool first = blockIdx.x == 0;
    xcc[first ? 0 : 1] = id & 0xf;
    long long t0 = wall_clock64();
    int done = 0;
    for (int i = 1; i <= iters; ++i) {
        slot_t x; x[0] = i; x[1] = __longlong_as_double((long long)i);
        bool ok = false;
        if (first) {
            st<MODE>(a, x);
            for (int spin = 0; spin < 20000; ++spin) { slot_t y = ld<MODE>(b); if (__double_as_longlong(y[1]) == i) { ok = true; break; } }
        } else {
            for (int spin = 0; spin < 20000; ++spin) { slot_t y = ld<MODE>(a); if (__double_as_longlong(y[1]) == i) { ok = true; break; } }
            st<MODE>(b, x);
        }
        if (!ok) break;   // the partner's store never became visible with these scope bits
        ++done;
    }
    if (first) { out[0] = wall_clock64() - t0; out[1] = done; }
}
template <int MODE> void run(const char* name, slot_t* a, slot_t* b, long long* out, int* xcc, int partner) {
    hipMemset(a, 0, 64); hipMemset(b, 0, 64);
    const int iters = 500;
    hipLaunchKernelGGL(pp<MODE>, dim3(partner + 1), dim3(64), 0, 0, a, b, partner, iters, out, xcc);
    hipDeviceSynchronize();
    long long t[2]; int x[2];
    hipMemcpy(t, out, 16, hipMemcpyDeviceToHost); hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost);
    if (t[1] == iters) printf("%-14s partner block %2d (xcc %d vs %d): round trip %.0f ns\n", name, partner, x[0], x[1], t[0] * 10.0 / iters);
    else printf("%-14s partner block %2d (xcc %d vs %d): NOT VISIBLE after %lld round trips\n", name, partner, x[0], x[1], t[1]);
    fflush(stdout);
}
int main() {
    slot_t *a, *b; long long* out; int* xcc;
    hipMalloc(&a, 4096); hipMalloc(&b, 4096); hipMalloc(&out, 64); hipMalloc(&xcc, 64);
    b = a + 64;  // different cache lines
    for (int partner : {1, 8, 16}) {
        run<0>("sc1", a, b, out, xcc, partner);
        run<1>("sc0 sc1", a, b, out, xcc, partner);
        run<2>("sc0", a, b, out, xcc, partner);
        run<3>("nt sc1", a, b, out, xcc, partner);
    }
    return 0;
}
