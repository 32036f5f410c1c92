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
//   for each panel:  qr_panel3_kernel (256-row slabs, one workgroup each, resident for the whole panel: rows in
//                    registers, one fused reduction and one slab-to-slab hand-off per column) produces the
//                    panel's reflectors, its PB x PB T block and its R block;
//                    trailing columns:  W2 -= V_p * (T_p^T * (V_p^T * W2)), at three widths of reflector (geqrt_core):
//                    the panel's own (k = 32: streaming MFMA kernels near_vtw / reduce_tt / near_update, ONE launch --
//                    near_fused_kernel -- for a single or small-batch factorisation), the OB = 128 wide block's and
//                    the SB wide superblock's (batched GEMMs on helper streams);
//   T off-diagonal blocks: DLARFT's recurrence per block / superblock column during the factorisation, or bottom-up
//                    T12 = -T1 * (V1^T V2) * T2 with V^T V from one big GEMM; not at all for T == NULL.
#include "npw_internal.h"

#include <algorithm>
#include <atomic>
#include <type_traits>
#include <chrono>

namespace npw {
namespace {

constexpr int PB = 32;    // panel width (columns one thread keeps in registers)
constexpr int OB = 128;   // outer block: PB-wide panels are aggregated into one OB-wide block reflector for the columns further right
// Superblock: OB-wide block reflectors are aggregated once more into one SB-wide reflector (V_S, T_S = the diagonal block
// of T) for everything right of the superblock's look-ahead window, so that the bulk of the trailing update is k = SB
// products on 128 x 128 tiles instead of k = 128 products (prologue / epilogue-bound, 4 x the passes over the trailing
// matrix).  $NPW_QR_SB overrides (a multiple of OB; OB itself gives the two-level scheme of rounds 1 - 3).
inline int64_t superblock_width(bool tri) {
    // Measured on batches of 32 4096^2 tiles (round 4, gpurun_out/r04a, r04d): dense 113.6 / 105.2 / 109.4 / 119.7 ms at
    // SB = 128 / 256 / 512 / 1024 (R only: 83.6 / 79.1 / 82.6 / 91.7) -- wider superblocks make the far products more
    // efficient and the k = OB updates inside the superblock's window more numerous; stacked triangles (a third of the
    // update work, growing row counts) 87.7 / 90.2 / 93.3 at 128 / 256 / 512.
    static const int64_t sb[2] = {[] {
        const char* e = getenv("NPW_QR_SB");
        int64_t v = e ? atoll(e) : 256;
        return (v < OB ? (int64_t)OB : v / OB * OB);
    }(), [] {
        const char* e = getenv("NPW_QR_SB_TRI") ? getenv("NPW_QR_SB_TRI") : getenv("NPW_QR_SB");
        int64_t v = e ? atoll(e) : 128;
        return (v < OB ? (int64_t)OB : v / OB * OB);
    }()};
    return sb[tri ? 1 : 0];
}
inline int64_t superblock_width_max() {
    const int64_t a = superblock_width(false), c = superblock_width(true);
    return a > c ? a : c;
}
constexpr int LA = 2 * OB;  // columns right of a superblock that its blocks' own (k = OB) updates keep current: the panel
                            // chain's two-block window must never reach into columns that still miss earlier reflectors
                            // (OB is NOT enough: the next superblock's first block already updates its successor per panel
                            //  -- measured: wrong factors at LA = OB; LA = 3 OB: the same speed)
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

constexpr int SLAB = 256;  // threads of a panel workgroup = rows of the panel per workgroup and row-per-thread
constexpr int PANEL_MAX_WGS = 256;  // workgroups of one panel launch with one row per thread (beyond: two rows)

// f(integral_constant<0>), f(integral_constant<1>), ... f(integral_constant<PB - 1>): a loop the compiler cannot refuse to unroll
template <int C, typename F>
__device__ __forceinline__ void unrolled_columns(F& f) {
    if constexpr (C < PB) {
        f(std::integral_constant<int, C>{});
        unrolled_columns<C + 1>(f);
    }
}

#ifdef NPW_QR_STAMPS  // developer timing (tools/qr_stamps.sh): where a column's time goes inside workgroup 0, 10 ns units
__device__ long long qr_stamps[8];
#define QR_STAMP(i, t0)                                                           \
    {                                                                             \
        const long long _t = wall_clock64();                                      \
        qr_acc[i] += _t - t0; /* registers: a global update here would itself cost ~1 us */ \
        t0 = _t;                                                                  \
    }
#else
#define QR_STAMP(i, t0)
#endif

// ------------------------------------------------------------------------------------------------
// Panel kernel: ONE launch per panel, the workgroups stay resident for all pb columns: every thread keeps its row of the panel (32 values) in
// registers, the per-column hand-off between the slabs (one partial-sum vector per slab + the next pivot row)
// goes through device-coherent memory, and the columns are separated by a grid barrier on a monotonic counter
// instead of a kernel boundary (~20 us per column -> a few).
//   publish : write-through stores (sc1), then s_waitcnt vmcnt(0) in every publishing wave, workgroup barrier,
//             one relaxed agent-scope atomic add by thread 0  (no release fence: see factor.hip)
//   consume : thread 0 spins on the counter (relaxed, s_sleep), workgroup barrier, sc1 loads of the partials
// The grid (<= 32 workgroups of 256 threads, no LDS to speak of) is always co-resident.
// ------------------------------------------------------------------------------------------------
// A hand-off slot: value + sequence tag, written with ONE 16-byte write-through store and read with one 16-byte
// device-coherent load, so a reader that sees the expected tag also sees the value: arrival of the data is the
// synchronisation (no counter, no fence, no separate barrier; one memory hop per column instead of three).
typedef double slot_t __attribute__((ext_vector_type(2)));  // {value, tag bits}

// Lanes whose bounded wait for a hand-off slot expired since the last reset.  A lost hand-off leaves the call's results
// undefined (the wait is bounded so that it cannot hang the GPU); the count is what lets the host fail loudly instead of
// returning them (npw_dgeqrt_handoff_timeouts; the executor checks it when a run with QR tasks settles).
__device__ int qr_handoff_timeouts = 0;

// (The trailing s_nop: a store of more than 8 bytes keeps reading its data registers for a wait state or two after
// issue; the compiler pads its own stores against a following VALU write of those registers but cannot see into the
// asm -- without the padding lanes 12..15 of every 16 stored the NEXT slot's contents when stores followed each other.)
__device__ inline void st_slot(slot_t* p, double v, unsigned long long tag) {
    slot_t x;
    x[0] = v;
    x[1] = __longlong_as_double((long long)tag);
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}
// poll up to four slots until each carries `tag` (slots whose `need` flag is false are not waited for).  NL = how many
// of the four can be needed at all: the others are not even loaded -- every poll of every lane is a device-coherent
// load that travels to the memory side, and a batch of factorisations polls from hundreds of workgroups at once.
// `nap`: s_sleep units between two polls (1 for a single factorisation: latency is everything; more in a batch).
template <int NL>
__device__ inline void ld_slots4(const slot_t* p0, const slot_t* p1, const slot_t* p2, const slot_t* p3, bool n0, bool n1,
                                 bool n2, bool n3, unsigned long long tag, int nap, double& a, double& b, double& c, double& d) {
    const double want = __longlong_as_double((long long)tag);
    for (int spin = 0; spin < (1 << 22); ++spin) {  // bounded: a lost hand-off must not hang the GPU
        slot_t x0, x1, x2, x3;
        if constexpr (NL == 1) {
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x0) : "v"(p0) : "memory");
            x1 = x2 = x3 = x0;
        } else if constexpr (NL == 2) {
            asm volatile(
                "global_load_dwordx4 %0, %2, off sc1\n\t"
                "global_load_dwordx4 %1, %3, off sc1\n\t"
                "s_waitcnt vmcnt(0)"
                : "=&v"(x0), "=&v"(x1)
                : "v"(p0), "v"(p1)
                : "memory");
            x2 = x3 = x0;
        } else {
            asm volatile(
                "global_load_dwordx4 %0, %4, off sc1\n\t"
                "global_load_dwordx4 %1, %5, off sc1\n\t"
                "global_load_dwordx4 %2, %6, off sc1\n\t"
                "global_load_dwordx4 %3, %7, off sc1\n\t"
                "s_waitcnt vmcnt(0)"
                : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)
                : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
                : "memory");
        }
        a = x0[0];
        b = x1[0];
        c = x2[0];
        d = x3[0];
        const bool ok = (!n0 || __double_as_longlong(x0[1]) == __double_as_longlong(want)) &&
                        (NL < 2 || !n1 || __double_as_longlong(x1[1]) == __double_as_longlong(want)) &&
                        (NL < 3 || !n2 || __double_as_longlong(x2[1]) == __double_as_longlong(want)) &&
                        (NL < 3 || !n3 || __double_as_longlong(x3[1]) == __double_as_longlong(want));
        if (ok) return;
        if (nap <= 1)
            __builtin_amdgcn_s_sleep(1);
        else
            __builtin_amdgcn_s_sleep(8);
    }
    atomicAdd(&qr_handoff_timeouts, 1);   // expired: the values returned above are not the ones waited for
}

// ---- wave-level reduce-scatter of 32 per-lane values over the 64 lanes of a wave ---------------------------------
// Lane l returns the sum over all lanes of acc[l >> 1].  Five halving steps (each lane keeps half of its values and
// adds its partner's copies of them: 16 + 8 + 4 + 2 + 1 additions) and one plain exchange, ~125 instructions, no LDS,
// no barrier: v_permlane32_swap / v_permlane16_swap move 32-bit halves between the wave's halves / odd and even
// rows of 16 lanes (gfx950), below that DPP row_mirror, row_half_mirror and two quad permutations -- any pairing across
// the two halves of the current group does.  (The 64-bit ALU takes no DPP modifier except row_newbcast, so the
// partner's value is fetched with two 32-bit moves.)  Fixed order: deterministic.
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const long long b = __double_as_longlong(v);
    int lo = (int)b, hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double swap_add32(double a, double b) {   // lanes < 32: a + a(lane + 32); lanes >= 32: b(lane - 32) + b
    const long long x = __double_as_longlong(a), y = __double_as_longlong(b);
    const auto r0 = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)y, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap((unsigned)(x >> 32), (unsigned)(y >> 32), false, false);
    return __longlong_as_double(((long long)r1[0] << 32) | r0[0]) + __longlong_as_double(((long long)r1[1] << 32) | r0[1]);
}
__device__ __forceinline__ double swap_add16(double a, double b) {   // the same between the odd and even rows of 16 lanes
    const long long x = __double_as_longlong(a), y = __double_as_longlong(b);
    const auto r0 = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)y, false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap((unsigned)(x >> 32), (unsigned)(y >> 32), false, false);
    return __longlong_as_double(((long long)r1[0] << 32) | r0[0]) + __longlong_as_double(((long long)r1[1] << 32) | r0[1]);
}
__device__ __forceinline__ double wave_reduce_scatter32(const double (&acc)[32], int lane) {
    double s1[16], s2[8], s3[4], s4[2];
#pragma unroll
    for (int k = 0; k < 16; ++k) s1[k] = swap_add32(acc[k], acc[k + 16]);
#pragma unroll
    for (int k = 0; k < 8; ++k) s2[k] = swap_add16(s1[k], s1[k + 8]);
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) s3[k] = (b3 ? s2[k + 4] : s2[k]) + dpp_mov_f64<0x140>(b3 ? s2[k] : s2[k + 4]);   // row_mirror
#pragma unroll
    for (int k = 0; k < 2; ++k) s4[k] = (b2 ? s3[k + 2] : s3[k]) + dpp_mov_f64<0x141>(b2 ? s3[k] : s3[k + 2]);   // row_half_mirror
    const double s5 = (b1 ? s4[1] : s4[0]) + dpp_mov_f64<0x1B>(b1 ? s4[0] : s4[1]);                              // quad [3,2,1,0]
    return s5 + dpp_mov_f64<0xB1>(s5);                                                                           // quad [1,0,3,2]
}

// RPT = rows per thread: a slab is SLAB * RPT rows (thread t holds rows t, t + SLAB, ... of it).  1 for a single matrix
// (shortest column step); 2 when a batch would otherwise ask for more workgroups than half the chip holds -- every
// workgroup of a launch has to be resident, and two batches on two streams must be able to be so side by side.
template <int RPT>
__global__ __launch_bounds__(SLAB) void qr_panel3_kernel(int mp, int pb, double* W, int64_t ldw, double* Tjj, int64_t ldt,
                                                         double* Rjj, int64_t ldr, slot_t* part /* [2][G][PB] */,
                                                         slot_t* rowbuf /* [2][PB] */, unsigned long long tag0,
                                                         int64_t sW, int64_t sT, int64_t sR, int64_t sPart, int64_t sRow,
                                                         int split, int64_t gap) {
    {   // blockIdx.y = matrix of a batch of independent factorisations (each with its own hand-off slots)
        const int64_t z = blockIdx.y;
        W += z * sW;
        Tjj += z * sT;
        Rjj += z * sR;
        part += z * sPart;
        rowbuf += z * sRow;
    }
    constexpr int TLD = PB + 1;  // row stride of T in LDS (odd: the eight rows a wave reads lie in different banks)
    __shared__ double q2[2][PB], d2[2][PB], rowl[PB], Tl[PB * TLD];  // q, d by column parity: the T step of column c
                                                                     // overlaps the hand-off of column c + 1
    __shared__ double wsum[2][SLAB / 64][PB];   // per-wave column sums, by column parity
    const int tid = threadIdx.x;
    const int G = gridDim.x;
    const int nap = gridDim.y > 1 ? 8 : 1;   // polling interval: a batch trades a little latency for far fewer coherent loads
    // thread's rows: r[0] < r[1] < ...; only r[0] can be a pivot row (r[i] >= SLAB > PB for i >= 1)
    int r[RPT];
    bool live[RPT];
    int64_t wrow[RPT];
    double pk[RPT][PB], acc[PB];
    const bool vec = pb == PB && (ldw & 1) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;   // (uniform)
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        r[i] = (blockIdx.x * RPT + i) * SLAB + tid;
        live[i] = r[i] < mp;
        // The panel's rows may be two segments of the matrix (two stacked triangles: the pb pivot rows, then the leading
        // rows of the lower block): row r of the panel is row r (r < split) or r + gap of W.
        wrow[i] = (int64_t)r[i] + (r[i] >= split ? gap : 0);
        if (vec) {   // a full panel with 16-byte aligned rows: 16 loads of 16 bytes per row instead of 32 of 8
            if (live[i]) {
                const double2* src = reinterpret_cast<const double2*>(W + wrow[i] * ldw);
#pragma unroll
                for (int k = 0; k < PB; k += 2) {
                    const double2 v = src[k >> 1];
                    pk[i][k] = v.x;
                    pk[i][k + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int k = 0; k < PB; ++k) pk[i][k] = 0.0;
            }
        } else {
#pragma unroll
            for (int k = 0; k < PB; ++k) pk[i][k] = (live[i] && k < pb) ? W[wrow[i] * ldw + k] : 0.0;
        }
    }
    if (blockIdx.x == 0)
        for (int i = tid; i < PB * TLD; i += SLAB) Tl[i] = 0.0;
    if (tid < 2 * PB) {  // entries beyond pb stay zero for the whole kernel (the first publish has the barrier)
        (&q2[0][0])[tid] = 0.0;
        (&d2[0][0])[tid] = 0.0;
    }

    // partial sums of (column c)^T (every column) over this slab's rows below the pivot, published for step c
    auto publish = [&](int c) {
        slot_t* mine = part + ((size_t)(c & 1) * G + blockIdx.x) * PB;
        // column sums over the slab's rows: inside each wave by the register-level reduce-scatter (lane l ends up with
        // column l >> 1), across the four waves through 1 KB of LDS.  (Until round 3 every thread dropped its 32 products
        // into a 64 KB LDS array and 8 lanes per column added 32 of them each: 0.6 us of a column's 2.9.)
        const int lane = tid & 63, wv = tid >> 6;
        const double w = wave_reduce_scatter32(acc, lane);
        if ((lane & 1) == 0) wsum[c & 1][wv][lane >> 1] = w;
        __syncthreads();
        if (tid < PB) {
            const double t = ((wsum[c & 1][0][tid] + wsum[c & 1][1][tid]) + wsum[c & 1][2][tid]) + wsum[c & 1][3][tid];
            if (tid < pb) st_slot(mine + tid, t, tag0 + (unsigned)c);
        }
    };

    {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const double x = (live[i] && r[i] > 0) ? pk[i][0] : 0.0;
#pragma unroll
            for (int k = 0; k < PB; ++k) acc[k] = (i == 0) ? x * pk[i][k] : fma(x, pk[i][k], acc[k]);
        }
        if (r[0] == 0) {
#pragma unroll
            for (int k = 0; k < PB; ++k) st_slot(rowbuf + k, pk[0][k], tag0);
        }
        publish(0);
    }

#ifdef NPW_QR_STAMPS
    long long qr_acc[4] = {0, 0, 0, 0};
    long long tq = wall_clock64();
#endif
    // The column loop is unrolled completely: with the column index a compile-time constant every access to the
    // thread's row (pk[], acc[]) is a plain register access -- with a run-time index each of them is a chain of
    // 32 compare-and-select pairs, and those ~1500 instructions per column were most of the 6.5 us a column took.
    auto column = [&](auto cc) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value;
        if (c >= pb) return;
        const slot_t* pin = part + (size_t)(c & 1) * G * PB;
        const slot_t* rin = rowbuf + (size_t)(c & 1) * PB;
        const unsigned long long tag = tag0 + (unsigned)c;
        double* const q = q2[c & 1];
        double* const d = d2[c & 1];
        {
            // 8 lanes per column gather the slab partials (fixed order: deterministic), then fold
            const int k = tid >> 3, sub = tid & 7;
            double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
            if (tid >= SLAB - PB) {  // the last wave's top lanes fetch the pivot row (overwritten below with d_k)
                const int kk = tid - (SLAB - PB);
                if (kk < pb) {
                    double u0, u1, u2, u3;
                    ld_slots4<1>(rin + kk, rin + kk, rin + kk, rin + kk, true, false, false, false, tag, nap, u0, u1, u2, u3);
                    d[kk] = u0;
                }
            }
            if (k < pb) {
                for (int gb = 0; gb < G; gb += 32) {  // 32 slabs (8192 rows) per round
                    const int g0 = gb + sub, g1 = g0 + 8, g2 = g0 + 16, g3 = g0 + 24;
                    const slot_t* base = pin + k;
                    double u0, u1, u2, u3;
                    const slot_t *s0 = base + (size_t)(g0 < G ? g0 : 0) * PB, *s1 = base + (size_t)(g1 < G ? g1 : 0) * PB,
                                 *s2 = base + (size_t)(g2 < G ? g2 : 0) * PB, *s3 = base + (size_t)(g3 < G ? g3 : 0) * PB;
                    if (G <= 8)         // (uniform over the launch) only the loads that can matter are issued
                        ld_slots4<1>(s0, s1, s2, s3, g0 < G, false, false, false, tag, nap, u0, u1, u2, u3);
                    else if (G <= 16)
                        ld_slots4<2>(s0, s1, s2, s3, g0 < G, g1 < G, false, false, tag, nap, u0, u1, u2, u3);
                    else
                        ld_slots4<4>(s0, s1, s2, s3, g0 < G, g1 < G, g2 < G, g3 < G, tag, nap, u0, u1, u2, u3);
                    t0 += (g0 < G) ? u0 : 0.0;
                    t1 += (g1 < G) ? u1 : 0.0;
                    t2 += (g2 < G) ? u2 : 0.0;
                    t3 += (g3 < G) ? u3 : 0.0;
                }
            }
            double t = ((t0 + t1) + t2) + t3;
            t += __shfl_down(t, 4, 8);
            t += __shfl_down(t, 2, 8);
            t += __shfl_down(t, 1, 8);
            if (sub == 0 && k < pb) q[k] = t;
        }
        __syncthreads();
        QR_STAMP(0, tq)   // hand-off: publish of the previous step -> all partials and the pivot row have arrived
        // Householder scalars: every thread computes them from the two broadcast values (a single thread + a barrier
        // + a broadcast through LDS costs more than the redundant sqrt and divisions)
        double tau = 0.0, scale = 0.0, beta;
        {
            const double alpha = d[c], ss = q[c];
            beta = alpha;
            if (ss != 0.0) {
                const double nrm = sqrt(fma(alpha, alpha, ss));
                beta = (alpha >= 0.0) ? -nrm : nrm;
                tau = (beta - alpha) / beta;
                scale = 1.0 / (alpha - beta);
            }
        }
        QR_STAMP(1, tq)   // scalar part: norm, tau
        // z_k = d_k + scale * q_k (the pivot row entry plus the scaled column sum) is formed where it is used

        // z_k = d_k + scale q_k for the columns to the right, fetched in one batch (q, d are zero beyond pb, and so are
        // the padded columns of the rows: no per-column bounds checks, which would serialise the LDS reads)
        double z[PB];
#pragma unroll
        for (int k = c + 1; k < PB; ++k) z[k] = fma(scale, q[k], d[k]);
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            if (live[i] && r[i] >= c) {
                const double v = (i == 0 && r[0] == c) ? 1.0 : pk[i][c] * scale;
                const double tv = -tau * v;
#pragma unroll
                for (int k = c + 1; k < PB; ++k) pk[i][k] = fma(z[k], tv, pk[i][k]);
                pk[i][c] = v;
            }
        }
        if (c + 1 < PB && live[0] && r[0] == c + 1) {  // the next pivot row (a lane of wave 0 in slab 0) goes out through LDS ...
#pragma unroll
            for (int k = 0; k < PB; ++k) rowl[k] = pk[0][k];
        }
        if (c + 1 < PB) {
            // products of the next column with every column, for the rows below the next pivot (0 elsewhere: rows above
            // hold finished R entries, rows outside the matrix hold zeros)
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const double xn = (live[i] && r[i] > c + 1) ? pk[i][c + 1 < PB ? c + 1 : c] : 0.0;
#pragma unroll
                for (int k = 0; k < PB; ++k) acc[k] = (i == 0) ? xn * pk[i][k] : fma(xn, pk[i][k], acc[k]);
            }
        }
        if (live[0] && r[0] == c) pk[0][c] = beta;  // the diagonal entry of R replaces the implicit 1 of v once the sums are formed
        if (c + 1 < pb && blockIdx.x == 0 && tid < PB) {
            // ... so that 32 lanes of the same wave store one slot each (LDS is in order within a wave: no barrier)
            slot_t* rout = rowbuf + (size_t)((c + 1) & 1) * PB;
            st_slot(rout + tid, rowl[tid], tag + 1);
        }
        QR_STAMP(2, tq)   // row update
        if (c + 1 < pb) publish(c + 1);
        QR_STAMP(3, tq)   // in-slab sums through LDS + slot stores
        if (blockIdx.x == 0) {
            // DLARFT: T[0:c, c] = -tau * T[0:c, 0:c] * z,  T[c][c] = tau -- after the publish, while the partial sums
            // travel.  Eight lanes per row of T.
            const int i = tid >> 3, sub = tid & 7;
            double sacc = 0.0;
            if (i < c)
                for (int j = i + sub; j < c; j += 8) sacc = fma(Tl[i * TLD + j], fma(scale, q[j], d[j]), sacc);
            sacc += __shfl_down(sacc, 4, 8);
            sacc += __shfl_down(sacc, 2, 8);
            sacc += __shfl_down(sacc, 1, 8);
            if (sub == 0) {
                if (i < c)
                    Tl[i * TLD + c] = -tau * sacc;
                else if (i == c)
                    Tl[c * TLD + c] = tau;
            }
        }
    };
    unrolled_columns<0>(column);
#ifdef NPW_QR_STAMPS
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)
        for (int i = 0; i < 4; ++i) qr_stamps[i] += qr_acc[i];
#endif

#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        if (live[i]) {
            if (vec) {
                double2* dst = reinterpret_cast<double2*>(W + wrow[i] * ldw);
#pragma unroll
                for (int k = 0; k < PB; k += 2) {
                    double2 v;
                    v.x = (r[i] > k) ? pk[i][k] : (r[i] == k ? 1.0 : 0.0);
                    v.y = (r[i] > k + 1) ? pk[i][k + 1] : (r[i] == k + 1 ? 1.0 : 0.0);
                    dst[k >> 1] = v;
                }
            }
#pragma unroll
            for (int k = 0; k < PB; ++k) {
                if (k < pb) {
                    if (!vec) W[wrow[i] * ldw + k] = (r[i] > k) ? pk[i][k] : (r[i] == k ? 1.0 : 0.0);
                    if (i == 0 && r[0] < pb) Rjj[(int64_t)r[0] * ldr + k] = (r[0] <= k) ? pk[0][k] : 0.0;
                }
            }
        }
    }
    if (blockIdx.x == 0) {
        __syncthreads();
        for (int i = tid; i < pb * pb; i += SLAB) {
            const int a = i / pb, b = i - a * pb;
            Tjj[(int64_t)a * ldt + b] = Tl[a * TLD + b];
        }
    }
}

// Element strides between the matrices of a batch of independent factorisations (count == 1: unused).
struct Batch {
    int count = 1;
    int64_t sV = 0, sT = 0, sR = 0;
};

// Scratch of `count` factorisations, one array per purpose (matrix z at base + z * stride): the batched GEMMs need a
// constant stride per operand, and their split-K partials want the per-matrix regions back to back.
struct QrWorkspace {
    double* X1;   // OB x n      far update temporaries; split-K scratch of the block Gram matrix
    double* X2;   // OB x n
    double* G;    // n x n       V^T V; split-K scratch of the far updates before that
    double* Tmp;  // (n/2 rounded up + PB) x n
    double* Gb;   // OB x OB     V_b^T V_b of the current outer block
    double* Part;    // 2 x slabs x PB hand-off slots (16 bytes each) of the panel kernel
    double* RowBuf;  // 2 x PB slots: the next pivot row
    double* XnA;     // PB x 4 OB   near update temporaries (2 OB columns; as 16-byte slots in the one-launch form)
    double* XnB;     // PB x 4 OB
    double* XnS;     // 64 x PB x 2 OB  split-K / per-slab partials of the near updates
    double* YnA;     // the same three for the next block's near updates, which run beside the own block's on another stream
    double* YnB;
    double* YnS;
    double* F1;      // SB x n     temporaries of the superblock reflector's far update (third helper stream)
    double* F2;      // SB x n
    double* Gf;      // 4 x SB x n  split-K scratch of that stream
    double* Ts;      // n x SB     the diagonal superblocks of T when the caller does not want T (T == NULL)
    int64_t sX, sG, sTmp, sGb, sPart, sRow, sXn, sXnS, sF, sGf, sTs;
};

inline size_t align2(size_t x) { return (x + 1) & ~(size_t)1; }

// `batch`: the size of the batch the strides are for (small batches run without the superblock level: SB = n)
QrWorkspace carve(void* ws, int64_t m, int64_t n, int count, int batch) {
    QrWorkspace q;
    q.sX = (int64_t)align2((size_t)OB * n);
    q.sG = (int64_t)align2((size_t)n * n);
    q.sTmp = (int64_t)align2((size_t)((n + 1) / 2 + PB) * n);
    q.sGb = (int64_t)OB * OB;
    q.sPart = (int64_t)align2((size_t)4 * ceil_div(m, SLAB) * PB);
    q.sRow = 4 * PB;
    q.sXn = (int64_t)PB * 4 * OB;          // (near_fused_kernel keeps X2 as 16-byte slots: twice the doubles)
    q.sXnS = (int64_t)64 * PB * 2 * OB;    // (... and up to 32 slabs' partials of 2 OB columns as slots)
    q.sF = (int64_t)align2((size_t)superblock_width_max() * n);   // (small batches: SB = n, no far update, unused)
    q.sGf = 4 * q.sF;
    q.sTs = batch >= 8 ? q.sF : (int64_t)align2((size_t)ceil_div(n, OB) * OB * n);   // ld = SB (= n rounded up for small batches)
    double* p = static_cast<double*>(ws);
    auto take = [&](int64_t stride) {
        double* r = p;
        p += (size_t)stride * count;
        return r;
    };
    q.X1 = take(q.sX);
    q.X2 = take(q.sX);
    q.G = take(q.sG);
    q.Tmp = take(q.sTmp);
    q.Gb = take(q.sGb);
    q.Part = take(q.sPart);
    q.RowBuf = take(q.sRow);
    q.XnA = take(q.sXn);
    q.XnB = take(q.sXn);
    q.XnS = take(q.sXnS);
    q.YnA = take(q.sXn);
    q.YnB = take(q.sXn);
    q.YnS = take(q.sXnS);
    q.F1 = take(q.sF);
    q.F2 = take(q.sF);
    q.Gf = take(q.sGf);
    q.Ts = take(q.sTs);
    return q;
}

size_t square_workspace_doubles(int64_t m, int64_t n, int batch) {
    const QrWorkspace q = carve(nullptr, m, n, 0, batch);   // count 0: only the strides are of interest
    return (size_t)(2 * q.sX + q.sG + q.sTmp + q.sGb + q.sPart + q.sRow + 4 * q.sXn + 2 * q.sXnS + 2 * q.sF + q.sGf + q.sTs);
}

inline GemmOpts batched(const Batch& b, int64_t sa, int64_t sb, int64_t sc, int64_t sd) {
    GemmOpts o;
    if (b.count > 1) {
        o.batch = b.count;
        o.batch_a = sa;
        o.batch_b = sb;
        o.batch_c = sc;
        o.batch_d = sd;
    }
    return o;
}

// T[lo:hi, lo:hi] is built from its final `leaf`-wide diagonal blocks by merging halves:
//   T12 = -T1 * G[lo:mid, mid:hi] * T2        (T1, T2 upper triangular, already final)
// G holds V^T V for the columns g0.. (G[0][0] is the entry of column g0 with itself); only its lower triangle is read
// (G12 = G21^T), so the Gram GEMM may skip the tiles above the diagonal.
int merge_t(const Batch& b, int64_t lo, int64_t hi, int64_t leaf, double* T, int64_t ldt, const double* G, int64_t ldg,
            int64_t sG, int64_t g0, double* Tmp, int64_t sTmp, hipStream_t s) {
    const int64_t nleaves = ceil_div(hi - lo, leaf);
    if (nleaves <= 1) return NPW_OK;
    const int64_t mid = lo + (nleaves / 2) * leaf;
    int rc = merge_t(b, lo, mid, leaf, T, ldt, G, ldg, sG, g0, Tmp, sTmp, s);
    if (rc) return rc;
    rc = merge_t(b, mid, hi, leaf, T, ldt, G, ldg, sG, g0, Tmp, sTmp, s);
    if (rc) return rc;
    const int64_t w1 = mid - lo, w2 = hi - mid;
    // Tmp (w1 x w2) = G12 * T2 = G21^T * T2;  T2 is upper triangular: column tile n0 of the product only sums k < n0 + tile
    GemmOpts o1 = batched(b, sG, b.sT, 0, sTmp);
    o1.b_lower_tri = true;
    rc = gemm<double>('T', 'N', w1, w2, w2, 1.0, G + (mid - g0) * ldg + (lo - g0), ldg, T + mid * ldt + mid, ldt, 0.0,
                      nullptr, 0, Tmp, w2, o1, s);
    if (rc) return rc;
    // T12 = -T1 * Tmp;  T1 is upper triangular: row tile m0 only sums k >= m0   (together: half the flops of the merge)
    GemmOpts o2 = batched(b, b.sT, sTmp, 0, b.sT);
    o2.a_upper_tri = true;
    return gemm<double>('N', 'N', w1, w2, w1, -1.0, T + lo * ldt + lo, ldt, Tmp, w2, 0.0, nullptr, 0,
                        T + lo * ldt + mid, ldt, o2, s);
}

// The top `rows` rows of the updated columns are final rows of R: move them to R and clear them in the working matrix
// (they lie above the diagonal of V).  blockIdx.z = matrix of the batch.
__global__ void move_rows_kernel(int rows, int64_t cols, double* W, int64_t ldw, int64_t sW, double* Rd, int64_t ldr,
                                 int64_t sR) {
    W += (int64_t)blockIdx.z * sW;
    Rd += (int64_t)blockIdx.z * sR;
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
        Rd[(int64_t)r * ldr + c] = W[(int64_t)r * ldw + c];
        W[(int64_t)r * ldw + c] = 0.0;
    }
}

// X2 = T^T (sum of the split-k partial products of V^T W  [+ C]) for pb <= PBMAX rows: the reduction of the partials and
// the small triangular product in ONE launch (they used to be two on the panel chain's critical path).
// One thread per column of X2; T is pb x pb upper triangular: X2[i] = sum_{l <= i} T[l][i] X1[l].
// Rd != NULL (the stacked-triangle update, C = the columns' pb top rows): the rows' final R entries C - X2 go to Rd and C is
// cleared here as well (each element of C is read and written by one thread only) -- one launch less per panel.
template <int PBMAX>
__global__ __launch_bounds__(PBMAX * 16) void reduce_tt_kernel(int pb, int64_t nc, int nsplit, const double* P, double* C,
                                                               int64_t ldc, int64_t sC, const double* Tjj, int64_t ldt,
                                                               int64_t sT, double* X2, int64_t sX2, double* Rd = nullptr,
                                                               int64_t ldr = 0, int64_t sR = 0) {
    // block = 16 columns x PBMAX rows: thread (l, j) first sums the partials of X1[l][j] (fixed order), then -- through
    // LDS -- forms X2[l][j] = sum_{i <= l} T[i][l] X1[i][j]
    __shared__ double Ts[PBMAX * (PBMAX + 1)], X1s[PBMAX * 17];
    const int z = blockIdx.y;
    P += (int64_t)z * nsplit * pb * nc;
    Tjj += (int64_t)z * sT;
    X2 += (int64_t)z * sX2;
    if (C) C += (int64_t)z * sC;
    const int j = threadIdx.x & 15, l = threadIdx.x >> 4;
    const int64_t col = (int64_t)blockIdx.x * 16 + j;
    for (int i = threadIdx.x; i < pb * pb; i += blockDim.x) Ts[(i / pb) * (PBMAX + 1) + (i % pb)] = Tjj[(int64_t)(i / pb) * ldt + (i % pb)];
    double a = 0.0, ctop = 0.0;
    if (l < pb && col < nc) {
        const double* p = P + (int64_t)l * nc + col;
        for (int sidx = 0; sidx < nsplit; ++sidx) a += p[(int64_t)sidx * pb * nc];
        if (C) {
            ctop = C[(int64_t)l * ldc + col];
            a += ctop;
        }
    }
    X1s[l * 17 + j] = a;
    __syncthreads();
    if (l < pb && col < nc) {
        double x = 0.0;
        for (int i = 0; i <= l; ++i) x = fma(Ts[i * (PBMAX + 1) + l], X1s[i * 17 + j], x);
        X2[(int64_t)l * nc + col] = x;
        if (Rd) {
            Rd[(int64_t)z * sR + (int64_t)l * ldr + col] = ctop - x;
            C[(int64_t)l * ldc + col] = 0.0;
        }
    }
}

// The near updates of a whole OB-wide block leave their R rows in the working matrix (one launch less per panel on the
// critical path); this moves them in one go: row r of the block (in the panel that ends at column pe(r)) holds final R
// entries in the columns [pe(r), c_end).  Wb, Rb point at (row b0, column b0).
__global__ void move_block_rows_kernel(int ob, int pbw, int64_t c_end, double* Wb, int64_t ldw, int64_t sW, double* Rb,
                                       int64_t ldr, int64_t sR) {
    Wb += (int64_t)blockIdx.z * sW;
    Rb += (int64_t)blockIdx.z * sR;
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // column relative to b0
    if (c >= c_end) return;
    for (int r = blockIdx.y; r < ob; r += gridDim.y) {
        const int pe = (r / pbw + 1) * pbw;  // first column right of this row's panel
        if (c >= pe) {
            Rb[(int64_t)r * ldr + c] = Wb[(int64_t)r * ldw + c];
            Wb[(int64_t)r * ldw + c] = 0.0;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Panel-wide (pb <= 32) reflector applied to a narrow column block: the two streaming kernels around reduce_tt_kernel.
// The generic tilings are a poor fit for these shapes (m = 32 rows of V^T W on 64 x 64 tiles with 256-byte row
// segments, k = 32 updates that are all prologue and epilogue: 3.5 TFLOP/s and 0.5 TB/s in the round-4 trace of a
// batch of 32, 48 ms of kernel time per batch for the next block's columns alone).  Both kernels read every element of
// W2 once, straight from global memory into the MFMA operand layout (16 consecutive lanes = 16 consecutive columns =
// one 128-byte segment), no LDS staging: they are HBM-bound by construction.
//   near_vtw_kernel   : P[z][slab] = V[slab rows]^T W2[slab rows]   (pb x nc partial products, one per row slab)
//   (reduce_tt_kernel : X2 = T^T (sum of the partials [+ Wtop]))
//   near_update_kernel: W2[slab rows] -= V[slab rows] X2
// blockIdx.x = row slab of `rs` rows, blockIdx.y = chunk of 64 columns, blockIdx.z = matrix of the batch.
// ------------------------------------------------------------------------------------------------------------------
typedef double nd4_t __attribute__((ext_vector_type(4)));
constexpr int NEAR_NT = 4;             // 16-column MFMA tiles per workgroup (one 64-column chunk)

// The slab's partial product V[rows of this wave]^T W2[same rows] for one 64-column chunk, summed over the four waves of
// the workgroup (fixed order: (w0 + w2) + (w1 + w3)); the result is left in wave 0's `acc` (rows 16 i + lg + 4 r of the
// pb x 64 block, column 16 j + li).  red: 2 x 2048 doubles of LDS.
// KEEP: the slab is one round of 16 rows per wave (rw == 16) and `keep` receives the wave's W2 values, which are then already
// in the layout of the update's C operand (keep[r][j] = W2[r_begin + lg + 4 r][c0 + 16 j + li]).
template <bool EDGE, bool KEEP = false>
__device__ __forceinline__ void near_vtw_body(int rows, int pb, int nc, int c0, int r_begin, int rw, const double* V, int64_t ldv,
                                              const double* W, int64_t ldw, double* red, nd4_t (&acc)[2][NEAR_NT],
                                              double (*keep)[NEAR_NT] = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NEAR_NT; ++j) acc[i][j] = nd4_t{0.0, 0.0, 0.0, 0.0};
    const double* vrow = V + (int64_t)(r_begin + lg) * ldv + li;
    const double* wrow = W + (int64_t)(r_begin + lg) * ldw + c0 + li;
    const int ntv = (nc - c0) / 16;              // fast form: 16-column tiles of this chunk inside the matrix
    // four MFMA steps (16 rows) per round: their 24 loads are in flight together
    for (int s0 = 0; s0 < rw; s0 += 16) {
        if (!EDGE && r_begin + s0 >= rows) break;   // (wave-uniform) the fast form only needs whole groups of 16 rows
        double a0[4], a1[4], bb[4][NEAR_NT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double* vr = vrow + (int64_t)(4 * u) * ldv;
            const double* wr = wrow + (int64_t)(4 * u) * ldw;
            if constexpr (!EDGE) {
                a0[u] = vr[0];
                a1[u] = vr[16];
#pragma unroll
                for (int j = 0; j < NEAR_NT; ++j) bb[u][j] = (j < ntv) ? wr[16 * j] : 0.0;   // (uniform) whole 16-column tiles
            } else {
                const bool rok = r_begin + s0 + 4 * u + lg < rows;
                a0[u] = (rok && li < pb) ? vr[0] : 0.0;
                a1[u] = (rok && 16 + li < pb) ? vr[16] : 0.0;
#pragma unroll
                for (int j = 0; j < NEAR_NT; ++j) bb[u][j] = (rok && c0 + 16 * j + li < nc) ? wr[16 * j] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < NEAR_NT; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], bb[u][j], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], bb[u][j], acc[1][j], 0, 0, 0);
                if constexpr (KEEP) keep[u][j] = bb[u][j];
            }
        vrow += 16 * ldv;
        wrow += 16 * ldw;
    }
    // (w0 + w2) + (w1 + w3): fixed order, deterministic
    auto put = [&](int slot) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NEAR_NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[slot * 2048 + ((i * NEAR_NT + j) * 4 + r) * 64 + lane] = acc[i][j][r];
    };
    auto add = [&](int slot) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NEAR_NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += red[slot * 2048 + ((i * NEAR_NT + j) * 4 + r) * 64 + lane];
    };
    if (wave >= 2) put(wave - 2);
    __syncthreads();
    if (wave < 2) add(wave);
    __syncthreads();
    if (wave == 1) put(0);
    __syncthreads();
    if (wave == 0) add(0);
}

// W2[rows of this wave] -= V[same rows] X2 for one 64-column chunk; bf = X2 in the B-operand layout: lane group g takes
// k = 8 g .. 8 g + 7 at the eight MFMA steps (any bijection does as long as both operands agree), so that a lane's eight
// values of V are 64 contiguous bytes of one row.
template <bool EDGE>
__device__ __forceinline__ void near_update_body(int rows, int pb, int nc, int c0, int r_begin, int rw, const double* V, int64_t ldv,
                                                 const double (&bf)[8][NEAR_NT], double* W, int64_t ldw) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    for (int t = 0; t < rw; t += 16) {
        if (!EDGE && r_begin + t >= rows) break;  // (wave-uniform) whole groups of 16 rows
        const int ra = r_begin + t + li;          // the row this lane feeds into the A operand
        double af[8];
        if constexpr (!EDGE) {
            const double2* src = reinterpret_cast<const double2*>(V + (int64_t)ra * ldv + 8 * lg);
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const double2 v = src[h];
                af[2 * h] = v.x;
                af[2 * h + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) af[s] = (ra < rows && 8 * lg + s < pb) ? V[(int64_t)ra * ldv + 8 * lg + s] : 0.0;
        }
        nd4_t acc[NEAR_NT];
        double cv[NEAR_NT][4];
#pragma unroll
        for (int j = 0; j < NEAR_NT; ++j) {
            acc[j] = nd4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r_begin + t + lg + 4 * r, col = c0 + 16 * j + li;
                cv[j][r] = (EDGE ? (row < rows && col < nc) : (16 * j < nc - c0)) ? W[(int64_t)row * ldw + col] : 0.0;
            }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int j = 0; j < NEAR_NT; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s], bf[s][j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NEAR_NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r_begin + t + lg + 4 * r, col = c0 + 16 * j + li;
                if (EDGE ? (row < rows && col < nc) : (16 * j < nc - c0)) W[(int64_t)row * ldw + col] = cv[j][r] - acc[j][r];
            }
    }
}

template <bool EDGE>
__global__ __launch_bounds__(256) void near_vtw_kernel(int rows, int pb, int nc, const double* V, int64_t ldv, int64_t sV,
                                                        const double* W, int64_t ldw, int64_t sW, double* P, int rs) {
    __shared__ double red[2 * 2048];   // two waves' accumulators (32 KB)
    const int z = blockIdx.z, slab = blockIdx.x, nslab = gridDim.x;
    const int c0 = blockIdx.y * 16 * NEAR_NT;
    V += (int64_t)z * sV;
    W += (int64_t)z * sW;
    P += ((int64_t)z * nslab + slab) * pb * nc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int rw = rs >> 2;                      // rows per wave (a multiple of 16)
    nd4_t acc[2][NEAR_NT];
    near_vtw_body<EDGE>(rows, pb, nc, c0, slab * rs + wave * rw, rw, V, ldv, W, ldw, red, acc);
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NEAR_NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mm = 16 * i + lg + 4 * r, col = c0 + 16 * j + li;
                    if (mm < pb && col < nc) P[(int64_t)mm * nc + col] = acc[i][j][r];
                }
    }
}

template <bool EDGE>
__global__ __launch_bounds__(256) void near_update_kernel(int rows, int pb, int nc, const double* V, int64_t ldv, int64_t sV,
                                                           const double* X2, int64_t ldx, int64_t sX, double* W, int64_t ldw,
                                                           int64_t sW, int rs) {
    const int z = blockIdx.z, slab = blockIdx.x;
    const int c0 = blockIdx.y * 16 * NEAR_NT;
    V += (int64_t)z * sV;
    W += (int64_t)z * sW;
    X2 += (int64_t)z * sX;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int rw = rs >> 2;                      // rows per wave (a multiple of 16)
    double bf[8][NEAR_NT];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int j = 0; j < NEAR_NT; ++j) {
            const int k = 8 * lg + s, col = c0 + 16 * j + li;
            bf[s][j] = (k < pb && col < nc) ? X2[(int64_t)k * ldx + col] : 0.0;
        }
    near_update_body<EDGE>(rows, pb, nc, c0, slab * rs + wave * rw, rw, V, ldv, bf, W, ldw);
}

// The three steps in ONE launch for a latency-bound (single or small-batch) factorisation, where the chain waits for
// each of them: per-slab partial products -> hand-off -> reduce / T^T -> hand-off -> update.  Every workgroup of the
// launch waits for the others (tagged slots, bounded spins, as in the panel kernel): the whole grid has to be resident,
// so the host only takes this form when the grid is at most one workgroup per compute unit of the stream.
//   part: [z][chunk][slab][32 x 64] slots -- the slabs' partial products of the chunk
//   xs  : [z][chunk][32 x 64] slots       -- X2 = T^T (sum of the partials [+ Wtop])
// The 64 columns of a chunk are dealt out to the slabs' workgroups for the reduction (cpw columns each).  Wtop != NULL:
// the stacked-triangle form (the reflectors' top part is the identity): X1 also takes the pb top rows of the columns, and
// the workgroup that reduces a column leaves its final R entries Wtop - X2 in Rdst and clears Wtop.
template <bool EDGE>
__global__ __launch_bounds__(256) void near_fused_kernel(int rows, int pb, int nc, const double* V, int64_t ldv, int64_t sV, double* W,
                                                          int64_t ldw, int64_t sW, double* Wtop, double* Rdst, int64_t ldr,
                                                          int64_t sR, const double* Tjj, int64_t ldt, int64_t sT, slot_t* part,
                                                          slot_t* xs, unsigned long long tag, int rs) {
    constexpr int CH = 16 * NEAR_NT;                 // columns of a chunk
    __shared__ double lds[32 * 33 + 2 * 32 * (CH + 1)];   // phase 1: two waves' accumulators (2 x 2048); then T, X1, X2
    static_assert(32 * 33 + 2 * 32 * (CH + 1) >= 2 * 2048, "the accumulators of phase 1 need 2 x 2048 doubles");
    double* const Ts = lds;                          // [32][33]
    double* const X1s = lds + 32 * 33;               // [32][CH + 1]
    double* const X2s = X1s + 32 * (CH + 1);         // [32][CH + 1]
    const int z = blockIdx.z, slab = blockIdx.x, nslab = gridDim.x, chunk = blockIdx.y, nchunk = gridDim.y;
    const int c0 = chunk * CH;
    V += (int64_t)z * sV;
    W += (int64_t)z * sW;
    Tjj += (int64_t)z * sT;
    if (Wtop) {
        Wtop += (int64_t)z * sW;
        Rdst += (int64_t)z * sR;
    }
    part += ((size_t)z * nchunk + chunk) * nslab * (32 * CH);
    xs += ((size_t)z * nchunk + chunk) * (32 * CH);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int rw = rs >> 2;
    const int r_begin = slab * rs + wave * rw;
    const bool one = rw == 16;                       // (uniform) one round per wave: W2 and V stay in registers for the update
    // this workgroup's columns of the chunk (the reduction's work is dealt out to the slabs)
    const int cpw = (CH + nslab - 1) / nslab;
    const int cb = slab * cpw, ce = (cb + cpw < CH) ? cb + cpw : CH;   // [cb, ce): empty for the last slabs when nslab > CH / cpw
    // T (needed after the first hand-off) is fetched now, behind the loads of phase 1
    double tpre[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int i = tid + 256 * h, a = i >> 5, bcol = i & 31;
        tpre[h] = (cb < ce && a < pb && bcol < pb) ? Tjj[(int64_t)a * ldt + bcol] : 0.0;
    }
    double keep[4][NEAR_NT], af[8];
    {
        nd4_t acc[2][NEAR_NT];
        if (one)
            near_vtw_body<EDGE, true>(rows, pb, nc, c0, r_begin, rw, V, ldv, W, ldw, lds, acc, keep);
        else
            near_vtw_body<EDGE>(rows, pb, nc, c0, r_begin, rw, V, ldv, W, ldw, lds, acc);
        if (wave == 0) {
            slot_t* mine = part + (size_t)slab * (32 * CH);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NEAR_NT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) st_slot(mine + (16 * i + lg + 4 * r) * CH + 16 * j + li, acc[i][j][r], tag);
        }
    }
    if (one) {
        // the update's A operand (lane group g: k = 8 g .. 8 g + 7 of row r_begin + li), requested before the waits
        const int ra = r_begin + li;
        if constexpr (!EDGE) {
            if (r_begin < rows) {
                const double2* src = reinterpret_cast<const double2*>(V + (int64_t)ra * ldv + 8 * lg);
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const double2 v = src[h];
                    af[2 * h] = v.x;
                    af[2 * h + 1] = v.y;
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) af[s] = (ra < rows && 8 * lg + s < pb) ? V[(int64_t)ra * ldv + 8 * lg + s] : 0.0;
        }
    }
    __syncthreads();   // (the accumulators in LDS are dead: T, X1, X2 take their place)
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int i = tid + 256 * h;
        Ts[(i >> 5) * 33 + (i & 31)] = tpre[h];
    }
    // ---- this workgroup's columns: sum of the slabs' partials (fixed order) [+ Wtop], T^T, publish ----
    // 32 rows x (ce - cb) columns x nslab slabs are about 2048 slots whatever the slab count: 8 per thread, two or three
    // batches of four loads in flight.  (One lane per element adding up all the slabs in turn was 16 dependent round
    // trips to the memory side for 64 slabs: 12 of the launch's 22 us.)
    // (Both waits first poll ONE slot per producer from one wave, with pauses, and only then read everything: thousands of
    //  lanes polling the same few KB with device-coherent loads queue up in front of the very stores they are waiting for.)
    if (cb < ce && tid < nslab) {
        double u0, u1, u2, u3;
        const slot_t* rep = part + (size_t)tid * (32 * CH) + cb;
        ld_slots4<1>(rep, rep, rep, rep, true, false, false, false, tag, 8, u0, u1, u2, u3);
    }
    __syncthreads();
    {
        const int l = tid >> 3, sub = tid & 7;   // row of X1, 8 lanes per row
        if (nslab >= 8) {
            // at most 8 columns: lane `sub` adds the slabs sub, sub + 8, ... of every column, the 8 lanes' sums are then added
            // in lane order (through LDS: the X2 area is still free)
            double* const lane_sums = X2s;       // [32][8 columns][8 lanes]
            const int ngi = (nslab + 7) >> 3, npair = (ce - cb) * ngi;
            double a = 0.0;
            int cur = 0;
            for (int q0 = 0; q0 < npair; q0 += 4) {
                const slot_t* ptr[4];
                bool need[4];
                int ci[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int q = q0 + e < npair ? q0 + e : npair - 1;
                    ci[e] = q / ngi;
                    const int g = sub + 8 * (q - ci[e] * ngi);
                    need[e] = q0 + e < npair && g < nslab;
                    ptr[e] = part + (size_t)(need[e] ? g : 0) * (32 * CH) + l * CH + cb + ci[e];
                }
                double u[4];
                ld_slots4<4>(ptr[0], ptr[1], ptr[2], ptr[3], need[0], need[1], need[2], need[3], tag, 1, u[0], u[1], u[2], u[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (q0 + e < npair) {           // (uniform)
                        if (ci[e] != cur) {         // (uniform) the column changes: the lane's sum of the previous one is complete
                            lane_sums[(l * 8 + cur) * 8 + sub] = a;
                            a = 0.0;
                            cur = ci[e];
                        }
                        if (need[e]) a += u[e];
                    }
                }
            }
            if (npair > 0) lane_sums[(l * 8 + cur) * 8 + sub] = a;
            __syncthreads();
            const int c = cb + sub;
            if (c < ce) {
                const double* ls = lane_sums + (l * 8 + sub) * 8;
                double t = ((((((ls[0] + ls[1]) + ls[2]) + ls[3]) + ls[4]) + ls[5]) + ls[6]) + ls[7];
                if (Wtop && l < pb && c0 + c < nc) t += Wtop[(int64_t)l * ldw + c0 + c];
                X1s[l * (CH + 1) + c] = t;
            }
        } else {
            // fewer than 8 slabs (more than 8 columns): a lane per column, its slabs in one or two batches
            for (int c = cb + sub; c < ce; c += 8) {
                const slot_t* base = part + l * CH + c;
                double a = 0.0;
                for (int g = 0; g < nslab; g += 4) {
                    double u0, u1, u2, u3;
                    const bool n1 = g + 1 < nslab, n2 = g + 2 < nslab, n3 = g + 3 < nslab;
                    ld_slots4<4>(base + (size_t)g * (32 * CH), base + (size_t)(n1 ? g + 1 : g) * (32 * CH),
                                 base + (size_t)(n2 ? g + 2 : g) * (32 * CH), base + (size_t)(n3 ? g + 3 : g) * (32 * CH), true, n1, n2,
                                 n3, tag, 1, u0, u1, u2, u3);
                    a += u0;
                    if (n1) a += u1;
                    if (n2) a += u2;
                    if (n3) a += u3;
                }
                if (Wtop && l < pb && c0 + c < nc) a += Wtop[(int64_t)l * ldw + c0 + c];
                X1s[l * (CH + 1) + c] = a;
            }
        }
    }
    __syncthreads();
    {
        const int l = tid >> 3, sub = tid & 7;
        for (int c = cb + sub; c < ce; c += 8) {
            double x = 0.0;
            for (int i = 0; i <= l; ++i) x = fma(Ts[i * 33 + l], X1s[i * (CH + 1) + c], x);
            st_slot(xs + l * CH + c, x, tag);
            if (Wtop && l < pb && c0 + c < nc) {
                double* wt = Wtop + (int64_t)l * ldw + c0 + c;
                Rdst[(int64_t)l * ldr + c0 + c] = *wt - x;
                *wt = 0.0;
            }
        }
    }
    // ---- all of X2 for the chunk ----
    if (tid < CH) {
        double u0, u1, u2, u3;
        ld_slots4<1>(xs + tid, xs + tid, xs + tid, xs + tid, true, false, false, false, tag, 8, u0, u1, u2, u3);
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int e0 = tid + 1024 * h;
        double u0, u1, u2, u3;
        ld_slots4<4>(xs + e0, xs + e0 + 256, xs + e0 + 512, xs + e0 + 768, true, true, true, true, tag, 1, u0, u1, u2, u3);
        X2s[(e0 >> 6) * (CH + 1) + (e0 & 63)] = u0;
        X2s[((e0 + 256) >> 6) * (CH + 1) + (e0 & 63)] = u1;
        X2s[((e0 + 512) >> 6) * (CH + 1) + (e0 & 63)] = u2;
        X2s[((e0 + 768) >> 6) * (CH + 1) + (e0 & 63)] = u3;
    }
    __syncthreads();
    double bf[8][NEAR_NT];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int j = 0; j < NEAR_NT; ++j) bf[s][j] = X2s[(8 * lg + s) * (CH + 1) + 16 * j + li];   // (zero beyond pb / nc: T and the partials are)
    if (!one) {
        near_update_body<EDGE>(rows, pb, nc, c0, r_begin, rw, V, ldv, bf, W, ldw);
        return;
    }
    if (!EDGE && r_begin >= rows) return;   // (wave-uniform)
    nd4_t acc[NEAR_NT];
#pragma unroll
    for (int j = 0; j < NEAR_NT; ++j) acc[j] = nd4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int j = 0; j < NEAR_NT; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s], bf[s][j], acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NEAR_NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = r_begin + lg + 4 * r, col = c0 + 16 * j + li;
            if (EDGE ? (row < rows && col < nc) : (16 * j < nc - c0)) W[(int64_t)row * ldw + col] = keep[r][j] - acc[j][r];
        }
}

// rows per workgroup of the two kernels above: 256 when that already gives the chip a few hundred workgroups, else 64;
// never more slabs than the partial-product scratch holds
inline int near_slab_rows(const Batch& b, int64_t rows, int64_t pb, int64_t nc, size_t skcap) {
    const int64_t chunks = ceil_div(nc, 16 * NEAR_NT);
    int rs = (ceil_div(rows, 256) * chunks * b.count >= 256) ? 256 : 64;
    // (no upper limit on rs: the callers have checked pb * nc <= skcap, so the loop ends at one slab at the latest -- a cap of
    //  4096 rows overran the scratch for single tall matrices on parts with more than 256 CUs)
    while ((size_t)(ceil_div(rows, rs) * pb * nc) > skcap) rs *= 2;
    return rs;
}
// the fast forms need whole 16-column tiles, a full panel, whole groups of 16 rows and 16-byte aligned rows of V
inline bool near_fast(int64_t rows, int64_t pb, int64_t nc, int rs, const double* Vp, int64_t ldv) {
    (void)rs;
    return pb == PB && nc % 16 == 0 && rows % 16 == 0 && ldv % 2 == 0 && (reinterpret_cast<uintptr_t>(Vp) & 15) == 0;
}
static const bool far_wide_tiles = [] {
    const char* e = getenv("NPW_QR_FAR_WIDE");
    return e == nullptr || atoi(e) != 0;
}();
static const bool far_nt_form = [] {
    const char* e = getenv("NPW_QR_FAR_NT");
    return e == nullptr || atoi(e) != 0;
}();
// (128 x 128 tiles for T's block columns: measured slower -- x16 61.3 vs 58.3 ms, x32 105.2 vs 103.6, gpurun_out/r04p: these
//  products run beside the far updates on the same stream set and the smaller workgroups fill the gaps better)
static const bool t_big_tiles = [] {
    const char* e = getenv("NPW_QR_T_BIG");
    return e != nullptr && atoi(e) != 0;
}();
static const bool near_kernels_on = [] {
    const char* e = getenv("NPW_QR_NEAR_KERNELS");
    return e == nullptr || atoi(e) != 0;
}();
static const bool near_fused_on = [] {
    const char* e = getenv("NPW_QR_NEAR_FUSED");
    return e == nullptr || atoi(e) != 0;
}();
// Rows per slab for near_fused_kernel, or 0 when the three-launch form has to run: the one launch is for latency-bound
// factorisations (fewer than 8 matrices), its whole grid has to be resident (at most one workgroup per compute unit of the
// stream), a chunk's 64 columns are dealt out to at most 64 slabs, and the slots have to fit the scratch (skcap doubles
// per matrix for the partials, sX2 for X2; a slot is two doubles).
inline int fused_slab_rows(const Batch& b, int64_t rows, int64_t nc, size_t skcap, int64_t sX2, unsigned long long tag, hipStream_t s) {
    if (!near_fused_on || tag == 0 || b.count >= 8) return 0;
    const int64_t chunks = ceil_div(nc, 16 * NEAR_NT);
    if (chunks * 2 * 32 * 16 * NEAR_NT > sX2) return 0;
    const int64_t cus = resident_cu_count(s);
    // (the smallest slab the limits allow: 4096 rows x 224 columns alone 17.4 / 17.9 / 18.3 ms from 64 / 128 / 256 rows up)
    for (int rs = 64; rs <= 1024; rs *= 2) {
        const int64_t nslab = ceil_div(rows, rs);
        if (nslab <= 16 * NEAR_NT && nslab * chunks * 2 * 32 * 16 * NEAR_NT <= (int64_t)skcap && nslab * chunks * b.count <= cus) return rs;
    }
    return 0;
}

// Split-K factor for X1 = V^T W2 (pb x nc, contraction over `rows`): enough workgroups for the chip.  Panel- and
// block-wide reflectors run on 64 x 64 tiles (one tile row counted, as tuned in rounds 2 - 3), superblock-wide ones on
// 128 x 128 tiles.
inline int64_t wanted_splits(const Batch& b, int64_t pb, int64_t nc, int64_t rows) {
    const int64_t wgs = (pb >= 256 ? ceil_div(pb, 128) * ceil_div(nc, 128) : ceil_div(nc, 64)) * b.count;
    int64_t want = 512 / (wgs > 0 ? wgs : 1);
    const int64_t per = pb <= PB ? 64 : 256;   // panel-wide: latency matters, cut finer
    if (want > rows / per) want = rows / per;
    if (want > 32) want = 32;
    return want;
}
// gemm() picks its tiling per problem; a batch of mid-size products fills the chip with 128 x 128 tiles much earlier
inline bool whole_chip_of_big_tiles(const Batch& b, int64_t m, int64_t n) {
    return m >= 256 && n >= 256 && ceil_div(m, 128) * ceil_div(n, 128) * b.count >= 256;
}

// W2 (mp x nc, ld ldv) -= V_p (T_p^T (V_p^T W2)), then its top pb rows (final rows of R) move to Rdst and are
// zeroed in place (V is zero there).  X1, X2: pb x nc scratch per matrix (strides sX1, sX2); skws: split-K scratch of
// skcap elements per matrix, the matrices' regions back to back.
int apply_panel(const Batch& b, const double* Wp, int64_t ldv, int64_t mp, int64_t pb, const double* Tjj, int64_t ldt,
                double* W2, int64_t nc, double* X1, int64_t sX1, double* X2, int64_t sX2, double* skws, size_t skcap,
                double* Rdst, int64_t ldr, hipStream_t s, unsigned long long tag = 0) {  // Rdst == nullptr: the caller moves the R rows later
    // tag != 0: a sequence tag for the hand-off slots of near_fused_kernel, unique to this call of this factorisation
    // X1 = V_p^T W2 is pb x nc with a contraction over all mp rows: split k so that the launch has a few hundred
    // workgroups instead of nc/64
    GemmOpts g1 = batched(b, b.sV, b.sV, 0, sX1);
    int64_t want = wanted_splits(b, pb, nc, mp);
    if (skws != nullptr && want > 1 && (size_t)want * pb * nc <= skcap) {
        g1.splitk = (int)want;
        g1.splitk_ws = skws;
    }
    const bool big = pb >= 256 && whole_chip_of_big_tiles(b, pb, nc);   // a superblock reflector: the batch fills the chip with 128 x 128 tiles
    g1.force_big = big;
    int rc;
    bool updated = false;
    if (near_kernels_on && pb <= PB && skws != nullptr && (size_t)pb * nc <= skcap) {
        // panel-wide reflector: the streaming kernels (per-slab partial products, one small kernel that reduces them in a
        // fixed order and applies T^T, the rank-pb update) -- as one launch where the chain waits for them
        if (const int frs = fused_slab_rows(b, mp, nc, skcap, sX2, tag, s)) {
            const dim3 grid((unsigned)ceil_div(mp, frs), (unsigned)ceil_div(nc, 16 * NEAR_NT), (unsigned)b.count);
            const bool fast = near_fast(mp, pb, nc, frs, Wp, ldv);
            hipLaunchKernelGGL(fast ? near_fused_kernel<false> : near_fused_kernel<true>, grid, dim3(256), 0, s, (int)mp, (int)pb, (int)nc,
                               Wp, ldv, b.sV, W2, ldv, b.sV, (double*)nullptr, (double*)nullptr, (int64_t)0, (int64_t)0, Tjj, ldt, b.sT,
                               reinterpret_cast<slot_t*>(skws), reinterpret_cast<slot_t*>(X2), tag, frs);
            NPW_LAUNCH_CHECK();
            if (Rdst == nullptr) return NPW_OK;
            const unsigned gy = (unsigned)(pb < 32 ? pb : 32);
            hipLaunchKernelGGL(move_rows_kernel, dim3((unsigned)ceil_div(nc, 256), gy, (unsigned)b.count), dim3(256), 0, s, (int)pb, nc,
                               W2, ldv, b.sV, Rdst, ldr, b.sR);
            NPW_LAUNCH_CHECK();
            return NPW_OK;
        }
        const int rs = near_slab_rows(b, mp, pb, nc, skcap);
        const dim3 grid((unsigned)ceil_div(mp, rs), (unsigned)ceil_div(nc, 16 * NEAR_NT), (unsigned)b.count);
        const bool fast = near_fast(mp, pb, nc, rs, Wp, ldv);
        hipLaunchKernelGGL(fast ? near_vtw_kernel<false> : near_vtw_kernel<true>, grid, dim3(256), 0, s, (int)mp, (int)pb, (int)nc, Wp,
                           ldv, b.sV, (const double*)W2, ldv, b.sV, skws, rs);
        NPW_LAUNCH_CHECK();
        hipLaunchKernelGGL(reduce_tt_kernel<PB>, dim3((unsigned)ceil_div(nc, 16), (unsigned)b.count), dim3(PB * 16), 0, s, (int)pb, nc,
                           (int)grid.x, skws, (double*)nullptr, (int64_t)0, (int64_t)0, Tjj, ldt, b.sT, X2, sX2);
        NPW_LAUNCH_CHECK();
        hipLaunchKernelGGL(fast ? near_update_kernel<false> : near_update_kernel<true>, grid, dim3(256), 0, s, (int)mp, (int)pb, (int)nc,
                           Wp, ldv, b.sV, (const double*)X2, nc, sX2, W2, ldv, b.sV, rs);
        NPW_LAUNCH_CHECK();
        if (Rdst == nullptr) return NPW_OK;
        const unsigned gy = (unsigned)(pb < 32 ? pb : 32);
        hipLaunchKernelGGL(move_rows_kernel, dim3((unsigned)ceil_div(nc, 256), gy, (unsigned)b.count), dim3(256), 0, s, (int)pb, nc,
                           W2, ldv, b.sV, Rdst, ldr, b.sR);
        NPW_LAUNCH_CHECK();
        return NPW_OK;
    }
    if (pb <= PB && skws != nullptr && (size_t)pb * nc <= skcap) {
        // (the same through the generic GEMM tilings: $NPW_QR_NEAR_KERNELS=0, A/B runs)  the partial products stay in the
        // scratch and ONE small kernel reduces them and applies T^T
        int nsplit = 1;
        g1.splitk_ws = skws;
        g1.splitk_keep = &nsplit;
        rc = gemm<double>('T', 'N', pb, nc, mp, 1.0, Wp, ldv, W2, ldv, 0.0, nullptr, 0, X1, nc, g1, s);
        if (rc) return rc;
        hipLaunchKernelGGL(reduce_tt_kernel<PB>, dim3((unsigned)ceil_div(nc, 16), (unsigned)b.count), dim3(PB * 16), 0, s, (int)pb, nc,
                           nsplit, skws, (double*)nullptr, (int64_t)0, (int64_t)0, Tjj, ldt, b.sT, X2, sX2);
        NPW_LAUNCH_CHECK();
    } else if (pb >= 256 && far_nt_form) {
        // superblock reflector: the temporaries are kept TRANSPOSED (X^T = W2^T V, nc x pb), so that the rank-pb update reads
        // both operands along k -- the N / T form, the only one with the pinned load / store interleave of gemm.hip
        g1.wide_n = far_wide_tiles;
        rc = gemm<double>('T', 'N', nc, pb, mp, 1.0, W2, ldv, Wp, ldv, 0.0, nullptr, 0, X1, pb, g1, s);
        if (rc) return rc;
        GemmOpts g2 = batched(b, sX1, b.sT, 0, sX2);
        g2.force_big = big;
        g2.b_lower_tri = true;   // T is upper triangular: column tile n0 of X1^T T only sums k < n0 + tile
        rc = gemm<double>('N', 'N', nc, pb, pb, 1.0, X1, pb, Tjj, ldt, 0.0, nullptr, 0, X2, pb, g2, s);
        if (rc) return rc;
        GemmOpts g3 = batched(b, b.sV, sX2, b.sV, b.sV);
        g3.force_big = whole_chip_of_big_tiles(b, mp, nc);
        rc = gemm<double>('N', 'T', mp, nc, pb, -1.0, Wp, ldv, X2, pb, 1.0, W2, ldv, W2, ldv, g3, s);
        if (rc) return rc;
        updated = true;
    } else {
        rc = gemm<double>('T', 'N', pb, nc, mp, 1.0, Wp, ldv, W2, ldv, 0.0, nullptr, 0, X1, nc, g1, s);
        if (rc) return rc;
        GemmOpts g2 = batched(b, b.sT, sX1, 0, sX2);
        g2.force_big = big;
        rc = gemm<double>('T', 'N', pb, nc, pb, 1.0, Tjj, ldt, X1, nc, 0.0, nullptr, 0, X2, nc, g2, s);
        if (rc) return rc;
    }
    if (!updated) {
        GemmOpts g3 = batched(b, b.sV, sX2, b.sV, b.sV);
        g3.force_big = pb >= 256 && whole_chip_of_big_tiles(b, mp, nc);
        rc = gemm<double>('N', 'N', mp, nc, pb, -1.0, Wp, ldv, X2, nc, 1.0, W2, ldv, W2, ldv, g3, s);
        if (rc) return rc;
    }
    if (Rdst == nullptr) return NPW_OK;
    const unsigned gy = (unsigned)(pb < 32 ? pb : 32);
    hipLaunchKernelGGL(move_rows_kernel, dim3((unsigned)ceil_div(nc, 256), gy, (unsigned)b.count), dim3(256), 0, s, (int)pb, nc,
                       W2, ldv, b.sV, Rdst, ldr, b.sR);
    NPW_LAUNCH_CHECK();
    return NPW_OK;
}

// Rd = Wtop - X2 for the top `rows` rows of the updated columns (final rows of R), Wtop cleared: the structured
// update's counterpart of move_rows_kernel (the reflectors' top part is the identity, so the top rows change by -X2).
__global__ void move_rows_sub_kernel(int rows, int64_t cols, double* W, int64_t ldw, int64_t sW, const double* X2, int64_t ldx,
                                     int64_t sX, double* Rd, int64_t ldr, int64_t sR) {
    W += (int64_t)blockIdx.z * sW;
    X2 += (int64_t)blockIdx.z * sX;
    Rd += (int64_t)blockIdx.z * sR;
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
        Rd[(int64_t)r * ldr + c] = W[(int64_t)r * ldw + c] - X2[(int64_t)r * ldx + c];
        W[(int64_t)r * ldw + c] = 0.0;
    }
}

// The same update for two stacked upper triangles.  The pb reflectors are [I; Vbot] with Vbot = the first `rows`
// rows of the lower block (everything below is zero), so
//   X1 = Wtop + Vbot^T Wbot,   X2 = T_p^T X1,   Wbot -= Vbot X2,   R rows = Wtop - X2 (Wtop cleared).
int apply_tri(const Batch& b, const double* Vbot, int64_t ldv, int64_t rows, int64_t pb, const double* Tjj, int64_t ldt,
              double* Wtop, double* Wbot, int64_t nc, double* X1, int64_t sX1, double* X2, int64_t sX2, double* skws,
              size_t skcap, double* Rdst, int64_t ldr, hipStream_t s, unsigned long long tag = 0) {
    GemmOpts g1 = batched(b, b.sV, b.sV, b.sV, sX1);
    int64_t want = wanted_splits(b, pb, nc, rows);
    if (skws != nullptr && want > 1 && (size_t)want * pb * nc <= skcap) {
        g1.splitk = (int)want;
        g1.splitk_ws = skws;
    }
    const bool big = pb >= 256 && whole_chip_of_big_tiles(b, pb, nc);
    g1.force_big = big;
    int rc;
    bool updated = false;   // Wbot -= Vbot X2 already done (the streaming kernels)
    if (near_kernels_on && pb <= PB && skws != nullptr && (size_t)pb * nc <= skcap) {
        if (const int frs = fused_slab_rows(b, rows, nc, skcap, sX2, tag, s)) {   // one launch, the R rows included
            const dim3 grid((unsigned)ceil_div(rows, frs), (unsigned)ceil_div(nc, 16 * NEAR_NT), (unsigned)b.count);
            const bool fast = near_fast(rows, pb, nc, frs, Vbot, ldv);
            hipLaunchKernelGGL(fast ? near_fused_kernel<false> : near_fused_kernel<true>, grid, dim3(256), 0, s, (int)rows, (int)pb,
                               (int)nc, Vbot, ldv, b.sV, Wbot, ldv, b.sV, Wtop, Rdst, ldr, b.sR, Tjj, ldt, b.sT,
                               reinterpret_cast<slot_t*>(skws), reinterpret_cast<slot_t*>(X2), tag, frs);
            NPW_LAUNCH_CHECK();
            return NPW_OK;
        }
        updated = true;
        const int rs = near_slab_rows(b, rows, pb, nc, skcap);
        const dim3 grid((unsigned)ceil_div(rows, rs), (unsigned)ceil_div(nc, 16 * NEAR_NT), (unsigned)b.count);
        const bool fast = near_fast(rows, pb, nc, rs, Vbot, ldv);
        hipLaunchKernelGGL(fast ? near_vtw_kernel<false> : near_vtw_kernel<true>, grid, dim3(256), 0, s, (int)rows, (int)pb, (int)nc,
                           Vbot, ldv, b.sV, (const double*)Wbot, ldv, b.sV, skws, rs);
        NPW_LAUNCH_CHECK();
        hipLaunchKernelGGL(reduce_tt_kernel<PB>, dim3((unsigned)ceil_div(nc, 16), (unsigned)b.count), dim3(PB * 16), 0, s, (int)pb, nc,
                           (int)grid.x, skws, Wtop, ldv, b.sV, Tjj, ldt, b.sT, X2, sX2, Rdst, ldr, b.sR);   // (the R rows included)
        NPW_LAUNCH_CHECK();
        hipLaunchKernelGGL(fast ? near_update_kernel<false> : near_update_kernel<true>, grid, dim3(256), 0, s, (int)rows, (int)pb,
                           (int)nc, Vbot, ldv, b.sV, (const double*)X2, nc, sX2, Wbot, ldv, b.sV, rs);
        NPW_LAUNCH_CHECK();
        return NPW_OK;
    } else if (pb <= PB && skws != nullptr && (size_t)pb * nc <= skcap) {
        int nsplit = 1;
        g1.splitk_ws = skws;
        g1.splitk_keep = &nsplit;
        rc = gemm<double>('T', 'N', pb, nc, rows, 1.0, Vbot, ldv, Wbot, ldv, 0.0, nullptr, 0, X1, nc, g1, s);
        if (rc) return rc;
        hipLaunchKernelGGL(reduce_tt_kernel<PB>, dim3((unsigned)ceil_div(nc, 16), (unsigned)b.count), dim3(PB * 16), 0, s, (int)pb, nc,
                           nsplit, skws, Wtop, ldv, b.sV, Tjj, ldt, b.sT, X2, sX2);
        NPW_LAUNCH_CHECK();
    } else {
        rc = gemm<double>('T', 'N', pb, nc, rows, 1.0, Vbot, ldv, Wbot, ldv, 1.0, Wtop, ldv, X1, nc, g1, s);
        if (rc) return rc;
        GemmOpts g2 = batched(b, b.sT, sX1, 0, sX2);
        g2.force_big = big;
        rc = gemm<double>('T', 'N', pb, nc, pb, 1.0, Tjj, ldt, X1, nc, 0.0, nullptr, 0, X2, nc, g2, s);
        if (rc) return rc;
    }
    if (!updated) {
        GemmOpts g3 = batched(b, b.sV, sX2, b.sV, b.sV);
        g3.force_big = pb >= 256 && whole_chip_of_big_tiles(b, rows, nc);
        rc = gemm<double>('N', 'N', rows, nc, pb, -1.0, Vbot, ldv, X2, nc, 1.0, Wbot, ldv, Wbot, ldv, g3, s);
        if (rc) return rc;
    }
    const unsigned gy = (unsigned)(pb < 32 ? pb : 32);
    hipLaunchKernelGGL(move_rows_sub_kernel, dim3((unsigned)ceil_div(nc, 256), gy, (unsigned)b.count), dim3(256), 0, s, (int)pb,
                       nc, Wtop, ldv, b.sV, X2, nc, sX2, Rdst, ldr, b.sR);
    NPW_LAUNCH_CHECK();
    return NPW_OK;
}

// `b.count` independent m x n (m >= n) factorisations in lock step; the working copies are already in V, T and R are
// cleared.  One sequence of launches serves the whole batch: the panel kernel runs count x slabs workgroups, every
// GEMM is a strided batch.
//
// tri: the matrices are two stacked n x n upper triangles (m == 2 n; LAPACK's DTPQRT with l = n).  Reflector j is then
// e_j on top of a vector with j + 1 leading non-zeros in the lower block: a panel at column j0 only touches its pb
// pivot rows and the first j0 + pb rows of the lower block, and the result is V = [I; V2] with V2 upper triangular --
// the same V, T, R as the dense algorithm gives, for about a third of the work.
int geqrt_core(const Batch& b_in, int64_t m, int64_t n, bool tri, double* V, int64_t ldv, double* T, int64_t ldt, double* R,
               int64_t ldr, void* workspace, hipStream_t s) {
    const QrWorkspace q = carve(workspace, m, n, b_in.count, b_in.count);
    double* const Vlow = V + n * ldv;  // tri: the lower block
    {
        // every workgroup of a panel launch waits for the others: the whole launch has to be resident on the stream's CUs
        // (a CU-masked stream of the executor offers fewer than the chip: refuse instead of timing out with wrong numbers)
        const int64_t slots = 2 * (int64_t)resident_cu_count(s);   // (minus RCCL's share while a communicator is live)
        const int64_t rows_max = tri ? n + PB : m;
        const int64_t need = (int64_t)b_in.count * ceil_div(rows_max, 2 * SLAB);
        NPW_REQUIRE(need <= slots, "batched QR: %lld x %lld rows need %lld resident panel workgroups, the stream's %d compute "
                    "units hold %lld", (long long)b_in.count, (long long)rows_max, (long long)need, resident_cu_count(s), (long long)slots);
    }
    // (small batches are bound by the latency of the panel chain, not by GEMM throughput: the extra launches and waits of
    //  the third level cost them time -- 4096^2: x1 17.6 ms without it, 19.9 / 27 ms with SB = 256 / 128; x4 26.3 vs 29.0)
    const int64_t SB = b_in.count >= 8 ? superblock_width(tri) : ((n + OB - 1) / OB) * OB;

    // T == NULL: the caller does not want the compact-WY factor (an "R only" request, see npw_hip.h).  The factorisation
    // itself needs T's diagonal superblocks; they then live in a strip of the workspace: element (r, c) of the superblock
    // that starts at column sb0 is Ts[r * SB + (c - sb0)], i.e. (Ts - sb0)[r * SB + c].
    const bool want_t = T != nullptr;
    Batch b = b_in;
    const int64_t ldtb = want_t ? ldt : SB;
    if (!want_t) {
        b.sT = q.sTs;
        NPW_HIP_CHECK(hipMemsetAsync(q.Ts, 0, (size_t)q.sTs * b.count * sizeof(double), s));
    }
    auto t_base = [&](int64_t sb0) { return want_t ? T : q.Ts - sb0; };

    // sequence tags of the panel kernel's hand-off slots: unique per call (process-wide counter seeded from the
    // clock), panel and column, so that stale slots in a recycled workspace can never look current
    static std::atomic<unsigned long long> call_counter{
        (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() & 0x3fffffffULL};
    const unsigned long long call_tag = (call_counter.fetch_add(1) + 1) << 24;
    // T outside its diagonal superblocks: superblock column by superblock column during the factorisation (DLARFT's
    // recurrence, behind the far updates on the third helper stream), or from the Gram matrix of all reflectors
    // afterwards (bottom-up merges), or -- T == NULL -- not at all.  Round-3 measurements of the per-block form of the
    // progressive variant on 4096^2 tiles: dense x1 20.09 -> 17.84 ms, x8 40.3 -> 37.4, x32 117.2 -> 116.8; stacked
    // triangles x1 19.6 -> 18.8, x8 35.6 -> 35.7, x32 89.8 -> 100.4.  NPW_QR_T_PROGRESSIVE=0 / 1 forces either form.
    static const int t_mode = [] {
        const char* e = getenv("NPW_QR_T_PROGRESSIVE");
        return e == nullptr ? -1 : (atoi(e) != 0 ? 1 : 0);
    }();
    const bool progressive_t = want_t && (t_mode >= 0 ? t_mode == 1 : (!tri || b.count < 8));
    SideStream* side = nullptr;
    SideStream serial_side;
    {
        int rc = side_stream(s, &side);
        if (rc) return rc;
        // $NPW_QR_SERIAL=1 (developer aid): every launch on the caller's stream -- a kernel trace then shows what each
        // kernel costs when it has the chip to itself
        static const bool serial = [] {
            const char* e = getenv("NPW_QR_SERIAL");
            return e != nullptr && atoi(e) != 0;
        }();
        if (serial) {
            serial_side = *side;
            serial_side.stream = serial_side.stream2 = serial_side.stream3 = s;
            side = &serial_side;
        }
        NPW_HIP_CHECK(hipEventRecord(side->join, s));  // so that the closing wait is valid for a single block
    }
    hipStream_t far = side->stream3;
    static const int64_t panel_wgs_env = [] {
        const char* e = getenv("NPW_QR_PANEL_MAX_WGS");
        return e ? (int64_t)atoll(e) : (int64_t)PANEL_MAX_WGS;
    }();
    const int64_t panel_max_wgs = std::min<int64_t>(panel_wgs_env, 2 * (int64_t)resident_cu_count(s));   // (default: half of the chip's slots)
    bool far_in_flight = false;   // a far update has been issued: the next `mid` update waits for its first part

    // T[r0:c0, c0:c0+w] = -T[r0:c0, r0:c0] * (V[:, r0:c0]^T V[:, c0:c0+w]) * T[c0:c0+w, c0:c0+w]   (DLARFT's recurrence for a
    // block column; both diagonal blocks are final).  Tq: base of T for these indices (t_base).  V[:, c0:] is zero above
    // row c0; tri: V = [I; V2] with V2 upper triangular, so the columns left of c0 end at row c0 of the lower block and the
    // identity parts contribute nothing off the diagonal.
    auto t_column = [&](double* Tq, int64_t r0, int64_t c0, int64_t w, double* X1, double* X2, int64_t sX, double* skws,
                        int64_t skcap, hipStream_t st) -> int {
        const int64_t rows_out = c0 - r0;
        if (rows_out <= 0 || w <= 0) return NPW_OK;
        const int64_t kk = tri ? c0 : m - c0;
        const double* Aop = tri ? Vlow + r0 : V + c0 * ldv + r0;
        const double* Bop = tri ? Vlow + c0 : V + c0 * ldv + c0;
        GemmOpts g1 = batched(b, b.sV, b.sV, 0, sX);
        const int64_t wgs = ceil_div(rows_out, 64) * ceil_div(w, 64) * b.count;
        int64_t want = 512 / (wgs > 0 ? wgs : 1);
        if (want > kk / 256) want = kk / 256;
        if (want > 32) want = 32;
        if (want > 1 && want * rows_out * w <= skcap) {
            g1.splitk = (int)want;
            g1.splitk_ws = skws;
        }
        const bool bigt = t_big_tiles && whole_chip_of_big_tiles(b, rows_out, w);
        g1.force_big = bigt;
        int rc = gemm<double>('T', 'N', rows_out, w, kk, 1.0, Aop, ldv, Bop, ldv, 0.0, nullptr, 0, X1, w, g1, st);
        if (rc) return rc;
        GemmOpts g2 = batched(b, sX, b.sT, 0, sX);
        g2.force_big = bigt;
        rc = gemm<double>('N', 'N', rows_out, w, w, 1.0, X1, w, Tq + c0 * ldtb + c0, ldtb, 0.0, nullptr, 0, X2, w, g2, st);
        if (rc) return rc;
        GemmOpts g3 = batched(b, b.sT, sX, 0, b.sT);
        g3.force_big = bigt;
        g3.a_upper_tri = true;
        return gemm<double>('N', 'N', rows_out, w, rows_out, -1.0, Tq + r0 * ldtb + r0, ldtb, X2, w, 0.0, nullptr, 0,
                            Tq + r0 * ldtb + c0, ldtb, g3, st);
    };

    // ---- the far updates: which superblock reflector reaches which columns when -------------------------------------
    // Reflector S (available once superblock S is factored) has to be applied to every column right of its own window
    // (mid_end(S)) -- in the order of S for any given column, and to the columns the chain touches next (up to mid_end(S)
    // + SB) right away.  Default: to ALL columns at once (right-looking), the window first (then the event the next
    // superblock's updates wait for).  That makes the first superblocks' steps several times longer than the panel chain
    // beside them and leaves the last ones nothing to do (r04 traces: blocks 0 - 15 of a batch of 32 are bound by the far
    // stream, 16 - 31 by the chain), so a deadline-ordered schedule with a per-step quota was built as well
    // ($NPW_QR_FAR_QUOTA = share factor, 1 = equal shares of what is left): step t brings the next window up to date
    // (every reflector <= t), then reflector t on the columns the older ones have already reached, then as many further
    // columns -- all reflectors <= t each -- as the step's share allows.  Measured on 32 x 4096^2 (gpurun_out/r04j): R only
    // 78.0 ms right-looking, 86.7 / 87.0 / 84.9 / 81.4 at quota 0.8 / 1 / 1.3 / 1.7 -- the narrow catch-up strips cost more
    // than the balance gains (the streams slow each other down whenever they overlap, whatever the order): not the default.
    // `far_front`: columns < far_front have every reflector < t.
    static const double far_quota = [] {
        const char* e = getenv("NPW_QR_FAR_QUOTA");
        return e ? atof(e) : 0.0;
    }();
    const int64_t nsb = ceil_div(n, SB);
    auto sb_mid_end = [&](int64_t S) {
        const int64_t e = (S * SB + SB < n ? S * SB + SB : n) + LA;
        return e < n ? e : n;
    };
    auto sb_rows = [&](int64_t S) { return tri ? ((S * SB + SB < n) ? S * SB + SB : n) : m - S * SB; };
    auto apply_sb = [&](int64_t S, int64_t c0, int64_t c1) -> int {
        if (c1 <= c0) return NPW_OK;
        const int64_t s0 = S * SB, sw = ((s0 + SB < n) ? s0 + SB : n) - s0;
        double* const Ts = t_base(s0) + s0 * ldtb + s0;
        return tri ? apply_tri(b, Vlow + s0, ldv, s0 + sw, sw, Ts, ldtb, V + s0 * ldv + c0, Vlow + c0, c1 - c0, q.F1, q.sF, q.F2, q.sF,
                               q.Gf, (size_t)q.sGf, R + s0 * ldr + c0, ldr, far)
                   : apply_panel(b, V + s0 * ldv + s0, ldv, m - s0, sw, Ts, ldtb, V + s0 * ldv + c0, c1 - c0, q.F1, q.sF, q.F2, q.sF,
                                 q.Gf, (size_t)q.sGf, R + s0 * ldr + c0, ldr, far);
    };
    int64_t far_front = sb_mid_end(0);
    auto far_step = [&](int64_t t) -> int {
        const int64_t me = sb_mid_end(t);
        const int64_t M = (me + SB < n) ? me + SB : n;                       // mandatory: the next window's columns
        int64_t target = n;
        if (far_quota > 0.0) {
            // work in units of rows x columns (the GEMMs' flops / (4 pb)): what is left over all reflectors, shared
            // equally among the steps that are left
            double left = 0.0;
            for (int64_t S = 0; S < nsb; ++S) {
                const int64_t from = S < t ? far_front : sb_mid_end(S);
                if (from < n) left += (double)sb_rows(S) * (double)(n - from);
            }
            int64_t steps_left = 0;
            for (int64_t S = t; S < nsb; ++S) steps_left += sb_mid_end(S) < n ? 1 : 0;
            const double budget = far_quota * left / (double)(steps_left > 0 ? steps_left : 1);
            double older = 0.0;                                                 // rows of the reflectors < t
            for (int64_t S = 0; S < t; ++S) older += (double)sb_rows(S);
            const int64_t base = far_front > M ? far_front : M;                 // columns < base are done after parts 1 - 3
            double spent = older * (double)(M > far_front ? M - far_front : 0) + (double)sb_rows(t) * (double)(base - me);
            const double per_col = older + (double)sb_rows(t);
            int64_t extra = budget > spent ? (int64_t)((budget - spent) / per_col) : 0;
            extra = (extra / OB) * OB;
            target = base + extra;
            if (target > n || n - target < 2 * OB) target = n;                  // no slivers at the end
        }
        int rc;
        for (int64_t S = 0; S < t; ++S) {                                       // 1. older reflectors on the window's new columns
            rc = apply_sb(S, far_front, M);
            if (rc) return rc;
        }
        rc = apply_sb(t, me, M);                                                // 2. this reflector on the window
        if (rc) return rc;
        if (M > me) {
            NPW_HIP_CHECK(hipEventRecord(side->join3, far));
            far_in_flight = true;
        }
        rc = apply_sb(t, M, far_front);                                         // 3. ... and on the columns the older ones have reached
        if (rc) return rc;
        const int64_t base = far_front > M ? far_front : M;
        for (int64_t S = 0; S <= t && target > base; ++S) {                     // 4. further columns: everything so far
            rc = apply_sb(S, base, target);
            if (rc) return rc;
        }
        far_front = target > base ? target : base;
        return NPW_OK;
    };

    // Three levels of blocking.  The panel chain (latency-bound: one small persistent launch per PB columns) runs on the
    // caller's stream and keeps only the rest of its OWN OB-wide block up to date (PB-wide reflectors, three small
    // launches per panel).  The NEXT block's columns get the same per-panel updates on a second helper stream, beside the
    // chain instead of inside it: the next panel does not read them, only the next block's first panel does (one wait
    // per block).  The columns right of that, up to LA columns past the end of the block's SB-wide superblock ("mid"),
    // are updated once per OB columns on the first helper stream with the block reflector (V_b, T_b).  Everything further
    // right ("far") is updated once per SUPERBLOCK on the third helper stream with (V_S, T_S): k = SB products on
    // 128 x 128 tiles that read and write the big trailing matrix SB / OB times less often than per-block updates.
    for (int64_t b0 = 0; b0 < n; b0 += OB) {
        const int64_t ob = (n - b0 < OB) ? n - b0 : OB;
        const int64_t own_end = b0 + ob;
        const int64_t near_end = (own_end + OB < n) ? own_end + OB : n;  // end of the next block
        const int64_t sb0 = (b0 / SB) * SB;
        const int64_t sb_end = (sb0 + SB < n) ? sb0 + SB : n;
        const int64_t mid_end = (sb_end + LA < n) ? sb_end + LA : n;    // end of the columns the block reflectors of this superblock update
        double* const Tq = t_base(sb0);
        // A batch runs the next block's near updates on the second helper stream (measured on 4096^2 tiles: x32 122.0 ->
        // 116.7 ms per batch, x16 67.0 -> 66.0); a single factorisation keeps them in the chain's own launches -- its
        // three near-update kernels are latency-bound whatever their width, and the extra event per panel costs more than
        // the narrower update saves (20.0 -> 21.0 ms).
        const bool split_near = b.count >= 8;
        const int64_t nnext = split_near ? near_end - own_end : 0;
        const int64_t chain_end = split_near ? own_end : near_end;   // columns the chain's own near update covers
        // the next block's columns were last written by the mid update of the previous block (its "part 0")
        // (a single factorisation waits on its own stream, after the block's first panel kernel: one panel time of slack)
        if (split_near && b0 > 0 && nnext > 0) NPW_HIP_CHECK(hipStreamWaitEvent(side->stream2, side->join, 0));
        for (int64_t j0 = b0; j0 < own_end; j0 += PB) {
            const int64_t pb = (own_end - j0 < PB) ? own_end - j0 : PB;
            const int64_t mp = tri ? pb + (j0 + pb) : m - j0;   // tri: pivot rows + the leading rows of the lower block
            double* Wp = V + j0 * ldv + j0;
            // every workgroup of the launch has to be resident: beyond a quarter of the chip's slots per launch (two
            // workgroups fit a CU, two batches may run side by side on two streams of the executor, the far updates' GEMM
            // workgroups hold slots as well) a slab takes two rows per thread
            const int rpt = (b.count * ceil_div(mp, SLAB) > panel_max_wgs) ? 2 : 1;
            const int G = (int)ceil_div(mp, SLAB * rpt);
            hipLaunchKernelGGL(rpt == 2 ? qr_panel3_kernel<2> : qr_panel3_kernel<1>, dim3(G, b.count), dim3(SLAB), 0, s, (int)mp, (int)pb, Wp, ldv,
                               Tq + j0 * ldtb + j0, ldtb, R + j0 * ldr + j0, ldr, reinterpret_cast<slot_t*>(q.Part),
                               reinterpret_cast<slot_t*>(q.RowBuf), call_tag + (unsigned long long)(j0 / PB) * 64, b.sV, b.sT,
                               b.sR, q.sPart / 2, q.sRow / 2, tri ? (int)pb : (int)mp, tri ? n - j0 - pb : (int64_t)0);
            NPW_LAUNCH_CHECK();
            if (nnext > 0) {
                NPW_HIP_CHECK(hipEventRecord(side->fork2, s));
                NPW_HIP_CHECK(hipStreamWaitEvent(side->stream2, side->fork2, 0));
                int rc = tri ? apply_tri(b, Vlow + j0, ldv, j0 + pb, pb, Tq + j0 * ldtb + j0, ldtb, V + j0 * ldv + own_end,
                                         Vlow + own_end, nnext, q.YnA, q.sXn, q.YnB, q.sXn, q.YnS, (size_t)q.sXnS,
                                         R + j0 * ldr + own_end, ldr, side->stream2)
                             : apply_panel(b, Wp, ldv, mp, pb, Tq + j0 * ldtb + j0, ldtb, V + j0 * ldv + own_end, nnext, q.YnA, q.sXn,
                                           q.YnB, q.sXn, q.YnS, (size_t)q.sXnS, nullptr, ldr, side->stream2);   // R rows: moved per block, below
                if (rc) return rc;
            }
            const int64_t nc = chain_end - j0 - pb;
            if (nc > 0) {
                if (!split_near && j0 == b0 && b0 > 0 && near_end > own_end) NPW_HIP_CHECK(hipStreamWaitEvent(s, side->join, 0));
                // (the panel kernel's columns use the tags 0 .. PB - 1 of the panel's 64; the fused near update takes the next one)
                const unsigned long long near_tag = call_tag + (unsigned long long)(j0 / PB) * 64 + PB;
                int rc = tri ? apply_tri(b, Vlow + j0, ldv, j0 + pb, pb, Tq + j0 * ldtb + j0, ldtb, Wp + pb, Vlow + j0 + pb, nc, q.XnA,
                                         q.sXn, q.XnB, q.sXn, q.XnS, (size_t)q.sXnS, R + j0 * ldr + j0 + pb, ldr, s, near_tag)
                             : apply_panel(b, Wp, ldv, mp, pb, Tq + j0 * ldtb + j0, ldtb, Wp + pb, nc, q.XnA, q.sXn, q.XnB, q.sXn,
                                           q.XnS, (size_t)q.sXnS, nullptr, ldr, s, near_tag);   // R rows: moved per block, below
                if (rc) return rc;
            }
        }
        if (nnext > 0) {
            // the next block's first panel (and this block's row move and mid update) start from the updated columns
            NPW_HIP_CHECK(hipEventRecord(side->join2, side->stream2));
            NPW_HIP_CHECK(hipStreamWaitEvent(s, side->join2, 0));
        }
        const int64_t nmid = mid_end - near_end;
        const bool move_near = !tri && near_end - b0 > PB;   // the dense near updates left R rows behind
        const bool last_in_sb = own_end == sb_end;
        const bool far_follows = last_in_sb && (n - mid_end > 0 || far_front < n || (progressive_t && sb0 > 0));
        // (T's block column inside the superblock is only ever read by the superblock's reflector or by the caller: an
        //  R-only call with a single superblock needs neither -- ADVICE r4)
        const bool t_col = b0 > sb0 && (want_t || nsb > 1);
        if (ob > PB || nmid > 0 || move_near || t_col || far_follows) {
            // block reflector on the side stream: T_b from the Gram matrix of the block's columns, then the mid update
            NPW_HIP_CHECK(hipEventRecord(side->fork, s));   // (behind the wait for join2: covers the second helper stream too)
            NPW_HIP_CHECK(hipStreamWaitEvent(side->stream, side->fork, 0));
            if (move_near) {
                const int64_t cw = near_end - b0;
                hipLaunchKernelGGL(move_block_rows_kernel, dim3((unsigned)ceil_div(cw, 256), 32, (unsigned)b.count), dim3(256), 0,
                                   side->stream, (int)ob, (int)PB, cw, V + b0 * ldv + b0, ldv, b.sV, R + b0 * ldr + b0, ldr, b.sR);
                NPW_LAUNCH_CHECK();
            }
            // tri: the block's reflectors are [I; Vlow(0 : b0 + ob, b0 : b0 + ob)]; the identity adds nothing to the
            // off-diagonal blocks of the Gram matrix, which are all merge_t reads
            const int64_t mb = tri ? b0 + ob : m - b0;
            double* Vb = tri ? Vlow + b0 : V + b0 * ldv + b0;
            if (ob > PB) {
                GemmOpts sk = batched(b, b.sV, b.sV, 0, q.sGb);
                int64_t want = mb / 256;
                if (want > 32) want = 32;
                if (want > q.sX / (ob * ob)) want = q.sX / (ob * ob);
                if (want > 1) {  // ob x ob output with a contraction over all rows: split k (scratch: X1, free until the mid update)
                    sk.splitk = (int)want;
                    sk.splitk_ws = q.X1;
                }
                int rc = gemm<double>('T', 'N', ob, ob, mb, 1.0, Vb, ldv, Vb, ldv, 0.0, nullptr, 0, q.Gb, OB, sk, side->stream);
                if (rc) return rc;
                rc = merge_t(b, b0, b0 + ob, PB, Tq, ldtb, q.Gb, OB, q.sGb, b0, q.Tmp, q.sTmp, side->stream);
                if (rc) return rc;
            }
            // the mid columns right of the previous superblock's window still miss that superblock's reflectors until the
            // first part of its far update is through
            if (nmid > 0 && far_in_flight) {
                NPW_HIP_CHECK(hipStreamWaitEvent(side->stream, side->join3, 0));
                far_in_flight = false;
            }
            // The mid update in two parts: first the block after next -- the only mid columns the panel chain touches
            // during the NEXT block -- then the join event, then the rest.  The caller's stream then waits for a short
            // chain of small launches, not for the bigger GEMMs.
            for (int part = 0; part < 2 && nmid > 0; ++part) {
                const int64_t c0 = (part == 0) ? near_end : ((near_end + OB < mid_end) ? near_end + OB : mid_end);
                const int64_t c1 = (part == 0) ? ((near_end + OB < mid_end) ? near_end + OB : mid_end) : mid_end;
                if (c1 > c0) {
                    int rc = tri ? apply_tri(b, Vb, ldv, mb, ob, Tq + b0 * ldtb + b0, ldtb, V + b0 * ldv + c0, Vlow + c0, c1 - c0, q.X1,
                                             q.sX, q.X2, q.sX, q.G, (size_t)q.sG, R + b0 * ldr + c0, ldr, side->stream)
                                 : apply_panel(b, Vb, ldv, mb, ob, Tq + b0 * ldtb + b0, ldtb, V + b0 * ldv + c0, c1 - c0, q.X1, q.sX,
                                               q.X2, q.sX, q.G, (size_t)q.sG, R + b0 * ldr + c0, ldr, side->stream);
                    if (rc) return rc;
                }
                if (part == 0) NPW_HIP_CHECK(hipEventRecord(side->join, side->stream));
            }
            if (nmid <= 0) NPW_HIP_CHECK(hipEventRecord(side->join, side->stream));
            // T's block column of this block inside its superblock (rows sb0 .. b0): T_S has to be complete before the
            // superblock's reflector is applied
            if (t_col) {
                int rc = t_column(Tq, sb0, b0, ob, q.X1, q.X2, q.sX, q.G, q.sG, side->stream);
                if (rc) return rc;
            }
            if (far_follows) {
                // the superblock reflectors (V_S, T_S) on the third helper stream (see far_step); then (progressive T) T's
                // rows above this superblock's diagonal block
                NPW_HIP_CHECK(hipEventRecord(side->fork3, side->stream));
                NPW_HIP_CHECK(hipStreamWaitEvent(far, side->fork3, 0));
                int rc = far_step(sb0 / SB);
                if (rc) return rc;
                if (progressive_t && sb0 > 0) {
                    rc = t_column(T, 0, sb0, sb_end - sb0, q.F1, q.F2, q.sF, q.Gf, q.sGf, far);
                    if (rc) return rc;
                }
            }
        }
    }
    NPW_HIP_CHECK(hipEventRecord(side->join, side->stream));  // everything the helper streams still have in flight
    NPW_HIP_CHECK(hipStreamWaitEvent(s, side->join, 0));
    NPW_HIP_CHECK(hipEventRecord(side->join3, far));
    NPW_HIP_CHECK(hipStreamWaitEvent(s, side->join3, 0));
    if (want_t && n > SB && !progressive_t) {
        // G = V^T V, then the off-diagonal SB-blocks of T bottom-up (the diagonal ones are final)
        // (lower triangle only, and V is lower trapezoidal: tile (i0, j0) sums over the rows from max(i0, j0) on --
        //  a sixth of the full product for a square matrix)
        GemmOpts gg = batched(b, b.sV, b.sV, 0, q.sG);
        gg.lower_only = true;
        int rc;
        if (tri) {  // V^T V = I + V2^T V2 with V2 upper triangular: column tile j0 only sums over rows < j0 + tile
            gg.b_lower_tri = true;
            rc = gemm<double>('T', 'N', n, n, n, 1.0, Vlow, ldv, Vlow, ldv, 0.0, nullptr, 0, q.G, n, gg, s);
        } else {
            gg.k_from_diag = true;
            rc = gemm<double>('T', 'N', n, n, m, 1.0, V, ldv, V, ldv, 0.0, nullptr, 0, q.G, n, gg, s);
        }
        if (rc) return rc;
        rc = merge_t(b, 0, n, SB, T, ldt, q.G, n, q.sG, 0, q.Tmp, q.sTmp, s);
        if (rc) return rc;
    }
    return NPW_OK;
}

}  // namespace
}  // namespace npw

using namespace npw;

extern "C" {

size_t npw_dgeqrt_workspace_bytes(int64_t m, int64_t n) {
    if (m <= 0 || n <= 0) return 0;
    if (m < n)  // wide: the square factorisation of the leading block + two m x (n - m) GEMM temporaries
        return npw_dgeqrt_workspace_bytes(m, m) + 2 * align2((size_t)m * (n - m)) * sizeof(double);
    return square_workspace_doubles(m, n, 1) * sizeof(double);
}

int npw_dgeqrt(int64_t m, int64_t n, const double* A, int64_t lda, double* V, int64_t ldv,
               double* T, int64_t ldt, double* R, int64_t ldr, void* workspace,
               npw_stream_t stream) {
    NPW_REQUIRE(m >= 0 && n >= 0, "npw_dgeqrt: negative dimension");
    if (n == 0) return NPW_OK;
    if (m == 0) return NPW_OK;
    NPW_REQUIRE(A && V && R && workspace, "npw_dgeqrt: NULL argument");
    NPW_REQUIRE(T != nullptr || m >= n, "npw_dgeqrt: T may only be omitted for m >= n");
    if (m < n) {
        // More columns than rows (reference kernels.py:94-95 -> slow_qr 67-84: DGEQRF + DLARFT): k = m reflectors, all
        // of them determined by the leading m x m block A1; the other columns only receive Q^T:
        //   V (m x m), T (m x m), R = [R1 | A2 - V (T^T (V^T A2))]  (m x n upper trapezoid).
        NPW_REQUIRE(lda >= n && ldr >= n && ldv >= m && ldt >= m, "npw_dgeqrt: leading dimension too small");
        NPW_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "npw_dgeqrt: workspace not 16B aligned");
        const int64_t n2 = n - m;
        int rc = npw_dgeqrt(m, m, A, lda, V, ldv, T, ldt, R, ldr, workspace, stream);
        if (rc) return rc;
        double* X = reinterpret_cast<double*>(static_cast<char*>(workspace) + npw_dgeqrt_workspace_bytes(m, m));
        double* W = X + align2((size_t)m * n2);
        hipStream_t s2 = as_stream(stream);
        rc = gemm<double>('T', 'N', m, n2, m, 1.0, V, ldv, A + m, lda, 0.0, nullptr, 0, X, n2, GemmOpts(), s2);
        if (rc) return rc;
        rc = gemm<double>('T', 'N', m, n2, m, 1.0, T, ldt, X, n2, 0.0, nullptr, 0, W, n2, GemmOpts(), s2);
        if (rc) return rc;
        return gemm<double>('N', 'N', m, n2, m, -1.0, V, ldv, W, n2, 1.0, A + m, lda, R + m, ldr, GemmOpts(), s2);
    }
    NPW_REQUIRE(lda >= n && ldv >= n && (T == nullptr || ldt >= n) && ldr >= n, "npw_dgeqrt: leading dimension too small");
    NPW_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "npw_dgeqrt: workspace not 16B aligned");
    NPW_REQUIRE((const void*)A != (const void*)V, "npw_dgeqrt: V must not alias A");
    hipStream_t s = as_stream(stream);
    // working copy: the factorisation runs in place inside V
    NPW_HIP_CHECK(hipMemcpy2DAsync(V, ldv * 8, A, lda * 8, n * 8, m, hipMemcpyDeviceToDevice, s));
    if (T) NPW_HIP_CHECK(hipMemset2DAsync(T, ldt * 8, 0, n * 8, n, s));
    NPW_HIP_CHECK(hipMemset2DAsync(R, ldr * 8, 0, n * 8, n, s));
    return geqrt_core(Batch(), m, n, false, V, ldv, T, ldt, R, ldr, workspace, s);
}

size_t npw_dgeqrt_batched_workspace_bytes(int count, int64_t m, int64_t n) {
    if (count <= 0 || m <= 0 || n <= 0 || m < n) return 0;
    return (size_t)count * square_workspace_doubles(m, n, count) * sizeof(double);
}

int npw_dgeqrt_batched(int count, int64_t m, int64_t n, const double* const* A, int64_t lda, double* V, int64_t ldv,
                       int64_t stride_v, double* T, int64_t ldt, int64_t stride_t, double* R, int64_t ldr,
                       int64_t stride_r, void* workspace, npw_stream_t stream) {
    NPW_REQUIRE(count >= 0 && m >= 0 && n >= 0, "npw_dgeqrt_batched: negative argument");
    if (count == 0 || n == 0 || m == 0) return NPW_OK;
    NPW_REQUIRE(m >= n, "npw_dgeqrt_batched: m (%lld) < n (%lld): use npw_dgeqrt", (long long)m, (long long)n);
    NPW_REQUIRE(count <= 65535, "npw_dgeqrt_batched: more than 65535 matrices");
    NPW_REQUIRE(A && V && R && workspace, "npw_dgeqrt_batched: NULL argument");
    NPW_REQUIRE(lda >= n && ldv >= n && (T == nullptr || ldt >= n) && ldr >= n, "npw_dgeqrt_batched: leading dimension too small");
    NPW_REQUIRE(stride_v >= m * ldv && (T == nullptr || stride_t >= n * ldt) && stride_r >= n * ldr, "npw_dgeqrt_batched: stride too small");
    NPW_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "npw_dgeqrt_batched: workspace not 16B aligned");
    hipStream_t s = as_stream(stream);
    for (int z = 0; z < count; ++z) {
        NPW_REQUIRE(A[z] != nullptr, "npw_dgeqrt_batched: A[%d] is NULL", z);
        NPW_HIP_CHECK(hipMemcpy2DAsync(V + (int64_t)z * stride_v, ldv * 8, A[z], lda * 8, n * 8, m, hipMemcpyDeviceToDevice, s));
    }
    if (T == nullptr) {
        // R only: no compact-WY factor
    } else if (stride_t == n * ldt && ldt == n) {
        NPW_HIP_CHECK(hipMemsetAsync(T, 0, (size_t)count * n * n * 8, s));
    } else {
        for (int z = 0; z < count; ++z) NPW_HIP_CHECK(hipMemset2DAsync(T + (int64_t)z * stride_t, ldt * 8, 0, n * 8, n, s));
    }
    if (stride_r == n * ldr && ldr == n) {
        NPW_HIP_CHECK(hipMemsetAsync(R, 0, (size_t)count * n * n * 8, s));
    } else {
        for (int z = 0; z < count; ++z) NPW_HIP_CHECK(hipMemset2DAsync(R + (int64_t)z * stride_r, ldr * 8, 0, n * 8, n, s));
    }
    Batch b;
    b.count = count;
    b.sV = stride_v;
    b.sT = stride_t;
    b.sR = stride_r;
    return geqrt_core(b, m, n, false, V, ldv, T, ldt, R, ldr, workspace, s);
}

size_t npw_dtpqrt_batched_workspace_bytes(int count, int64_t n) {
    return npw_dgeqrt_batched_workspace_bytes(count, 2 * n, n);
}

int npw_dtpqrt_batched(int count, int64_t n, const double* const* A1, const double* const* A2, int64_t lda, double* V,
                       int64_t ldv, int64_t stride_v, double* T, int64_t ldt, int64_t stride_t, double* R, int64_t ldr,
                       int64_t stride_r, void* workspace, npw_stream_t stream) {
    NPW_REQUIRE(count >= 0 && n >= 0, "npw_dtpqrt_batched: negative argument");
    if (count == 0 || n == 0) return NPW_OK;
    NPW_REQUIRE(count <= 65535, "npw_dtpqrt_batched: more than 65535 matrices");
    NPW_REQUIRE(A1 && A2 && V && R && workspace, "npw_dtpqrt_batched: NULL argument");
    NPW_REQUIRE(lda >= n && ldv >= n && (T == nullptr || ldt >= n) && ldr >= n, "npw_dtpqrt_batched: leading dimension too small");
    NPW_REQUIRE(stride_v >= 2 * n * ldv && (T == nullptr || stride_t >= n * ldt) && stride_r >= n * ldr, "npw_dtpqrt_batched: stride too small");
    NPW_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "npw_dtpqrt_batched: workspace not 16B aligned");
    hipStream_t s = as_stream(stream);
    for (int z = 0; z < count; ++z) {
        NPW_REQUIRE(A1[z] != nullptr && A2[z] != nullptr, "npw_dtpqrt_batched: A[%d] is NULL", z);
        double* Vz = V + (int64_t)z * stride_v;
        NPW_HIP_CHECK(hipMemcpy2DAsync(Vz, ldv * 8, A1[z], lda * 8, n * 8, n, hipMemcpyDeviceToDevice, s));
        NPW_HIP_CHECK(hipMemcpy2DAsync(Vz + n * ldv, ldv * 8, A2[z], lda * 8, n * 8, n, hipMemcpyDeviceToDevice, s));
        if (T) NPW_HIP_CHECK(hipMemset2DAsync(T + (int64_t)z * stride_t, ldt * 8, 0, n * 8, n, s));
        NPW_HIP_CHECK(hipMemset2DAsync(R + (int64_t)z * stride_r, ldr * 8, 0, n * 8, n, s));
    }
    Batch b;
    b.count = count;
    b.sV = stride_v;
    b.sT = stride_t;
    b.sR = stride_r;
    return geqrt_core(b, 2 * n, n, true, V, ldv, T, ldt, R, ldr, workspace, s);
}

int npw_dgeqrt_handoff_timeouts(int* count, int reset) {
    NPW_REQUIRE(count != nullptr, "npw_dgeqrt_handoff_timeouts: NULL");
    NPW_HIP_CHECK(hipMemcpyFromSymbol(count, HIP_SYMBOL(qr_handoff_timeouts), sizeof(int)));
    if (reset && *count != 0) {
        const int zero = 0;
        NPW_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(qr_handoff_timeouts), &zero, sizeof(int)));
    }
    return NPW_OK;
}

#ifdef NPW_QR_STAMPS
int npw_debug_qr_stamps(long long* out, int reset) {
    if (out) NPW_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(qr_stamps), sizeof(long long) * 8));
    if (reset) {
        long long z[8] = {0};
        NPW_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(qr_stamps), z, sizeof(z)));
    }
    return NPW_OK;
}
#endif

}  // extern "C"
