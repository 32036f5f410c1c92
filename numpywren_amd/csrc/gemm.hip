// gemm.hip -- MFMA tile GEMM for gfx950 (MI355X):  D = alpha * op(A) op(B) + beta * C
//
// This is the arithmetic behind kernels.gemm / kernels.syrk / the GEMM-rich parts of
// trsm, chol and the QR family (reference numpywren/kernels.py:239-244, 212-215).
//
// Design (CDNA4, wave64):
//   * one workgroup = 256 threads = 4 waves in a 2x2 arrangement; block tile BM x BN x BK,
//     each wave owns a (BM/2) x (BN/2) sub-tile as TM x TN MFMA 16x16 accumulators.
//   * fp64: v_mfma_f64_16x16x4_f64 (A: lane l holds A[l&15][l>>4], B: B[l>>4][l&15],
//     D: col = l&15, row = (l>>4) + 4*reg).  fp32: v_mfma_f32_16x16x4_f32 (same A/B
//     maps, D row = 4*(l>>4) + reg).
//   * operands are staged global -> registers -> LDS in 16-byte chunks, double-buffered:
//     the loads of k-tile t+1 are issued before the MFMAs of k-tile t and written to the
//     other LDS buffer after them; one barrier per k-tile.  Two workgroups per CU
//     (launch_bounds(256, 2)) cover each other's barrier stalls.
//   * each operand keeps its *natural* layout in LDS, so no transposition is needed
//     while staging:
//       "KC" (k contiguous in memory: A of op N, B of op T):  Xs[row][k]; fp64: unpadded rows with
//            XOR-swizzled 16-byte chunks (kc_off), fp32: ld = BK + 4
//       "MC" (m/n contiguous in memory: A of op T, B of op N): Xs[k][col], ld = BMN + 16
//     MFMA step pairs (2p, 2p+1) use adjacent k (k_of), so a KC fragment for two steps is one
//     ds_read_b128 (fp64) / ds_read_b64 (fp32); SQ_LDS_BANK_CONFLICT = 0 for the trailing update.
//   * 1-D grid with an XCD-aware, grouped block->tile map so the 8 private L2s each see a
//     compact patch of the output.
//   * EDGE instantiations (any shape / alignment) guard every global access; the fast
//     instantiations require full tiles and 16-byte aligned rows.
#include "npw_internal.h"

namespace npw {
namespace {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));

template <typename T>
struct MfmaTraits;

template <>
struct MfmaTraits<double> {
    using acc_t = d4_t;
    using vec_t = d2_t;  // 16-byte chunk
    static constexpr int VEC = 2;
    static constexpr int PADK = 0;  // KC rows are not padded: 16-byte chunks are XOR-swizzled instead (kc_off)
    static constexpr int PAD_MC = 8;
    __device__ static inline acc_t mfma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // row of accumulator register r for lane group g (= lane >> 4)
    __device__ static inline int acc_row(int g, int r) { return g + 4 * r; }
};

template <>
struct MfmaTraits<float> {
    using acc_t = f4_t;
    using vec_t = f4_t;
    static constexpr int VEC = 4;
    static constexpr int PADK = 0;
    static constexpr int PAD_MC = 4;
    __device__ static inline acc_t mfma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    __device__ static inline int acc_row(int g, int r) { return 4 * g + r; }
};

// k index inside a BK tile consumed by lane group g at MFMA step s.  Any bijection works as long as A and B
// agree: VEC consecutive steps of lane group g use VEC adjacent k (VEC = elements per 16 bytes), so a KC fragment
// for VEC steps is ONE ds_read_b128.
template <typename T>
__device__ inline int k_of(int s, int g) {
    constexpr int VEC = MfmaTraits<T>::VEC;
    return (s / VEC) * (4 * VEC) + VEC * g + (s % VEC);
}

// Element offset of (row, k) inside a KC tile (`Xs[row][k]`): rows of exactly BK elements whose 16-byte chunks are
// XOR-swizzled with the row number.  (A padded layout that is conflict-free for ds_read_b64 does not survive
// hipcc fusing the fragment loads of consecutive steps into ds_read2_b64, which is serviced in 16-lane groups
// over 32 banks at half the rate: PMC showed ~40 % of the LDS cycles of the fp64 and fp32 kernels as bank
// conflicts.)  A fragment is one ds_read_b128, whose 16-lane groups {0-3,12-15,20-27},... mix rows li and
// lane groups g, g+1; chunk ^ f(row) makes each group cover all 64 banks exactly once, and the 8-lane groups of
// ds_write_b128 stay conflict-free.  f depends on the chunks per row: 4 -> -(row>>2) & 3, 8 -> (row>>1) & 7,
// 16 -> row & 15 (checked by enumeration against the bank rules of MI355X_MICROARCH.md).
template <typename T, int BK>
__device__ inline int kc_off(int row, int k) {
    constexpr int VEC = MfmaTraits<T>::VEC;
    constexpr int CPR = BK / VEC;
    static_assert(CPR == 4 || CPR == 8 || CPR == 16, "kc_off: unsupported k-tile depth");
    const int f = (CPR == 4) ? ((-(row >> 2)) & 3) : (CPR == 8) ? ((row >> 1) & 7) : (row & 15);
    return row * BK + ((((k / VEC) ^ f) * VEC) | (k % VEC));
}

template <typename T>
struct GemmParams {
    const T* A;
    const T* B;
    const T* C;
    T* D;
    int64_t lda, ldb, ldc, ldd;
    int M, N, K;
    T alpha, beta;
    const int32_t* skip0;
    const int32_t* skip1;
    int tiles_m, tiles_n;
    int tiles_batch;
    int lower_only;  // 0 all tiles | 1 rectangular grid, early exit above the diagonal | 2 lower triangle | 3 strictly lower
    int tag;
    int k_chunk;           // split-K: k range handled per blockIdx.y (0 = no split)
    int64_t split_stride;  // split-K: element stride between the partial outputs
    int64_t batch_a, batch_b, batch_c, batch_d;  // element strides between the problems of a batch (blockIdx.z)
    int batch_inner;                                  // two-level batch: z -> (z / inner, z % inner)
    int64_t batch2_a, batch2_b, batch2_c, batch2_d;
    int b_lower_tri;                                  // k range of column tile n0 ends at n0 + BN
    int b_blockdiag;                                  // > 0: op(B) = W^T, W block diagonal with lower triangular blocks of this width;
                                                      // < 0: op(A) = W instead (width = -b_blockdiag)
                                                      // (in the slot of a former pad word: without a word here the arrays below
                                                      // start 8 bytes earlier and the 128 x 128 trailing update measured 0.3 %
                                                      // slower, 1.9233 vs 1.9176 ms)
    int k_from_diag;                                  // k range of tile (m0, n0) starts at max(m0, n0) (trapezoidal operands)
    int a_upper_tri;                                  // k range of row tile m0 starts at m0 (op(A) upper triangular)
    int use_delta;                                    // irregular batch: element offsets per problem instead of strides
    int pad_;                                         // (sixth int of the group: the padding before the int64 arrays, named)
    int64_t da[16], db[16], dc[16], dd[16];
    int64_t ds0[16], ds1[16];                         // ... and of the skip flags (int32 units)
    // (at the END: nothing above moves)  tag 2: diagonal blocks in the same launch -- see GemmOpts::diag_ws
    T* diag_ws;
    int diag_split, diag_chunk;
};

// XCD-aware + grouped mapping of the linear block id to an output tile.
// Blocks are dispatched round-robin over the 8 XCDs (observed; speed only): give each XCD a
// contiguous range of the tile sequence, and order the sequence in GROUP-row bands walked
// column-major so concurrently running blocks share A row-panels and B column-panels.
__device__ inline void block_to_tile(int tiles_m, int tiles_n, bool spread, int& tm, int& tn) {
    const int nwg = tiles_m * tiles_n;
    const int b = blockIdx.x;
    constexpr int NXCD = 8;
    int id;
    if (spread) {
        // tiles of unequal length (triangular operands: the k range depends on the tile's row or column): a contiguous
        // run per XCD would hand one XCD all the long tiles; consecutive ids go to consecutive XCDs instead
        id = b;
    } else {
        const int q = nwg / NXCD, r = nwg % NXCD;
        const int xcd = b % NXCD, within = b / NXCD;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    constexpr int GROUP = 8;
    const int band = id / (GROUP * tiles_n);
    const int first_m = band * GROUP;
    const int rows = min(GROUP, tiles_m - first_m);
    const int in_band = id - band * GROUP * tiles_n;
    tm = first_m + in_band % rows;
    tn = in_band / rows;
}

// lower-triangle enumeration for square tile grids (BM == BN): the launch holds exactly T (T + 1) / 2
// workgroups, every XCD gets a contiguous, equally long run of the row-major triangle.  (With the
// rectangular mapping + early exit the XCD that owns the bottom tile rows does several times the work of
// the one that owns the top rows, and the launch takes as long as the full square.)
__device__ inline void block_to_tile_tri(int tiles, bool spread, int& tm, int& tn) {
    const int nwg = tiles * (tiles + 1) / 2;
    const int b = blockIdx.x;
    constexpr int NXCD = 8;
    const int q = nwg / NXCD, r = nwg % NXCD;
    const int xcd = b % NXCD, within = b / NXCD;
    const int id = spread ? b : (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    int row = (int)((sqrtf(8.0f * (float)id + 1.0f) - 1.0f) * 0.5f);
    while ((row + 1) * (row + 2) / 2 <= id) ++row;
    while (row * (row + 1) / 2 > id) --row;
    tm = row;
    tn = id - row * (row + 1) / 2;
}

// TAG gives the kernel a distinct symbol (TAG 4 also a schedule: see launch()): TAG 1 = the tile-level trailing update issued by
// npw_dgemm_nt_sub (kernels.syrk), so profilers report it separately from the many smaller GEMMs that
// trsm / potrf / geqrt run through the same tiling; TAG 2 = its symmetric form (X == Y, lower tiles only).
// NW = waves along n: 2 (4 waves as 2 x 2, two workgroups per CU) or 4 (8 waves as 2 x 4, ONE workgroup per CU holding a
// 128 x 256 tile: the same waves per CU, but an A panel is fetched once for 256 output columns -- for products whose n is
// exactly two 128-wide tile columns, where the two workgroups that share an A panel drift apart in their k loops and both
// fetch it from memory: QR's far V^T W product read its operands 2.1 x, profiles/r04_qr32r_hbm_bytes.txt).
template <typename T, int BM, int BN, int BK, bool A_KC, bool B_KC, bool EDGE, int TAG = 0, int NW = 2>
__global__ __launch_bounds__(128 * NW, 2) void gemm_kernel(const GemmParams<T> p_in) {
    constexpr int NT = 128 * NW;         // threads per workgroup
    constexpr int WAVES_N = NW;
    GemmParams<T> p = p_in;
    if (gridDim.z > 1 && p.use_delta) {
        // (indexed in the kernel-argument segment itself: a dynamically indexed array inside the local copy `p` would
        //  push the whole parameter block out of the scalar registers -- the trailing update lost 8 % that way)
        const int bz = blockIdx.z;
        p.A += p_in.da[bz];
        p.B += p_in.db[bz];
        if (p.C) p.C += p_in.dc[bz];
        p.D += p_in.dd[bz];
        if (p.skip0) p.skip0 += p_in.ds0[bz];
        if (p.skip1) p.skip1 += p_in.ds1[bz];
    } else if (gridDim.z > 1) {
        int64_t bz = blockIdx.z, b2 = 0;
        if (p.batch_inner > 0) {
            b2 = bz % p.batch_inner;
            bz = bz / p.batch_inner;
        }
        p.A += bz * p.batch_a + b2 * p.batch2_a;
        p.B += bz * p.batch_b + b2 * p.batch2_b;
        if (p.C) p.C += bz * p.batch_c + b2 * p.batch2_c;
        p.D += bz * p.batch_d + b2 * p.batch2_d;
    }
    using TR = MfmaTraits<T>;
    using acc_t = typename TR::acc_t;
    using vec_t = typename TR::vec_t;
    constexpr int VEC = TR::VEC;
    constexpr int WM = BM / 2, WN = BN / WAVES_N;
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr int KSTEPS = BK / 4;
    constexpr int LDKC = BK + TR::PADK;
    // MC row stride: VEC-step fragments read rows VEC apart in neighbouring lane groups; + 16 B / + 32 B of padding
    // (fp32 / fp64) puts them on disjoint banks
    constexpr int LDA_MC = BM + TR::PAD_MC, LDB_MC = BN + TR::PAD_MC;
    constexpr int A_ELEMS = A_KC ? BM * LDKC : BK * LDA_MC;
    constexpr int B_ELEMS = B_KC ? BN * LDKC : BK * LDB_MC;
    constexpr int A_CHUNKS = BM * BK / VEC / NT;
    constexpr int B_CHUNKS = BN * BK / VEC / NT;
    static_assert(A_CHUNKS >= 1 && B_CHUNKS >= 1, "tile too small for the workgroup");
    static_assert(BK % 8 == 0 && KSTEPS % VEC == 0, "BK");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);
    // stage `buf` = [A tile | B tile] at element offset buf * STAGE.  Everything below indexes `smem`
    // directly with integer offsets: going through an array of per-stage pointers makes hipcc lose the
    // LDS address space and emit flat_load / flat_store (+ 64-bit address arithmetic) for every fragment.
    constexpr int STAGE = A_ELEMS + B_ELEMS;

    int tile_m = 0, tile_n = 0;
    // (only when the launch takes several rounds of workgroups: within one round everybody is resident anyway and the
    //  contiguous map's L2 locality is worth more)
    const bool spread = (p.b_lower_tri || p.k_from_diag || p.a_upper_tri) && (gridDim.x * gridDim.y * gridDim.z > 1024u);
    int diag_part = -1;   // >= 0: this workgroup is a k chunk of a diagonal block (tag 2 with diag_ws), not a tile pair
    if constexpr (TAG == 2) {
        const int ntri = p.tiles_m * (p.tiles_m - 1) / 2;
        if ((int)blockIdx.x >= ntri) {
            const int e = (int)blockIdx.x - ntri;
            tile_m = tile_n = e / p.diag_split;
            diag_part = e - tile_m * p.diag_split;
        }
    }
    if (diag_part >= 0) {
    } else if (p.lower_only == 2) {
        block_to_tile_tri(p.tiles_m, spread, tile_m, tile_n);
    } else if (p.lower_only == 3) {  // strictly lower tiles: the triangle of order tiles - 1, one row down
        block_to_tile_tri(p.tiles_m - 1, spread, tile_m, tile_n);
        tile_m += 1;
    } else if (p.b_blockdiag > 0) {
        // block-diagonal triangular B: the tiles' k ranges are BN, 2 BN, ... b_blockdiag in every diagonal block.  Block
        // ids in order of falling length -- class 0 = the last column tile of every block, all tile rows, then class 1 ...
        // -- so the dispatcher, which starts workgroups in id order as slots free up, does longest-first list scheduling:
        // a slot that ran a long tile picks up a short one (1024 tiles on 512 slots: 1024 + 128 = 896 + 256 = ...)
        const int tpg = p.b_blockdiag / BN;
        const int per_class = p.tiles_m * (p.tiles_n / tpg);
        const int b = (int)blockIdx.x;
        const int cls = b / per_class, rem = b - cls * per_class;
        tile_m = rem % p.tiles_m;
        tile_n = (rem / p.tiles_m) * tpg + (tpg - 1 - cls);
    } else {
        block_to_tile(p.tiles_m, p.tiles_n, spread, tile_m, tile_n);
        // triangular B: the k range grows with the tile column -- longest tiles first, the short ones fill the tail
        if (p.b_lower_tri) tile_n = p.tiles_n - 1 - tile_n;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    if (p.lower_only && n0 > m0 + BM - 1) return;  // tile entirely above the diagonal

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    const int li = lane & 15, lg = lane >> 4;

    // split-K: blockIdx.y selects the k range [kb, kend) and its own partial output
    int kb = p.k_chunk > 0 ? (int)blockIdx.y * p.k_chunk : 0;
    if (p.k_from_diag) kb = max(kb, (max(m0, n0) / BK) * BK);  // V^T V with V lower trapezoidal: rows above are zero
    if (p.a_upper_tri) kb = max(kb, (m0 / BK) * BK);           // T1 * X with T1 upper triangular: columns left of the diagonal are zero
    int kend = p.k_chunk > 0 ? min(p.K, kb + p.k_chunk) : p.K;
    if (p.b_lower_tri) kend = min(kend, n0 + BN);  // rows of W^T below the diagonal block are zero
    if (p.b_blockdiag > 0) {
        // W's block g = n0 / width sits at B + g width^2 with ld = width: element (n, k) of the virtual n x n matrix is at
        // B + n ld + (k - g width), and the block only has k in [g width, n0 + BN)
        const int g0 = (n0 / p.b_blockdiag) * p.b_blockdiag;
        kb = max(kb, g0);
        kend = min(kend, n0 + BN);
        p.B -= g0;
    } else if (p.b_blockdiag < 0) {
        // the same W on the left: element (r, k) at A + r ld + (k - g width), k in [g width, m0 + BM); tiles right of the
        // block's own column are nobody's business
        const int w = -p.b_blockdiag;
        const int g0 = (m0 / w) * w;
        if (n0 >= g0 + w) return;
        kb = max(kb, g0);
        kend = min(kend, m0 + BM);
        p.A -= g0;
    }
    if constexpr (TAG == 2) {
        if (diag_part >= 0) {
            // raw partial product of the block's k chunk into the scratch: no alpha, no C, 128 x 128 with ld = BN
            kb = diag_part * p.diag_chunk;
            kend = min(p.K, kb + p.diag_chunk);
            p.alpha = T(1);
            p.beta = T(0);
            p.ldd = BN;
            p.D = p.diag_ws + (((int64_t)blockIdx.z * p.tiles_m + tile_m) * p.diag_split + diag_part) * (int64_t)(BM * BN) -
                  ((int64_t)m0 * BN + n0);
        }
    }
    int nk = (kend - kb + BK - 1) / BK;
    if ((p.skip0 && *p.skip0) || (p.skip1 && *p.skip1)) nk = 0;

    acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0;

    vec_t ra[A_CHUNKS], rb[B_CHUNKS];

    // ---- global -> registers ------------------------------------------------------------
    auto load_tiles = [&](int kt) {
        const int k0 = kb + kt * BK;
#pragma unroll
        for (int c = 0; c < A_CHUNKS; ++c) {
            const int id = tid + NT * c;
            if constexpr (A_KC) {  // A stored M x K
                const int row = id / (BK / VEC), kc = (id % (BK / VEC)) * VEC;
                const T* src = p.A + (int64_t)(m0 + row) * p.lda + (k0 + kc);
                if constexpr (!EDGE) {
                    ra[c] = *reinterpret_cast<const vec_t*>(src);
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        ra[c][v] = (m0 + row < p.M && k0 + kc + v < kend) ? src[v] : T(0);
                }
            } else {  // A stored K x M
                const int kk = id / (BM / VEC), mc = (id % (BM / VEC)) * VEC;
                const T* src = p.A + (int64_t)(k0 + kk) * p.lda + (m0 + mc);
                if constexpr (!EDGE) {
                    ra[c] = *reinterpret_cast<const vec_t*>(src);
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        ra[c][v] = (k0 + kk < kend && m0 + mc + v < p.M) ? src[v] : T(0);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < B_CHUNKS; ++c) {
            const int id = tid + NT * c;
            if constexpr (B_KC) {  // B stored N x K
                const int row = id / (BK / VEC), kc = (id % (BK / VEC)) * VEC;
                const T* src = p.B + (int64_t)(n0 + row) * p.ldb + (k0 + kc);
                if constexpr (!EDGE) {
                    rb[c] = *reinterpret_cast<const vec_t*>(src);
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        rb[c][v] = (n0 + row < p.N && k0 + kc + v < kend) ? src[v] : T(0);
                }
            } else {  // B stored K x N
                const int kk = id / (BN / VEC), nc = (id % (BN / VEC)) * VEC;
                const T* src = p.B + (int64_t)(k0 + kk) * p.ldb + (n0 + nc);
                if constexpr (!EDGE) {
                    rb[c] = *reinterpret_cast<const vec_t*>(src);
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        rb[c][v] = (k0 + kk < kend && n0 + nc + v < p.N) ? src[v] : T(0);
                }
            }
        }
    };

    // ---- registers -> LDS -----------------------------------------------------------------
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int c = 0; c < A_CHUNKS; ++c) {
            const int id = tid + NT * c;
            if constexpr (A_KC) {
                const int row = id / (BK / VEC), kc = (id % (BK / VEC)) * VEC;
                *reinterpret_cast<vec_t*>(&smem[buf * STAGE + kc_off<T, BK>(row, kc)]) = ra[c];
            } else {
                const int kk = id / (BM / VEC), mc = (id % (BM / VEC)) * VEC;
                *reinterpret_cast<vec_t*>(&smem[buf * STAGE + kk * LDA_MC + mc]) = ra[c];
            }
        }
#pragma unroll
        for (int c = 0; c < B_CHUNKS; ++c) {
            const int id = tid + NT * c;
            if constexpr (B_KC) {
                const int row = id / (BK / VEC), kc = (id % (BK / VEC)) * VEC;
                *reinterpret_cast<vec_t*>(&smem[buf * STAGE + A_ELEMS + kc_off<T, BK>(row, kc)]) = rb[c];
            } else {
                const int kk = id / (BN / VEC), nc = (id % (BN / VEC)) * VEC;
                *reinterpret_cast<vec_t*>(&smem[buf * STAGE + A_ELEMS + kk * LDB_MC + nc]) = rb[c];
            }
        }
    };

    // ---- LDS -> fragments -> MFMA ----------------------------------------------------------
    auto compute = [&](int buf) {
        const T* a_s = smem + buf * STAGE;
        const T* b_s = smem + buf * STAGE + A_ELEMS;
        // VEC MFMA steps per fragment load: their k are adjacent (one 16-byte LDS read for KC operands)
#pragma unroll
        for (int s = 0; s < KSTEPS; s += VEC) {
            const int kk = k_of<T>(s, lg);
            vec_t af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (A_KC) {
                    af[i] = *reinterpret_cast<const vec_t*>(&a_s[kc_off<T, BK>(wm0 + 16 * i + li, kk)]);
                } else {
#pragma unroll
                    for (int h = 0; h < VEC; ++h) af[i][h] = a_s[(kk + h) * LDA_MC + wm0 + 16 * i + li];
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (B_KC) {
                    bf[j] = *reinterpret_cast<const vec_t*>(&b_s[kc_off<T, BK>(wn0 + 16 * j + li, kk)]);
                } else {
#pragma unroll
                    for (int h = 0; h < VEC; ++h) bf[j][h] = b_s[(kk + h) * LDB_MC + wn0 + 16 * j + li];
                }
            }
#pragma unroll
            for (int h = 0; h < VEC; ++h)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = TR::mfma(af[i][h], bf[j][h], acc[i][j]);
        }
    };

    if (nk > 0) {
        load_tiles(0);
        store_tiles(0);
        __syncthreads();
        if constexpr ((A_KC && B_KC && BM == 128 && BN == 128) || TAG == 4) {
            // The LDS stores of the next k-tile are spread between the MFMAs of the LAST step group instead of leaving
            // as a burst in front of the barrier.  Measured on the 4096^3 trailing update (tools/syrk_time.py): the
            // loop without the stores runs at 74.7 TFLOP/s -- the bare-MFMA ceiling; the barrier costs nothing -- with
            // the burst at 68.5, because 32 back-to-back ds_write_b128 per workgroup keep the LDS busy for ~400 cycles
            // during which the OTHER workgroup's fragment reads (and with them its MFMAs) wait, and the storing
            // workgroup sits at its barrier and cannot fill in.  One store per few MFMAs hides in the matrix pipe's
            // 64-cycle shadow: 67.3 -> 69.0 TFLOP/s.  (Direct-to-LDS loads, global_load_lds_dwordx4, were tried
            // instead of the register staging: +1.4 % only -- the LDS array is the contended resource, not the VGPR
            // path -- and 8 waves per workgroup changed nothing: latency hiding is not what is missing.)
            // The global loads of the next k-tile get the same treatment in the FIRST half of the MFMAs (0x020 = VMEM
            // read): 69.9 -> 71.5 TFLOP/s.  Spreading the second group's fragment reads as well gains nothing.
            // The last iteration is peeled so that loads, MFMAs and stores of the others share one basic block, and
            // the issue order is pinned with sched_group_barrier (0x100 = LDS read, 0x008 = MFMA, 0x200 = LDS write, 0x020 = VMEM read).
            // (Only for the 128 x 128 tiling -- long k loops, two workgroups per CU.  The small tilings run grids of
            //  well under two workgroups per CU where pinning the order costs more than the burst: trsm 1.29 -> 1.51 ms.)
            constexpr int GROUPS = KSTEPS / VEC;               // fragment loads per k-tile (VEC MFMA steps each)
            constexpr int MF = VEC * TM * TN;                  // MFMAs per group
            constexpr int NST = A_CHUNKS + B_CHUNKS;           // 16-byte LDS stores per thread and k-tile
            constexpr int HALF = GROUPS * MF / 2;              // the stores go between the MFMAs of the second half
            constexpr int PER = HALF / NST;
            static_assert(PER >= 1 && GROUPS <= 2, "store interleave: unexpected tile geometry");
            // LDS reads per fragment group: one 16-byte read per fragment for k-contiguous operands, VEC 8-byte reads otherwise
            constexpr int FRAGS = (A_KC ? TM : TM * VEC) + (B_KC ? TN : TN * VEC);
            for (int kt = 0; kt + 1 < nk; ++kt) {
                const int cur = kt & 1;
                load_tiles(kt + 1);
                compute(cur);
                store_tiles(cur ^ 1);
                __builtin_amdgcn_sched_group_barrier(0x100, FRAGS, 0);       // fragments of the first group
                // first half of the MFMAs, the global loads of the next k-tile between them (one per PER MFMAs: as a burst
                // at the top of the iteration they cost 1.5 %; spread, the wave's memory instructions never queue)
#pragma unroll
                for (int g = 0; g < NST; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                if constexpr (GROUPS == 2) __builtin_amdgcn_sched_group_barrier(0x100, FRAGS, 0);
#pragma unroll
                for (int g = 0; g < NST; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
                __syncthreads();
            }
            compute((nk - 1) & 1);
            __syncthreads();
        } else {
            for (int kt = 0; kt < nk; ++kt) {
                const int cur = kt & 1;
                if (kt + 1 < nk) load_tiles(kt + 1);
                compute(cur);
                if (kt + 1 < nk) store_tiles(cur ^ 1);
                __syncthreads();
            }
        }
    }

    // ---- epilogue: D = alpha * acc + beta * C ------------------------------------------------
    const T alpha = p.alpha, beta = p.beta;
    T* const Dp = p.D + (int64_t)blockIdx.y * p.split_stride;

    // ---- symmetric product (TAG 2, X == Y): a strictly-lower tile also writes its mirror tile --------------------
    // acc(r, c) = sum_k X[m0 + r, k] X[n0 + c, k] is bit for bit what the workgroup of tile (tile_n, tile_m) would have
    // summed for element (c, r) (same products, same order), so  D[n0 + c, m0 + r] = beta * C[n0 + c, m0 + r] +
    // alpha * acc(r, c)  is exactly the full product's upper tile for ANY C -- symmetric or not.  The accumulators
    // go through LDS (free after the k loop's last barrier) 16 rows at a time so that both the read of C and the
    // store of D are 128-byte row segments.  Row block i is finished (both tiles) before i + 1 starts, so the
    // accumulator registers retire as the epilogue proceeds.
    if constexpr (TAG == 2) {
        constexpr int LDT = 17;                      // odd stride: ds_write_b64 of a 16 x 16 block is conflict-free
        T* stage = smem + wave * (64 * LDT);         // 64 x 16 transposed block per wave
        const bool mirror = (tile_m != tile_n);
        const bool has_c = (beta != T(0));
        const int64_t urow0 = n0 + wn0, ucol0 = m0 + wm0;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            T cv[TN][4];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) cv[j][r] = T(0);
            if (has_c) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        cv[j][r] = p.C[(int64_t)(m0 + wm0 + 16 * i + TR::acc_row(lg, r)) * p.ldc + n0 + wn0 + 16 * j + li];
            }
            if (mirror) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) stage[(16 * j + li) * LDT + TR::acc_row(lg, r)] = acc[i][j][r];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Dp[(int64_t)(m0 + wm0 + 16 * i + TR::acc_row(lg, r)) * p.ldd + n0 + wn0 + 16 * j + li] =
                        fma(beta, cv[j][r], alpha * acc[i][j][r]);
            if (mirror) {
                __syncthreads();
                const T* crow = p.C + (urow0 + lg) * p.ldc + ucol0 + 16 * i + li;
                T* drow = Dp + (urow0 + lg) * p.ldd + ucol0 + 16 * i + li;
#pragma unroll
                for (int h = 0; h < 2; ++h) {        // two batches of 8 rows: the loads of a batch are in flight together
                    T av[8], uv[8];
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        av[it] = stage[(4 * (8 * h + it) + lg) * LDT + li];
                        uv[it] = T(0);
                    }
                    if (has_c) {
#pragma unroll
                        for (int it = 0; it < 8; ++it) uv[it] = crow[(int64_t)(4 * (8 * h + it)) * p.ldc];
                    }
#pragma unroll
                    for (int it = 0; it < 8; ++it)
                        drow[(int64_t)(4 * (8 * h + it)) * p.ldd] = fma(beta, uv[it], alpha * av[it]);
                }
                __syncthreads();
            }
        }
        return;
    }

    // (the beta test is hoisted out of the unrolled loops: a per-element "load or not" select
    //  makes hipcc branch around and wait for every single load)
    if (beta != T(0)) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            T cv[TN][4];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn0 + 16 * j + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm0 + 16 * i + TR::acc_row(lg, r);
                    const bool ok = !EDGE || (row < p.M && col < p.N);
                    cv[j][r] = ok ? p.C[(int64_t)row * p.ldc + col] : T(0);
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn0 + 16 * j + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm0 + 16 * i + TR::acc_row(lg, r);
                    if (EDGE && (row >= p.M || col >= p.N)) continue;
                    Dp[(int64_t)row * p.ldd + col] = fma(beta, cv[j][r], alpha * acc[i][j][r]);
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn0 + 16 * j + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm0 + 16 * i + TR::acc_row(lg, r);
                    if (EDGE && (row >= p.M || col >= p.N)) continue;
                    Dp[(int64_t)row * p.ldd + col] = alpha * acc[i][j][r];
                }
            }
        }
    }
}

template <typename T, int BM, int BN, int BK, bool A_KC, bool B_KC, bool EDGE, int NW = 2>
int launch(const GemmParams<T>& p, hipStream_t stream) {
    using TR = MfmaTraits<T>;
    constexpr int LDKC = BK + TR::PADK;
    constexpr int A_ELEMS = A_KC ? BM * LDKC : BK * (BM + TR::PAD_MC);
    constexpr int B_ELEMS = B_KC ? BN * LDKC : BK * (BN + TR::PAD_MC);
    constexpr size_t smem = 2 * (A_ELEMS + B_ELEMS) * sizeof(T);
    // lower_only == 2: only the tiles touching the lower triangle are launched (see block_to_tile_tri)
    const int nwg = (p.lower_only == 2)   ? p.tiles_m * (p.tiles_m + 1) / 2
                    : (p.lower_only == 3) ? p.tiles_m * (p.tiles_m - 1) / 2
                                          : p.tiles_m * p.tiles_n;
    if (nwg == 0) return NPW_OK;
    const unsigned nbatch = (unsigned)p.tiles_batch;
    const int nsplit = p.k_chunk > 0 ? (p.K + p.k_chunk - 1) / p.k_chunk : 1;
    if constexpr (sizeof(T) == 8 && BM == 128 && BN == 128 && A_KC && B_KC && !EDGE) {
        if (p.tag == 1 || p.tag == 2) {
            auto tagged = (p.tag == 1) ? gemm_kernel<T, BM, BN, BK, A_KC, B_KC, EDGE, 1>
                                       : gemm_kernel<T, BM, BN, BK, A_KC, B_KC, EDGE, 2>;
            static thread_local bool tagged_attr[3] = {false, false, false};
            if (!tagged_attr[p.tag]) {
                NPW_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tagged),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
                tagged_attr[p.tag] = true;
            }
            // (tag 2 with diag_ws: the diagonal blocks' k chunks ride behind the tile pairs)
            const int extra = (p.tag == 2 && p.diag_ws != nullptr) ? p.tiles_m * p.diag_split : 0;
            hipLaunchKernelGGL(tagged, dim3(nwg + extra, nsplit, nbatch), dim3(256), smem, stream, p);
            NPW_LAUNCH_CHECK();
            return NPW_OK;
        }
    }
    if constexpr (sizeof(T) == 8 && NW == 4 && BN == 256 && !A_KC && !B_KC && !EDGE) {
        // TAG 4 = the same kernel with the pinned load / store interleave of the 128 x 128 N / T tiling (its fragment groups are
        // 2 x (TM + TN) 8-byte LDS reads instead of TM + TN 16-byte ones).  Round 6, same-box A/B on a batch of 32 factorisations:
        // every launch on one stream 91.8 -> 90.3 ms (the product itself ~7 % faster), four streams 76.0 / 76.4 -> 75.7 / 75.9
        // R only, 103.2 / 103.7 -> 102.6 / 102.7 with T; the same sums in the same order.  $NPW_GEMM_WIDE_PINNED=0: the plain loop.
        static const bool pinned = [] { const char* e = getenv("NPW_GEMM_WIDE_PINNED"); return e == nullptr || atoi(e) != 0; }();
        if (pinned) {
            auto k4 = gemm_kernel<T, BM, BN, BK, A_KC, B_KC, EDGE, 4, NW>;
            static thread_local bool a4 = false;
            if (!a4) {
                NPW_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
                a4 = true;
            }
            hipLaunchKernelGGL(k4, dim3(nwg, nsplit, nbatch), dim3(128 * NW), smem, stream, p);
            NPW_LAUNCH_CHECK();
            return NPW_OK;
        }
    }
    auto kern = gemm_kernel<T, BM, BN, BK, A_KC, B_KC, EDGE, 0, NW>;
    static thread_local bool attr_set = false;
    if (!attr_set && smem > 48 * 1024) {
        NPW_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(nwg, nsplit, nbatch), dim3(128 * NW), smem, stream, p);
    NPW_LAUNCH_CHECK();
    return NPW_OK;
}

template <typename T, int BM, int BN, int BK, bool EDGE, int NW = 2>
int dispatch_layout(bool a_kc, bool b_kc, const GemmParams<T>& p, hipStream_t s) {
    if (a_kc && b_kc) return launch<T, BM, BN, BK, true, true, EDGE, NW>(p, s);
    if (a_kc && !b_kc) return launch<T, BM, BN, BK, true, false, EDGE, NW>(p, s);
    if (!a_kc && b_kc) return launch<T, BM, BN, BK, false, true, EDGE, NW>(p, s);
    return launch<T, BM, BN, BK, false, false, EDGE, NW>(p, s);
}

// D = alpha * sum_s P[s] + beta * C over the split-K partial products P[s] (each m x n, contiguous)
template <typename T>
__global__ void splitk_reduce_kernel(int nsplit, const T* P, int64_t stride, int64_t m, int64_t n, T alpha, T beta,
                                     const T* C, int64_t ldc, T* D, int64_t ldd, int64_t batch_c, int64_t batch_d) {
    // blockIdx.z = problem of a batch: its partials are contiguous (nsplit * stride), C / D move by the batch strides
    P += (int64_t)blockIdx.z * nsplit * stride;
    if (C) C += (int64_t)blockIdx.z * batch_c;
    D += (int64_t)blockIdx.z * batch_d;
    for (int64_t r = blockIdx.y; r < m; r += gridDim.y)
        for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (int64_t)gridDim.x * blockDim.x) {
            T acc = T(0);
            for (int sidx = 0; sidx < nsplit; ++sidx) acc += P[sidx * stride + r * n + c];  // fixed order: deterministic
            T v = alpha * acc;
            if (beta != T(0)) v = fma(beta, C[r * ldc + c], v);
            D[r * ldd + c] = v;
        }
}

}  // namespace

template <typename T>
int gemm(char transA, char transB, int64_t m, int64_t n, int64_t k, T alpha, const T* A,
         int64_t lda, const T* B, int64_t ldb, T beta, const T* C, int64_t ldc, T* D, int64_t ldd,
         const GemmOpts& opts, hipStream_t stream) {
    const bool ta = (transA == 'T' || transA == 't');
    const bool tb = (transB == 'T' || transB == 't');
    NPW_REQUIRE(ta || transA == 'N' || transA == 'n', "gemm: bad transA '%c'", transA);
    NPW_REQUIRE(tb || transB == 'N' || transB == 'n', "gemm: bad transB '%c'", transB);
    NPW_REQUIRE(m >= 0 && n >= 0 && k >= 0, "gemm: negative dimension");
    NPW_REQUIRE(m < (1LL << 30) && n < (1LL << 30) && k < (1LL << 30), "gemm: dimension too large");
    if (m == 0 || n == 0) return NPW_OK;
    NPW_REQUIRE(D != nullptr, "gemm: D is NULL");
    NPW_REQUIRE(beta == T(0) || C != nullptr, "gemm: C is NULL with beta != 0");
    NPW_REQUIRE(k == 0 || (A != nullptr && B != nullptr), "gemm: A/B NULL");
    // (a block diagonal operand -- b_blockdiag / a_blockdiag -- is stored as its diagonal blocks back to back: ld = block width)
    NPW_REQUIRE((lda >= (ta ? m : k) || opts.a_blockdiag > 0) && (ldb >= (tb ? k : n) || opts.b_blockdiag > 0) && ldd >= n && (beta == T(0) || ldc >= n),
                "gemm: leading dimension too small (m=%lld n=%lld k=%lld lda=%lld ldb=%lld ldc=%lld "
                "ldd=%lld)",
                (long long)m, (long long)n, (long long)k, (long long)lda, (long long)ldb,
                (long long)ldc, (long long)ldd);

    // split-K for skinny outputs with a long contraction (e.g. V_p^T W in the QR update): the k range is
    // cut into chunks that run as independent workgroups into partial products, summed in a fixed order.
    if (opts.splitk > 1 && opts.splitk_ws != nullptr && opts.k_chunk_ == 0 && k >= 128) {
        const int64_t chunk = ceil_div(ceil_div(k, (int64_t)opts.splitk), 32) * 32;
        const int nsplit = (int)ceil_div(k, chunk);
        if (nsplit > 1) {
            GemmOpts inner = opts;
            inner.splitk = 1;
            inner.k_chunk_ = (int)chunk;
            inner.lower_only = false;  // the reduction reads every element of the partial products
            inner.strict_lower = false;
            inner.batch_c = inner.batch2_c = 0;
            inner.batch_d = (int64_t)nsplit * m * n;  // partials of one problem are contiguous
            inner.batch2_d = 0;
            NPW_REQUIRE(opts.batch_inner == 0, "gemm: split-K with a two-level batch is not supported");
            T* P = static_cast<T*>(opts.splitk_ws);
            inner.splitk_keep = nullptr;
            int rc = gemm<T>(transA, transB, m, n, k, T(1), A, lda, B, ldb, T(0), nullptr, 0, P, n, inner, stream);
            if (rc) return rc;
            if (opts.splitk_keep != nullptr) {
                *opts.splitk_keep = nsplit;
                return NPW_OK;
            }
            const unsigned gx = (unsigned)(ceil_div(n, 256) > 16 ? 16 : ceil_div(n, 256));
            const unsigned gy = (unsigned)(m > 1024 ? 1024 : m);
            hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3(gx, gy, (unsigned)opts.batch), dim3(256), 0, stream, nsplit, P,
                               m * n, m, n, alpha, beta, C, ldc, D, ldd, opts.batch_c, opts.batch_d);
            NPW_LAUNCH_CHECK();
            return NPW_OK;
        }
    }

    if (opts.splitk_keep != nullptr && opts.k_chunk_ == 0) {
        // the caller reduces the partial products itself: without a split there is exactly one, the plain product
        NPW_REQUIRE(opts.splitk_ws != nullptr, "gemm: splitk_keep without scratch");
        GemmOpts one = opts;
        one.splitk_keep = nullptr;
        one.splitk = 1;
        one.batch_c = one.batch2_c = 0;
        one.batch_d = m * n;
        one.batch2_d = 0;
        *opts.splitk_keep = 1;
        return gemm<T>(transA, transB, m, n, k, T(1), A, lda, B, ldb, T(0), nullptr, 0, static_cast<T*>(opts.splitk_ws), n, one,
                       stream);
    }

    GemmParams<T> p;
    p.A = A;
    p.B = B;
    p.C = C;
    p.D = D;
    p.lda = lda;
    p.ldb = ldb;
    p.ldc = ldc;
    p.ldd = ldd;
    p.M = (int)m;
    p.N = (int)n;
    p.K = (int)k;
    p.alpha = alpha;
    p.beta = beta;
    p.skip0 = opts.skip0;
    p.skip1 = opts.skip1;
    p.lower_only = opts.lower_only ? (m == n && !opts.inplace_a ? (opts.strict_lower ? 3 : 2) : 1) : 0;
    NPW_REQUIRE(!opts.strict_lower || p.lower_only == 3, "gemm: strict_lower needs a square lower_only product");
    NPW_REQUIRE(opts.batch >= 1, "gemm: bad batch");
    p.tiles_batch = opts.batch;
    p.batch_a = opts.batch_a;
    p.batch_b = opts.batch_b;
    p.batch_c = opts.batch_c;
    p.batch_d = opts.batch_d;
    p.batch_inner = opts.batch_inner;
    p.batch2_a = opts.batch2_a;
    p.batch2_b = opts.batch2_b;
    p.batch2_c = opts.batch2_c;
    p.batch2_d = opts.batch2_d;
    p.b_lower_tri = opts.b_lower_tri ? 1 : 0;
    p.b_blockdiag = opts.b_blockdiag > 0 ? opts.b_blockdiag : -opts.a_blockdiag;
    NPW_REQUIRE(opts.a_blockdiag == 0 || (opts.b_blockdiag == 0 && !ta && opts.a_blockdiag % 128 == 0 && m % opts.a_blockdiag == 0 && k == m &&
                                          lda == opts.a_blockdiag && opts.force_big && !opts.lower_only && opts.k_chunk_ == 0 && n % 128 == 0),
                "gemm: a_blockdiag needs op(A) = N, whole 128-column tiles, m == k a multiple of the block width, lda == width and force_big");
    p.pad_ = 0;
    NPW_REQUIRE(opts.b_blockdiag == 0 || (tb && !ta && opts.b_blockdiag % 128 == 0 && n % opts.b_blockdiag == 0 && k == n &&
                                          ldb == opts.b_blockdiag && opts.force_big && !opts.lower_only && opts.k_chunk_ == 0 && m % 128 == 0),
                "gemm: b_blockdiag needs op(A) = N, op(B) = T, whole 128-row tiles, n == k a multiple of the block width, ldb == width and force_big");
    p.k_from_diag = opts.k_from_diag ? 1 : 0;
    p.a_upper_tri = opts.a_upper_tri ? 1 : 0;
    p.use_delta = 0;
    if (opts.delta_a || opts.delta_b || opts.delta_c || opts.delta_d) {
        NPW_REQUIRE(opts.batch <= 16 && opts.batch_inner == 0, "gemm: an irregular batch holds at most 16 problems");
        for (int z = 0; z < opts.batch; ++z)   // the vectorised tilings need every problem's rows 16-byte aligned
            NPW_REQUIRE((!opts.delta_a || opts.delta_a[z] % (16 / (int)sizeof(T)) == 0) &&
                            (!opts.delta_b || opts.delta_b[z] % (16 / (int)sizeof(T)) == 0),
                        "gemm: irregular batch offsets must keep 16-byte alignment");
        p.use_delta = 1;
        for (int z = 0; z < 16; ++z) {
            const bool in = z < opts.batch;
            p.da[z] = (in && opts.delta_a) ? opts.delta_a[z] : (in ? (int64_t)z * opts.batch_a : 0);
            p.db[z] = (in && opts.delta_b) ? opts.delta_b[z] : (in ? (int64_t)z * opts.batch_b : 0);
            p.dc[z] = (in && opts.delta_c) ? opts.delta_c[z] : (in ? (int64_t)z * opts.batch_c : 0);
            p.dd[z] = (in && opts.delta_d) ? opts.delta_d[z] : (in ? (int64_t)z * opts.batch_d : 0);
            p.ds0[z] = (in && opts.delta_skip0) ? opts.delta_skip0[z] : 0;
            p.ds1[z] = (in && opts.delta_skip1) ? opts.delta_skip1[z] : 0;
        }
    }
    p.tag = opts.tag;
    p.diag_ws = nullptr;
    p.diag_split = p.diag_chunk = 0;
    if (opts.diag_ws != nullptr && opts.diag_split > 1) {
        NPW_REQUIRE(opts.tag == 2 && opts.strict_lower && opts.k_chunk_ == 0 && k % (16 * opts.diag_split) == 0,
                    "gemm: diag_ws goes with the symmetric (tag 2) product and k a multiple of 16 * diag_split");
        p.diag_ws = static_cast<T*>(opts.diag_ws);
        p.diag_split = opts.diag_split;
        p.diag_chunk = (int)(k / opts.diag_split);
    }
    p.k_chunk = opts.k_chunk_;
    p.split_stride = opts.k_chunk_ > 0 ? m * n : 0;

    const bool a_kc = !ta;  // A stored M x K  => k contiguous
    const bool b_kc = tb;   // B stored N x K  => k contiguous
    constexpr int VEC = 16 / (int)sizeof(T);
    auto aligned16 = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; };
    const bool vec_ok = aligned16(A) && aligned16(B) && (lda % VEC == 0) && (ldb % VEC == 0);
    // k-tile depth: 16 for the 128 x 128 tiling (2 workgroups per CU hide each other's barrier and
    // load latency), 32 for the small tilings, whose grids are often far below one workgroup per CU:
    // there every k-tile is a dependent global-memory round trip, so fewer, deeper k-tiles win.
    constexpr int BK = 16, BKS = 32;
    const bool k_ok = (k % BK == 0), ks_ok = (k % BKS == 0);

    if (opts.inplace_a) {
        // D aliases A: legal because with a single tile column every workgroup reads only the rows
        // it later writes, and its reads complete (k-loop) before its epilogue stores.
        NPW_REQUIRE(!ta && n <= 128 && k <= 128 && (void*)D == (void*)A && ldd == lda,
                    "gemm: inplace_a needs op(A)=N, n,k <= 128 and D == A");
        p.tiles_m = (int)ceil_div(m, 64);
        p.tiles_n = 1;
        const bool full = vec_ok && ks_ok && (m % 64 == 0) && (n == 128);
        if (full) return dispatch_layout<T, 64, 128, BKS, false>(a_kc, b_kc, p, stream);
        return dispatch_layout<T, 64, 128, BKS, true>(a_kc, b_kc, p, stream);
    }
    // tile selection: 128x128 when it fills the chip (or the problem is large), else 64x64
    const int64_t t128 = ceil_div(m, 128);
    // (per problem, also for batches: counting the batch in moved trsm's batched 512-wide leaves to 128 x 128 tiles, whose
    //  coarser k-limits cost more than the fuller grid gains: 1.21 -> 1.41 ms per right-hand side)
    const int64_t wg128 = (p.lower_only >= 2) ? t128 * (t128 + 1) / 2 : t128 * ceil_div(n, 128);
    static const int64_t big_min = [] {
        const char* e = getenv("NPW_GEMM_BIG_MIN");
        return e ? (int64_t)atoi(e) : (int64_t)192;
    }();
    static const int64_t big_min_shortk = [] {
        const char* e = getenv("NPW_GEMM_BIG_MIN_SHORTK");
        return e ? (int64_t)atoi(e) : (int64_t)512;  // k <= 256: prologue/epilogue-bound, more workgroups win
    }();
    const bool big = !opts.force_small && (opts.force_big || (wg128 >= (k <= 256 ? big_min_shortk : big_min)));
    if constexpr (sizeof(T) == 8) {
        if (opts.wide_n && big && vec_ok && k_ok && (m % 128 == 0) && (n % 256 == 0) && !p.lower_only && opts.tag == 0 && ta && !tb && opts.k_chunk_ == 0) {
            // op(A) = T, op(B) = N with n a multiple of 256: one 8-wave workgroup per 128 x 256 tile (see gemm_kernel)
            p.tiles_m = (int)(m / 128);
            p.tiles_n = (int)(n / 256);
            return launch<T, 128, 256, BK, false, false, false, 4>(p, stream);
        }
    }
    if (big) {
        p.tiles_m = (int)ceil_div(m, 128);
        p.tiles_n = (int)ceil_div(n, 128);
        const bool full = vec_ok && k_ok && (m % 128 == 0) && (n % 128 == 0);
        NPW_REQUIRE(full || opts.tag != 2, "gemm: the symmetric (tag 2) product needs full, aligned 128 x 128 tiles");
        if constexpr (sizeof(T) == 4) {
            // fp32, both operands k-contiguous: k-tiles of 32 (the same 128 bytes per row as fp64's 16) -- a k-tile of 16 floats
            // holds half the matrix-pipe time of the fp64 one behind the same barrier.  Measured (tools/sgemm_time.py, 4096^3):
            // N / T 140.4 -> 141.5 TFLOP/s; the other three forms LOSE 3.5 % with it (134 -> 129) and keep 16.
            // $NPW_SGEMM_BK32=0 restores 16 everywhere for A/B runs.
            static const bool bk32 = [] {
                const char* e = getenv("NPW_SGEMM_BK32");
                return e == nullptr || atoi(e) != 0;
            }();
            if (bk32 && full && a_kc && b_kc && k % 32 == 0 && opts.k_chunk_ % 32 == 0) return launch<T, 128, 128, 32, true, true, false>(p, stream);
        }
        if (full) return dispatch_layout<T, 128, 128, BK, false>(a_kc, b_kc, p, stream);
        return dispatch_layout<T, 128, 128, BK, true>(a_kc, b_kc, p, stream);
    }
    p.tiles_m = (int)ceil_div(m, 64);
    p.tiles_n = (int)ceil_div(n, 64);
    const bool full = vec_ok && ks_ok && (m % 64 == 0) && (n % 64 == 0);
    if (full) return dispatch_layout<T, 64, 64, BKS, false>(a_kc, b_kc, p, stream);
    // whole 32 x 32 tiles (the 32-column panels of QR: V_p^T W with m = 32, the rank-32 updates on 32-aligned
    // ranges): unguarded 16-byte accesses on a quarter-size tile instead of the element-wise guards of the EDGE form
    // (batched QR x32: these two products 32.4 + 20.4 -> 21.6 + 13.5 ms of kernel time; the factorisation's wall time did
    //  not move -- the panel chain and the far updates of the other stream fill the chip either way)
    if (vec_ok && ks_ok && (m % 32 == 0) && (n % 32 == 0) && !p.lower_only) {
        p.tiles_m = (int)(m / 32);
        p.tiles_n = (int)(n / 32);
        return dispatch_layout<T, 32, 32, BKS, false>(a_kc, b_kc, p, stream);
    }
    return dispatch_layout<T, 64, 64, BKS, true>(a_kc, b_kc, p, stream);
}

template int gemm<double>(char, char, int64_t, int64_t, int64_t, double, const double*, int64_t,
                          const double*, int64_t, double, const double*, int64_t, double*, int64_t,
                          const GemmOpts&, hipStream_t);
template int gemm<float>(char, char, int64_t, int64_t, int64_t, float, const float*, int64_t,
                         const float*, int64_t, float, const float*, int64_t, float*, int64_t,
                         const GemmOpts&, hipStream_t);

}  // namespace npw

namespace {
template <typename T>
int gemm_batched_impl(const char* who, int count, char transA, char transB, int64_t m, int64_t n, int64_t k, const T* const* A,
                      int64_t lda, const T* const* B, int64_t ldb, T* const* D, int64_t ldd, npw_stream_t stream) {
    NPW_REQUIRE(count >= 0 && m >= 0 && n >= 0 && k >= 0, "%s: negative argument", who);
    if (count == 0 || m == 0 || n == 0) return NPW_OK;
    NPW_REQUIRE(count <= 16, "%s: at most 16 problems per call", who);
    NPW_REQUIRE(A && B && D, "%s: NULL argument", who);
    int64_t da[16], db[16], dd[16];
    for (int z = 0; z < count; ++z) {
        NPW_REQUIRE(A[z] && B[z] && D[z], "%s: NULL tile (problem %d)", who, z);
        NPW_REQUIRE(((reinterpret_cast<uintptr_t>(A[z]) | reinterpret_cast<uintptr_t>(B[z]) | reinterpret_cast<uintptr_t>(D[z])) & 15) == 0,
                    "%s: tiles must be 16-byte aligned", who);
        da[z] = A[z] - A[0];
        db[z] = B[z] - B[0];
        dd[z] = D[z] - D[0];
    }
    npw::GemmOpts o;
    if (count > 1) {
        o.batch = count;
        o.delta_a = da;
        o.delta_b = db;
        o.delta_d = dd;
    }
    return npw::gemm<T>(transA, transB, m, n, k, T(1), A[0], lda, B[0], ldb, T(0), nullptr, 0, D[0], ldd, o, npw::as_stream(stream));
}

}  // namespace

extern "C" {

int npw_dgemm(char transA, char transB, int64_t m, int64_t n, int64_t k, double alpha,
              const double* A, int64_t lda, const double* B, int64_t ldb, double beta,
              const double* C, int64_t ldc, double* D, int64_t ldd, const int32_t* skip_flag,
              npw_stream_t stream) {
    npw::GemmOpts o;
    o.skip0 = skip_flag;
    return npw::gemm<double>(transA, transB, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, D, ldd, o,
                             npw::as_stream(stream));
}

int npw_sgemm(char transA, char transB, int64_t m, int64_t n, int64_t k, float alpha,
              const float* A, int64_t lda, const float* B, int64_t ldb, float beta, const float* C,
              int64_t ldc, float* D, int64_t ldd, const int32_t* skip_flag, npw_stream_t stream) {
    npw::GemmOpts o;
    o.skip0 = skip_flag;
    return npw::gemm<float>(transA, transB, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, D, ldd, o,
                            npw::as_stream(stream));
}

namespace {
constexpr int kDiagSplit = 8;  // k chunks of the diagonal-block launch of the symmetric trailing update (the separate-launch form)
// ... and of the form that rides in the pair launch (GemmOpts::diag_ws): 16 chunks of k / 16 -- 60 us each for k = 4096, short
// enough that what the free slots have not worked off when the pairs finish is one more short round on the whole chip
constexpr int kDiagSplitFused = 16;
bool diag_fused_on() {
    static const bool on = [] {
        const char* e = getenv("NPW_SYRK_DIAG_FUSED");
        return e == nullptr || atoi(e) != 0;
    }();
    return on;
}
bool diag_fused_ok(int64_t m, int64_t k, const void* workspace) {
    return diag_fused_on() && workspace != nullptr && m >= 1024 && k >= 1024 && k % (16 * kDiagSplitFused) == 0 &&
           (reinterpret_cast<uintptr_t>(workspace) & 15) == 0;
}
// X == Y takes the symmetric route when the tagged full-tile kernel applies: square, whole 128 x 128 tiles, k a multiple
// of the k-tile, 16-byte aligned rows; big enough that halving the tile count matters.
bool nt_sub_symmetric(int64_t m, int64_t n, int64_t k, const double* X, int64_t ldx, const double* Y, int64_t ldy) {
    return X == Y && ldx == ldy && m == n && m % 128 == 0 && m >= 1024 && k > 0 && k % 16 == 0 && ldx % 2 == 0 &&
           (reinterpret_cast<uintptr_t>(X) & 15) == 0;
}
}  // namespace

size_t npw_dgemm_nt_sub_workspace_bytes(int64_t m, int64_t n, int64_t k);

namespace {
// the diagonal blocks' partial products left in `partials` by the pair launch (GemmOpts::diag_ws), summed in a fixed order:
// D_block = S_block - sum of the chunks
int nt_sub_diag_reduce(int64_t m, const double* S, int64_t lds, double* D, int64_t ldd, const double* partials, hipStream_t stream) {
    hipLaunchKernelGGL(npw::splitk_reduce_kernel<double>, dim3(1, 128, (unsigned)(m / 128)), dim3(128), 0, stream, kDiagSplitFused, partials,
                       (int64_t)128 * 128, (int64_t)128, (int64_t)128, -1.0, 1.0, S, lds, D, ldd, (int64_t)128 * (lds + 1),
                       (int64_t)128 * (ldd + 1));
    NPW_LAUNCH_CHECK();
    return NPW_OK;
}

// the m / 128 diagonal 128 x 128 blocks of  D = S - X X^T  (the symmetric route's second launch)
int nt_sub_diag_blocks(int64_t m, int64_t k, const double* S, int64_t lds, const double* X, int64_t ldx, double* D, int64_t ldd,
                       const int32_t* skip_x, const int32_t* skip_y, void* workspace, hipStream_t stream) {
    npw::GemmOpts d;
    d.skip0 = skip_x;
    d.skip1 = skip_y;
    d.batch = (int)(m / 128);
    d.batch_a = d.batch_b = 128 * ldx;
    d.batch_c = 128 * (lds + 1);
    d.batch_d = 128 * (ldd + 1);
    if (workspace != nullptr && npw_dgemm_nt_sub_workspace_bytes(m, m, k) > 0) {
        // 128 workgroups with the full k would run alone for 0.18 ms: cut k into chunks (8x the workgroups,
        // partial products summed in a fixed order)
        NPW_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "npw_dgemm_nt_sub: workspace not 16B aligned");
        d.splitk = kDiagSplit;
        d.splitk_ws = workspace;
    }
    return npw::gemm<double>('N', 'T', 128, 128, k, -1.0, X, ldx, X, ldx, 1.0, S, lds, D, ldd, d, stream);
}
}  // namespace

size_t npw_dgemm_nt_sub_workspace_bytes(int64_t m, int64_t n, int64_t k) {
    if (m != n || m % 128 != 0 || m < 1024 || k < 1024) return 0;
    return (size_t)(m / 128) * (kDiagSplit > kDiagSplitFused ? kDiagSplit : kDiagSplitFused) * 128 * 128 * sizeof(double);
}

int npw_dgemm_nt_sub(int64_t m, int64_t n, int64_t k, const double* S, int64_t lds,
                     const double* X, int64_t ldx, const double* Y, int64_t ldy, double* D,
                     int64_t ldd, const int32_t* skip_x, const int32_t* skip_y, void* workspace,
                     npw_stream_t stream) {
    npw::GemmOpts o;
    o.skip0 = skip_x;
    o.skip1 = skip_y;
    o.tag = 1;
    // X == Y (the diagonal tiles of the trailing matrix): X X^T is symmetric bit for bit (element (i, j) and (j, i)
    // sum the same products in the same order), so only the strictly-lower 128 x 128 tiles are multiplied; each of
    // their workgroups writes BOTH D tiles of its pair from its accumulators (S is read at both positions, so the
    // result is the full S - X X^T for any S, symmetric or not; reference kernels.py:212-215).
    if (!nt_sub_symmetric(m, n, k, X, ldx, Y, ldy))
        return npw::gemm<double>('N', 'T', m, n, k, -1.0, X, ldx, Y, ldy, 1.0, S, lds, D, ldd, o,
                                 npw::as_stream(stream));
    o.tag = 2;
    o.lower_only = true;
    o.strict_lower = true;
    o.force_big = true;
    // 2 workgroups share a CU, so the chip holds 512 tiles at a time: the 496 strictly-lower tiles of a 4096^2
    // output are one full wave of work, the 32 diagonal tiles would be a second, nearly empty one.  They go into
    // their own small batched launch instead (full 128 x 128 blocks).
    // ... or, with a workspace, ride in the pair launch as short k chunks behind the pairs (GemmOpts::diag_ws): the 16 slots
    // the pairs leave free work through them meanwhile, the rest fills the tail; then one small reduction launch.
    const bool fused = diag_fused_ok(m, k, workspace);
    if (fused) {
        o.diag_ws = workspace;
        o.diag_split = kDiagSplitFused;
    }
    int rc = npw::gemm<double>('N', 'T', m, n, k, -1.0, X, ldx, Y, ldy, 1.0, S, lds, D, ldd, o, npw::as_stream(stream));
    if (rc) return rc;
    if (fused) return nt_sub_diag_reduce(m, S, lds, D, ldd, static_cast<const double*>(workspace), npw::as_stream(stream));
    return nt_sub_diag_blocks(m, k, S, lds, X, ldx, D, ldd, skip_x, skip_y, workspace, npw::as_stream(stream));
}

// `count` independent trailing updates  D[z] = S[z] - X[z] Y[z]^T  of one shape as ONE launch (blockIdx.z = problem):
// workgroups flow from one problem's tiles into the next one's, so the chip drains once per batch instead of once per
// tile (measured: 1.917 -> 1.882 ms per 4096^3 update in launches of 16, profiles/r02_syrk_launch_forms.md).  Each problem is computed exactly as
// npw_dgemm_nt_sub computes it (same tiles, same order of products).  When EVERY problem has X[z] == Y[z] (and the
// symmetric route's shape) the batch takes that route: one launch over all problems' strictly-lower tile pairs, then
// the diagonal blocks problem by problem (workspace: npw_dgemm_nt_sub_workspace_bytes(m, n, k), may be NULL).
int npw_dgemm_batched(int count, char transA, char transB, int64_t m, int64_t n, int64_t k, const double* const* A, int64_t lda,
                      const double* const* B, int64_t ldb, double* const* D, int64_t ldd, npw_stream_t stream) {
    return gemm_batched_impl<double>("npw_dgemm_batched", count, transA, transB, m, n, k, A, lda, B, ldb, D, ldd, stream);
}

int npw_sgemm_batched(int count, char transA, char transB, int64_t m, int64_t n, int64_t k, const float* const* A, int64_t lda,
                      const float* const* B, int64_t ldb, float* const* D, int64_t ldd, npw_stream_t stream) {
    return gemm_batched_impl<float>("npw_sgemm_batched", count, transA, transB, m, n, k, A, lda, B, ldb, D, ldd, stream);
}

size_t npw_dgemm_nt_sub_batched_workspace_bytes(int count, int64_t m, int64_t n, int64_t k) {
    return count <= 0 ? 0 : (size_t)count * npw_dgemm_nt_sub_workspace_bytes(m, n, k);
}

int npw_dgemm_nt_sub_batched(int count, int64_t m, int64_t n, int64_t k, const double* const* S, int64_t lds,
                             const double* const* X, int64_t ldx, const double* const* Y, int64_t ldy, double* const* D,
                             int64_t ldd, const int32_t* const* skip_x, const int32_t* const* skip_y, void* workspace,
                             npw_stream_t stream) {
    NPW_REQUIRE(count >= 0 && m >= 0 && n >= 0 && k >= 0, "npw_dgemm_nt_sub_batched: negative argument");
    if (count == 0 || m == 0 || n == 0) return NPW_OK;
    NPW_REQUIRE(count <= 16, "npw_dgemm_nt_sub_batched: at most 16 problems per call");
    NPW_REQUIRE(S && X && Y && D, "npw_dgemm_nt_sub_batched: NULL argument");
    NPW_REQUIRE((skip_x == nullptr) == (skip_y == nullptr), "npw_dgemm_nt_sub_batched: skip_x and skip_y go together");
    int64_t da[16], db[16], dc[16], dd[16], ds0[16], ds1[16];
    for (int z = 0; z < count; ++z) {
        NPW_REQUIRE(S[z] && X[z] && Y[z] && D[z], "npw_dgemm_nt_sub_batched: NULL tile (problem %d)", z);
        NPW_REQUIRE(((reinterpret_cast<uintptr_t>(S[z]) | reinterpret_cast<uintptr_t>(X[z]) | reinterpret_cast<uintptr_t>(Y[z]) |
                      reinterpret_cast<uintptr_t>(D[z])) & 15) == 0, "npw_dgemm_nt_sub_batched: tiles must be 16-byte aligned");
        da[z] = X[z] - X[0];
        db[z] = Y[z] - Y[0];
        dc[z] = S[z] - S[0];
        dd[z] = D[z] - D[0];
        ds0[z] = ds1[z] = 0;
        if (skip_x) {
            NPW_REQUIRE(skip_x[z] && skip_y[z], "npw_dgemm_nt_sub_batched: NULL skip flag (problem %d)", z);
            ds0[z] = skip_x[z] - skip_x[0];
            ds1[z] = skip_y[z] - skip_y[0];
        }
    }
    // every problem with X[z] == Y[z] and the symmetric route's shape: ONE launch over the strictly-lower tiles of all
    // problems (each workgroup writes its tile and the mirror tile), then each problem's diagonal blocks
    bool symmetric = true;
    for (int z = 0; z < count; ++z) symmetric = symmetric && nt_sub_symmetric(m, n, k, X[z], ldx, Y[z], ldy);
    npw::GemmOpts o;
    o.tag = symmetric ? 2 : 1;
    if (symmetric) o.lower_only = o.strict_lower = o.force_big = true;
    o.skip0 = skip_x ? skip_x[0] : nullptr;
    o.skip1 = skip_y ? skip_y[0] : nullptr;
    if (count > 1) {
        o.batch = count;
        o.delta_a = da;
        o.delta_b = db;
        o.delta_c = dc;
        o.delta_d = dd;
        o.delta_skip0 = ds0;
        o.delta_skip1 = ds1;
    }
    // (symmetric, with a workspace of `count` x npw_dgemm_nt_sub_workspace_bytes: every problem's diagonal blocks ride in the
    //  one launch, as in npw_dgemm_nt_sub; then a reduction launch per problem)
    const bool fused = symmetric && diag_fused_ok(m, k, workspace);
    if (fused) {
        o.diag_ws = workspace;
        o.diag_split = kDiagSplitFused;
    }
    int rc = npw::gemm<double>('N', 'T', m, n, k, -1.0, X[0], ldx, Y[0], ldy, 1.0, S[0], lds, D[0], ldd, o, npw::as_stream(stream));
    if (rc || !symmetric) return rc;
    if (fused) {
        const size_t per_problem = (size_t)(m / 128) * kDiagSplitFused * 128 * 128;
        for (int z = 0; z < count; ++z) {
            rc = nt_sub_diag_reduce(m, S[z], lds, D[z], ldd, static_cast<const double*>(workspace) + z * per_problem, npw::as_stream(stream));
            if (rc) return rc;
        }
        return NPW_OK;
    }
    for (int z = 0; z < count; ++z) {   // (one workspace: the launches of one stream run one after the other)
        rc = nt_sub_diag_blocks(m, k, S[z], lds, X[z], ldx, D[z], ldd, skip_x ? skip_x[z] : nullptr, skip_y ? skip_y[z] : nullptr,
                                workspace, npw::as_stream(stream));
        if (rc) return rc;
    }
    return NPW_OK;
}

}  // extern "C"
