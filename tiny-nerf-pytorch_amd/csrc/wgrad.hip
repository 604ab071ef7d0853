// Weight-gradient GEMMs, slab reduction, weight packing, loss gradient and Adam.
//
// wgrad: for every layer  dW[n][k] = sum_m dZ^T[n][m] * X^T[k][m]   (X = the layer's input)
// Both operands are rows of the block-major stash ([32-sample block][row][32], tnerf_internal.h), so an MFMA
// fragment "lane = row, 4 consecutive samples" is one 16-byte piece of a 128-byte line.  A workgroup
// (8 waves, two per SIMD) owns one (A rows x B rows) block of up to 256x256 outputs for one chunk
// of samples: the 8 waves split it 4x2 / 8x1 / 1x8 (<= 2x4 tiles of 32x32 = 128 accumulator
// registers per wave).  Per 32-sample step the workgroup stages [rows][32] of A and B through LDS
// in whole 128-byte lines by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip, the next block in flight during
// the MFMAs), XOR-swizzled — on the source address — so that the ds_read_b128 fragment reads are bank-conflict free.  Each workgroup
// writes its partial block to its own slab; a gather-reduce kernel sums the slabs in a fixed
// order into the flat gradient (deterministic: no float atomics).
#include "mlp_core.hpp"
#include "mlp_args.hpp"

// The WX_* knobs below strip parts of the split-bf16 kernel for ablation timings (tools/wgrad_x3_probe.py) and produce WRONG
// numerics.  They only compile in a diagnostic build (tools/build_variant.sh passes -DTN_DIAG): a stray -DWX_... on the product
// build is a hard error instead of a library that silently ships garbage.
#if (defined(WX_ONE_MFMA) || defined(WX_NO_CONVERT) || defined(WX_NO_PIN)) && !defined(TN_DIAG)
#error "WX_ONE_MFMA / WX_NO_CONVERT / WX_NO_PIN are diagnostic knobs with wrong numerics: build with -DTN_DIAG (tools/build_variant.sh)"
#endif

#define WG_LDS_ROWS 256                       // per operand
#define WG_LDS_FLOATS (2 * 2 * WG_LDS_ROWS * 32)

// LDS image of one operand block: row r holds its 8 16-byte chunks permuted by (r>>1)&7.
__device__ __forceinline__ int wg_lds_off(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 1) & 7)) << 2); }

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

template <int TA, int TB>
__device__ __forceinline__ void wgrad_body(const float* __restrict__ stash, int64_t stash_rows, int64_t M, const int32_t* __restrict__ job,
                                           float* __restrict__ slabs, float* lds) {
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // 0..7: two waves per SIMD
    const int n_at = job[JOB_N_AT], n_bt = job[JOB_N_BT], WA = job[JOB_WA];
    const int a_rows = job[JOB_A_ROWS], b_rows = job[JOB_B_ROWS];
    const int64_t a_row0 = job[JOB_A_ROW0], b_row0 = job[JOB_B_ROW0];
    const int blk0 = job[JOB_MBLK0], nblk = job[JOB_MBLKN];
    const int wa = wave % WA, wb = wave / WA;
    const int a_t0 = wa * TA, b_t0 = wb * TB;                  // first tile of this wave
    const int rows_a = n_at * 32, rows_b = n_bt * 32, rows = rows_a + rows_b;
    f32x16 acc[TA][TB];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float bsum[TA];
#pragma unroll
    for (int i = 0; i < TA; ++i) bsum[i] = 0.0f;

    // Staging by LDS-DMA: the rows [A | B] of a block are cut into pieces of 8 rows (1 KB of LDS = one wave-wide
    // global_load_lds_dwordx4); wave w moves pieces w, w+8, ...  Lane l of a piece fills row (l>>3), chunk position l&7,
    // i.e. it fetches source chunk (l&7) ^ ((row>>1)&7) of that row: the XOR swizzle is applied on the source address,
    // the LDS image stays lane-linear.  Rows that do not exist (head: 4 of 32, enc: 40 of 64) are clamped to a valid
    // row: they only feed slab rows / columns that the reduce table never references.
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
    constexpr int MAXP = (2 * WG_LDS_ROWS) / 64;               // pieces per wave
    const int npieces = rows / 8;
    uint32_t voff[MAXP], pdst[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int pc = wave + 8 * i;
        const int cr = 8 * pc + (lane >> 3);                    // combined row
        const bool isA = cr < rows_a;
        const int lr = isA ? cr : cr - rows_a;
        const int lim = (isA ? a_rows : b_rows) - 1;
        const int srow_ = (int)(isA ? a_row0 : b_row0) + (lr < lim ? lr : lim);
        voff[i] = (uint32_t)srow_ * 128u + (uint32_t)(((lane & 7) ^ ((lr >> 1) & 7)) << 4);
        const int lr0 = (8 * pc < rows_a) ? 8 * pc : 8 * pc - rows_a;       // first row of the piece (wave-uniform)
        pdst[i] = ((8 * pc < rows_a) ? 0u : (uint32_t)(WG_LDS_ROWS * 128)) + (uint32_t)lr0 * 128u;
    }
    // pieces [P0, P1) of this wave's share of block `blk` -> slot `buf`
    auto stage_dma_part = [&](int blk, int buf, auto p0c, auto p1c) TN_INLINE_LAMBDA {
        constexpr int P0 = decltype(p0c)::value, P1 = decltype(p1c)::value;
        const float* blkbase = stash + (int64_t)blk * stash_rows * 32;
        const uint32_t slot = lds0 + (uint32_t)buf * (2 * WG_LDS_ROWS * 128);
        tn_static_for<P1 - P0>([&](auto ic) TN_INLINE_LAMBDA {
            constexpr int i = P0 + decltype(ic)::value;
            if (wave + 8 * i < npieces) tn_glds16(blkbase, voff[i], slot + __builtin_amdgcn_readfirstlane(pdst[i]));   // wave-uniform
        });
    };
    auto stage_dma = [&](int blk, int buf) TN_INLINE_LAMBDA {
        stage_dma_part(blk, buf, std::integral_constant<int, 0>{}, std::integral_constant<int, MAXP>{});
    };
    // The last block of the batch may hold fewer than 32 samples: the slots behind M were never written by the
    // forward / dgrad kernels.  Each lane clears them in the 16 bytes it has just DMA'd (after its own vmcnt wait).
    auto clear_tail = [&](int blk, int buf) TN_INLINE_LAMBDA {
        const int nvalid = (int)(M - (int64_t)blk * 32);         // 1..31 here
        float* base = lds + buf * (2 * WG_LDS_ROWS * 32);
        tn_static_for<MAXP>([&](auto ic) TN_INLINE_LAMBDA {
            constexpr int i = decltype(ic)::value;
            if (wave + 8 * i < npieces) {
                const int pc = wave + 8 * i, cr = 8 * pc + (lane >> 3);
                const int lr = cr < rows_a ? cr : cr - rows_a;
                const int s0 = (((lane & 7) ^ ((lr >> 1) & 7)) << 2);           // first sample of this lane's chunk
                f32x4* q = reinterpret_cast<f32x4*>(base + (__builtin_amdgcn_readfirstlane(pdst[i]) >> 2) + lane * 4);
                f32x4 v = *q;
                v[0] = s0 + 0 < nvalid ? v[0] : 0.f; v[1] = s0 + 1 < nvalid ? v[1] : 0.f;
                v[2] = s0 + 2 < nvalid ? v[2] : 0.f; v[3] = s0 + 3 < nvalid ? v[3] : 0.f;
                *q = v;
            }
        });
    };

    const int frow = lane & 31, fh = lane >> 5;
    const bool active = (a_t0 < n_at) && (b_t0 < n_bt);
    const bool do_bias = job[JOB_HAS_BIAS] && wb == 0;
    auto is_tail = [&](int blk) TN_INLINE_LAMBDA { return (int64_t)blk * 32 + 32 > M; };          // wave-uniform

    // Two LDS slots: block b is read from slot b&1 while block b+1 lands in the other one.  Issuing its 8 DMAs costs a
    // wave several hundred cycles during which it feeds no MFMA: waves 0..3 issue theirs behind the fragment groups
    // q = 0, 1 of block b, waves 4..7 (their SIMD partners) behind q = 2, 3, so that a SIMD always has one wave on the
    // matrix pipe.  The last pieces still have a quarter of a block (> 4k cycles) to land.
    const bool late = wave >= 4;
    if (nblk > 0) stage_dma(blk0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (nblk > 0 && is_tail(blk0)) { clear_tail(blk0, 0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    __builtin_amdgcn_s_barrier();
    for (int b = 0; b < nblk; ++b) {
        const bool more = b + 1 < nblk;
        const float* A = lds + (b & 1) * (2 * WG_LDS_ROWS * 32);
        const float* B = A + WG_LDS_ROWS * 32;
        tn_static_for<4>([&](auto qc) TN_INLINE_LAMBDA {
            constexpr int q = decltype(qc)::value;
            f32x4 fa[TA], fb[TB];
            if (active) {
#pragma unroll
                for (int i = 0; i < TA; ++i) fa[i] = *reinterpret_cast<const f32x4*>(A + wg_lds_off((a_t0 + i) * 32 + frow, 2 * q + fh));
#pragma unroll
                for (int j = 0; j < TB; ++j) fb[j] = *reinterpret_cast<const f32x4*>(B + wg_lds_off((b_t0 + j) * 32 + frow, 2 * q + fh));
            }
            if (more && late == (q >= 2)) {
                if constexpr ((q & 1) == 0) stage_dma_part(blk0 + b + 1, (b + 1) & 1, std::integral_constant<int, 0>{}, std::integral_constant<int, MAXP / 2>{});
                else                        stage_dma_part(blk0 + b + 1, (b + 1) & 1, std::integral_constant<int, MAXP / 2>{}, std::integral_constant<int, MAXP>{});
            }
            if (active) {
                if (do_bias) {
#pragma unroll
                    for (int i = 0; i < TA; ++i) bsum[i] += (fa[i][0] + fa[i][1]) + (fa[i][2] + fa[i][3]);
                }
                // k-step outermost: consecutive MFMAs go to different accumulators
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int i = 0; i < TA; ++i)
#pragma unroll
                        for (int j = 0; j < TB; ++j) acc[i][j] = TN_MFMA(fa[i][p], fb[j][p], acc[i][j]);
            }
        });
        // this wave's share of block b+1 has landed; after the barrier everybody's has, and everybody is done reading
        // slot b&1, which block b+2 may overwrite
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (more && is_tail(blk0 + b + 1)) { clear_tail(blk0 + b + 1, (b + 1) & 1); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
    }

    // epilogue: partial block -> this workgroup's slab  [n_at*32][n_bt*32] then bias [n_at*32]
    if (active) {
        float* slab = slabs + job[JOB_SLAB_OFF];
        const int ld = n_bt * 32;
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int j = 0; j < TB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    slab[(int64_t)((a_t0 + i) * 32 + TN_ACC_ROW(r, fh)) * ld + (b_t0 + j) * 32 + frow] = acc[i][j][r];
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < TA; ++i) {
                const float tot = bsum[i] + __shfl_xor(bsum[i], 32, 64);
                if (fh == 0) slab[(int64_t)n_at * 32 * ld + (a_t0 + i) * 32 + frow] = tot;
            }
        }
    }
}

// ------------------------------------------------------------------------------------- "x3" body: three partial products
// The same job (an (A rows x B rows) block of <= 256 x 256 outputs for a chunk of samples, 8 waves, <= 2 x 4 tiles per wave)
// with the products on the fp16 matrix pipe, as in the chain kernels (mlpx3_core.hpp): every fp32 operand value, scaled by a
// power of two into the fp16 range, is two fp16 pieces (round to nearest), a product = a1 b1 + a1 b2 + a2 b1, fp32 accumulation.
// The scales are per ROW GROUP and batch: 2^(14 - exponent(bound)) with the magnitude bounds the chain kernels left behind the
// stash (TNB_*): dZ_l rows and H_l rows each share one scale, so a sum over samples needs no per-sample bookkeeping; elements far
// below the bound lose relative, never absolute precision (<= 2^-39 of the bound).  ONE accumulator: unlike in the chain
// kernels nothing here is cut by the matrix pipe's floor that the split would save (tools/microbench/split_schemes.py: the
// weight gradients are as close to fp64 with one accumulator as with two, or with an fp32 fma chain).
//   * every operand element is split ONCE per workgroup: the 512 threads fetch a 16-sample STAGE of the job's rows straight
//     from the stash into registers (buffer loads, 4 lanes per 64-byte row half), two stages ahead; scale and split it; and
//     write the two pieces into an LDS image laid out as MFMA operands (row-piece = 16 samples = 32 B);
//   * the MFMA phase is VALU-free: one ds_read_b128 per (tile, piece) and k-step, three MFMAs per tile pair;
//   * the image is double-buffered, so converting stage s+1 overlaps the MFMAs of stage s: one raw barrier per stage.
// Image: [buffer 2][piece 2][row 512] x 32 B; the two 16-byte halves of a row-piece are swapped when (row >> 3) & 1 — that
// makes the ds_read_b128 of 16 consecutive rows conflict-free (the 8-byte writes of a wave cover 512 contiguous bytes).
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
#define TN_MFMA16H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define WX_IMG_BYTES (2 * 2 * WG_LDS_ROWS * 32)            // 32 KB per buffer
static_assert(2 * WX_IMG_BYTES <= WG_LDS_FLOATS * 4, "the x3 operand images must fit the kernel's LDS");
__device__ __forceinline__ uint32_t wx_img_off(int piece, int crow, int half) {
    return (uint32_t)((piece * (2 * WG_LDS_ROWS) + crow) * 32 + ((half ^ ((crow >> 3) & 1)) << 4));
}
__device__ __forceinline__ unsigned wx_cvt2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
}
// 2^(14 - exponent(bound)): bound 2^e stays below 2^14 (fp16 overflows at 2^16); bound = 0 gives a harmless finite scale
__device__ __forceinline__ int wx_scale_exp(float bound) {
    const int e = 14 - __builtin_amdgcn_frexp_expf(bound);
    return e < -100 ? -100 : (e > 100 ? 100 : e);
}
__device__ __forceinline__ float wx_exp2i(int e) { return __uint_as_float((uint32_t)((e < -126 ? -126 : (e > 127 ? 127 : e)) + 127) << 23); }

// PD = register sets of the fetch pipeline = stages between a fetch and its use: 2 for the 256 x 256 blocks (an iteration is
// > 1.5k cycles of MFMA), 4 for the small, bandwidth-hungry input / head classes whose iterations are a few hundred cycles.
// NI = converter items per thread = ceil(job rows / 128): 4 covers the 512 combined rows of a 256 x 256 job; jobs of <= 256 combined
// rows (every class of a 128-wide network) run NI = 2 — with 4, half of their fetches and conversions worked on clamped duplicate rows
// that nobody reads, and their stages are conversion-bound (6 MFMAs against ~90 VALU per wave).
template <int TA, int TB, int PD, int NI>
__device__ __forceinline__ void wgrad_x3_body(const float* __restrict__ stash, int64_t stash_rows, int64_t M, const int32_t* __restrict__ job,
                                              float* __restrict__ slabs, float* lds_f, const float* __restrict__ bounds) {
    static_assert(PD % 2 == 0, "the image parity must follow the unrolled stage index");
    unsigned char* lds = reinterpret_cast<unsigned char*>(lds_f);
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_at = job[JOB_N_AT], n_bt = job[JOB_N_BT], WA = job[JOB_WA];
    const int a_rows = job[JOB_A_ROWS], b_rows = job[JOB_B_ROWS];
    const int a_row0 = job[JOB_A_ROW0], b_row0 = job[JOB_B_ROW0];
    const int blk0 = job[JOB_MBLK0], nblk = job[JOB_MBLKN];
    const int wa = wave % WA, wb = wave / WA;
    const int a_t0 = wa * TA, b_t0 = wb * TB;
    const int rows_a = n_at * 32, rows_b = n_bt * 32, rows = rows_a + rows_b;
    const bool has_bias = job[JOB_HAS_BIAS] != 0;
    const int ea = wx_scale_exp(bounds[job[JOB_A_BOUND]]), eb = wx_scale_exp(bounds[job[JOB_B_BOUND]]);
    f32x16 acc[TA][TB];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // Converter side: item k of thread t = (combined row (t >> 2) + 128 k of [A | B], 16-byte chunk t & 3 of the stage's 64 B).
    // Rows that do not exist (head: 4 of 32, input: 40 of 64) are clamped to a real row: they only feed slab rows / columns
    // that the reduce table never references.
    const int chunk = tid & 3;
    static_assert(NI % TB == 0 && NI <= 4, "items are dealt to the B-tile groups");
    int goff[NI]; bool isA[NI]; int crow[NI]; float gsc[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        crow[k] = (tid >> 2) + 128 * k;
        isA[k] = crow[k] < rows_a;
        const int lr = isA[k] ? crow[k] : crow[k] - rows_a;
        const int lim = (isA[k] ? a_rows : b_rows) - 1;
        const int srow_ = (isA[k] ? a_row0 : b_row0) + (lr < lim ? lr : lim);          // combined rows past the job's (dead items) land on B's last row
        goff[k] = srow_ * 128 + chunk * 16;
        gsc[k] = wx_exp2i(isA[k] ? ea : eb);
    }
    float bs[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) bs[k] = 0.f;
    const int nst = 2 * nblk;                                   // stages of 16 samples
    const int64_t blk_bytes = stash_rows * 128;
    f32x4 raw[PD][NI];
    // The hot loop must be ONE basic block (a branch ends the scheduling region: the conversion would no longer be placed in the
    // MFMAs' shadow): dead items (combined rows past the job's) are converted like live ones into image rows nobody reads, the
    // bias sums are formed for every item and only A rows are stored at the end, and the stages that need care — the batch's
    // ragged last block, the last PD+1 stages of the job — run in a guarded copy of the iteration.
    auto fetch1 = [&](int st, auto setc, auto kc) TN_INLINE_LAMBDA {      // item k of stage st -> register set
        constexpr int set = decltype(setc)::value, k = decltype(kc)::value;
        const unsigned char* base = reinterpret_cast<const unsigned char*>(stash) + (int64_t)(blk0 + (st >> 1)) * blk_bytes;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 0, (int)blk_bytes, 0x00020000);
        raw[set][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, goff[k], (st & 1) * 64, 0));
    };
    // Conversion of one item in three steps (so that the hot loop can place one step behind every MFMA of a group):
    //   0: scale, first pieces     1: residuals     2: second pieces, the bias row sum, both pieces to the image
    struct Item { f32x4 x, s; u32x2_t p1, p2; };
    auto conv_begin = [&](Item& it, int st, auto setc, auto kc, auto guardc) TN_INLINE_LAMBDA {
        constexpr int set = decltype(setc)::value, k = decltype(kc)::value;
        it.x = raw[set][k];
        if constexpr (decltype(guardc)::value) {                 // slots behind M were never written by the chain kernels
            const int nvalid = (int)(M - (int64_t)(blk0 + (st >> 1)) * 32) - (st & 1) * 16;
#pragma unroll
            for (int e = 0; e < 4; ++e) it.x[e] = 4 * chunk + e < nvalid ? it.x[e] : 0.0f;
        }
    };
    auto conv_step = [&](Item& it, int st, auto kc, auto stepc) TN_INLINE_LAMBDA {
        constexpr int k = decltype(kc)::value, step = decltype(stepc)::value;
        if constexpr (step == 0) {
            it.s = it.x * gsc[k];
            it.p1[0] = wx_cvt2(it.s[0], it.s[1]); it.p1[1] = wx_cvt2(it.s[2], it.s[3]);
        } else if constexpr (step == 1) {
            // (vector elements are copied to scalars first: __builtin_bit_cast applied directly to it.p1[i] reads element 0, hipcc 7.2)
            const unsigned q0 = it.p1[0], q1 = it.p1[1];
            const f16x2_t h0 = __builtin_bit_cast(f16x2_t, q0), h1 = __builtin_bit_cast(f16x2_t, q1);
            it.s[0] -= (float)h0[0]; it.s[1] -= (float)h0[1]; it.s[2] -= (float)h1[0]; it.s[3] -= (float)h1[1];
        } else {
            it.p2[0] = wx_cvt2(it.s[0], it.s[1]); it.p2[1] = wx_cvt2(it.s[2], it.s[3]);
            bs[k] += (it.x[0] + it.x[1]) + (it.x[2] + it.x[3]);
            unsigned char* img = lds + (st & 1) * WX_IMG_BYTES;
            const uint32_t o = wx_img_off(0, crow[k], chunk >> 1) + (chunk & 1) * 8;
            *reinterpret_cast<u32x2_t*>(img + o) = it.p1;
            *reinterpret_cast<u32x2_t*>(img + o + 2 * WG_LDS_ROWS * 32) = it.p2;
        }
    };
    auto convert1 = [&](int st, auto setc, auto kc, auto guardc) TN_INLINE_LAMBDA {    // a whole item at once (prologue)
        Item it;
        conv_begin(it, st, setc, kc, guardc);
        tn_static_for<3>([&](auto sc) TN_INLINE_LAMBDA { conv_step(it, st, kc, sc); });
    };
    const int frow = lane & 31, fh = lane >> 5;
    const bool active = (a_t0 < n_at) && (b_t0 < n_bt);
    uint32_t oa[TA], ob[TB];                                     // piece-0 image offsets of this lane's fragments
#pragma unroll
    for (int i = 0; i < TA; ++i) oa[i] = wx_img_off(0, (a_t0 + i) * 32 + frow, fh);
#pragma unroll
    for (int j = 0; j < TB; ++j) ob[j] = wx_img_off(0, rows_a + (b_t0 + j) * 32 + frow, fh);
    constexpr uint32_t PS = 2 * WG_LDS_ROWS * 32;                // bytes between pieces

    // One iteration: the MFMAs of stage s (image s & 1) with the conversion of stage s+1 (register set -> image (s+1) & 1)
    // and the re-fill of that set with stage s+1+PD woven BETWEEN the MFMA groups: the two waves of a SIMD run in lockstep
    // (one barrier per stage), so conversion work that is not in an MFMA's shadow leaves the matrix pipe idle.
    auto iteration = [&](int s, auto uc, auto activec, auto guardc) TN_INLINE_LAMBDA {
        constexpr int u = decltype(uc)::value;                   // s % PD
        constexpr int set = (u + 1) % PD;
        constexpr bool ACTIVE = decltype(activec)::value, GUARD = decltype(guardc)::value;
        using SetC = std::integral_constant<int, set>;
        const bool conv = !GUARD || s + 1 < nst, fill = !GUARD || s + 1 + PD < nst;
        const unsigned char* img = lds + (u & 1) * WX_IMG_BYTES;
        f16x8_t a1[TA], a2[TA];
        if constexpr (ACTIVE) {
#pragma unroll
            for (int i = 0; i < TA; ++i) {
                a1[i] = *reinterpret_cast<const f16x8_t*>(img + oa[i]);
                a2[i] = *reinterpret_cast<const f16x8_t*>(img + oa[i] + PS);
            }
        }
        // Group j: the 3 TA MFMAs of B tile j, with the conversion of items j (4/TB) .. of stage s+1 cut into three steps, one
        // behind every TA MFMAs; the order is pinned (left alone, the scheduler emits the MFMAs back to back and the whole
        // conversion after them, in the shadow of the last one only).
        constexpr int IPG = NI / TB;                                 // items per group
        f16x8_t b1, b2;
        if constexpr (ACTIVE) {
            b1 = *reinterpret_cast<const f16x8_t*>(img + ob[0]);
            b2 = *reinterpret_cast<const f16x8_t*>(img + ob[0] + PS);
        }
        tn_static_for<TB>([&](auto jc) TN_INLINE_LAMBDA {
            constexpr int j = decltype(jc)::value;
            Item it[IPG];
#ifndef WX_NO_CONVERT    // diagnostic build: MFMA phase only
            if (conv) tn_static_for<IPG>([&](auto qc) TN_INLINE_LAMBDA {
                conv_begin(it[decltype(qc)::value], s + 1, SetC{}, std::integral_constant<int, j * IPG + decltype(qc)::value>{}, guardc);
            });
#endif
            tn_static_for<3>([&](auto tc) TN_INLINE_LAMBDA {
                constexpr int term = decltype(tc)::value;           // small terms first: a2 b1, a1 b2, a1 b1
                if constexpr (ACTIVE) {
#ifdef WX_ONE_MFMA       // diagnostic build: only the leading product
                    if constexpr (term == 2)
#endif
#pragma unroll
                    for (int i = 0; i < TA; ++i) {
                        const f16x8_t& av = term == 0 ? a2[i] : a1[i];
                        const f16x8_t& bv = term == 1 ? b2 : b1;
                        acc[i][j] = TN_MFMA16H(av, bv, acc[i][j]);
                    }
                }
#ifndef WX_NO_CONVERT
                if (conv) tn_static_for<IPG>([&](auto qc) TN_INLINE_LAMBDA {
                    conv_step(it[decltype(qc)::value], s + 1, std::integral_constant<int, j * IPG + decltype(qc)::value>{}, tc);
                });
#endif
                if constexpr (term == 2) {
#ifndef WX_NO_CONVERT
                    if (fill) tn_static_for<IPG>([&](auto qc) TN_INLINE_LAMBDA {
                        fetch1(s + 1 + PD, SetC{}, std::integral_constant<int, j * IPG + decltype(qc)::value>{});
                    });
#endif
                    if constexpr (ACTIVE && j + 1 < TB) {            // the next group's B fragments
                        b1 = *reinterpret_cast<const f16x8_t*>(img + ob[j + 1]);
                        b2 = *reinterpret_cast<const f16x8_t*>(img + ob[j + 1] + PS);
                    }
                }
#ifndef WX_NO_PIN
                if constexpr (!GUARD) __builtin_amdgcn_sched_barrier(0);
#endif
            });
        });
    };

    // Prologue: stages 0 .. PD-1 into the register sets, stage 0 converted (guarded), its set re-filled with stage PD.
    using T_ = std::true_type; using F_ = std::false_type;
    tn_static_for<PD>([&](auto pc) TN_INLINE_LAMBDA {
        constexpr int p_ = decltype(pc)::value;
        using PC = std::integral_constant<int, p_>;
        tn_static_for<NI>([&](auto kc) TN_INLINE_LAMBDA { if (p_ < nst) fetch1(p_, PC{}, kc); });
    });
    using C0 = std::integral_constant<int, 0>;
    tn_static_for<NI>([&](auto kc) TN_INLINE_LAMBDA {
        if (nst > 0) convert1(0, C0{}, kc, T_{});
        if (PD < nst) fetch1(PD, C0{}, kc);
    });
    // Unguarded iterations: s + 1 + PD < nst, and stage s + 1 lies in a full block (the ragged last block of the batch, if this
    // job reaches it, is left to the guarded loop).
    const int64_t full_blocks = M / 32 - blk0;
    const int n_safe = 2 * (int)(full_blocks < nblk ? (full_blocks > 0 ? full_blocks : 0) : nblk);      // stages 0 .. n_safe-1 are in full blocks
    int n_fast = nst - 1 - PD < n_safe - 1 ? nst - 1 - PD : n_safe - 1;                                   // iterations 0 .. n_fast-1 need no guard
    n_fast = n_fast > 0 ? n_fast / PD * PD : 0;
    // (a raw s_barrier does not wait for this wave's outstanding LDS writes: lgkmcnt(0) in front of every barrier.  The barrier
    // orders "image s & 1 complete" (written in iteration s-1) and "everyone is done reading image (s+1) & 1" (read in s-1).)
    auto run = [&](auto activec) TN_INLINE_LAMBDA {
#pragma unroll 1
        for (int s0 = 0; s0 < n_fast; s0 += PD) {
            tn_static_for<PD>([&](auto uc) TN_INLINE_LAMBDA {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                iteration(s0 + decltype(uc)::value, uc, activec, F_{});
            });
        }
#pragma unroll 1
        for (int s0 = n_fast; s0 < nst; s0 += PD) {
            tn_static_for<PD>([&](auto uc) TN_INLINE_LAMBDA {
                if (s0 + decltype(uc)::value < nst) {            // wave-uniform
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    iteration(s0 + decltype(uc)::value, uc, activec, T_{});
                }
            });
        }
    };
    if (active) run(T_{}); else run(F_{});

    // epilogue: partial block (scaled back) -> this workgroup's slab  [n_at*32][n_bt*32] then bias [n_at*32]   (as wgrad_body)
    float* slab = slabs + job[JOB_SLAB_OFF];
    const int ld = n_bt * 32;
    if (active) {
        const float back = wx_exp2i(-(ea + eb));
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int j = 0; j < TB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    slab[(int64_t)((a_t0 + i) * 32 + TN_ACC_ROW(r, fh)) * ld + (b_t0 + j) * 32 + frow] = acc[i][j][r] * back;
    }
    if (has_bias) {                                              // row sums of A: the 4 lanes of a row hold its 4 chunks
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            float t = bs[k];
            t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64);
            if (chunk == 0 && isA[k]) slab[(int64_t)n_at * 32 * ld + crow[k]] = t;          // (isA implies the row exists in the slab)
        }
    }
}

// 512 threads = 8 waves = TWO per SIMD (<= 8 accumulator tiles = 128 registers per wave): while one wave of a SIMD
// waits for LDS fragments, the staging writes or the barrier, the other one keeps the matrix pipe busy.
// (Packed fp32 VALU stays ON here, unlike in the backward chain kernels (mlpx3.hip TX_PLAIN_F32): with two waves per SIMD a packed
// instruction that cannot overlap its own wave's MFMA overlaps the other wave's.  A/B in the bench's step, three runs each: 0.886-0.890 ms
// packed against 0.909-0.915 ms unpacked — although a probe on synthetic operands had said the opposite by 2 %.)
// MODE 0: the fp32-MFMA body, 1: the x3 body — the stash's pipe tag (tnerf_internal.h TNB_TAG) must agree; 2: the body the tag names
// (tnerf_wgrad: the per-call entry point cannot know which forward filled the stash).  A stash whose tag does not fit — another
// pipe's forward, a dgrad kernel that refused it, memory no training forward ever wrote — gives NaN slabs, not garbage.
template <int MODE>
__global__ __launch_bounds__(512, 2) void k_wgrad(const float* __restrict__ stash, int64_t stash_rows, int64_t M,
                                                  const int32_t* __restrict__ jobs, float* __restrict__ slabs, int64_t* step_inc,
                                                  const float* __restrict__ bounds) {
    __shared__ __attribute__((aligned(16))) float lds[WG_LDS_FLOATS];
    // Dataset mode: the step counter advances HERE — the forward and dgrad kernels of this step (which read it) are done,
    // the finishing kernel (which needs the 1-based count for Adam's bias correction) has not started.
    if (step_inc && blockIdx.x == 0 && threadIdx.x == 0) *step_inc += 1;
#ifdef TN_STAMPS   // diagnostic build: per-workgroup duration, written over the (unused) tail of the job record
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
    const int32_t* job = jobs + (int64_t)blockIdx.x * TN_JOB_INTS;
    const int n_at = job[JOB_N_AT], n_bt = job[JOB_N_BT], WA = job[JOB_WA], WB = 8 / WA;
    const unsigned tag = reinterpret_cast<const unsigned*>(bounds)[TNB_TAG];
    const bool X3 = MODE == 2 ? tag == TN_TAG_X3 : MODE == 1;                 // uniform
    if (tag != (X3 ? TN_TAG_X3 : TN_TAG_F32)) {
        float* slab = slabs + job[JOB_SLAB_OFF];
        const int n = n_at * 32 * n_bt * 32 + (job[JOB_HAS_BIAS] ? n_at * 32 : 0);
        for (int i = threadIdx.x; i < n; i += 512) slab[i] = __builtin_nanf("");
        return;
    }
    const int ta = (n_at + WA - 1) / WA, tb = (n_bt + WB - 1) / WB;     // the host plan only emits full-or-idle waves
    const bool small = (n_at + n_bt) * 32 <= 256;                      // combined rows of the job (uniform): two converter items per thread cover them
    switch (ta * 8 + tb) {
        case 2 * 8 + 4: if (MODE != 0 && X3) wgrad_x3_body<2, 4, 2, 4>(stash, stash_rows, M, job, slabs, lds, bounds); else if (MODE != 1) wgrad_body<2, 4>(stash, stash_rows, M, job, slabs, lds); break;
        case 1 * 8 + 2: if (MODE != 0 && X3) { if (small) wgrad_x3_body<1, 2, 4, 2>(stash, stash_rows, M, job, slabs, lds, bounds); else wgrad_x3_body<1, 2, 4, 4>(stash, stash_rows, M, job, slabs, lds, bounds); }
                        else if (MODE != 1) wgrad_body<1, 2>(stash, stash_rows, M, job, slabs, lds); break;
        case 2 * 8 + 1: if (MODE != 0 && X3) { if (small) wgrad_x3_body<2, 1, 4, 2>(stash, stash_rows, M, job, slabs, lds, bounds); else wgrad_x3_body<2, 1, 4, 4>(stash, stash_rows, M, job, slabs, lds, bounds); }
                        else if (MODE != 1) wgrad_body<2, 1>(stash, stash_rows, M, job, slabs, lds); break;
        case 1 * 8 + 1: if (MODE != 0 && X3) { if (small) wgrad_x3_body<1, 1, 4, 2>(stash, stash_rows, M, job, slabs, lds, bounds); else wgrad_x3_body<1, 1, 4, 4>(stash, stash_rows, M, job, slabs, lds, bounds); }
                        else if (MODE != 1) wgrad_body<1, 1>(stash, stash_rows, M, job, slabs, lds); break;
        default: break;   // unreachable: shapes are validated on the host (tnerf_plan_fill)
    }
#ifdef TN_STAMPS
    if (threadIdx.x == 0) {
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_start;
        int32_t* rec = const_cast<int32_t*>(jobs) + (int64_t)blockIdx.x * TN_JOB_INTS;
        rec[14] = (int32_t)(dt & 0xffffffffu); rec[15] = (int32_t)(dt >> 32);      // (slots 12, 13 are JOB_A_BOUND / JOB_B_BOUND: not ours)
    }
#endif
}

// bounds: the stash's magnitude-bound words (TN_BOUND_OFF), whose last word is the pipe tag.  mode: 0 fp32-MFMA body, 1 x3 body, 2 by the tag.
int tn_launch_wgrad(const float* stash, int64_t stash_rows, int64_t M, const int32_t* jobs, int64_t n_jobs, float* slabs, int64_t* step_inc, hipStream_t stream, int mode,
                    const float* bounds) {
    if (!bounds) { tn_set_error("wgrad: the stash's bound words are required"); return TNERF_EINVAL; }
    if (mode == 1)      hipLaunchKernelGGL(k_wgrad<1>, dim3((unsigned)n_jobs), dim3(512), 0, stream, stash, stash_rows, M, jobs, slabs, step_inc, bounds);
    else if (mode == 0) hipLaunchKernelGGL(k_wgrad<0>, dim3((unsigned)n_jobs), dim3(512), 0, stream, stash, stash_rows, M, jobs, slabs, step_inc, bounds);
    else                hipLaunchKernelGGL(k_wgrad<2>, dim3((unsigned)n_jobs), dim3(512), 0, stream, stash, stash_rows, M, jobs, slabs, step_inc, bounds);
    TN_HIP_CHECK_LAUNCH("wgrad");
    return TNERF_OK;
}

// ------------------------------------------------------------------------------- finishing kernel
// One thread per parameter i:
//   REDUCE: grads[i] = sum over the chunks of its job class of slab[off + c * stride]   (fixed order: deterministic)
//   ADAM  : torch.optim.Adam (no amsgrad, no weight decay, maximize=False), single-tensor formulation:
//             m = lerp(m, g, 1-b1); v = b2 v + (1-b2) g^2; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
//           and the updated weight goes straight to its places in the packed (MFMA-fragment-ordered) copy the next
//           forward reads — no separate pack launch.
// One extra workgroup sums the per-ray squared errors (fixed order) into the loss.
template <bool REDUCE, bool ADAM>
__global__ __launch_bounds__(256) void k_finish(FinishArgs f) {
    const int nb = (int)((f.n_params + 255) / 256);
    if ((int)blockIdx.x == nb) {                                  // the loss block (only launched when loss_out != NULL)
        __shared__ float part[4];
        float s = 0.0f;
        for (int64_t r = threadIdx.x; r < f.R; r += 256) s += f.ray_ws[4 * r + 3];
        s = tn_wave_sum(s);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) f.loss_out[0] = ((part[0] + part[1]) + (part[2] + part[3])) * f.inv_denom;
        return;
    }
    // Which 256 parameters this workgroup takes: NOT blockIdx.x.  Workgroups go to the eight XCDs round-robin (blockIdx.x % 8), each XCD
    // has its own L2, and the re-pack below scatters 2- and 4-byte values into MFMA-fragment order, where one 128-byte line holds the
    // same k range of 8 consecutive weight ROWS — 8 consecutive workgroups for a 256-wide layer, i.e. one partial line in each of the 8
    // L2s, written back as eight masked partial lines.  With the workgroups of one XCD taking a CONTIGUOUS range of parameter blocks the
    // lines are completed inside one L2.  (A bijection of the block index: every parameter is still handled exactly once.)
    const int q_ = nb / 8, r_ = nb % 8, x_ = (int)blockIdx.x % 8, j_ = (int)blockIdx.x / 8;
    const int blk = x_ * q_ + (x_ < r_ ? x_ : r_) + j_;
    const int64_t i = (int64_t)blk * 256 + threadIdx.x;
    __shared__ float bc[2];
    if (ADAM) {
        if (threadIdx.x == 0) {
            const double t = (double)(f.step ? *f.step : f.step_host);
            bc[0] = (float)((double)f.lr / (1.0 - pow((double)f.b1, t)));         // step size
            bc[1] = (float)(1.0 / sqrt(1.0 - pow((double)f.b2, t)));
        }
        __syncthreads();
    }
    __shared__ unsigned smax[2 * (TN_MAXD + 1)];
    const bool track = ADAM && f.scatter3 != nullptr;             // uniform
    if (track) {
        for (int j = threadIdx.x; j < 2 * (TN_MAXD + 1); j += 256) smax[j] = 0u;
        __syncthreads();
    }
    int x3_key = -1; unsigned x3_bits = 0u;
    if (i < f.n_params) {
    float g;
    if (REDUCE) {
        const int32_t* __restrict__ table = f.reduce_table;
        const float* __restrict__ slabs = f.slabs;
        const int32_t off = table[TN_RED_HDR + 2 * i], cls = table[TN_RED_HDR + 2 * i + 1];
        const int64_t base = (int64_t)table[1 + 4 * cls] + off;
        const int64_t stride = table[1 + 4 * cls + 1];
        const int n = table[1 + 4 * cls + 2];
        // eight loads in flight per thread (the kernel is latency-bound: 64 MB of slabs, ~30 dependent-free loads per thread)
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
        const float* q = slabs + base;
        int c = 0;
        for (; c + 7 < n; c += 8, q += 8 * stride) {
            const float v0 = q[0], v1 = q[stride], v2 = q[2 * stride], v3 = q[3 * stride];
            const float v4 = q[4 * stride], v5 = q[5 * stride], v6 = q[6 * stride], v7 = q[7 * stride];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3; s4 += v4; s5 += v5; s6 += v6; s7 += v7;
        }
        for (; c < n; ++c, q += stride) s0 += q[0];
        g = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
        f.grads[i] = g;
    } else {
        g = f.grads[i];
    }
    if (ADAM) {
        const float gi = g * f.gscale;
        const float mi = f.m[i] + (gi - f.m[i]) * (1.0f - f.b1);
        const float vi = f.v[i] * f.b2 + (gi * gi) * (1.0f - f.b2);
        f.m[i] = mi; f.v[i] = vi;
        const float denom = sqrtf(vi) * bc[1] + f.eps;
        const float pn = f.params[i] - bc[0] * (mi / denom);
        f.params[i] = pn;
        if (f.scatter) {
            for (int k = 0; k < f.width; ++k) {
                const int32_t d = f.scatter[i * f.width + k];
                if (d < 0) break;
                if (d < f.bf16_elems) {                            // bf16 fragment stream: round to nearest even
                    reinterpret_cast<unsigned short*>(f.packed)[d] = __builtin_bit_cast(unsigned short, (__bf16)pn);   // as k_pack16
                } else {
                    reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(f.packed) + f.bias_off_bytes)[d - f.bf16_elems] = pn;
                }
            }
        }
        if (f.scatter3) {                                          // the x3 record stream: two scaled fp16 pieces (as k_packx3)
            const int32_t d0 = f.scatter3[i * f.width3];
            float* meta = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(f.packed3) + f.n3.meta_off);
            float wsc = 1.0f;
            int key = -1;                                          // 2 * layer + (bias ? 1 : 0): whose running maximum |pn| belongs to
            if (d0 >= 0 && d0 < f.x3_elems) {                      // every position of a weight lies in the same layer
                const int l = tx_record_layer(&f.n3, (int)(d0 / (f.n3.rec_frags * 512)));
                wsc = meta[l * TX_META + 3]; key = 2 * l;
            } else if (d0 >= f.x3_elems) {
                const int bi = (int)(d0 - f.x3_elems);
                key = 2 * (bi < f.n3.depth * f.n3.hidden ? bi / f.n3.hidden : f.n3.depth) + 1;
            }
            for (int k = 0; k < f.width3; ++k) {
                const int32_t d = f.scatter3[i * f.width3 + k];
                if (d < 0) break;
                if (d < f.x3_elems) {
                    reinterpret_cast<unsigned short*>(f.packed3)[d] = tx_piece_bits(pn, wsc, (d >> 9) % TX_NP);
                } else {
                    reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(f.packed3) + f.n3.bias_off)[d - f.x3_elems] = pn;
                }
            }
            // the layer's running maxima (k_x3stats_final, launched behind this kernel, turns them into the next scale), reduced per
            // WORKGROUP in LDS first: a wave-level reduction alone left ~7500 atomics per step on the same 18 addresses
            x3_key = key; x3_bits = __float_as_uint(fabsf(pn));
        }
    }
    }
    if (track) {
        float* meta = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(f.packed3) + f.n3.meta_off);
        if (x3_key >= 0) atomicMax(&smax[x3_key], x3_bits);
        __syncthreads();
        // Small networks (f.fold_stats: at most TN_FOLD_STATS_BLOCKS workgroups): the LAST parameter workgroup to get here publishes the
        // scale records itself — what k_x3stats_final otherwise does in a launch of its own, 4 us of a 0.3 ms step; every workgroup has
        // scattered its weights with the old scales and added its maxima by then.  Counter: the spare word TX_META_DONE of record 0, left
        // at zero for the next step.  Ordering without a fence (an agent-scope release writes back the XCD's whole L2: measured +110 us
        // per step with one in each of 1850 workgroups): the maxima are RETURNING atomics — performed at the memory side, where the eight
        // L2s agree, before their results come back — and the count is taken behind the wait for those results; the last workgroup
        // reads the maxima with atomic loads.  Large networks keep the separate launch: 1850 returning atomics on ONE counter word
        // serialise, and every workgroup waits out two round trips to memory before it can retire (measured: +17 us on the 8x256
        // finishing kernel against the 4 us launch it saves).
        if (!f.fold_stats) {
            for (int j = threadIdx.x; j < 2 * (f.n3.depth + 1); j += 256)
                if (smax[j]) atomicMax(reinterpret_cast<unsigned*>(meta) + (j >> 1) * TX_META + 4 + (j & 1), smax[j]);
            return;
        }
        unsigned seen = 0u;
        for (int j = threadIdx.x; j < 2 * (f.n3.depth + 1); j += 256)
            if (smax[j]) seen |= atomicMax(reinterpret_cast<unsigned*>(meta) + (j >> 1) * TX_META + 4 + (j & 1), smax[j]);
        asm volatile("s_waitcnt vmcnt(0)" :: "v"(seen) : "memory");
        __shared__ unsigned last_;
        __syncthreads();
        unsigned* done = reinterpret_cast<unsigned*>(meta) + TX_META_DONE;
        if (threadIdx.x == 0) last_ = atomicAdd(done, 1u) == (unsigned)nb - 1u;
        __syncthreads();
        if (last_) {
            tx_stats_final(meta, (int)threadIdx.x, f.n3.depth + 1, 1, f.scale_floor);
            if (threadIdx.x == 0) *done = 0u;
        }
    }
}

int tn_launch_finish(const FinishArgs& f, hipStream_t stream) {
    const unsigned nb = (unsigned)((f.n_params + 255) / 256) + (f.loss_out ? 1u : 0u);
    const bool red = f.slabs != nullptr, adam = f.params != nullptr;
    if (red && adam)       hipLaunchKernelGGL((k_finish<true, true>), dim3(nb), dim3(256), 0, stream, f);
    else if (red)          hipLaunchKernelGGL((k_finish<true, false>), dim3(nb), dim3(256), 0, stream, f);
    else if (adam)         hipLaunchKernelGGL((k_finish<false, true>), dim3(nb), dim3(256), 0, stream, f);
    else { tn_set_error("finish: nothing to do"); return TNERF_EINVAL; }
    TN_HIP_CHECK_LAUNCH("step/finish");
    return TNERF_OK;
}

// ------------------------------------------------------------------------------- pack
__global__ __launch_bounds__(256) void k_pack(const float* __restrict__ params, const int32_t* __restrict__ table, int64_t n,
                                              float* __restrict__ packed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t s = table[i];
    packed[i] = s >= 0 ? params[s] : 0.0f;
}

extern "C" int tnerf_mlp_pack(const float* params, const int32_t* pack_table, int64_t packed_floats, float* packed, tnerf_stream_t stream) {
    if (!params || !pack_table || !packed || packed_floats < 1) {
        tn_set_error("tnerf_mlp_pack: params=%p table=%p packed=%p n=%lld", (const void*)params, (const void*)pack_table, (void*)packed, (long long)packed_floats);
        return TNERF_EINVAL;
    }
    hipLaunchKernelGGL(k_pack, dim3((unsigned)((packed_floats + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, pack_table, packed_floats, packed);
    TN_HIP_CHECK_LAUNCH("tnerf_mlp_pack");
    return TNERF_OK;
}

// ------------------------------------------------------------------------------- Adam
extern "C" int tnerf_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                               float beta1, float beta2, float eps, int64_t step, float grad_scale, tnerf_stream_t stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || n < 1 || step < 1) {
        tn_set_error("tnerf_adam_step: n=%lld step=%lld or NULL buffer", (long long)n, (long long)step);
        return TNERF_EINVAL;
    }
    FinishArgs f{};
    f.n_params = n; f.grads = const_cast<float*>(grads);
    f.params = params; f.m = exp_avg; f.v = exp_avg_sq; f.lr = lr; f.b1 = beta1; f.b2 = beta2; f.eps = eps; f.gscale = grad_scale;
    f.step_host = step;
    return tn_launch_finish(f, (hipStream_t)stream);
}
