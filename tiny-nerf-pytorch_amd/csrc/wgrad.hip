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

#define WG_LDS_ROWS 256                       // per operand
#define WG_LDS_FLOATS (2 * 2 * WG_LDS_ROWS * 32)

// LDS image of one operand block: row r holds its 8 16-byte chunks permuted by (r>>1)&7.
__device__ __forceinline__ int wg_lds_off(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 1) & 7)) << 2); }

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

// 512 threads = 8 waves = TWO per SIMD (<= 8 accumulator tiles = 128 registers per wave): while one wave of a SIMD
// waits for LDS fragments, the staging writes or the barrier, the other one keeps the matrix pipe busy.
__global__ __launch_bounds__(512, 2) void k_wgrad(const float* __restrict__ stash, int64_t stash_rows, int64_t M,
                                                  const int32_t* __restrict__ jobs, float* __restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) float lds[WG_LDS_FLOATS];
#ifdef TN_STAMPS   // diagnostic build: per-workgroup duration, written over the (unused) tail of the job record
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
    const int32_t* job = jobs + (int64_t)blockIdx.x * TN_JOB_INTS;
    const int n_at = job[JOB_N_AT], n_bt = job[JOB_N_BT], WA = job[JOB_WA], WB = 8 / WA;
    const int ta = (n_at + WA - 1) / WA, tb = (n_bt + WB - 1) / WB;     // the host plan only emits full-or-idle waves
    switch (ta * 8 + tb) {
        case 2 * 8 + 4: wgrad_body<2, 4>(stash, stash_rows, M, job, slabs, lds); break;
        case 1 * 8 + 2: wgrad_body<1, 2>(stash, stash_rows, M, job, slabs, lds); break;
        case 2 * 8 + 1: wgrad_body<2, 1>(stash, stash_rows, M, job, slabs, lds); break;
        case 1 * 8 + 1: wgrad_body<1, 1>(stash, stash_rows, M, job, slabs, lds); break;
        default: break;   // unreachable: shapes are validated on the host (tnerf_plan_fill)
    }
#ifdef TN_STAMPS
    if (threadIdx.x == 0) {
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_start;
        int32_t* rec = const_cast<int32_t*>(jobs) + (int64_t)blockIdx.x * TN_JOB_INTS;
        rec[14] = (int32_t)(dt & 0xffffffffu); rec[15] = (int32_t)(dt >> 32);
    }
#endif
}

int tn_launch_wgrad(const float* stash, int64_t stash_rows, int64_t M, const int32_t* jobs, int64_t n_jobs, float* slabs, hipStream_t stream) {
    hipLaunchKernelGGL(k_wgrad, dim3((unsigned)n_jobs), dim3(512), 0, stream, stash, stash_rows, M, jobs, slabs);
    TN_HIP_CHECK_LAUNCH("wgrad");
    return TNERF_OK;
}

// grads[i] = sum over the chunks of its job class of slab[off + c * stride]   (fixed order)
__global__ __launch_bounds__(256) void k_reduce(const float* __restrict__ slabs, const int32_t* __restrict__ table,
                                                int64_t n_params, float* __restrict__ grads) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_params) return;
    const int32_t off = table[TN_RED_HDR + 2 * i], cls = table[TN_RED_HDR + 2 * i + 1];
    const int64_t base = (int64_t)table[1 + 4 * cls] + off;
    const int64_t stride = table[1 + 4 * cls + 1];
    const int n = table[1 + 4 * cls + 2];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = 0;
    for (; c + 3 < n; c += 4) {
        s0 += slabs[base + (int64_t)c * stride];       s1 += slabs[base + (int64_t)(c + 1) * stride];
        s2 += slabs[base + (int64_t)(c + 2) * stride]; s3 += slabs[base + (int64_t)(c + 3) * stride];
    }
    for (; c < n; ++c) s0 += slabs[base + (int64_t)c * stride];
    grads[i] = (s0 + s1) + (s2 + s3);
}

int tn_launch_reduce(const float* slabs, const int32_t* table, int64_t n_params, float* grads, hipStream_t stream) {
    hipLaunchKernelGGL(k_reduce, dim3((unsigned)((n_params + 255) / 256)), dim3(256), 0, stream, slabs, table, n_params, grads);
    TN_HIP_CHECK_LAUNCH("wgrad/reduce");
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

// ------------------------------------------------------------------------------- loss
// loss = sum((comp - target)^2) / denom ; g = 2 (comp - target) / denom      [reference src/train.py:122]
// One workgroup, fixed summation order (deterministic).
__global__ __launch_bounds__(1024) void k_loss_grad(const float* __restrict__ comp, const float* __restrict__ target, const int64_t* __restrict__ tindex, int64_t n,
                                                    float inv_denom, float* __restrict__ g, float* __restrict__ loss_out) {
    __shared__ float part[16];
    float s = 0.0f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const float d = comp[i] - (tindex ? target[3 * tindex[i / 3] + i % 3] : target[i]);
        s += d * d;
        g[i] = (2.0f * d) * inv_denom;
    }
    s = tn_wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int w = 0; w < 16; ++w) t += part[w];
        loss_out[0] = t * inv_denom;
    }
}

int tn_launch_loss_grad(const float* comp, const float* target, const int64_t* target_index, int64_t R, double denom, float* g_comp, float* loss_out, hipStream_t stream) {
    hipLaunchKernelGGL(k_loss_grad, dim3(1), dim3(1024), 0, stream, comp, target, target_index, R * 3, (float)(1.0 / denom), g_comp, loss_out);
    TN_HIP_CHECK_LAUNCH("train_step/loss");
    return TNERF_OK;
}

// ------------------------------------------------------------------------------- Adam
// torch.optim.Adam (no amsgrad, no weight decay, maximize=False), single-tensor formulation:
//   m = lerp(m, g, 1-b1); v = b2 v + (1-b2) g^2; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, int64_t n, float b1, float b2, float eps,
                                              float step_size, float inv_sqrt_bc2, float gscale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * gscale;
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
    const float vi = v[i] * b2 + (gi * gi) * (1.0f - b2);
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] = p[i] - step_size * (mi / denom);
}

extern "C" int tnerf_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                               float beta1, float beta2, float eps, int64_t step, float grad_scale, tnerf_stream_t stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || n < 1 || step < 1) {
        tn_set_error("tnerf_adam_step: n=%lld step=%lld or NULL buffer", (long long)n, (long long)step);
        return TNERF_EINVAL;
    }
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n,
                       beta1, beta2, eps, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), grad_scale);
    TN_HIP_CHECK_LAUNCH("tnerf_adam_step");
    return TNERF_OK;
}
