// Device-side helpers shared by the HIP kernels (gfx950 only; wavefront = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "tnerf_internal.h"

typedef float f32x4  __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TN_WAVE 64

#define TN_HIP_CHECK_LAUNCH(name)                                              \
    do {                                                                       \
        hipError_t e_ = hipGetLastError();                                     \
        if (e_ != hipSuccess) {                                                \
            tn_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return (int)e_;                                                    \
        }                                                                      \
    } while (0)

// ------------------------------------------------------------------------------- host-side launch helpers
// Per-device facts are cached in atomics indexed by the device ordinal (idempotent values: a race only repeats a query).
#include <atomic>
#define TN_MAX_DEVICES 64
// The device a launch on `stream` runs on (the NULL stream belongs to the current device).
inline int tn_stream_device(hipStream_t stream) {
    hipDevice_t d = 0; int cur = 0;
    if (stream && hipStreamGetDevice(stream, &d) == hipSuccess) return (int)d;
    (void)hipGetDevice(&cur);
    return cur;
}
inline int tn_device_cus(int dev) {
    static std::atomic<int> cus[TN_MAX_DEVICES];
    if (dev < 0 || dev >= TN_MAX_DEVICES) return 256;
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: `seen` (one array per kernel, indexed by
// device) remembers the largest size already granted there.  Returns TNERF_OK or the hipError_t.
inline int tn_grant_dyn_lds(const void* kernel, size_t bytes, int dev, std::atomic<uint32_t>* seen, const char* who) {
    if (dev >= 0 && dev < TN_MAX_DEVICES && seen[dev].load(std::memory_order_acquire) >= bytes) return TNERF_OK;
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != dev) { tn_set_error("%s: the stream belongs to device %d but the current device is %d", who, dev, cur); return TNERF_EINVAL; }
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { tn_set_error("%s: cannot grant %zu B of dynamic LDS: %s", who, bytes, hipGetErrorString(e)); return (int)e; }
    if (dev >= 0 && dev < TN_MAX_DEVICES) seen[dev].store((uint32_t)bytes, std::memory_order_release);
    return TNERF_OK;
}

__device__ __forceinline__ int tn_lane() { return (int)(threadIdx.x & 63); }

// "x3" weight stream (tnerf_internal.h): fp16 bits of piece 0 / 1 of x 2^s (wsc = 2^s), both rounded to nearest.  The clamp only
// matters for a weight that grew sixteen-fold since the scale was chosen: it stays finite.
__device__ __forceinline__ unsigned short tx_piece_bits(float x, float wsc, int piece) {
    const float s = fminf(fmaxf(x * wsc, -65000.0f), 65000.0f);
    const _Float16 p1 = (_Float16)s;
    const _Float16 p = piece == 0 ? p1 : (_Float16)(s - (float)p1);
    return __builtin_bit_cast(unsigned short, p);
}

// Running maxima of the scale records -> the records (thread l = layer l, l < n_layers), and the maxima cleared for the next round
// (the semantics of post / floor: k_x3stats_final in mlpx3.hip, which is this function as a kernel; the finishing kernel's last
// workgroup calls it directly).  The maxima were written with atomics by other workgroups: read them at the L2, not from a line
// this CU may have cached when it read the old scale.
#define TX_META_DONE 6          // record 0's spare word: the finishing kernel's count of workgroups that are through
__device__ __forceinline__ void tx_stats_final(float* __restrict__ meta, int l, int n_layers, int post, float floor) {
    if (l >= n_layers) return;
    float* m = meta + l * TX_META;
    unsigned* mu = reinterpret_cast<unsigned*>(m);
    const float mw = __uint_as_float(__hip_atomic_load(mu + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const float mb = __uint_as_float(__hip_atomic_load(mu + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const float wsc_old = m[3];
    const int e = 12 - __builtin_amdgcn_frexp_expf(fmaxf(mw, floor));      // max|W| = f 2^e', f in [0.5, 1): max|W| 2^(12-e') in [2^11, 2^12)
    const float wsc_new = __uint_as_float((uint32_t)(min(max(e, -126), 127) + 127) << 23);
    m[0] = 1.0f / (post ? wsc_old : wsc_new);                               // a power of two: exact
    m[1] = mw; m[2] = mb; m[3] = wsc_new;
    mu[4] = 0u; mu[5] = 0u;
}

// ----------------------------------------------------------------------------- Philox4x32-10
// Counter-based generator for the "speed mode" jitter (t_rand == NULL).  One 128-bit block per
// (sample index / 4); lane takes word (index & 3).  u in [0,1) with 24 random bits, like
// torch's uniform_ for float.
__device__ __forceinline__ uint32_t tn_mulhi(uint32_t a, uint32_t b) { return __umulhi(a, b); }

__device__ __forceinline__ uint32_t tn_philox_u32(uint64_t seed, uint64_t index);
__device__ __forceinline__ float tn_philox_uniform(uint64_t seed, uint64_t index) {
    return (float)(tn_philox_u32(seed, index) & 0xFFFFFFu) * (1.0f / 16777216.0f);
}
// 32 random bits for counter `index` (the word (index & 3) of Philox block index >> 2)
__device__ __forceinline__ uint32_t tn_philox_u32(uint64_t seed, uint64_t index) {
    uint32_t c0 = (uint32_t)(index >> 2), c1 = (uint32_t)(index >> 34), c2 = 0u, c3 = 0u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    // The seed is a kernel argument (wave-uniform).  Left visible, the ten round keys are loop-invariant scalars that hipcc computes once
    // per kernel and keeps — 18 SGPRs per seed, which the chain kernels do not have: they went to VGPR lanes (v_writelane), i.e. into
    // the vector registers the layer walk needs.  Behind this barrier the keys are formed where they are used (18 s_add per call).
    asm volatile("" : "+s"(k0), "+s"(k1));
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = tn_mulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = tn_mulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n1 = l1, n2 = h0 ^ c3 ^ k1, n3 = l0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const uint32_t w = (uint32_t)(index & 3);
    return w == 0 ? c0 : (w == 1 ? c1 : (w == 2 ? c2 : c3));
}

// ------------------------------------------------------------------------------ sin / cos
// Both sin and cos of the same argument from ONE range reduction, branch-free and call-free (a call inside the MFMA
// kernels would spill the live activation registers): the reduction x - j*pi/2 is done in fp64 (2-term pi/2, exact
// enough for every fp32 argument up to ~1e15), the minimax polynomials on [-pi/4, pi/4] in fp32.
// <= 1.5 ulp / 7.3e-8 absolute against fp64 on 3e7 points up to |x| = 1e9 (checked on the host);
// inf / nan give nan like libm.  The encoder's arguments 2^k * p are exact in fp32.
__device__ __forceinline__ void tn_sincos(float x, float& sn, float& cs) {
    const double xd = (double)x;
    const double jd = rint(xd * 0.63661977236758134308);
    double rd = fma(-jd, 1.57079632679489655800e+00, xd);
    rd = fma(-jd, 6.12323399573676603587e-17, rd);
    const float a = (float)rd;
    const int q = (int)((long long)jd & 3);
    const float s2 = __fmul_rn(a, a);
    float r = 2.86567956e-6f;
    r = fmaf(r, s2, -1.98559923e-4f); r = fmaf(r, s2, 8.33338592e-3f); r = fmaf(r, s2, -1.66666672e-1f);
    const float ps = fmaf(r, __fmul_rn(a, s2), a);
    float pc = 2.44677067e-5f;
    pc = fmaf(pc, s2, -1.38877297e-3f); pc = fmaf(pc, s2, 4.16666567e-2f); pc = fmaf(pc, s2, -0.5f); pc = fmaf(pc, s2, 1.0f);
    float S = (q & 1) ? pc : ps, C = (q & 1) ? ps : pc;
    if (q & 2) S = -S;
    if ((q + 1) & 2) C = -C;
    sn = S; cs = C;
}

// The same arithmetic for N independent arguments, written STAGE BY STAGE: with one wave per SIMD a dependent instruction waits out
// the pipeline latency of its predecessor, and tn_sincos is one serial chain of ~33 instructions (measured in the chain kernels:
// ~240 cycles per argument, 4.3 k cycles per 32-sample tile at L = 6).  N chains advancing together fill those gaps.  Every
// element goes through exactly the operations of tn_sincos: results are bit-identical.
template <int N>
__device__ __forceinline__ void tn_sincos_n(const float (&x)[N], float (&sn)[N], float (&cs)[N]) {
    double xd[N], jd[N], rd[N];
    float a[N], s2[N], r[N], pc[N], ps[N];
    int q[N];
#pragma unroll
    for (int i = 0; i < N; ++i) xd[i] = (double)x[i];
#pragma unroll
    for (int i = 0; i < N; ++i) jd[i] = rint(xd[i] * 0.63661977236758134308);
#pragma unroll
    for (int i = 0; i < N; ++i) rd[i] = fma(-jd[i], 1.57079632679489655800e+00, xd[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) rd[i] = fma(-jd[i], 6.12323399573676603587e-17, rd[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) { a[i] = (float)rd[i]; q[i] = (int)((long long)jd[i] & 3); }
#pragma unroll
    for (int i = 0; i < N; ++i) { s2[i] = __fmul_rn(a[i], a[i]); r[i] = 2.86567956e-6f; pc[i] = 2.44677067e-5f; }
#pragma unroll
    for (int i = 0; i < N; ++i) { r[i] = fmaf(r[i], s2[i], -1.98559923e-4f); pc[i] = fmaf(pc[i], s2[i], -1.38877297e-3f); }
#pragma unroll
    for (int i = 0; i < N; ++i) { r[i] = fmaf(r[i], s2[i], 8.33338592e-3f); pc[i] = fmaf(pc[i], s2[i], 4.16666567e-2f); }
#pragma unroll
    for (int i = 0; i < N; ++i) { r[i] = fmaf(r[i], s2[i], -1.66666672e-1f); pc[i] = fmaf(pc[i], s2[i], -0.5f); }
#pragma unroll
    for (int i = 0; i < N; ++i) { ps[i] = fmaf(r[i], __fmul_rn(a[i], s2[i]), a[i]); pc[i] = fmaf(pc[i], s2[i], 1.0f); }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float S = (q[i] & 1) ? pc[i] : ps[i], C = (q[i] & 1) ? ps[i] : pc[i];
        if (q[i] & 2) S = -S;
        if ((q[i] + 1) & 2) C = -C;
        sn[i] = S; cs[i] = C;
    }
}

// ------------------------------------------------------------------------- sampling arithmetic
// Everything that defines a "sample bin" is rounded op by op exactly like the reference's
// separate ATen kernels (no FMA contraction): sampling.py:25 and :27.
struct SampleArgs {
    const float* ztab;      // [3*S]: z | lo | hi
    const float* t_rand;    // [R*S] or NULL
    uint64_t seed, offset;
    int32_t S;
    int32_t randomized;
    // dataset mode (tnerf_train_step_dataset): the Philox offset of this step is offset + *step * per_step, resolved once
    // per kernel by tn_resolve_step
    const int64_t* step; uint64_t per_step;
};

__device__ __forceinline__ float tn_depth(const SampleArgs& a, int64_t ray, int s) {
    if (!a.randomized) return a.ztab[s];
    const float lo = a.ztab[a.S + s], hi = a.ztab[2 * a.S + s];
    const int64_t idx = ray * a.S + s;
    const float u = a.t_rand ? a.t_rand[idx] : tn_philox_uniform(a.seed, a.offset + (uint64_t)idx);
    return __fadd_rn(lo, __fmul_rn(__fsub_rn(hi, lo), u));
}

__device__ __forceinline__ float tn_point(float o, float d, float z) { return __fadd_rn(o, __fmul_rn(d, z)); }

// --------------------------------------------------------------------------- wave primitives
__device__ __forceinline__ float tn_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float tn_wave_prod(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v *= __shfl_xor(v, o, 64);
    return v;
}
// inclusive product scan over the 64 lanes
__device__ __forceinline__ float tn_wave_scan_mul(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float n = __shfl_up(v, o, 64);
        if (lane >= o) v *= n;
    }
    return v;
}
// inclusive suffix sum: out[l] = sum_{k>=l} v[k]
__device__ __forceinline__ float tn_wave_suffix_sum(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float n = __shfl_down(v, o, 64);
        if (lane + o < 64) v += n;
    }
    return v;
}

// Per-sample compositing terms of volume_render (volume.py:18-34) for one lane.
struct CompTerms { float delta, e, alpha, om; };
__device__ __forceinline__ CompTerms tn_comp_terms(float sigma, float z, float z_next, bool last, float dnorm) {
    CompTerms c;
    const float gap = last ? 1e10f : __fsub_rn(z_next, z);
    c.delta = __fmul_rn(gap, dnorm);
    c.e = expf(__fmul_rn(-sigma, c.delta));
    c.alpha = __fsub_rn(1.0f, c.e);
    c.om = __fadd_rn(__fsub_rn(1.0f, c.alpha), 1e-10f);
    return c;
}

__device__ __forceinline__ float tn_norm3(float x, float y, float z) {
    return sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
}

// Address of (row 0, sample m) in the block-major training stash (layout: tnerf_internal.h).
__device__ __forceinline__ float* tn_stash_at(float* stash, int64_t rows, int64_t m) {
    return stash + ((m >> 5) * rows) * 32 + (m & 31);
}
__device__ __forceinline__ const float* tn_stash_at(const float* stash, int64_t rows, int64_t m) {
    return stash + ((m >> 5) * rows) * 32 + (m & 31);
}

// The stash's pipe tag (tnerf_internal.h TNB_TAG): true if it is `want`; otherwise the tag is marked BAD (whoever consumes the stash
// next sees the mismatch too) and the caller returns before doing anything.
__device__ __forceinline__ bool tn_stash_tag_is(float* stash, const MlpLayout& L, int64_t Mp, unsigned want) {
    unsigned* w = reinterpret_cast<unsigned*>(stash + TN_BOUND_OFF(L, Mp)) + TNB_TAG;
    const unsigned tag = __builtin_nontemporal_load(w);
    if (tag == want) return true;
    if (blockIdx.x == 0 && threadIdx.x == 0) *w = TN_TAG_BAD;
    return false;
}
__global__ void k_stash_tag(unsigned* __restrict__ w, unsigned tag);      // mlp_fwd.hip

// ---------------------------------------------------------------------------------------- LDS-DMA
// global_load_lds_dwordx4: 64 lanes x 16 B from (wave-uniform base + per-lane 32-bit offset) straight into LDS at
// lds_dst + lane * 16 (wave-uniform destination, lane-linear image), no VGPR round trip.  Issued from inline asm: hipcc
// neither counts it in its s_waitcnt bookkeeping nor drains it at barriers — the caller waits with a counted vmcnt and
// publishes with a barrier.  M0 (the DMA's LDS base) is compiler-reserved: saved and restored around the load.
__device__ __forceinline__ void tn_glds16(const void* src, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(src), "s"(lds_dst) : "memory");
}

// ------------------------------------------------------------------------------------- ray source
// Where a fused kernel gets ray r from:
//   tables : rays_o / rays_d [n,3] (what get_rays precomputed, reference src/train.py:94-101), row = index ? index[r] : r
//   camera : c2w != NULL -> the ray of flat pixel p = index ? index[r] : first + r is generated in the kernel with
//            exactly the arithmetic of k_get_rays (reference src/rays.py:15-32): no (N,HW,3) tables, no gathers.
//   dataset: step != NULL (tnerf_train_step_dataset) -> c2w is the pose TABLE [n_images,16]; tn_resolve_step picks pose
//            step % n_images and ray r trains on pixel Philox(seed, step * rays_global + ray_first + r) % (H*W) — the draw
//            the reference makes with torch.randint (src/train.py:108-109), taken in the kernel.
struct RaySource {
    const float* rays_o; const float* rays_d; const int64_t* index;
    const float* c2w; int64_t first; int32_t H, W; float focal;
    const int64_t* step; int32_t n_images; int32_t image; uint64_t seed; int64_t ray_first, rays_global;
};
#define TN_PIX_KEY 0x9E3779B97F4A7C15ull      // the pixel draw and the jitter draw use different Philox keys

// The null checks of these wave-uniform pointers sit inside the chain kernels' tile loops.  Left visible, hipcc evaluates each once per
// kernel and keeps the RESULT as a 64-bit lane mask: two SGPRs per check on top of the pointer, in kernels whose SGPRs already spill
// into VGPR lanes (the registers the layer walk needs).  Called on the local copies at the top of the loop, the checks are redone there.
__device__ __forceinline__ void tn_opaque_sources(RaySource& rs, SampleArgs& sa) {
    asm volatile("" : "+s"(rs.rays_o), "+s"(rs.index), "+s"(rs.c2w), "+s"(rs.step), "+s"(sa.t_rand));
}

// Dataset mode: read the device step counter ONCE and bind this launch's image / Philox counters (local copies).
__device__ __forceinline__ int64_t tn_uniform64(int64_t v) {      // a wave-uniform value, moved to SGPRs
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
// (Call it on LOCAL copies of the two small structs: writing into the kernel-argument block itself would turn the whole
// block — with its runtime-indexed layout tables — into a scratch alloca.)
__device__ __forceinline__ void tn_resolve_step(RaySource& rs, SampleArgs& sa) {
    if (rs.step) {
        const int64_t s = tn_uniform64(*rs.step);
        rs.image = (int32_t)(s % rs.n_images);
        rs.c2w += 16 * rs.image;
        rs.first = s * rs.rays_global + rs.ray_first;           // Philox counter of ray 0 of this rank
    }
    if (sa.step) sa.offset += (uint64_t)tn_uniform64(*sa.step) * sa.per_step;
}

__device__ __forceinline__ void tn_pixel_ray(const float* __restrict__ c2w, int H, int W, float focal, int64_t p,
                                             float (&o)[3], float (&d)[3]) {
    const int col = (int)(p % W), row = (int)(p / W);
    const float cx = __fdiv_rn(__fsub_rn((float)col, (float)(W * 0.5)), focal);
    const float cy = __fdiv_rn(-__fsub_rn((float)row, (float)(H * 0.5)), focal);
    const float cz = -1.0f;
    float w[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float acc = __fmul_rn(cx, c2w[4 * c + 0]);
        acc = fmaf(cy, c2w[4 * c + 1], acc);
        acc = fmaf(cz, c2w[4 * c + 2], acc);
        w[c] = acc;
    }
    float s = __fmul_rn(w[0], w[0]);
    s = fmaf(w[1], w[1], s);
    s = fmaf(w[2], w[2], s);
    const float n = fmaxf(sqrtf(s), 1e-12f);
#pragma unroll
    for (int c = 0; c < 3; ++c) { d[c] = __fdiv_rn(w[c], n); o[c] = c2w[4 * c + 3]; }
}

// Flat pixel index of ray r (camera / dataset sources).
__device__ __forceinline__ int64_t tn_ray_pixel(const RaySource& rs, int64_t r) {
    if (rs.step) return (int64_t)(tn_philox_u32(rs.seed ^ TN_PIX_KEY, (uint64_t)(rs.first + r)) % (uint32_t)(rs.H * rs.W));
    return rs.index ? rs.index[r] : rs.first + r;
}

// (Both sources fill SCALARS and the arrays are written once behind the branch: with stores to o[c] / d[c] in both branches hipcc
// merges them into one store with a run-time index, which keeps the two small arrays in scratch memory — 12 bytes of stack in
// every kernel that fetches a ray.)
__device__ __forceinline__ void tn_fetch_ray(const RaySource& rs, int64_t r, float (&o)[3], float (&d)[3]) {
    float o0, o1, o2, d0, d1, d2;
    if (rs.c2w) {
        float po[3], pd[3];
        tn_pixel_ray(rs.c2w, rs.H, rs.W, rs.focal, tn_ray_pixel(rs, r), po, pd);
        o0 = po[0]; o1 = po[1]; o2 = po[2]; d0 = pd[0]; d1 = pd[1]; d2 = pd[2];
    } else {
        const int64_t i = rs.index ? rs.index[r] : r;
        o0 = rs.rays_o[3 * i]; o1 = rs.rays_o[3 * i + 1]; o2 = rs.rays_o[3 * i + 2];
        d0 = rs.rays_d[3 * i]; d1 = rs.rays_d[3 * i + 1]; d2 = rs.rays_d[3 * i + 2];
    }
    o[0] = o0; o[1] = o1; o[2] = o2; d[0] = d0; d[1] = d1; d[2] = d2;
}

// ------------------------------------------------------------------------------ loss folded into the forward
// MSE of the train step (reference src/train.py:122) evaluated where the composited colour is produced: the ray's wave
// writes  ray_ws[4r + c] = dL/dC_c = 2 (C_c - target_c) / denom  and  ray_ws[4r + 3] = sum_c (C_c - target_c)^2;
// the finishing kernel sums the fourth column in a fixed order into the loss.
struct LossArgs {
    const float* target;            // table source: [R,3] (or [.,3] indexed by target_index); camera: this image's [H*W,3]; dataset: [n_images,H*W,3]
    const int64_t* target_index;    // NULL: row = ray (tables) / the ray's pixel (camera, dataset)
    float inv_denom;
    float* ray_ws;                  // [R,4]; NULL = no loss (inference / autograd path)
    int32_t* pix_out;               // [R] or NULL: the pixel each ray trained on (dataset mode; lets a test replay the step)
};
__device__ __forceinline__ void tn_ray_loss(const LossArgs& la, const RaySource& rs, int64_t ray, float cr, float cg, float cb) {
    int64_t row = ray;
    if (rs.c2w) {
        const int64_t p = tn_ray_pixel(rs, ray);
        row = rs.step ? (int64_t)rs.image * rs.H * rs.W + p : p;
        if (la.pix_out) la.pix_out[ray] = (int32_t)p;
    } else if (la.target_index) {
        row = la.target_index[ray];
    }
    const float d0 = cr - la.target[3 * row], d1 = cg - la.target[3 * row + 1], d2 = cb - la.target[3 * row + 2];
    f32x4 w;
    w[0] = (2.0f * d0) * la.inv_denom; w[1] = (2.0f * d1) * la.inv_denom; w[2] = (2.0f * d2) * la.inv_denom;
    w[3] = (d0 * d0 + d1 * d1) + d2 * d2;
    *reinterpret_cast<f32x4*>(la.ray_ws + 4 * ray) = w;
}
