// TinyNeRF.forward / its backward LAYER BY LAYER, for what the register-resident chain kernels do not cover (reference src/nerf.py:10
// takes any widths): hidden > 256, in_dim > 64, and — for any shape — the gradient w.r.t. the network input (which the reference's
// training never asks for: its points carry no grad, train.py:114-117, but nerf.py:29-41 is differentiable in x).
//
// This is not the hot path: every layer is ONE plain library SGEMM (hipBLAS -> rocBLAS, fp32, atomics off: results
// are deterministic) between small elementwise kernels, activations make a round trip through HBM per layer.  Row-major tensors
// as the reference has them: x [M, in_dim], W_l [out, in] (nn.Linear), H_l [M, hidden].
//   forward  : Z_l = A_l W_l[:, :K]^T (+ X W_l[:, K:]^T for the skip layer, cat([h, x]) nerf.py:37-38), H_l = relu(Z_l + b_l)
//              rgb = sigmoid(H W_rgb^T + b), sigma = relu(H W_sigma^T + b)                                  nerf.py:39-40
//   backward : dZ_head, dH = dZ_rgb W_rgb + dZ_sigma W_sigma, then per layer dW_l = dZ_l^T A_l, db_l = dZ_l^T 1,
//              dA_l = dZ_l W_l[:, :K], dX += dZ_l W_l[:, K:], dZ_{l-1} = dA_l * (H_{l-1} > 0)
// The hipBLAS handle is the CALLER's (torch.cuda.current_blas_handle() in the Python binding): the library creates none, so it
// still allocates no device memory.  The hipBLAS entry points are bound on first use — to the copy already in the process (blas()).
#include <hip/hip_runtime.h>
#include <hipblas/hipblas.h>
#include <dlfcn.h>
#include <link.h>
#include <string.h>
#include <mutex>
#include "../../include/tnerf.h"
#include "tnerf_internal.h"
#include "dev_common.hpp"

namespace {
typedef hipblasStatus_t (*fn_sgemm)(hipblasHandle_t, hipblasOperation_t, hipblasOperation_t, int, int, int, const float*, const float*, int,
                                    const float*, int, const float*, float*, int);
typedef hipblasStatus_t (*fn_set_stream)(hipblasHandle_t, hipStream_t);
typedef hipblasStatus_t (*fn_get_stream)(hipblasHandle_t, hipStream_t*);
typedef hipblasStatus_t (*fn_set_atomics)(hipblasHandle_t, hipblasAtomicsMode_t);
typedef hipblasStatus_t (*fn_get_atomics)(hipblasHandle_t, hipblasAtomicsMode_t*);
typedef hipblasStatus_t (*fn_set_pm)(hipblasHandle_t, hipblasPointerMode_t);
typedef hipblasStatus_t (*fn_get_pm)(hipblasHandle_t, hipblasPointerMode_t*);
struct Blas { void* h; fn_sgemm sgemm; fn_set_stream set_stream; fn_get_stream get_stream; fn_set_atomics set_atomics; fn_get_atomics get_atomics;
              fn_set_pm set_pm; fn_get_pm get_pm; };

Blas* blas() {
    static Blas b{};
    static std::once_flag once;
    std::call_once(once, [] {
        // The handle is the CALLER's: the functions must come from the copy of hipBLAS that created it.  A torch wheel ships its own
        // libhipblas under torch/lib (its SONAME need not be the system's), so first look for a copy that is ALREADY mapped into the
        // process (dl_iterate_phdr: any object whose file name contains "libhipblas") and bind to that one (RTLD_NOLOAD: no second copy
        // is ever loaded beside it); only a process that has none yet loads one by name.
        struct Found { char path[1024]; } found{};
        dl_iterate_phdr([](struct dl_phdr_info* info, size_t, void* out) -> int {
            if (info->dlpi_name && strstr(info->dlpi_name, "libhipblas") && !strstr(info->dlpi_name, "libhipblaslt")) {
                strncpy(static_cast<Found*>(out)->path, info->dlpi_name, sizeof(Found::path) - 1);
                return 1;
            }
            return 0;
        }, &found);
        if (found.path[0]) b.h = dlopen(found.path, RTLD_NOW | RTLD_NOLOAD);
        if (!b.h) {
            const char* names[] = {"libhipblas.so", "libhipblas.so.3", "/opt/rocm/lib/libhipblas.so"};
            for (const char* n : names) { b.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (b.h) break; }
        }
        if (b.h) {
            b.sgemm = (fn_sgemm)dlsym(b.h, "hipblasSgemm");
            b.set_stream = (fn_set_stream)dlsym(b.h, "hipblasSetStream");
            b.get_stream = (fn_get_stream)dlsym(b.h, "hipblasGetStream");
            b.set_atomics = (fn_set_atomics)dlsym(b.h, "hipblasSetAtomicsMode");
            b.get_atomics = (fn_get_atomics)dlsym(b.h, "hipblasGetAtomicsMode");
            b.set_pm = (fn_set_pm)dlsym(b.h, "hipblasSetPointerMode");
            b.get_pm = (fn_get_pm)dlsym(b.h, "hipblasGetPointerMode");
        }
    });
    if (!b.h || !b.sgemm || !b.set_stream || !b.get_stream || !b.set_atomics || !b.get_atomics || !b.set_pm || !b.get_pm) {
        tn_set_error("hipBLAS (libhipblas.so) could not be loaded: %s", dlerror()); return nullptr; }
    return &b;
}

// The caller's handle, borrowed for one call: stream, atomics mode and pointer mode set for it and put back afterwards.
struct Borrow {
    Blas* b; hipblasHandle_t h; hipStream_t old_stream; hipblasAtomicsMode_t old_atomics; hipblasPointerMode_t old_pm; bool ok;
    Borrow(Blas* b_, void* handle, hipStream_t s) : b(b_), h((hipblasHandle_t)handle), ok(false) {
        if (b->get_stream(h, &old_stream) || b->get_atomics(h, &old_atomics) || b->get_pm(h, &old_pm)) return;
        ok = !b->set_stream(h, s) && !b->set_atomics(h, HIPBLAS_ATOMICS_NOT_ALLOWED) && !b->set_pm(h, HIPBLAS_POINTER_MODE_HOST);
    }
    ~Borrow() { if (ok) { b->set_stream(h, old_stream); b->set_atomics(h, old_atomics); b->set_pm(h, old_pm); } }
};

struct Shape { int in_dim, hidden, depth, skip; };

int check_shape(const char* who, const tnerf_mlp_desc* d, int64_t M, Shape* s) {
    if (!d || d->in_dim < 1 || d->hidden < 1 || d->depth < 1 || d->depth > 64 || d->in_dim > 4096 || d->hidden > 4096 || d->skip_at < 0 ||
        d->skip_at >= d->depth || M < 1) {
        tn_set_error("%s: in_dim=%d hidden=%d depth=%d skip_at=%d rows=%lld (1 <= in_dim, hidden <= 4096, depth <= 64, 0 <= skip_at < depth)", who,
                     d ? d->in_dim : 0, d ? d->hidden : 0, d ? d->depth : 0, d ? d->skip_at : 0, (long long)M);
        return TNERF_EINVAL;
    }
    if (M * (int64_t)(d->hidden + d->in_dim) >= ((int64_t)1 << 31)) {
        tn_set_error("%s: %lld rows x %d columns exceed the 32-bit extents of one SGEMM; split the batch", who, (long long)M, d->hidden + d->in_dim);
        return TNERF_EUNSUPPORTED;
    }
    *s = Shape{d->in_dim, d->hidden, d->depth, d->skip_at};
    return TNERF_OK;
}
inline int in_width(const Shape& s, int l) { return l == 0 ? s.in_dim : (l == s.skip ? s.hidden + s.in_dim : s.hidden); }

// row-major C[M,N] (ldc) = alpha-less  A[M,K] (lda) . B[N,K]^T (ldb) + beta C
int gemm_nt(Blas* b, hipblasHandle_t h, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float beta, float* Cm, int ldc) {
    const float one = 1.0f;
    return (int)b->sgemm(h, HIPBLAS_OP_T, HIPBLAS_OP_N, N, M, K, &one, B, ldb, A, lda, &beta, Cm, ldc);
}
// row-major C[M,K] (ldc) = A[M,N] (lda) . B[N,K] (ldb) + beta C
int gemm_nn(Blas* b, hipblasHandle_t h, int M, int K, int N, const float* A, int lda, const float* B, int ldb, float beta, float* Cm, int ldc) {
    const float one = 1.0f;
    return (int)b->sgemm(h, HIPBLAS_OP_N, HIPBLAS_OP_N, K, M, N, &one, B, ldb, A, lda, &beta, Cm, ldc);
}
// row-major C[N,K] (ldc) = A[M,N]^T (lda) . B[M,K] (ldb)          (contraction over the M rows)
int gemm_tn(Blas* b, hipblasHandle_t h, int N, int K, int M, const float* A, int lda, const float* B, int ldb, float* Cm, int ldc) {
    const float one = 1.0f, zero = 0.0f;
    return (int)b->sgemm(h, HIPBLAS_OP_N, HIPBLAS_OP_T, K, N, M, &one, B, ldb, A, lda, &zero, Cm, ldc);
}

template <int ACT>   // 0: relu, 1: sigmoid
__global__ __launch_bounds__(256) void k_bias_act(float* __restrict__ z, const float* __restrict__ bias, int64_t n, int N) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = z[i] + bias[(int)(i % N)];
        z[i] = ACT == 0 ? fmaxf(v, 0.0f) : 1.0f / (1.0f + expf(-v));
    }
}
// dZ_rgb = d_rgb rgb (1 - rgb), dZ_sigma = d_sigma [sigma > 0]                     (what autograd derives from nerf.py:39-40)
__global__ __launch_bounds__(256) void k_head_bwd(const float* __restrict__ d_rgb, const float* __restrict__ d_sigma, const float* __restrict__ rgb,
                                                  const float* __restrict__ sigma, float* __restrict__ dzr, float* __restrict__ dzs, int64_t M) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < 4 * M; i += (int64_t)gridDim.x * 256) {
        if (i < 3 * M) { const float r = rgb[i]; dzr[i] = d_rgb[i] * (r * (1.0f - r)); }
        else { const int64_t m = i - 3 * M; dzs[m] = sigma[m] > 0.0f ? d_sigma[m] : 0.0f; }
    }
}
__global__ __launch_bounds__(256) void k_relu_mask(float* __restrict__ dh, const float* __restrict__ h, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dh[i] = h[i] > 0.0f ? dh[i] : 0.0f;
}
__global__ __launch_bounds__(256) void k_fill(float* __restrict__ p, float v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}
inline unsigned grid_for(int64_t n) { const int64_t g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }

#define TN_BLAS(expr, who) do { const int rc_ = (expr); if (rc_) { tn_set_error("%s: hipblasSgemm failed with status %d", who, rc_); return 20000 + rc_; } } while (0)
}  // namespace

// HOST.  Floats of the activation buffer H_0 .. H_{depth-1} ([n_rows, hidden] each) the generic forward fills and the backward reads.
extern "C" int64_t tnerf_mlp_generic_acts_floats(const tnerf_mlp_desc* d, int64_t n_rows) {
    Shape s; if (check_shape("tnerf_mlp_generic_acts_floats", d, n_rows, &s)) return TNERF_EINVAL;
    return (int64_t)s.depth * s.hidden * n_rows;
}
// HOST.  Floats of the backward's scratch: two [n_rows, hidden] gradient buffers, the head gradients [n_rows, 4], a ones vector.
extern "C" int64_t tnerf_mlp_generic_scratch_floats(const tnerf_mlp_desc* d, int64_t n_rows) {
    Shape s; if (check_shape("tnerf_mlp_generic_scratch_floats", d, n_rows, &s)) return TNERF_EINVAL;
    return 2 * (int64_t)s.hidden * n_rows + 5 * n_rows;
}

extern "C" int tnerf_mlp_fwd_generic(const tnerf_mlp_desc* d, void* blas_handle, const float* const* params, const float* x, int64_t M,
                                     float* rgb, float* sigma, float* acts, int64_t acts_floats, tnerf_stream_t stream_) {
    const char* who = "tnerf_mlp_fwd_generic";
    Shape s; int rc = check_shape(who, d, M, &s); if (rc) return rc;
    if (!blas_handle || !params || !x || !rgb || !sigma || !acts) { tn_set_error("%s: NULL argument", who); return TNERF_EINVAL; }
    if (acts_floats < (int64_t)s.depth * s.hidden * M) {
        tn_set_error("%s: activation buffer of %lld floats, tnerf_mlp_generic_acts_floats() = %lld", who, (long long)acts_floats, (long long)s.depth * s.hidden * M);
        return TNERF_ESMALL; }
    Blas* b = blas(); if (!b) return TNERF_EUNSUPPORTED;
    hipStream_t stream = (hipStream_t)stream_;
    Borrow br(b, blas_handle, stream);
    if (!br.ok) { tn_set_error("%s: the hipBLAS handle could not be configured", who); return TNERF_EINVAL; }
    const int Mi = (int)M, H = s.hidden;
    for (int l = 0; l < s.depth; ++l) {
        const float* W = params[2 * l]; const float* bias = params[2 * l + 1];
        float* out = acts + (int64_t)l * H * M;
        const int ldw = in_width(s, l);
        const float* A = l == 0 ? x : acts + (int64_t)(l - 1) * H * M;
        const int K = l == 0 ? s.in_dim : H;
        TN_BLAS(gemm_nt(b, br.h, Mi, H, K, A, K, W, ldw, 0.0f, out, H), who);
        if (l > 0 && l == s.skip) TN_BLAS(gemm_nt(b, br.h, Mi, H, s.in_dim, x, s.in_dim, W + H, ldw, 1.0f, out, H), who);      // cat([h, x]): the x columns follow
        hipLaunchKernelGGL(k_bias_act<0>, dim3(grid_for(M * H)), dim3(256), 0, stream, out, bias, M * H, H);
        TN_HIP_CHECK_LAUNCH(who);
    }
    const float* last = acts + (int64_t)(s.depth - 1) * H * M;
    TN_BLAS(gemm_nt(b, br.h, Mi, 1, H, last, H, params[2 * s.depth], H, 0.0f, sigma, 1), who);
    hipLaunchKernelGGL(k_bias_act<0>, dim3(grid_for(M)), dim3(256), 0, stream, sigma, params[2 * s.depth + 1], M, 1);
    TN_HIP_CHECK_LAUNCH(who);
    TN_BLAS(gemm_nt(b, br.h, Mi, 3, H, last, H, params[2 * s.depth + 2], H, 0.0f, rgb, 3), who);
    hipLaunchKernelGGL(k_bias_act<1>, dim3(grid_for(3 * M)), dim3(256), 0, stream, rgb, params[2 * s.depth + 3], 3 * M, 3);
    TN_HIP_CHECK_LAUNCH(who);
    return TNERF_OK;
}

extern "C" int tnerf_mlp_bwd_generic(const tnerf_mlp_desc* d, void* blas_handle, const float* const* params, const float* x, int64_t M,
                                     const float* rgb, const float* sigma, const float* d_rgb, const float* d_sigma, const float* acts,
                                     int64_t acts_floats, float* scratch, int64_t scratch_floats, float* const* grads, float* dx,
                                     tnerf_stream_t stream_) {
    const char* who = "tnerf_mlp_bwd_generic";
    Shape s; int rc = check_shape(who, d, M, &s); if (rc) return rc;
    if (!blas_handle || !params || !x || !rgb || !sigma || !d_rgb || !d_sigma || !acts || !scratch || !grads) { tn_set_error("%s: NULL argument", who); return TNERF_EINVAL; }
    const int H = s.hidden, Mi = (int)M;
    if (acts_floats < (int64_t)s.depth * H * M || scratch_floats < 2 * (int64_t)H * M + 5 * M) {
        tn_set_error("%s: activations %lld floats (need %lld), scratch %lld floats (need %lld)", who, (long long)acts_floats, (long long)s.depth * H * M,
                     (long long)scratch_floats, (long long)(2 * (int64_t)H * M + 5 * M));
        return TNERF_ESMALL; }
    Blas* b = blas(); if (!b) return TNERF_EUNSUPPORTED;
    hipStream_t stream = (hipStream_t)stream_;
    Borrow br(b, blas_handle, stream);
    if (!br.ok) { tn_set_error("%s: the hipBLAS handle could not be configured", who); return TNERF_EINVAL; }
    float* dz[2] = {scratch, scratch + (int64_t)H * M};
    float* dzr = scratch + 2 * (int64_t)H * M; float* dzs = dzr + 3 * M; float* ones = dzs + M;
    hipLaunchKernelGGL(k_fill, dim3(grid_for(M)), dim3(256), 0, stream, ones, 1.0f, M);
    hipLaunchKernelGGL(k_head_bwd, dim3(grid_for(4 * M)), dim3(256), 0, stream, d_rgb, d_sigma, rgb, sigma, dzr, dzs, M);
    TN_HIP_CHECK_LAUNCH(who);
    const float* last = acts + (int64_t)(s.depth - 1) * H * M;
    const int D = s.depth;
    // heads: parameter gradients, then dH of the last layer
    TN_BLAS(gemm_tn(b, br.h, 1, H, Mi, dzs, 1, last, H, grads[2 * D], H), who);
    TN_BLAS(gemm_tn(b, br.h, 1, 1, Mi, dzs, 1, ones, 1, grads[2 * D + 1], 1), who);
    TN_BLAS(gemm_tn(b, br.h, 3, H, Mi, dzr, 3, last, H, grads[2 * D + 2], H), who);
    TN_BLAS(gemm_tn(b, br.h, 3, 1, Mi, dzr, 3, ones, 1, grads[2 * D + 3], 1), who);
    int cur = 0;
    TN_BLAS(gemm_nn(b, br.h, Mi, H, 1, dzs, 1, params[2 * D], H, 0.0f, dz[cur], H), who);
    TN_BLAS(gemm_nn(b, br.h, Mi, H, 3, dzr, 3, params[2 * D + 2], H, 1.0f, dz[cur], H), who);
    bool dx_started = false;
    for (int l = D - 1; l >= 0; --l) {
        const float* Hl = acts + (int64_t)l * H * M;
        hipLaunchKernelGGL(k_relu_mask, dim3(grid_for(M * H)), dim3(256), 0, stream, dz[cur], Hl, M * H);        // dZ_l = dH_l [H_l > 0]
        TN_HIP_CHECK_LAUNCH(who);
        const float* W = params[2 * l]; const int ldw = in_width(s, l);
        const float* A = l == 0 ? x : acts + (int64_t)(l - 1) * H * M;
        const int K = l == 0 ? s.in_dim : H;
        TN_BLAS(gemm_tn(b, br.h, H, K, Mi, dz[cur], H, A, K, grads[2 * l], ldw), who);                            // dW_l[:, :K]
        if (l > 0 && l == s.skip) TN_BLAS(gemm_tn(b, br.h, H, s.in_dim, Mi, dz[cur], H, x, s.in_dim, grads[2 * l] + H, ldw), who);
        TN_BLAS(gemm_tn(b, br.h, H, 1, Mi, dz[cur], H, ones, 1, grads[2 * l + 1], 1), who);                       // db_l
        if (dx && (l == 0 || l == s.skip)) {
            const float* Wx = l == 0 ? W : W + H;
            TN_BLAS(gemm_nn(b, br.h, Mi, s.in_dim, H, dz[cur], H, Wx, ldw, dx_started ? 1.0f : 0.0f, dx, s.in_dim), who);
            dx_started = true;
        }
        if (l > 0) {
            TN_BLAS(gemm_nn(b, br.h, Mi, H, H, dz[cur], H, W, ldw, 0.0f, dz[cur ^ 1], H), who);                   // dH_{l-1}
            cur ^= 1;
        }
    }
    return TNERF_OK;
}
