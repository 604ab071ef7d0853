// Microbenchmark: raw input -> output pairs of v_mfma_f32_32x32x16_{bf16,f16} for fitting a bit-exact model of the matrix pipe's
// accumulation offline (tools/microbench/mfma_model.py).  One wave per case block: D = A(32x16) . B(16x32) + C(32x32).
//   mfma_dump <bf16|f16> <in.bin> <out.bin>
// in.bin : int32 n, then per block A[32][16] u16 (row m, k), B[16][32] u16 (k, col n), C[32][32] f32 (m, n)
// out.bin: per block D[32][32] f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;
struct Block { u16 A[32][16]; u16 B[16][32]; float C[32][32]; };
template <bool F16>
__global__ void k(const Block* in, float* out) {
    const Block& b = in[blockIdx.x];
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    u16 av[8], bv[8];
    for (int i = 0; i < 8; ++i) { av[i] = b.A[j][8 * h + i]; bv[i] = b.B[8 * h + i][j]; }
    f32x16 C, D;
    for (int r = 0; r < 16; ++r) C[r] = b.C[(r & 3) + 8 * (r >> 2) + 4 * h][j];
    if (F16) { f16x8 A_, B_; memcpy(&A_, av, 16); memcpy(&B_, bv, 16); D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, C, 0, 0, 0); }
    else     { bf16x8 A_, B_; memcpy(&A_, av, 16); memcpy(&B_, bv, 16); D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, C, 0, 0, 0); }
    float* o = out + (size_t)blockIdx.x * 1024;
    for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = D[r];
}
int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: mfma_dump <bf16|f16> in.bin out.bin\n"); return 2; }
    const bool f16 = !strcmp(argv[1], "f16");
    FILE* f = fopen(argv[2], "rb"); if (!f) { perror(argv[2]); return 1; }
    int n = 0; if (fread(&n, 4, 1, f) != 1 || n <= 0) return 1;
    std::vector<Block> blocks(n);
    if (fread(blocks.data(), sizeof(Block), n, f) != (size_t)n) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);
    Block* d; float* o;
    if (hipMalloc(&d, sizeof(Block) * n) != hipSuccess || hipMalloc(&o, (size_t)n * 4096) != hipSuccess) return 1;
    hipMemcpy(d, blocks.data(), sizeof(Block) * n, hipMemcpyHostToDevice);
    if (f16) hipLaunchKernelGGL(k<true>, dim3(n), dim3(64), 0, 0, d, o); else hipLaunchKernelGGL(k<false>, dim3(n), dim3(64), 0, 0, d, o);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    std::vector<float> out((size_t)n * 1024);
    hipMemcpy(out.data(), o, out.size() * 4, hipMemcpyDeviceToHost);
    f = fopen(argv[3], "wb"); fwrite(out.data(), 4, out.size(), f); fclose(f);
    printf("%s: %d blocks\n", argv[1], n);
    return 0;
}
