// Microtest: how does v_mfma_f32_32x32x16_bf16 round D = A.B + C?  One non-zero product p added to C, for p a fraction of ulp(C):
// IEEE round-to-nearest-even would give the same results as the fp32 expression C + p; a truncating adder would not.
// Also: many tiny products whose SUM is above half an ulp although each is below it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;
static u16 bf(float f) { unsigned u; memcpy(&u, &f, 4); return (u16)(u >> 16); }     // exact for the values used here
struct Case { float c, a[16], b[16]; };
__global__ void k(const Case* cs, int n, float* out, float* out32) {
    const int lane = threadIdx.x;
    for (int i = 0; i < n; ++i) {
        // output element (row 0, col 0): A row 0 = lane 0 (k 0..7) and lane 32 (k 8..15); B col 0 likewise
        u16 av[8] = {0,0,0,0,0,0,0,0}, bv[8] = {0,0,0,0,0,0,0,0};
        if (lane == 0 || lane == 32) for (int j = 0; j < 8; ++j) {
            const int kk = (lane >> 5) * 8 + j;
            unsigned ua, ub; memcpy(&ua, &cs[i].a[kk], 4); memcpy(&ub, &cs[i].b[kk], 4);
            av[j] = (u16)(ua >> 16); bv[j] = (u16)(ub >> 16);
        }
        bf16x8 A, B; memcpy(&A, av, 16); memcpy(&B, bv, 16);
        f32x16 C;
        for (int r = 0; r < 16; ++r) C[r] = 0.f;
        if (lane == 0) C[0] = cs[i].c;
        f32x16 D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0);
        if (lane == 0) out[i] = D[0];
        // the same sum on the fp32 MFMA (32x32x2: k = 0 on lanes 0..31, k = 1 on lanes 32..63), 8 instructions
        f32x16 E = C;
        for (int s = 0; s < 8; ++s) {
            float a1 = 0.f, b1 = 0.f;
            if (lane == 0)  { a1 = cs[i].a[2 * s];     b1 = cs[i].b[2 * s]; }
            if (lane == 32) { a1 = cs[i].a[2 * s + 1]; b1 = cs[i].b[2 * s + 1]; }
            E = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, E, 0, 0, 0);
        }
        if (lane == 0) out32[i] = E[0];
    }
}
int main() {
    std::vector<Case> cs;
    auto one = [&](float c, float a, float b) { Case x{}; x.c = c; x.a[0] = a; x.b[0] = b; cs.push_back(x); };
    const float e = 5.9604644775390625e-08f;   // 2^-24 = half an ulp of 1.0
    one(1.0f, 0.000244140625f, 1.5f * 0.000244140625f);      // + 1.5 * 2^-24 : RNE -> 1 + 2^-23
    one(1.0f, 0.000244140625f, 0.75f * 0.000244140625f);     // + 0.75 * 2^-24: RNE -> 1.0
    one(1.0f, 0.000244140625f, 1.0f * 0.000244140625f);      // + exactly half an ulp: RNE -> 1.0 (even)
    one(1.0f, 0.000244140625f, 3.0f * 0.000244140625f);      // + 1.5 ulp: RNE -> 1 + 2 ulp (even)
    one(-1.0f, 0.000244140625f, -1.5f * 0.000244140625f);    // mirror
    one(1.0f, 0.000244140625f, -1.25f * 0.000244140625f);    // 1 - 1.25 * 2^-24 (ulp below 1 is 2^-24): RNE -> 1 - 2^-24
    one(1.0f, 0.000244140625f, -0.75f * 0.000244140625f);    // 1 - 0.75 * 2^-24: RNE -> 1 - 2^-24
    { Case x{}; x.c = 1.0f; for (int j = 0; j < 16; ++j) { x.a[j] = 0.000244140625f; x.b[j] = 0.125f * 0.000244140625f; } cs.push_back(x); }   // 16 x 0.125 * 2^-24 = 2 * 2^-24 = one ulp: exact sum -> 1 + 2^-23
    { Case x{}; x.c = 1.0f; for (int j = 0; j < 16; ++j) { x.a[j] = 0.000244140625f; x.b[j] = 0.09375f * 0.000244140625f; } cs.push_back(x); } // 16 x 0.09375 = 1.5 * 2^-24 -> RNE 1 + 2^-23
    { Case x{}; x.c = 1024.0f; for (int j = 0; j < 16; ++j) { x.a[j] = 0.0078125f; x.b[j] = 0.5f * 0.0078125f; } cs.push_back(x); }          // 1024 + 16 * 2^-15 = 1024 + 2^-11 (ulp(1024) = 2^-13): exact
    (void)e;
    Case* d; float *o, *o32; hipMalloc(&d, cs.size() * sizeof(Case)); hipMalloc(&o, cs.size() * 4); hipMalloc(&o32, cs.size() * 4);
    hipMemcpy(d, cs.data(), cs.size() * sizeof(Case), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, (int)cs.size(), o, o32);
    std::vector<float> h(cs.size()), h32(cs.size()); hipMemcpy(h.data(), o, cs.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(h32.data(), o32, cs.size() * 4, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < cs.size(); ++i) {
        double exact = cs[i].c; for (int j = 0; j < 16; ++j) exact += (double)cs[i].a[j] * cs[i].b[j];
        const float rne = (float)exact;
        unsigned ub, u3, ur; memcpy(&ub, &h[i], 4); memcpy(&u3, &h32[i], 4); memcpy(&ur, &rne, 4);
        printf("case %2zu: bf16 MFMA %.9g (0x%08x)  fp32 MFMA %.9g (0x%08x)  correctly rounded %.9g (0x%08x)  %s %s\n", i, h[i], ub, h32[i], u3, rne, ur,
               ub == ur ? "bf16:ok" : "bf16:DIFFERS", u3 == ur ? "fp32:ok" : "fp32:differs");
    }
    return 0;
}
