// Confirms the operand layout of v_mfma_f32_32x32x16_bf16 on gfx950: lane l supplies A[row = l%32][k = 8*(l/32)+i] and
// B[k = 8*(l/32)+i][col = l%32], i = 0..7; D[row = (r&3) + 8*(r>>2) + 4*(l/32)][col = l%32] in register r.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, c = l & 31, h = l >> 5;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)A[c * 16 + 8 * h + i]; b[i] = (__bf16)B[(8 * h + i) * 32 + c]; }
    f32x16 d;
    for (int r = 0; r < 16; ++r) d[r] = 0.f;
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + c] = d[r];
}
int main() {
    float hA[32 * 16], hB[16 * 32], hD[32 * 32], ref[32 * 32];
    for (int i = 0; i < 512; ++i) { hA[i] = (float)(rand() % 17 - 8); hB[i] = (float)(rand() % 13 - 6); }
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) { float s = 0; for (int kk = 0; kk < 16; ++kk) s += hA[r * 16 + kk] * hB[kk * 32 + c]; ref[r * 32 + c] = s; }
    float *A, *B, *D; (void)hipMalloc(&A, sizeof hA); (void)hipMalloc(&B, sizeof hB); (void)hipMalloc(&D, sizeof hD);
    (void)hipMemcpy(A, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(B, hB, sizeof hB, hipMemcpyHostToDevice);
    k<<<1, 64>>>(A, B, D);
    (void)hipMemcpy(hD, D, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += hD[i] != ref[i];
    printf("mismatches: %d of 1024\n", bad);
    return bad != 0;
}
