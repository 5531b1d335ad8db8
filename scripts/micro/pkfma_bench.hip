// Issue-rate micro-benchmark: v_fma_f32 vs v_pk_fma_f32 with VGPR / SGPR weight operands (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, const float* w, int iters) {
    v2f acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v2f){(float)threadIdx.x, 1.0f};
    v2f x = (v2f){0.5f + threadIdx.x * 1e-3f, 0.25f};
    float ws0 = w[0], ws1 = w[1];                      // uniform -> SGPRs
    v2f wv = (v2f){w[threadIdx.x & 1], w[2]};          // per-lane -> VGPRs
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) { acc[i].x = __builtin_fmaf(ws0, x.x, acc[i].x); acc[i].y = __builtin_fmaf(ws1, x.x, acc[i].y); }       // 2 v_fma, SGPR weight
                if (MODE == 1) acc[i] = __builtin_elementwise_fma((v2f){ws0, ws1}, (v2f){x.x, x.x}, acc[i]);                          // pk, SGPR pair + splat
                if (MODE == 2) acc[i] = __builtin_elementwise_fma(wv, x, acc[i]);                                                      // pk, all VGPR
                if (MODE == 3) { acc[i].x = __builtin_fmaf(wv.x, x.x, acc[i].x); acc[i].y = __builtin_fmaf(wv.y, x.y, acc[i].y); }     // 2 v_fma, VGPR
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, float* out, float* w) {
    const int iters = 2000, blocks = 256 * 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(out, w, 10);
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(out, w, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double fma = (double)blocks * 256 * iters * 64 * 2;      // scalar FMAs
    printf("%-34s %8.3f ms  %7.1f TFLOP/s\n", name, ms, 2 * fma / (ms * 1e-3) / 1e12);
}
int main() {
    float *out, *w; hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&w, 16);
    float hw[4] = {1.0001f, 0.9999f, 1.00005f, 0.f}; hipMemcpy(w, hw, 16, hipMemcpyHostToDevice);
    run<0>("v_fma_f32, SGPR weight", out, w);
    run<1>("v_pk_fma_f32, SGPR pair + splat", out, w);
    run<2>("v_pk_fma_f32, VGPR operands", out, w);
    run<3>("v_fma_f32, VGPR operands", out, w);
    return 0;
}
