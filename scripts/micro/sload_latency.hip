// sload_latency.hip -- latency of scalar loads from the constant cache on gfx950 (what the weight streams of K9-K14 wait for).
// A dependent chain of s_load_dword (each address comes from the previous result, table of zeros -> always the same line) gives the
// scalar-cache HIT latency; the same chain over a 1 MB stride table gives the miss (L2) latency; a chain of ds_read_b32 gives the
// LDS latency for comparison.  One wave, s_memtime around 256 loads.
// Build: hipcc --offload-arch=gfx950 -O3 -o sload_latency sload_latency.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef const uint32_t __attribute__((address_space(4))) * cmem_t;

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

__global__ void chain(const uint32_t* tab, int n, unsigned long long* out, int mode) {
    __shared__ uint32_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    cmem_t T = (cmem_t)(uintptr_t)tab;
    uint32_t idx = 0;
    // warm-up
    for (int i = 0; i < 64; ++i) idx = T[idx];
    unsigned long long t0 = now();
    if (mode == 0) {
        for (int i = 0; i < n; ++i) idx = T[idx];                                   // dependent scalar loads
    } else if (mode == 1) {
        uint32_t v = threadIdx.x & 0;
        for (int i = 0; i < n; ++i) v = lds[v];                                     // dependent LDS reads
        idx += v;
    } else {
        // 16 independent x16 loads issued back to back, one wait: throughput of the stream
        for (int i = 0; i < n / 16; ++i) {
            uint32_t acc = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += T[(i & 3) * 256 + j * 16 + (idx & 1)];
            idx += acc;
        }
    }
    unsigned long long t1 = now();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = idx; }
}

// throughput: every wave of the grid streams s_load_dwordx16 from a 16 KB table (the weight streams of K9-K14), 4 loads per wait
typedef uint32_t u16v __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) stream(const uint32_t* tab, int iters, unsigned long long* out, uint32_t* sink) {
    const uint64_t base = (uint64_t)(uintptr_t)tab;
    uint32_t acc = 0;
    unsigned long long t0 = now();
    for (int i = 0; i < iters; ++i) {
        u16v a, b, c, d;
        const uint64_t p = base + (uint64_t)((i & 15) * 1024);
        asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\ts_load_dwordx16 %2, %4, 0x80\n\ts_load_dwordx16 %3, %4, 0xc0\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=s"(a), "=s"(b), "=s"(c), "=s"(d) : "s"(p) : "memory");
        acc += a[0] + b[1] + c[2] + d[3];
    }
    unsigned long long t1 = now();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (acc == 12345u) sink[0] = acc;
}

// do scalar loads overlap with VALU work of the same wave?  [2 x16 loads][NF FMAs][wait] against the two parts alone
template <int NF, bool LOADS>
__global__ void __launch_bounds__(64) overlap(const uint32_t* tab, int iters, unsigned long long* out, float* sink) {
    const uint64_t base = (uint64_t)(uintptr_t)tab;
    float f[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) f[j] = threadIdx.x + j;
    uint32_t acc = 0;
    unsigned long long t0 = now();
    for (int i = 0; i < iters; ++i) {
        u16v a, b;
        const uint64_t p = base + (uint64_t)((i & 15) * 1024);
        if (LOADS) asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=s"(a), "=s"(b) : "s"(p) : "memory");
#pragma unroll
        for (int k = 0; k < NF; ++k) asm volatile("v_fmac_f32 %0, 0x3f800347, %0" : "+v"(f[k & 15]));
        if (LOADS) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b)); acc += a[0] + b[1]; }
    }
    unsigned long long t1 = now();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    float s = 0; for (int j = 0; j < 16; ++j) s += f[j];
    if (s == 1.2345f || acc == 12345u) sink[0] = s;
}

// a lone wave: 32 multiply-adds as 32 v_fmac_f32 or as 16 v_pk_fma_f32 (SGPR-pair weights as in the weight streams)
typedef float f2v __attribute__((ext_vector_type(2)));
template <bool PK>
__global__ void __launch_bounds__(64) lone_fma(const float* w, int iters, unsigned long long* out, float* sink) {
    f2v acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (f2v){(float)threadIdx.x + j, 1.0f};
    const f2v w0 = (f2v){w[0], w[1]}, w1 = (f2v){w[2], w[3]};            // uniform -> SGPR pairs
    const float x0 = 0.5f + threadIdx.x * 1e-3f, x1 = 0.25f;
    unsigned long long t0 = now();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f2v ww = r ? w1 : w0;
                const float xx = r ? x1 : x0;
                if (PK) acc[j] = __builtin_elementwise_fma(ww, (f2v){xx, xx}, acc[j]);
                else { acc[j].x = __builtin_fmaf(ww.x, xx, acc[j].x); acc[j].y = __builtin_fmaf(ww.y, xx, acc[j].y); }
            }
        }
    }
    unsigned long long t1 = now();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    float s = 0; for (int j = 0; j < 8; ++j) s += acc[j].x + acc[j].y;
    if (s == 1.2345f) sink[0] = s;
}

int main() {
    const int n = 4096;
    uint32_t* tab; unsigned long long* out;
    hipMalloc(&tab, 64 << 20); hipMemset(tab, 0, 64 << 20); hipMalloc(&out, 16);
    unsigned long long h[2];
    // hit: zeros -> idx stays 0
    chain<<<1, 64>>>(tab, n, out, 0); hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("scalar load, cache hit, dependent chain : %.1f cycles (s_memtime ticks) per load\n", (double)h[0] / n);
    // miss: stride chain through 64 MB (each entry points 1 MB + 64 B further)
    {
        uint32_t* hst = (uint32_t*)calloc(16 << 20, 4);
        uint32_t cur = 0;
        for (int i = 0; i < 8192; ++i) { uint32_t nxt = (uint32_t)(((uint64_t)cur + (1u << 18) + 16) % (16u << 20)); hst[cur] = nxt; cur = nxt; }
        hipMemcpy(tab, hst, 64 << 20, hipMemcpyHostToDevice);
        chain<<<1, 64>>>(tab, n, out, 0); hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        printf("scalar load, 1 MB stride (cache miss)    : %.1f ticks per load\n", (double)h[0] / n);
        hipMemset(tab, 0, 64 << 20);
        free(hst);
    }
    chain<<<1, 64>>>(tab, n, out, 1); hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("ds_read_b32, dependent chain             : %.1f ticks per read\n", (double)h[0] / n);
    chain<<<1, 64>>>(tab, n, out, 2); hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("16 independent scalar loads + one wait   : %.1f ticks per group of 16\n", (double)h[0] / (n / 16));
    {
        unsigned long long* o2; uint32_t* sink; hipMalloc(&o2, 8 * 4 * 4096); hipMalloc(&sink, 4);
        const int iters = 2000;
        for (int wpc = 1; wpc <= 16; wpc *= 2) {                 // waves per CU: 256 CUs x wpc waves (blocks of 1..4 waves)
            const int threads = wpc >= 4 ? 256 : 64 * wpc, blocks = 256 * (wpc >= 4 ? wpc / 4 : 1);
            stream<<<blocks, threads>>>(tab, 10, o2, sink); hipDeviceSynchronize();
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a); stream<<<blocks, threads>>>(tab, iters, o2, sink); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            unsigned long long w0; hipMemcpy(&w0, o2, 8, hipMemcpyDeviceToHost);
            printf("x16 stream, %2d waves per CU: wave 0 %.1f ticks per x16 load; kernel %.3f ms -> %.2f x16 loads per us per CU\n", wpc,
                   (double)w0 / (iters * 4.0), ms, (double)wpc * iters * 4 / (ms * 1e3));
        }
    }
    {
        unsigned long long* o3; float* fs; hipMalloc(&o3, 64); hipMalloc(&fs, 4);
        unsigned long long r[3]; const int it = 4000;
        overlap<0, true><<<1, 64>>>(tab, it, o3, fs); hipMemcpy(&r[0], o3, 8, hipMemcpyDeviceToHost);
        overlap<32, false><<<1, 64>>>(tab, it, o3, fs); hipMemcpy(&r[1], o3, 8, hipMemcpyDeviceToHost);
        overlap<32, true><<<1, 64>>>(tab, it, o3, fs); hipMemcpy(&r[2], o3, 8, hipMemcpyDeviceToHost);
        printf("one wave: 2 x16 loads + wait %.1f ticks; 32 v_fmac %.1f ticks; loads, 32 v_fmac, wait %.1f ticks per round\n",
               (double)r[0] / it, (double)r[1] / it, (double)r[2] / it);
        overlap<64, false><<<1, 64>>>(tab, it, o3, fs); hipMemcpy(&r[1], o3, 8, hipMemcpyDeviceToHost);
        overlap<64, true><<<1, 64>>>(tab, it, o3, fs); hipMemcpy(&r[2], o3, 8, hipMemcpyDeviceToHost);
        printf("one wave: 64 v_fmac %.1f ticks; loads, 64 v_fmac, wait %.1f ticks per round\n", (double)r[1] / it, (double)r[2] / it);
    }
    {
        unsigned long long* o4; float* fs2; float* wd; hipMalloc(&o4, 64); hipMalloc(&fs2, 4); hipMalloc(&wd, 16);
        float hw[4] = {1.0001f, 0.9999f, 1.00005f, 0.99995f}; hipMemcpy(wd, hw, 16, hipMemcpyHostToDevice);
        unsigned long long r[2]; const int it = 4000;
        lone_fma<false><<<1, 64>>>(wd, it, o4, fs2); hipMemcpy(&r[0], o4, 8, hipMemcpyDeviceToHost);
        lone_fma<true><<<1, 64>>>(wd, it, o4, fs2); hipMemcpy(&r[1], o4, 8, hipMemcpyDeviceToHost);
        printf("one wave, 32 multiply-adds per round: 32 v_fmac_f32 %.1f ticks; 16 v_pk_fma_f32 (SGPR pair x splat) %.1f ticks\n",
               (double)r[0] / it, (double)r[1] / it);
    }
    // s_memtime runs at a fixed 100 MHz on this part: print the ratio to the shader clock measured with a VALU loop elsewhere
    return 0;
}
