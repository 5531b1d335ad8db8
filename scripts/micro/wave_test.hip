// wave_test.hip -- the DPP reductions / prefix sum of csrc/rs_wave.hpp against host sums (40 active lanes, as the particle filter uses them).
// Build: hipcc --offload-arch=gfx950 -O3 -o wave_test wave_test.hip ; run on the GPU box (exit code 0 = all lanes agree).
#include "../../radiation_ppo_amd/csrc/rs_wave.hpp"
#include <stdio.h>
__global__ void k(const float* x, const double* d, float* s, float* m, double* sc, double* tot) {
    const int l = threadIdx.x;
    s[l] = rs_wave_sum(l < 40 ? x[l] : 0.0f);
    m[l] = rs_wave_max(l < 40 ? x[l] : -INFINITY);
    const double c = rs_wave_scan(l < 40 ? d[l] : 0.0);
    sc[l] = c; tot[l] = rs_lane_d<39>(c);
}
int main() {
    float hx[64], hs[64], hm[64]; double hd[64], hsc[64], ht[64];
    for (int i = 0; i < 64; ++i) { hx[i] = (float)((i * 37) % 11) - 3.5f + 0.01f * i; hd[i] = 0.001 * (i + 1) + 1e-9 * i * i; }
    float *x, *s, *m; double *d, *sc, *t;
    hipMalloc(&x, 256); hipMalloc(&s, 256); hipMalloc(&m, 256); hipMalloc(&d, 512); hipMalloc(&sc, 512); hipMalloc(&t, 512);
    hipMemcpy(x, hx, 256, hipMemcpyHostToDevice); hipMemcpy(d, hd, 512, hipMemcpyHostToDevice);
    k<<<1, 64>>>(x, d, s, m, sc, t);
    hipMemcpy(hs, s, 256, hipMemcpyDeviceToHost); hipMemcpy(hm, m, 256, hipMemcpyDeviceToHost);
    hipMemcpy(hsc, sc, 512, hipMemcpyDeviceToHost); hipMemcpy(ht, t, 512, hipMemcpyDeviceToHost);
    double rs = 0, rm = -1e30; for (int i = 0; i < 40; ++i) { rs += hx[i]; if (hx[i] > rm) rm = hx[i]; }
    int bad = 0; double run = 0;
    for (int i = 0; i < 64; ++i) {
        if (fabs(hs[i] - rs) > 1e-4 || hm[i] != (float)rm) ++bad;
        if (i < 40) { run += hd[i]; if (fabs(hsc[i] - run) > 1e-15) ++bad; }
        if (fabs(ht[i] - ([&]{double q=0; for(int j=0;j<40;++j) q+=hd[j]; return q;})()) > 1e-15) ++bad;
    }
    printf("sum %.6f (ref %.6f) max %.3f (ref %.3f) scan39 %.12f bad %d\n", hs[5], rs, hm[7], rm, hsc[39], bad);
    return bad != 0;
}
