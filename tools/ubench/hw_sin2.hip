// sin^2 for Snake by the hardware's v_sin_f32 / v_cos_f32 (argument in revolutions) against float64: is one transcendental
// instruction accurate enough to replace the ~17-instruction Cody-Waite + polynomial of tap_gemm.h sin2_f32?
// hipcc --offload-arch=gfx950 -O3 tools/ubench/hw_sin2.hip -o /tmp/hw_sin2 && /tmp/hw_sin2
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* t, float* a, float* b, float* c, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float f = t[i] * 0.15915494309189535f;
    const float s = __builtin_amdgcn_sinf(f);
    a[i] = s * s;                                                 // sin^2 by v_sin
    b[i] = __builtin_fmaf(-0.5f, __builtin_amdgcn_cosf(f + f), 0.5f);   // (1 - cos 2t) / 2 by v_cos
    // two-term Cody-Waite reduction by pi (exact for |t| < 2^15), then ONE transcendental on [-1/4, 1/4] revolutions
    const float kk = __builtin_rintf(t[i] * 0.3183098861837907f);
    float r = __builtin_fmaf(kk, -0x1.921fb6p+1f, t[i]);
    r = __builtin_fmaf(kk, 0x1.777a5cp-24f, r);
#ifdef COSFORM
    c[i] = __builtin_fmaf(-0.5f, __builtin_amdgcn_cosf(r * 0.3183098861837907f), 0.5f);      // (1 - cos 2r) / 2
#else
    const float s3 = __builtin_amdgcn_sinf(r * 0.15915494309189535f);
    c[i] = s3 * s3;
#endif
}
int main() {
    const int n = 1 << 23;
    std::vector<float> t(n);
    unsigned long long x = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const double u = (double)(x >> 11) / 9007199254740992.0;
        const double span = i < n / 4 ? 0.02 : i < n / 2 ? 10.0 : i < 3 * n / 4 ? 100.0 : 1500.0;
        t[i] = (float)((2.0 * u - 1.0) * span);
    }
    float *dt, *da, *db, *dc;
    hipMalloc(&dt, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4);
    hipMemcpy(dt, t.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dt, da, db, dc, n);
    std::vector<float> a(n), b(n), c(n);
    hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    const char* names[4] = {"|t| < 0.02", "|t| < 10", "|t| < 100", "|t| < 1500"};
    for (int q = 0; q < 4; ++q) {
        double ea = 0, eb = 0, ec = 0, er = 0, rc = 0;
        for (int i = q * (n / 4); i < (q + 1) * (n / 4); ++i) {
            const double s = std::sin((double)t[i]), ref = s * s;
            const float sf = sinf(t[i]);
            ea = std::fmax(ea, std::fabs(a[i] - ref)); eb = std::fmax(eb, std::fabs(b[i] - ref)); ec = std::fmax(ec, std::fabs(c[i] - ref));
            er = std::fmax(er, std::fabs((double)(sf * sf) - ref));
            if (ref > 1e-30) rc = std::fmax(rc, std::fabs(c[i] - ref) / ref);
        }
        printf("           reduced form: max RELATIVE error %.3e\n", rc);
        printf("%-10s max abs error: v_sin^2 %.3e | (1 - v_cos 2t)/2 %.3e | reduced v_sin^2 %.3e | host sinf^2 %.3e\n", names[q], ea, eb, ec, er);
    }
    return 0;
}
