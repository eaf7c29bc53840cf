// Developer tool: per-stage s_memtime trace of tap_gemm3's producer/consumer waves on one layer.
#define TAP4_TRACE 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "../../audiocodecs_amd/csrc/tap_gemm4.h"
using namespace ac;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main() {
    // enc.down4: L=120000 cin=64 s=4 J=2 N=128, B=8
    const int B = 8, L = 120000, cin = 64, s = 4, J = 2, N = 128, M = L / s, K = J * s * cin;
    size_t nx = (size_t)B * L * cin, nw = (size_t)N * K, ny = (size_t)B * M * N;
    float *dx, *dw, *db, *dy; CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&db, N * 4)); CK(hipMalloc(&dy, ny * 4));
    std::vector<float> hx(nx); unsigned r = 1; for (auto& v : hx) { r = r * 1664525u + 1013904223u; v = ((r >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hx.data(), nw * 4, hipMemcpyHostToDevice)); CK(hipMemset(db, 0, N * 4));
    unsigned long long* dtr; const size_t NTR = 2 * 4 * 64 * 5; CK(hipMalloc(&dtr, NTR * 8)); CK(hipMemset(dtr, 0, NTR * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_tap4_trace), &dtr, sizeof(dtr)));
    TapGemmParams p{};
    p.nseg = 1; TapSeg& g = p.seg[0];
    g.x = dx; g.bs = (long long)L * cin; g.ts = cin; g.L = L; g.cin = cin; g.cin_shift = 6; g.s = s; g.J = J; g.Lp = L; g.lim = L; g.reflect = 1; g.elu = 0; g.kofs = 0;
    p.w = dw; p.bias = db; p.y = dy; p.y_elu = nullptr; p.y_bs = (long long)M * N; p.y_rs = N; p.B = B; p.M = M; p.N = N; p.Ktot = K;
    using Cfg = Tap4Cfg<2, 2, 4, 4>;
    p.mtiles = (M + Cfg::BM - 1) / Cfg::BM; p.ntiles = 1;
    CK(hipFuncSetAttribute((const void*)tap_gemm4_kernel<2, 2, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((tap_gemm4_kernel<2, 2, 4, 4>), dim3(B * p.mtiles), dim3(Cfg::NT), getenv("ONE_PER_CU") ? 120 * 1024 : Cfg::lds_bytes, 0, p);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("kernel %.1f us  (%.1f TF/s)\n", ms * 1e3, 2.0 * B * M * N * K / ms / 1e9);
    }
    std::vector<unsigned long long> t(NTR);
    CK(hipMemcpy(t.data(), dtr, NTR * 8, hipMemcpyDeviceToHost));
    for (int slot = 0; slot < 2; ++slot)
        for (int wave = 0; wave < 4; wave += 3) {
            unsigned long long t0 = t[((slot * 4 + 0) * 64 + 0) * 5 + 0];
            printf("block %d wave %d: stage | top  loads_issued(d)  mfma_issued(d)  lds_written(d)  after_barrier(d) | stage total\n", 700 + slot, wave);
            for (int st = 0; st < 15; ++st) {
                auto T = [&](int k) { return (long long)(t[((slot * 4 + wave) * 64 + st) * 5 + k] - t0); };
                printf("%5d | %8lld %6lld %6lld %6lld %6lld | %6lld\n", st, T(0), T(1) - T(0), T(2) - T(1), T(3) - T(2), T(4) - T(3), T(4) - T(0));
            }
        }
    return 0;
}
