// Developer micro-benchmark (not part of the product): times tap_gemm variants on the layer shapes
// of BASELINE.json config 2 (EnCodec-24k, 64 x 10 s; batch reduced with -b) with interleaved A/B
// rounds in ONE process, and checks the variants against each other.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/dev_bench.hip -o dev_bench_bin && ./dev_bench_bin -b 64
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../audiocodecs_amd/csrc/tap_gemm.h"
#include "../../audiocodecs_amd/csrc/tap_gemm4.h"

using namespace ac;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Layer { const char* name; int L, cin, s, J, N, reflect, elu, cin2; };

static const Layer LAYERS[] = {
    {"enc.rb32.c3", 240000, 32, 1, 3, 16, 1, 1, 0},   {"enc.rb32.fused", 240000, 16, 1, 1, 32, 1, 1, 32},
    {"enc.down2", 240000, 32, 2, 2, 64, 1, 1, 0},     {"enc.rb64.c3", 120000, 64, 1, 3, 32, 1, 1, 0},
    {"enc.rb64.fused", 120000, 32, 1, 1, 64, 1, 1, 64}, {"enc.down4", 120000, 64, 4, 2, 128, 1, 1, 0},
    {"enc.rb128.c3", 30000, 128, 1, 3, 64, 1, 1, 0},  {"enc.rb128.fused", 30000, 64, 1, 1, 128, 1, 1, 128},
    {"enc.down5", 30000, 128, 5, 2, 256, 1, 1, 0},    {"enc.rb256.c3", 6000, 256, 1, 3, 128, 1, 1, 0},
    {"enc.rb256.fused", 6000, 128, 1, 1, 256, 1, 1, 256}, {"enc.down8", 6000, 256, 8, 2, 512, 1, 1, 0},
    {"lstm.ih", 750, 512, 1, 1, 2048, 0, 0, 0},       {"enc.final", 750, 512, 1, 7, 128, 1, 1, 0},
    {"dec.first", 750, 128, 1, 7, 512, 1, 0, 0},      {"dec.up8", 750, 512, 1, 2, 2048, 0, 1, 0},
    {"dec.up5", 6000, 256, 1, 2, 640, 0, 1, 0},       {"dec.up4", 30000, 128, 1, 2, 256, 0, 1, 0},
    {"dec.up2", 120000, 64, 1, 2, 64, 0, 1, 0},       {"dec.head", 240000, 32, 1, 7, 1, 1, 1, 0},
};

static TapSeg mkseg(const float* x, int L, int cin, int s, int J, int reflect, int elu, int extra, int kofs) {
    TapSeg g{};
    g.x = x; g.bs = (long long)L * cin; g.ts = cin; g.rel_len = nullptr; g.L = L; g.cin = cin; g.cin_shift = -1;
    for (int sh = 0; sh < 30; ++sh) if ((1 << sh) == cin) g.cin_shift = sh;
    g.s = s; g.J = J;
    const int pl = (J - 1) * s, mp = std::max(pl, extra);
    g.Lp = (reflect && L <= mp) ? mp + 1 : L;
    g.lim = reflect ? L + extra : L;
    g.reflect = reflect; g.elu = elu; g.kofs = kofs;
    return g;
}

template <int WGM, int WGN, int WM, int WN>
void run_v1(TapGemmParams p, hipStream_t st) {
    constexpr int BM = WGM * WM * 16, BN = WGN * WN * 16;
    p.mtiles = (p.M + BM - 1) / BM; p.ntiles = (p.N + BN - 1) / BN;
    const size_t lds = tap_gemm_lds_bytes<WGM, WGN, WM, WN>();
    hipLaunchKernelGGL((tap_gemm_kernel<WGM, WGN, WM, WN, true>), dim3(p.B * p.mtiles * p.ntiles), dim3(WGM * WGN * 64), lds, st, p);
}
template <int WGM, int WGN, int WM, int WN>
void run_v4(TapGemmParams p, hipStream_t st) {
    using Cfg = Tap4Cfg<WGM, WGN, WM, WN>;
    p.mtiles = (p.M + Cfg::BM - 1) / Cfg::BM; p.ntiles = (p.N + Cfg::BN - 1) / Cfg::BN;
    static bool once = false;
    if (!once) { once = true; CK(hipFuncSetAttribute((const void*)tap_gemm4_kernel<WGM, WGN, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::lds_bytes)); }
    hipLaunchKernelGGL((tap_gemm4_kernel<WGM, WGN, WM, WN>), dim3(p.B * p.mtiles * p.ntiles), dim3(Cfg::NT), Cfg::lds_bytes, st, p);
}

struct Variant { std::string name; std::function<void(TapGemmParams, hipStream_t)> fn; };

int main(int argc, char** argv) {
    int B = 8, reps = 5; const char* only = nullptr;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-b")) B = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-r")) reps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-l")) only = argv[++i];
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%-16s %-22s %9s %8s %8s %10s\n", "layer", "variant", "us", "TF/s", "GB/s", "maxdiff");
    for (const Layer& ly : LAYERS) {
        if (only && !strstr(ly.name, only)) continue;
        const int M = (ly.L + ly.s - 1) / ly.s, extra = M * ly.s - ly.L;
        const int K1 = ly.J * ly.s * ly.cin, Ktot = K1 + ly.cin2;
        const size_t nx = (size_t)B * ly.L * ly.cin, nx2 = (size_t)B * ly.L * std::max(ly.cin2, 1), nw = (size_t)ly.N * Ktot, ny = (size_t)B * M * ly.N;
        std::vector<float> hx(nx), hx2(nx2), hw(nw), hb(ly.N);
        unsigned s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
        for (auto& v : hx) v = rnd() * 0.5f;
        for (auto& v : hx2) v = rnd() * 0.5f;
        for (auto& v : hw) v = rnd() / std::sqrt((float)Ktot);
        for (auto& v : hb) v = rnd() * 0.02f;
        float *dx, *dx2, *dw, *db, *dy, *dyref;
        CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dx2, nx2 * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&db, ly.N * 4));
        CK(hipMalloc(&dy, ny * 4)); CK(hipMalloc(&dyref, ny * 4));
        CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dx2, hx2.data(), nx2 * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), ly.N * 4, hipMemcpyHostToDevice));
        TapGemmParams p{};
        p.nseg = ly.cin2 ? 2 : 1;
        p.seg[0] = mkseg(dx, ly.L, ly.cin, ly.s, ly.J, ly.reflect, 0, extra, 0);
        if (ly.cin2) p.seg[1] = mkseg(dx2, ly.L, ly.cin2, 1, 1, 1, 0, 0, K1);
        p.w = dw; p.bias = db; p.y = dy; p.y_elu = nullptr; p.y_bs = (long long)M * ly.N; p.y_rs = ly.N; p.B = B; p.M = M; p.N = ly.N; p.Ktot = Ktot;
        std::vector<Variant> vs;
        const bool fast = (ly.s * ly.cin) % 32 == 0 && (ly.cin2 % 32) == 0 && ly.N % 4 == 0;
        if (ly.N <= 16) { vs.push_back({"v1<4,1,2,1>", run_v1<4, 1, 2, 1>}); if (fast) vs.push_back({"v4<4,1,2,1>", run_v4<4, 1, 2, 1>}); }
        else if (ly.N <= 32) { vs.push_back({"v1<4,1,2,2>", run_v1<4, 1, 2, 2>}); if (fast) { vs.push_back({"v4<4,1,2,2>", run_v4<4, 1, 2, 2>}); vs.push_back({"v4<4,1,4,2>", run_v4<4, 1, 4, 2>}); } }
        else if (ly.N <= 64) { vs.push_back({"v1<2,2,2,2>", run_v1<2, 2, 2, 2>}); if (fast) { vs.push_back({"v4<2,2,2,2>", run_v4<2, 2, 2, 2>}); vs.push_back({"v4<4,1,2,4>", run_v4<4, 1, 2, 4>}); vs.push_back({"v4<2,2,4,2>", run_v4<2, 2, 4, 2>}); } }
        else { vs.push_back({"v1<2,2,4,4>", run_v1<2, 2, 4, 4>}); vs.push_back({"v4<2,2,4,4>", run_v4<2, 2, 4, 4>}); vs.push_back({"v4<4,2,4,4>", run_v4<4, 2, 4, 4>}); vs.push_back({"v4<2,4,4,4>", run_v4<2, 4, 4, 4>}); vs.push_back({"v4<4,2,2,4>", run_v4<4, 2, 2, 4>}); }
        const double flops = 2.0 * B * (double)M * ly.N * Ktot;
        const double bytes = (double)(nx + (ly.cin2 ? nx2 : 0) + ny + nw) * 4.0;
        std::vector<double> best(vs.size(), 1e30);
        std::vector<float> diff(vs.size(), 0.f);
        std::vector<float> href(std::min(ny, (size_t)1 << 22)), hy(href.size());
        for (int r = 0; r < reps; ++r)
            for (size_t v = 0; v < vs.size(); ++v) {
                p.y = v == 0 ? dyref : dy;
                if (r == 0 && v) CK(hipMemsetAsync(dy, 0xFF, ny * 4, st));
                CK(hipEventRecord(e0, st));
                vs[v].fn(p, st);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                CK(hipGetLastError());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best[v] = std::min(best[v], (double)ms);
                if (r == 0) {
                    // compare the head and the tail of the output against variant 0
                    const size_t n = href.size();
                    if (v == 0) { CK(hipMemcpy(href.data(), dyref + (ny - n), n * 4, hipMemcpyDeviceToHost)); }
                    else {
                        CK(hipMemcpy(hy.data(), dy + (ny - n), n * 4, hipMemcpyDeviceToHost));
                        float d = 0.f; for (size_t i = 0; i < n; ++i) { float x = std::fabs(hy[i] - href[i]); if (!(x <= d)) d = x; } diff[v] = d;
                    }
                }
            }
        for (size_t v = 0; v < vs.size(); ++v)
            printf("%-16s %-22s %9.1f %8.1f %8.0f %10.2e\n", ly.name, vs[v].name.c_str(), best[v] * 1e3, flops / best[v] / 1e9, bytes / best[v] / 1e6, diff[v]);
        CK(hipFree(dx)); CK(hipFree(dx2)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dy)); CK(hipFree(dyref));
    }
    return 0;
}
