// Round 5: what does a CU pay per KB fetched from L2 into LDS, by path and by the number of waves that issue the requests?
//   MODE 0: global_load_lds_dwordx4 (LDS-DMA, 1 KB per wave-instruction)       -- tap_gemm8's weight stage
//   MODE 1: global_load_dwordx4 into registers + ds_write_b128                   -- the same bytes through the register file
//   MODE 2: global_load_dwordx4 into registers only                              -- the bare load path
//   MODE 3: global_store_dwordx4 of the same bytes (private streams only)
// One workgroup of W waves per CU (256 workgroups); every wave fetches PIECES x 1 KB per round from a 64 KB window of a buffer that
// stays in L2 (every workgroup reads the same window, like the weight tiles of a GEMM), waits for all of it, ROUNDS times.
// Prints shader clocks per 1 KB piece PER CU (wall clocks of the slowest wave / pieces of all waves of the workgroup) and B/clk/CU.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 lds_dma_rate.hip -o lds_dma_rate_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int MODE, int PIECES>
__global__ __launch_bounds__(1024, 1) void k(const char* __restrict__ src, float* sink, int rounds, unsigned long long* clk, int loaders, size_t region = 0) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) *reinterpret_cast<volatile int*>(lds + 159 * 1024) = 0;
    __syncthreads();
    if (wave >= loaders) {      // READER waves (a GEMM's consumers): 16 ds_read_b128 per round from the loaders' region, until the loaders are done
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        volatile int* done = reinterpret_cast<volatile int*>(lds + 159 * 1024);
        for (int r = 0; r < rounds * 64 && !*done; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a += *reinterpret_cast<const f32x4*>(lds + (((r + i) * 37 + wave * 11) & 127) * 1024 + lane * 16);
        }
        if (a.x == 12345.678f) sink[tid] = a.x;
        return;
    }
    const unsigned base = (unsigned)(size_t)lds + (unsigned)wave * PIECES * 1024u;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
        const char* p = src + ((size_t)((r * 7 + wave * 3) & 63)) * 1024 + lane * 16;      // somewhere in the 64 KB window
        if (region)         // STREAM: this workgroup's own region, walked once (PIECES x 64 KB apart per round; pieces of a round 64 KB apart... contiguous per wave)
            p = src + (size_t)blockIdx.x * region + ((size_t)(r * loaders + wave) * PIECES * 1024) % region + lane * 16;
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < PIECES; ++i) glds16(p + (region ? (size_t)i * 1024 : (size_t)((i * 5) & 63) * 1024), base + i * 1024u);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            f32x4 v[PIECES];
#pragma unroll
            for (int i = 0; i < PIECES; ++i) if (MODE != 3) v[i] = *reinterpret_cast<const f32x4*>(p + (region ? (size_t)i * 1024 : (size_t)((i * 5) & 63) * 1024));
            if (MODE == 3) {   // the same bytes WRITTEN to the region (global_store_dwordx4)
                char* q = const_cast<char*>(p);
#pragma unroll
                for (int i = 0; i < PIECES; ++i) {
                    *reinterpret_cast<f32x4*>(q + (region ? (size_t)i * 1024 : (size_t)((i * 5) & 63) * 1024)) = acc;
                }
            } else if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < PIECES; ++i) *reinterpret_cast<f32x4*>(lds + (size_t)wave * PIECES * 1024 + i * 1024 + lane * 16) = v[i];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
#pragma unroll
                for (int i = 0; i < PIECES; ++i) acc += v[i];
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) clk[blockIdx.x * 16 + wave] = t1 - t0;
    if (tid == 0) *reinterpret_cast<volatile int*>(lds + 159 * 1024) = 1;
    if (MODE == 2 && acc.x == 12345.678f) sink[tid] = acc.x + acc.y + acc.z + acc.w;
    if (MODE != 2 && lds[tid * 16] == 77 && sink) sink[tid] = 1.f;
}

template <int MODE, int PIECES>
static void run(const char* src, float* sink, unsigned long long* clk, int waves, double mhz_ratio, int readers = 0, size_t region = 0) {
    const int rounds = region ? (int)(region / ((size_t)waves * PIECES * 1024)) : 400;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, PIECES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t lds = 160 * 1024;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(clk, 0, 256 * 16 * 8));
        hipLaunchKernelGGL((k<MODE, PIECES>), dim3(256), dim3(64 * (waves + readers)), lds, 0, src, sink, rounds, clk, waves, region);
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h(256 * 16);
    CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    double sum = 0;
    for (int b = 0; b < 256; ++b) { unsigned long long m = 0; for (int w = 0; w < waves; ++w) m = h[b * 16 + w] > m ? h[b * 16 + w] : m; sum += (double)m; }
    const double ticks = sum / 256.0;                         // s_memtime ticks (100 MHz) of the slowest wave, averaged over the CUs
    const double clocks = ticks * mhz_ratio;                  // -> shader clocks
    const double pieces = (double)rounds * PIECES * waves;
    printf("mode %d  loaders %2d  readers %2d  pieces/round/wave %2d  %s: %7.1f clocks per KB per CU = %5.1f B/clk/CU\n", MODE, waves, readers, PIECES,
           region ? (region >= (4u << 20) ? "private 4 MB stream per CU (1 GB: HBM) " : "private 64 KB region per CU (16 MB: L2 / Infinity Cache) ") : "", clocks / pieces, 1024.0 * pieces / clocks);
}

int main() {
    char* src; float* sink; unsigned long long* clk;
    CK(hipMalloc(&src, (size_t)1 << 30)); CK(hipMemset(src, 1, (size_t)1 << 30));
    CK(hipMalloc(&sink, 4096 * 4)); CK(hipMalloc(&clk, 256 * 16 * 8));
    // (s_memtime counts at the shader clock's rate on this chip -- 1.97 GHz against the 100 MHz of s_memrealtime, profiles/r5a_* -- so its
    //  ticks are taken as clocks)
    const double ratio = 1.0;
    for (int w : {1, 2, 4, 8, 12, 16}) run<0, 8>(src, sink, clk, w, ratio);
    for (int w : {1, 2, 4, 8, 12, 16}) run<1, 8>(src, sink, clk, w, ratio);
    for (int w : {1, 2, 4, 8, 12, 16}) run<2, 8>(src, sink, clk, w, ratio);
    for (int w : {4, 8}) run<0, 4>(src, sink, clk, w, ratio);
    // beside waves that read fragments from the same LDS (no MFMAs): 4 or 8 loaders + 8 readers
    for (int w : {4, 8}) run<0, 8>(src, sink, clk, w, ratio, 8);
    for (int w : {4, 8}) run<1, 8>(src, sink, clk, w, ratio, 8);
    for (int w : {4, 8}) run<2, 8>(src, sink, clk, w, ratio, 8);
    // every CU streams data of its own: 64 KB regions (16 MB in all, re-walked: cache-resident) and 4 MB regions (1 GB in all: HBM)
    for (int w : {4, 8, 16}) run<0, 8>(src, sink, clk, w, ratio, 0, (size_t)64 << 10);
    for (int w : {4, 8, 16}) run<0, 8>(src, sink, clk, w, ratio, 0, (size_t)4 << 20);
    for (int w : {4, 8, 16}) run<2, 8>(src, sink, clk, w, ratio, 0, (size_t)4 << 20);
    for (int w : {8, 16}) run<3, 8>(src, sink, clk, w, ratio, 0, (size_t)4 << 20);      // stores only
    return 0;
}
