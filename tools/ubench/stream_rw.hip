// Round 6: what the MEMORY SKELETON of the stream kernels (rb_stream6.h: 256 workgroups x 16 waves, a wave moves 4 KB tiles of a
// [rows][64] fp32 tensor) can reach, by access pattern and walk -- next to the 6.2 TB/s torch's elementwise ELU moves the same 3.9 GB at.
//   PAT 0: rb_stream6's operand-shaped access: an instruction covers 16 rows x 64 B (lane (li, kq): row li, bytes 16 kq of a 64-byte group)
//   PAT 1: lane-linear: an instruction covers 1 KB contiguous (4 rows)
//   WALK 0: a wave owns a contiguous segment of tiles;  WALK 1: tile = iteration x (all waves) + wave id (the chip sweeps memory in order)
//   MODE 1 read only (sum into a register), 2 write only, 3 read + write (copy: the wait for a tile's rows in front of its stores)
//   DEPTH: tiles requested ahead (1 or 2)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 stream_rw.hip -o stream_rw_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PAT, int WALK, int MODE, int WPB>
__global__ __launch_bounds__(64 * WPB) void k(const float* __restrict__ x, float* __restrict__ y, long long tiles, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
    const long long nw = (long long)gridDim.x * WPB, gw = (long long)blockIdx.x * WPB + wave;
    const long long per = (tiles + nw - 1) / nw;
    const long long t0 = WALK == 0 ? gw * per : gw, t1 = WALK == 0 ? (t0 + per < tiles ? t0 + per : tiles) : tiles, ts = WALK == 0 ? 1 : nw;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto off = [&](long long t, int i) -> long long {          // float offset of this lane's 16 bytes of instruction i of tile t
        return PAT == 0 ? t * 1024 + li * 64 + i * 16 + kq * 4 : t * 1024 + i * 256 + lane * 4;
    };
    f32x4 r[4];
    if (MODE & 1) for (int i = 0; i < 4; ++i) r[i] = t0 < t1 ? *reinterpret_cast<const f32x4*>(x + off(t0, i)) : acc;
    for (long long t = t0; t < t1; t += ts) {
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (MODE & 1) ? r[i] : f32x4{(float)t, 1.f, 2.f, 3.f};
        if (MODE & 1) {
            const long long tn = t + ts < t1 ? t + ts : t;
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const f32x4*>(x + off(tn, i));
        }
        if (MODE & 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(y + off(t, i)) = v[i] * 1.5f;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += v[i];
        }
    }
    if (acc.x == 1234.5f) sink[0] = acc.x + acc.y + acc.z + acc.w;
}

// one-shot blocks (no loop): a wave moves U tiles (U x 4 loads in flight, then U x 4 stores) and retires -- torch's elementwise shape
template <int U, int NT>
__global__ __launch_bounds__(256) void oneshot(const float* __restrict__ x, float* __restrict__ y, long long tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long t = ((long long)blockIdx.x * 4 + wave) * U;
    if (t >= tiles) return;
    f32x4 r[U * 4];
#pragma unroll
    for (int i = 0; i < U * 4; ++i) r[i] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + t * 1024 + i * 256 + lane * 4)) : *reinterpret_cast<const f32x4*>(x + t * 1024 + i * 256 + lane * 4);
#pragma unroll
    for (int i = 0; i < U * 4; ++i) {
        if (NT) __builtin_nontemporal_store(r[i] * 1.5f, reinterpret_cast<f32x4*>(y + t * 1024 + i * 256 + lane * 4));
        else *reinterpret_cast<f32x4*>(y + t * 1024 + i * 256 + lane * 4) = r[i] * 1.5f;
    }
}
template <int U, int NT>
void run1(const float* x, float* y, long long tiles) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = (int)((tiles + 4 * U - 1) / (4 * U));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((oneshot<U, NT>), dim3(grid), dim3(256), 0, 0, x, y, tiles);
    CK(hipEventRecord(e0));
    const int n = 10;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((oneshot<U, NT>), dim3(grid), dim3(256), 0, 0, x, y, tiles);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= n;
    printf("one-shot copy, %d tiles per wave, nt %d, grid %d: %.3f ms  %.2f TB/s\n", U, NT, grid, ms, tiles * 8192.0 / 1e9 / ms);
}

template <int PAT, int WALK, int MODE, int WPB>
void run(const float* x, float* y, long long tiles, float* sink, int grid) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<PAT, WALK, MODE, WPB>), dim3(grid), dim3(64 * WPB), 0, 0, x, y, tiles, sink);
    CK(hipEventRecord(e0));
    const int n = 10;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((k<PAT, WALK, MODE, WPB>), dim3(grid), dim3(64 * WPB), 0, 0, x, y, tiles, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= n;
    const double gb = tiles * 4096.0 * ((MODE & 1 ? 1 : 0) + (MODE & 2 ? 1 : 0)) / 1e9;
    printf("pat %d walk %d mode %d waves/wg %2d grid %5d: %.3f ms  %.2f TB/s\n", PAT, WALK, MODE, WPB, grid, ms, gb / ms);
}

int main() {
    const long long tiles = 64LL * 120000 / 16;      // 480000 tiles of 16 rows x 256 B = 1.97 GB
    float *x, *y, *sink;
    CK(hipMalloc(&x, tiles * 4096)); CK(hipMalloc(&y, tiles * 4096)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(x, 0, tiles * 4096));
#define ALL(MODE) \
    run<0, 0, MODE, 16>(x, y, tiles, sink, 256); run<1, 0, MODE, 16>(x, y, tiles, sink, 256); \
    run<0, 1, MODE, 16>(x, y, tiles, sink, 256); run<1, 1, MODE, 16>(x, y, tiles, sink, 256); \
    run<1, 1, MODE, 4>(x, y, tiles, sink, 2048); run<1, 1, MODE, 4>(x, y, tiles, sink, 8192); run<0, 0, MODE, 4>(x, y, tiles, sink, 2048);
    ALL(3)
    run1<1, 0>(x, y, tiles); run1<2, 0>(x, y, tiles); run1<4, 0>(x, y, tiles); run1<1, 1>(x, y, tiles); run1<4, 1>(x, y, tiles);
    return 0;
}
