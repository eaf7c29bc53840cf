// How fast do the store patterns of a tap-GEMM epilogue drain?  Every workgroup (256 threads, __launch_bounds__(256, 2)) writes
// [128][256] fp32 tiles of a [M][256] matrix, values from registers (no loads):
//   pattern 0: the direct epilogue of tap_gemm6.h -- wave w owns columns 64 w .. 64 w + 63; one buffer_store_b32 per value,
//              an instruction covers two rows x 128 contiguous bytes (128 instructions per wave)
//   pattern 1: 16 bytes per lane, an instruction covers one whole 1 KB row (32 instructions per wave)
//   pattern 2: as 0 with global_store_dword (no buffer descriptor)
//   pattern 3: pattern 0, and a second copy of the tile to another matrix (y and y_elu)
//   hipcc --offload-arch=gfx950 -O3 -o store_patterns_bin store_patterns.hip && ./store_patterns_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int PAT>
__global__ __launch_bounds__(256, 2) void k(float* __restrict__ y, float* __restrict__ y2, int mtiles, float seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = blockIdx.x; t < mtiles; t += gridDim.x) {
        const int m0 = t * 128;
        float v = seed + t;
        if (PAT == 0 || PAT == 3) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, 0x7ffffff0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)y2, 0, 0x7ffffff0, 0x00020000);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int voff = (m0 + a * 32 + 4 * (lane >> 5)) * 1024 + (wave * 64 + c * 32 + (lane & 31)) * 4;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = 8 * (r / 4) + (r % 4);
                        v = v * 1.0001f + 1.f;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, voff, dr * 1024, 0);
                        if (PAT == 3) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v + 1.f), rs2, voff, dr * 1024, 0);
                    }
                }
        } else if (PAT == 2) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = 8 * (r / 4) + (r % 4);
                        v = v * 1.0001f + 1.f;
                        y[(long long)(m0 + a * 32 + 4 * (lane >> 5) + dr) * 256 + wave * 64 + c * 32 + (lane & 31)] = v;
                    }
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                v = v * 1.0001f + 1.f;
                *reinterpret_cast<f32x4*>(y + (long long)(m0 + wave * 32 + i) * 256 + lane * 4) = f32x4{v, v, v, v};
            }
        }
    }
}
// per-wave issue rate: grid 256 (one workgroup per CU), `nw` waves per workgroup, every wave issues the 128 b32 stores of a tile
// column slice per tile, 64 tiles; cycles per store instruction from s_memtime
__global__ __launch_bounds__(512) void rate(float* __restrict__ y, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, 0x7ffffff0, 0x00020000);
    float v = lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < 64; ++t) {
        const int m0 = (blockIdx.x * 64 + t) * 128;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int voff = (m0 + a * 32 + 4 * (lane >> 5)) * 1024 + ((wave & 3) * 64 + c * 32 + (lane & 31)) * 4;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    v = v * 1.0001f + 1.f;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, voff, (8 * (r / 4) + (r % 4)) * 1024, 0);
                }
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 100) cyc[0] = t1 - t0;
}
int main() {
    {
        float* yy; unsigned long long* cyc; hipMalloc(&yy, (size_t)256 * 64 * 128 * 1024); hipMalloc(&cyc, 8);
        for (int nw : {1, 2, 4, 8}) for (int grid : {1, 256}) {
            unsigned long long h = 0;
            for (int rep = 0; rep < 2; ++rep) { rate<<<grid == 1 ? 101 : 256, 64 * nw>>>(yy, cyc); hipDeviceSynchronize(); }
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            printf("%d wave(s) per CU storing, %s: %.1f cycles per b32 store instruction per wave\n", nw, grid == 1 ? "101 CUs busy" : "256 CUs busy", (double)h / (64 * 128));
        }
    }
    const int M = 64 * 30000, mtiles = M / 128;
    float *y, *y2;
    hipMalloc(&y, (size_t)M * 1024); hipMalloc(&y2, (size_t)M * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {512, 15000}) for (int pat = 0; pat < 4; ++pat) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (pat == 0) k<0><<<grid, 256>>>(y, y2, mtiles, 1.f);
            if (pat == 1) k<1><<<grid, 256>>>(y, y2, mtiles, 1.f);
            if (pat == 2) k<2><<<grid, 256>>>(y, y2, mtiles, 1.f);
            if (pat == 3) k<3><<<grid, 256>>>(y, y2, mtiles, 1.f);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        const double gb = (double)M * 1024 * (pat == 3 ? 2 : 1) / 1e9;
        printf("grid %5d pattern %d: %.3f ms  %.2f TB/s written\n", grid, pat, ms, gb / ms);
    }
    return 0;
}
