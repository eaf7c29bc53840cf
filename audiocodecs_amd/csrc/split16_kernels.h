// The stand-alone kernels of split16.h (amax of a tensor nobody reported one for, per-row amax, bound fills): compiled in
// core.hip only; the helpers every kernel header uses are in split16.h.
#pragma once
#include "split16.h"

namespace ac {

// Clearing the amax slots / row words: a kernel of the library's own, not hipMemsetAsync.  A memset node captured into a hipGraph does not
// replay reliably on this runtime (ROCm 7.0.2 / HIP 7.0.51831: after the first replay part of the range holds garbage --
// tools/experiments/r5i_memset_in_graph.py), and slots that are not cleared keep the atomicMax of every earlier replay: a captured
// encode replayed on NEW data was wrong by 0.5 absolute (tools/experiments/r5j_graph_newdata.py) while every test that replayed the data it
// had captured with stayed green.  16 bytes per thread; n16 = number of 16-byte units.
__global__ __launch_bounds__(256) void zero16_kernel(s16_f32x4* __restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = s16_f32x4{0.f, 0.f, 0.f, 0.f};
}

// Fallback for tensors whose producer reports no amax: slot[b] = max |x[b]| over [L][C] rows of pitch ts.
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long long bs, long long ts, int L, int C, unsigned* __restrict__ slot) {
    const int b = blockIdx.y;
    const float* xb = x + (long long)b * bs;
    const long long n = (long long)L * C;
    unsigned mx = 0;
    if (ts == C && (C & 3) == 0 && ((reinterpret_cast<uintptr_t>(xb) & 15) == 0)) {
        const s16_f32x4* v = reinterpret_cast<const s16_f32x4*>(xb);
        const long long n4 = n / 4, step = (long long)gridDim.x * 256;
        long long i = (long long)blockIdx.x * 256 + threadIdx.x;
        for (; i + 3 * step < n4; i += 4 * step) {          // four independent 16-byte loads in flight per thread
            const s16_f32x4 a0 = v[i], a1 = v[i + step], a2 = v[i + 2 * step], a3 = v[i + 3 * step];
            amax_acc4(mx, a0); amax_acc4(mx, a1); amax_acc4(mx, a2); amax_acc4(mx, a3);
        }
        for (; i < n4; i += step) amax_acc4(mx, v[i]);
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) amax_acc(mx, xb[(i / C) * ts + i % C]);
    }
    // one flush per workgroup
    __shared__ unsigned s_mx[4];
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)mx, o);
        mx = t > mx ? t : mx;
    }
    if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned m01 = s_mx[0] > s_mx[1] ? s_mx[0] : s_mx[1], m23 = s_mx[2] > s_mx[3] ? s_mx[2] : s_mx[3];
        const unsigned m = m01 > m23 ? m01 : m23;
        unsigned* w = amax_at(slot, b);
        if (m > __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(w, m);
    }
}

// Row mode (linear layers over a merged row matrix): out[r] = max |x[r][0 .. C)|, one wave per row
__global__ __launch_bounds__(256) void rowmax_kernel(const float* __restrict__ x, long long pitch, long long rows, int C, unsigned* __restrict__ out) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + r * pitch;
    unsigned mx = 0;
    if ((C & 3) == 0 && (pitch & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        for (int i = lane; i < C / 4; i += 64) amax_acc4(mx, reinterpret_cast<const s16_f32x4*>(xr)[i]);
    } else {
        for (int i = lane; i < C; i += 64) amax_acc(mx, xr[i]);
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)mx, o);
        mx = t > mx ? t : mx;
    }
    if (lane == 0) out[r] = mx;
}

// slot[b] = bits (a bound known without looking at the data)
__global__ void amax_fill_kernel(unsigned* __restrict__ slot, unsigned bits, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) *amax_at(slot, b) = bits;
}

// slot_out[b] = bits of (|a| + add): the bound of a sum whose second term is bounded by `add` (LSTM skip: h in (-1, 1))
__global__ void amax_add_kernel(const unsigned* __restrict__ in, float add, unsigned* __restrict__ out, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const unsigned v = *amax_at(in, b);
    *amax_at(out, b) = ((v >> 23) & 0xffu) == 255 ? v : __float_as_uint((__uint_as_float(v) + add) * 1.0000002f);   // rounded up
}

}  // namespace ac
