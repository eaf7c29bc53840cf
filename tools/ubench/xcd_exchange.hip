// Micro-benchmark: fence-free producer/consumer exchange between workgroups through L2 on MI355X.
// Question (persistent LSTM design): how long does one "publish 1 KB, wait for the 32 peers of my group, read
// their 32 KB" round take when the group lives on ONE XCD (same L2), and does data published with agent-scope
// relaxed atomic stores (no wbl2 / inv fences) arrive intact (a) inside the XCD, (b) on the neighbouring XCD?
// 256 workgroups (one per CU); each reads its XCC_ID, takes a slot on its XCD, and runs `iters` rounds.
// Every spin is bounded so a residency problem cannot hang the box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Ctl {
    unsigned slot_cnt[8 * 16];      // per-XCD slot allocator
    unsigned flag[8][64 * 32];      // flag[x][slot * FLAG_STRIDE] = last round published
    unsigned timeout[16];
    unsigned errors[16];
    unsigned xcd_hist[16];
};

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }   // hwreg(HW_REG_XCC_ID, 0, 4)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifndef FLAG_STRIDE
#define FLAG_STRIDE 1    // words between the flags of a group: 1 = one hot 128-byte line, 32 = a line per flag
#endif
#ifndef AUX
#define AUX 16   // buffer cache-policy bits on gfx940+: 1 = sc0, 16 = sc1 (sc1 alone = agent scope: miss the per-CU L1)
#endif

// MODE 0: consumers read their own XCD's payload; MODE 1: consumers read the payload of XCD x^1 (cross-XCD)
template <int MODE>
__global__ __launch_bounds__(256) void k(Ctl* C, float* buf, int iters, int per_xcd) {
    __shared__ unsigned s_slot, s_x;
    if (threadIdx.x == 0) {
        s_x = xcc_id();
        s_slot = __hip_atomic_fetch_add(&C->slot_cnt[s_x * 16], 1u, RLX);
        atomicAdd(&C->xcd_hist[s_x], 1u);
    }
    __syncthreads();
    const unsigned x = s_x, slot = s_slot;
    if (slot >= (unsigned)per_xcd) return;                       // more than per_xcd workgroups landed here: sit out
    const unsigned src_x = MODE == 0 ? x : (x ^ 1);
    float acc = 0.f;
    unsigned bad = 0;
    for (int it = 1; it <= iters; ++it) {
        // publish 1 KB: value encodes (round, xcd, slot, lane)
        float* mine = buf + (((size_t)(it & 1) * 8 + x) * per_xcd + slot) * 256;
        if (threadIdx.x < 64) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, 1024, 0x00020000);
            const float b = (float)(it * 7 + x * 3 + slot);
            const int l = threadIdx.x * 4;
            const f32x4 v = f32x4{b + l * 0.001f, b + (l + 1) * 0.001f, b + (l + 2) * 0.001f, b + (l + 3) * 0.001f};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, threadIdx.x * 16, 0, AUX);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // stores acknowledged by L2
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&C->flag[x][slot * FLAG_STRIDE], (unsigned)it, RLX);
        // wait for all peers of the SOURCE group
        if (threadIdx.x < 64) {
            const unsigned lane = threadIdx.x;
            for (unsigned spins = 0;; ++spins) {
                const unsigned f = lane < (unsigned)per_xcd ? __hip_atomic_load(&C->flag[src_x][lane * FLAG_STRIDE], RLX) : 0xffffffffu;
                if (__all(f >= (unsigned)it)) break;
                if (spins > (1u << 20)) { if (lane == 0) __hip_atomic_store(&C->timeout[0], 1u, RLX); break; }
            }
        }
        __syncthreads();
        if (__hip_atomic_load(&C->timeout[0], RLX)) return;
        // read the group's payload (per_xcd KB) through L2
        const float* src = buf + ((size_t)(it & 1) * 8 + src_x) * per_xcd * 256;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, per_xcd * 1024, 0x00020000);
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (threadIdx.x + 256 * j) * 16, 0, AUX));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i4 = threadIdx.x + 256 * j;                 // float4 index inside the group payload
            if (i4 * 4 < per_xcd * 256) {
                const int s = (i4 * 4) >> 8, l = (i4 * 4) & 255;
                const float b = (float)(it * 7 + src_x * 3 + s);
                bad += (v[j].x != b + l * 0.001f) + (v[j].y != b + (l + 1) * 0.001f) + (v[j].z != b + (l + 2) * 0.001f) + (v[j].w != b + (l + 3) * 0.001f);
                acc += v[j].x + v[j].w;
            }
        }
    }
    if (bad) atomicAdd(&C->errors[0], bad);
    if (acc == 12345.f) buf[0] = acc;
}

// MODE "mixed": a role's 32 workgroups are split 16/16 over an XCD pair (each XCD hosts halves of two roles), so every
// consumer reads 16 slices produced on its own XCD and 16 from the neighbour.
__global__ __launch_bounds__(256) void kmix(Ctl* C, float* buf, int iters, int sleep) {
    __shared__ unsigned s_slot, s_x;
    if (threadIdx.x == 0) {
        s_x = xcc_id();
        s_slot = __hip_atomic_fetch_add(&C->slot_cnt[s_x * 16], 1u, RLX);
    }
    __syncthreads();
    const unsigned x = s_x, slot = s_slot;
    if (slot >= 32) return;
    const unsigned pr = x >> 1, role = slot >> 4, idx = (x & 1) * 16 + (slot & 15);
    unsigned* flags = &C->flag[pr * 2 + role][0];
    float acc = 0.f;
    unsigned bad = 0;
    for (int it = 1; it <= iters; ++it) {
        float* base = buf + ((size_t)(it & 1) * 8 + pr * 2 + role) * 32 * 256;
        if (threadIdx.x < 64) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + idx * 256), 0, 1024, 0x00020000);
            const float b = (float)(it * 7 + pr * 3 + idx);
            const int l = threadIdx.x * 4;
            const f32x4 v = f32x4{b + l * 0.001f, b + (l + 1) * 0.001f, b + (l + 2) * 0.001f, b + (l + 3) * 0.001f};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, threadIdx.x * 16, 0, AUX);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&flags[idx * FLAG_STRIDE], (unsigned)it, RLX);
        if (threadIdx.x < 64) {
            const unsigned lane = threadIdx.x;
            for (unsigned spins = 0;; ++spins) {
                const unsigned f = lane < 32 ? __hip_atomic_load(&flags[lane * FLAG_STRIDE], RLX) : 0xffffffffu;
                if (__all(f >= (unsigned)it)) break;
                if (spins > (1u << 20)) { if (lane == 0) __hip_atomic_store(&C->timeout[0], 1u, RLX); break; }
                if (sleep) __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (__hip_atomic_load(&C->timeout[0], RLX)) return;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 32 * 1024, 0x00020000);
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (threadIdx.x + 256 * j) * 16, 0, AUX));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i4 = threadIdx.x + 256 * j;
            const int sl = (i4 * 4) >> 8, l = (i4 * 4) & 255;
            const float b = (float)(it * 7 + pr * 3 + sl);
            bad += (v[j].x != b + l * 0.001f) + (v[j].w != b + (l + 3) * 0.001f);
            acc += v[j].x + v[j].w;
        }
    }
    if (bad) atomicAdd(&C->errors[0], bad);
    if (acc == 12345.f) buf[0] = acc;
}

int main() {
    Ctl* C; float* buf;
    CK(hipMalloc(&C, sizeof(Ctl)));
    CK(hipMalloc(&buf, 2ull * 8 * 64 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 3000;
    for (int mode = 0; mode < 2; ++mode)
        for (int per_xcd : {32, 16}) {
            CK(hipMemset(C, 0, sizeof(Ctl)));
            CK(hipMemset(buf, 0, 2ull * 8 * 64 * 256 * 4));
            CK(hipEventRecord(e0));
            void* args[] = {&C, &buf, (void*)&iters, (void*)&per_xcd};
            if (mode == 0) CK(hipLaunchCooperativeKernel((const void*)k<0>, dim3(256), dim3(256), args, 0, 0));
            else CK(hipLaunchCooperativeKernel((const void*)k<1>, dim3(256), dim3(256), args, 0, 0));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            Ctl h; CK(hipMemcpy(&h, C, sizeof(Ctl), hipMemcpyDeviceToHost));
            printf("%s, %d workgroups per XCD: %.2f us per round, errors %u, timeout %u, xcd histogram", mode ? "cross-XCD (x^1)" : "same XCD", per_xcd,
                   ms * 1e3 / iters, h.errors[0], h.timeout[0]);
            for (int i = 0; i < 8; ++i) printf(" %u", h.xcd_hist[i]);
            printf("\n");
        }
    for (int sleep = 0; sleep < 2; ++sleep) {
        CK(hipMemset(C, 0, sizeof(Ctl)));
        CK(hipEventRecord(e0));
        void* args[] = {&C, &buf, (void*)&iters, (void*)&sleep};
        CK(hipLaunchCooperativeKernel((const void*)kmix, dim3(256), dim3(256), args, 0, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        Ctl h; CK(hipMemcpy(&h, C, sizeof(Ctl), hipMemcpyDeviceToHost));
        printf("mixed 16+16 over an XCD pair (2 roles per pair), s_sleep=%d: %.2f us per round, errors %u, timeout %u\n", sleep, ms * 1e3 / iters, h.errors[0], h.timeout[0]);
    }
    return 0;
}
