// Micro-benchmark 2: the SELF-VALIDATING exchange of lstm_persist6.h (no flags: consumers poll the payload itself) under
// different placements of a 32-workgroup role and cache policies.  Each round: publish 1 KB tagged with the round number,
// then re-read the role's 32 KB until every 16-byte piece carries the tag.
//   placement 0: the role's 32 workgroups all on ONE XCD (recurrent exchange stays inside one L2)
//   placement 1: 16 + 16 over an XCD pair (what lstm_persist6.h does today)
// AUX: cache-policy bits of the payload loads / stores (1 = sc0, 16 = sc1, 17 = both)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }

struct Ctl { unsigned slot_cnt[8 * 16]; unsigned timeout[16]; unsigned spins[16]; };

// FRESH 0: a 4-deep ring of slots (lines stay in the L2 that wrote them); 1: a fresh slot every round, filled with a "not yet
// written" pattern by the host (what lstm_persist6.h does); 2: fresh slot, but every workgroup pre-writes the pattern into its
// slice of the slot 4 rounds ahead (sc0 store: the line is then dirty in the local L2 when the early polls arrive)
template <int PLACE, int AUXS, int AUXL, int FRESH = 0>
__global__ __launch_bounds__(256) void k(Ctl* C, unsigned* buf, int iters) {
    __shared__ unsigned s_slot, s_x;
    if (threadIdx.x == 0) { s_x = xcc_id(); s_slot = __hip_atomic_fetch_add(&C->slot_cnt[s_x * 16], 1u, RLX); }
    __syncthreads();
    const unsigned x = s_x & 7, slot = s_slot;
    if (slot >= 32) return;
    unsigned role, idx;
    if (PLACE == 0) { role = x; idx = slot; }                                  // role = XCD
    else { role = (x >> 1) * 2 + (slot >> 4); idx = (x & 1) * 16 + (slot & 15); }   // two roles per XCD pair
    unsigned long long total_spins = 0;
    for (int it = 1; it <= iters; ++it) {
        unsigned* base = buf + ((size_t)(FRESH ? it : (it & 3)) * 8 + role) * 32 * 256;       // [slot][role][32 slices][256 words]
        if (FRESH == 2 && threadIdx.x >= 64 && threadIdx.x < 128 && it + 4 <= iters) {
            unsigned* nb = buf + ((size_t)(it + 4) * 8 + role) * 32 * 256;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(nb + idx * 256), 0, 1024, 0x00020000);
            const u32x4 v = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, (threadIdx.x - 64) * 16, 0, 1);
        }
        if (FRESH == 3) {   // publish like lstm_persist6.h: every thread two 2-byte stores (a wave instruction = one 128-byte line)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + idx * 256), 0, 1024, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)it, rs, threadIdx.x * 2, 0, AUXS);
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)it, rs, 512 + threadIdx.x * 2, 0, AUXS);
        } else if (threadIdx.x < 64) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + idx * 256), 0, 1024, 0x00020000);
            const u32x4 v = {(unsigned)it, (unsigned)it, (unsigned)it, (unsigned)it};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, threadIdx.x * 16, 0, AUXS);
        }
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 32 * 1024, 0x00020000);
        // every wave validates its own quarter (8 KB = 8 slices), like the K quarters of the LSTM
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (unsigned spins = 0;; ++spins) {
            unsigned bad = 0;
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, wave * 8192 + j * 1024 + lane * 16, 0, AUXL);
            const unsigned want = FRESH == 3 ? ((unsigned)it & 0xffffu) * 0x10001u : (unsigned)it;
#pragma unroll
            for (int j = 0; j < 8; ++j) bad |= (v[j].x ^ want) | (v[j].w ^ want);
            if (__all(bad == 0u)) { total_spins += spins; break; }
            if (spins > (1u << 18)) { __hip_atomic_store(&C->timeout[0], 1u, RLX); return; }
            if ((spins & 63) == 63 && __hip_atomic_load(&C->timeout[0], RLX)) return;
            __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(&C->spins[0], (unsigned)(total_spins / iters));
}

template <int PLACE, int AUXS, int AUXL, int FRESH = 0>
void run(const char* what) {
    Ctl* C; unsigned* buf;
    const int iters_ = FRESH == 3 ? 3000 : 4000;
    const size_t words = (size_t)(FRESH ? iters_ + 1 : 4) * 8 * 32 * 256;
    CK(hipMalloc(&C, sizeof(Ctl)));
    CK(hipMalloc(&buf, words * 4));
    CK(hipMemset(C, 0, sizeof(Ctl)));
    CK(hipMemset(buf, FRESH ? 0xff : 0, words * 4));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int iters = iters_;
    void* args[] = {&C, &buf, (void*)&iters};
    CK(hipEventRecord(e0));
    CK(hipLaunchCooperativeKernel((const void*)k<PLACE, AUXS, AUXL, FRESH>, dim3(256), dim3(256), args, 0, 0));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    Ctl h; CK(hipMemcpy(&h, C, sizeof(Ctl), hipMemcpyDeviceToHost));
    printf("[%s] %-46s store aux %2d, load aux %2d: %.2f us per round, timeout %u, mean polls/round/wg %.1f\n", FRESH == 0 ? "ring of 4 slots" : FRESH == 1 ? "fresh slot/round" : FRESH == 2 ? "fresh + pre-touch" : "fresh, b16 stores", what, AUXS, AUXL, ms * 1e3 / iters, h.timeout[0], h.spins[0] / 256.0);
    CK(hipFree(C)); CK(hipFree(buf));
}

int main() {
    run<1, 16, 16>("16+16 over an XCD pair");
    run<0, 16, 16>("all 32 on one XCD");
    run<0, 1, 1>("all 32 on one XCD");
    run<0, 0, 1>("all 32 on one XCD");
    run<0, 16, 1>("all 32 on one XCD");
    run<0, 1, 16>("all 32 on one XCD");
    run<0, 17, 17>("all 32 on one XCD");
    run<1, 17, 17>("16+16 over an XCD pair");
    run<1, 16, 16>("16+16 over an XCD pair (again)");
    run<0, 1, 16, 1>("all 32 on one XCD");
    run<0, 1, 16, 2>("all 32 on one XCD");
    run<1, 16, 16, 1>("16+16 over an XCD pair");
    run<0, 16, 16, 1>("all 32 on one XCD");
    run<0, 1, 16, 0>("all 32 on one XCD");
    run<0, 1, 16, 3>("all 32 on one XCD, 2-byte stores");
    run<1, 16, 16, 3>("16+16 over an XCD pair, 2-byte stores");
    return 0;
}
