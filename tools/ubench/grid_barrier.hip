// Micro-benchmark: cost of an in-kernel grid barrier on MI355X (256 workgroups, one per CU), flat
// counter vs XCD-hierarchical.  Every spin is bounded (timeout word) so a residency problem cannot hang.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Bar { unsigned cnt_x[8 * 16]; unsigned top[16]; unsigned gen_x[8 * 16]; unsigned flat[16]; unsigned timeout[16]; };

__device__ __forceinline__ bool wait_ge(unsigned* w, unsigned v, unsigned* tmo) {
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(w, RLX) >= v) return true;
        if (spins > (1u << 22)) { __hip_atomic_store(tmo, 1u, RLX); return false; }
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ bool barrier_flat(Bar* B, unsigned epoch, unsigned nblk) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&B->flat[0], 1u, RLX);
        ok = wait_ge(&B->flat[0], epoch * nblk, &B->timeout[0]);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}
__device__ __forceinline__ bool barrier_xcd(Bar* B, unsigned epoch, unsigned nblk) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned x = blockIdx.x & 7, per = nblk >> 3;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned old = __hip_atomic_fetch_add(&B->cnt_x[x * 16], 1u, RLX);
        if (old == epoch * per - 1) {
            const unsigned old2 = __hip_atomic_fetch_add(&B->top[0], 1u, RLX);
            if (old2 == epoch * 8 - 1)
                for (int i = 0; i < 8; ++i) __hip_atomic_store(&B->gen_x[i * 16], epoch, RLX);
        }
        ok = wait_ge(&B->gen_x[x * 16], epoch, &B->timeout[0]);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}
template <int KIND>
__global__ __launch_bounds__(256) void k(Bar* B, float* buf, int iters, int payload) {
    float acc = 0.f;
    for (int it = 1; it <= iters; ++it) {
        if (payload) {   // each block publishes 512 B, then reads 64 KB written by the others
            buf[((it & 1) * gridDim.x + blockIdx.x) * 128 + (threadIdx.x & 127)] = acc + it;
        }
        const bool ok = KIND == 0 ? barrier_flat(B, it, gridDim.x) : barrier_xcd(B, it, gridDim.x);
        if (!ok) return;
        if (payload) {
            const float* src = buf + (it & 1) * gridDim.x * 128;
            for (int i = threadIdx.x; i < 16384; i += 256) acc += src[i];
        }
    }
    if (acc == 12345.f) buf[0] = acc;
}
int main() {
    Bar* B; float* buf; CK(hipMalloc(&B, sizeof(Bar))); CK(hipMalloc(&buf, 2 * 256 * 128 * 4 + 65536 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int payload = 0; payload < 2; ++payload)
        for (int kind = 0; kind < 2; ++kind) {
            CK(hipMemset(B, 0, sizeof(Bar)));
            CK(hipEventRecord(e0));
            void* args[] = {&B, &buf, (void*)&iters, &payload};
            if (kind == 0) CK(hipLaunchCooperativeKernel((const void*)k<0>, dim3(256), dim3(256), args, 0, 0));
            else CK(hipLaunchCooperativeKernel((const void*)k<1>, dim3(256), dim3(256), args, 0, 0));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned tmo; CK(hipMemcpy(&tmo, &B->timeout[0], 4, hipMemcpyDeviceToHost));
            printf("%s barrier, payload=%d : %.2f us per iteration (timeout flag %u)\n", kind ? "xcd-hierarchical" : "flat counter", payload, ms * 1e3 / iters, tmo);
        }
    return 0;
}
