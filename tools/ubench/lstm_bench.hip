// Developer tool: per-launch cost of lstm_step_kernel<16> (D=512, B=64) by number of active roles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../audiocodecs_amd/csrc/lstm.h"
using namespace ac;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main() {
    const int B = 64, D = 512, T = 300;
    float *w, *h, *gin, *c, *y, *bias;
    CK(hipMalloc(&w, 3ull * 4 * D * D * 4)); CK(hipMalloc(&h, 2ull * (T + 1) * B * D * 4)); CK(hipMalloc(&gin, 2ull * T * B * 4 * D * 4));
    CK(hipMalloc(&c, 2ull * B * D * 4)); CK(hipMalloc(&y, 1ull * T * B * D * 4)); CK(hipMalloc(&bias, 4 * D * 4));
    CK(hipMemset(w, 0, 3ull * 4 * D * D * 4)); CK(hipMemset(h, 0, 2ull * (T + 1) * B * D * 4)); CK(hipMemset(gin, 0, 2ull * T * B * 4 * D * 4));
    CK(hipMemset(c, 0, 2ull * B * D * 4)); CK(hipMemset(bias, 0, 4 * D * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long BD = (long long)B * D, B4D = (long long)B * 4 * D;
    for (int nroles = 1; nroles <= 3; ++nroles)
        for (int mask : {7, 1, 2, 4, 3, 5, 6}) {
            if (nroles == 1 && mask != 7) continue;
            if (nroles == 2) continue;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                for (int s = 1; s < T; ++s) {
                    LstmLaunchParams q{}; q.B = B; q.D = D;
                    LstmRole& r0 = q.role[0]; r0.active = mask & 1; r0.kind = 0; r0.a = h + (s - 1) * BD; r0.wpk = w; r0.gin = gin + s * B4D; r0.hnext = h + s * BD; r0.c = c; r0.first = 0;
                    LstmRole& r1 = q.role[1]; r1.active = (mask >> 1) & 1; r1.kind = 1; r1.a = h + (s - 1) * BD; r1.wpk = w + 4ull * D * D; r1.bias = bias; r1.gout = gin + (T + s) * B4D;
                    LstmRole& r2 = q.role[2]; r2.active = (mask >> 2) & 1; r2.kind = 0; r2.a = h + (T + 1 + s - 1) * BD; r2.wpk = w + 8ull * D * D; r2.gin = gin + (T + s - 1) * B4D; r2.hnext = h + (T + 1 + s) * BD; r2.c = c + BD; r2.first = 0;
                    r2.skip = h; r2.skip_bs = D; r2.yout_elu = y + (long long)s * D; r2.y_bs = (long long)T * D;
                    hipLaunchKernelGGL(lstm_step_kernel<16>, dim3(D / 4, 2, nroles), dim3(256), 0, 0, q);
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) printf("grid.z=%d active-mask=%d : %.2f us per launch\n", nroles, mask, ms * 1e3 / (T - 1));
            }
        }
    return 0;
}
