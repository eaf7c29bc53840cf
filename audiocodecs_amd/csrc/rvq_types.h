// Parameter blocks of the codebook kernels (rvq.h, rvq16.h): plain data, shared with the host-side headers of every
// translation unit (the kernels themselves are compiled in core.hip only).
#pragma once

namespace ac {

struct RvqEncParams {
    const float* x;      // [F][H] frames (channels-last encoder output), F = B*N
    const float* epk;    // packed codebooks: [K][C/16 code tiles][H/16 ksteps][64 lanes][4]
    const float* e;      // plain codebooks [K][C][H] (for the residual update)
    const float* ee;     // [K][C] squared norms of the code vectors
    long long* toks;     // [F][tK]: stage k of this launch writes column tk0 + k
    int F, H, C, K;
    int xs;              // row pitch of x (floats)
    int tK, tk0;
};

struct RvqDecParams {
    const long long* toks;  // [F][tK]: stage k reads column tk0 + k
    const float* e;         // [K][C][H] codebooks of the stages summed here
    float* out;             // [F][H]
    int F, H, C, K;
    int tK, tk0;
    int os;                 // row pitch of out (floats)
    unsigned* bad;          // sticky counter (host-mapped) raised when an id is outside [0, C); may be null
};

}  // namespace ac
