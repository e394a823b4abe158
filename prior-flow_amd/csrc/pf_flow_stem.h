// Shared between pf_elem_kernels.hip (the dispatcher of pf_conv2d_direct[_group]) and pf_flow_stem.hip (the kernel).
#pragma once

struct PfFlowStemProblem {
    const float* in; int ld_in, c_in_off;          // channel-last rows [B*H*W][ld_in], the two flow channels at c_in_off
    const float* w; const float* bias;             // [7*7*2][128] fp32 (k = (ky*7 + kx)*2 + c), [128]
    float* out; int ld_out, c_out_off;             // fp32 rows (may be NULL when the twin is given)
    void* out_split; int lds_out;                  // bf16 hi|lo split twin of `out` (may be NULL)
    int relu;
};
struct PfFlowStemMulti {
    PfFlowStemProblem p[4];                        // blockIdx.y = problem; all share B, H, W
    int B, H, W;
};

// PF_OK / error code
int pf_flow_stem_launch(const PfFlowStemMulti& m, int n, void* stream);
