// Kernel-side argument block shared by the weight-gradient partial-product kernels (conv_wgrad.hip, wgrad_bf16.hip):
// every one of them fills the same private slabs [split][tap][CoP][CiP] (+ bias slabs [split][CoP]).
#pragma once

struct WgKArgs {
    const float* g; int ldG, Co;
    const float* inA; const float* inB; int ldA, ldB, Ca, Ctot;
    float* slabs; float* bslabs;
    int N, H, W, Ho, Wo, pad;
    int tilesX, tilesY, ntiles, nsplit;
    int CoP, CiP;
    int accum;                 // add into the slabs instead of overwriting them
};

// wgrad_bf16.hip: 3x3 / stride-1 partial products with bf16 MFMA operands; geometry = the fp32 W3 plan
// (64 x 64 channel tile, 2 x 32 pixel tiles), grid (nsplit, nciT, ncoT)
int refid_wgrad_bf16_launch(const WgKArgs& a, int nciT, int ncoT, hipStream_t st);
