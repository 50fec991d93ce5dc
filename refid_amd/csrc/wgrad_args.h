// Kernel-side argument block shared by the weight-gradient partial-product kernels (conv_wgrad.hip, wgrad_bf16.hip):
// every one of them fills the same private slabs [split][tap][CoP][CiP] (+ bias slabs [split][CoP]).
#pragma once

#include "../../include/refid_hip.h"

struct WgKArgs {
    // up to REFID_WGRAD_MAX_GROUPS time steps of the same convolution (same geometry): their tiles form one K range, the
    // slabs are read-modify-written once per group instead of once per step (refid_wgrad_desc.groups)
    const float* g[REFID_WGRAD_MAX_GROUPS]; const float* inA[REFID_WGRAD_MAX_GROUPS]; const float* inB[REFID_WGRAD_MAX_GROUPS];
    int groups;
    int ldG, Co;
    int ldA, ldB, Ca, Ctot;
    float* slabs; float* bslabs;
    int N, H, W, Ho, Wo, pad;
    int tilesX, tilesY, ntiles, nsplit;
    int CoP, CiP;
    int accum;                 // add into the slabs instead of overwriting them
    // wgrad_pws.hip only, patch form (refid_wgrad_desc.algo 8): input pixel p sits at (p / patchW) * patchRow + (p % patchW) * ld
    // floats instead of p * ld (both sources: the even / odd rows of ONE tensor); 0 = dense
    int patchW = 0, patchRow = 0;
};

// wgrad_bf16.hip: 3x3 / stride-1 partial products with bf16 MFMA operands; geometry = the fp32 W3 plan
// (64 x 64 channel tile, 2 x 32 pixel tiles), grid (nsplit, nciT, ncoT)
int refid_wgrad_bf16_launch(const WgKArgs& a, int nciT, int ncoT, hipStream_t st);

// wgrad_wino24.hip: streaming first stage of a split-K slab reduction (S partial slabs out of nsplit; 0 = nothing to fold)
int refid_slab_fold_count(long long slabFloats, int nsplit);
int refid_launch_slab_fold(const float* slabs, float* part, long long slabFloats, int nsplit, int S, hipStream_t st);

// wgrad_pws.hip: streaming 1x1 weight gradient (LDS-DMA ring, fp32 MFMA); geometry of its slabs [split][CoP][CiP]
bool refid_wgrad_pws_ok(const refid_wgrad_desc* d);
int refid_wgrad_pws_pixels_per_buffer(const refid_wgrad_desc* d);
void refid_wgrad_pws_geo(const refid_wgrad_desc* d, int* ncoT, int* nciT, int* nsplit, int* CoP, int* CiP);
int refid_wgrad_pws_launch(const refid_wgrad_desc* d, const WgKArgs& a, int nciT, int ncoT, hipStream_t st);

// Deferred second stage of the slab reductions (refid_wgrad_desc.phase = 4 + refid_wgrad_finish_flush): a phase-4 call runs its
// streaming fold at once and QUEUES its element-wise stage; the flush issues every queued stage of a family as ONE launch
// (the job blocks travel as kernel arguments, at most REFID_FINISH_BATCH per launch: no device table, graph-capturable).
// ~130 dependent 10-300 us launches of 1-30 workgroups per step become three or four.
constexpr int REFID_FINISH_BATCH = 40;
bool refid_finish_defer_now();                          // conv_wgrad.hip: is the running refid_conv2d_wgrad call a phase-4 call?
int refid_wino24_finish_flush(hipStream_t st);          // wgrad_wino24.hip: its queue
int refid_slab_fold_flush(hipStream_t st);              // wgrad_wino24.hip: the queued first stages (before either family's second)
