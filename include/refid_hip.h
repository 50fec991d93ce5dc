/*
 * refid_hip.h -- C ABI of the MI355X-native REFID hot path (librefid_hip.so).
 *
 * The reference (AHupuJR/REFID) is pure Python/PyTorch and has no FFI of its own:
 * every arithmetic op on its hot path is a torch.nn call that lands in cuDNN/ATen
 * ("Native / CUDA kernel inventory: empty").  The entry points
 * below are therefore the operator-level boundary a maintainer would bind instead of
 * those torch.nn calls; each one cites the reference call site(s) it replaces
 * (paths relative to /root/reference/basicsr/models/archs unless noted).
 *
 * Conventions
 *  - plain C: raw DEVICE pointers, ints, floats; no torch types.
 *  - activations are NHWC fp32 ("pixel-major": channel is the fastest axis), every
 *    tensor is described by a base pointer and a pixel pitch `ld` (floats) so channel
 *    slices of wider buffers can be addressed; base pointers and pitches are multiples
 *    of 4 floats (16 B).
 *  - every function is asynchronous on the caller's `stream` (a hipStream_t passed as
 *    void*), allocates nothing, never synchronises, keeps no global mutable state and is
 *    re-entrant: scratch memory is always the CALLER's (refid_conv_workspace_bytes,
 *    refid_wgrad_workspace_bytes), so launches on different streams use different buffers.
 *  - return value: 0 = ok, non-zero = error; refid_last_error() returns a thread-local
 *    message for the last failing call on this thread.
 */
#ifndef REFID_HIP_H
#define REFID_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REFID_ABI_VERSION 9     /* 9: refid_conv_desc.algo 5 with mfma_terms 3 (Winograd x three fp16 products) / algo 4 with mfma_terms 19 (conv_down on three fp16 products), refid_pack_conv_weights_wino3h / _split_f16, pack-table kinds 5 / 6 + refid_pack_batch_prepass; 8: refid_wgrad_desc.phase 4 + refid_wgrad_finish_flush (batched slab reductions), refid_*_tb layout conversions, REFID_ROLE_CONVT_DGRAD_PW, refid_wgrad_desc.algo 8, refid_rows_sum_defer / _flush; 7: refid_wgrad_desc.algo 5 (2x4 Winograd tiles) / 6; thin-output 3x3 weight gradient */

const char* refid_last_error(void);
int refid_abi_version(void);
/* Always 0 since ABI 9: the tiles that were measured and lost (the persistent and the wide Winograd tiles, the six-product /
 * LDS-DMA / F(3x3,4x4) weight gradients; REFID_EXPERIMENTAL_TILES=1 builds of ABI <= 8) are no longer in the tree.  Kept so that
 * callers written against earlier versions still link. */
int refid_experimental_tiles(void);

/* Device facts used by the host-side scheduler (CU count etc.); -1 on failure. */
int refid_device_cu_count(void);

/* ------------------------------------------------------------------------------------
 * Fused convolution tile:   out = mask( post( pre(conv(src) + bias) + res ) )
 *
 *   src   = in_a                    (c_b == 0)
 *         = [in_a | in_b]           channel concatenation WITHOUT materialising the cat
 *   pre   = LeakyReLU(slope_pre)    (1.0 = identity, 0.0 = ReLU)
 *   post  = LeakyReLU(slope_post)   applied after the residual add
 *   mask  = multiply by (mask_src > 0 ? 1 : slope_mask)   (activation derivative, backward)
 *
 * Replaces (forward): nn.Conv2d + LeakyReLU/ReLU + residual adds + torch.cat at
 *   recurrent_sub_modules.py:41-49 (ImageEncoderConvBlock), :74-84 (ConvLayer),
 *   :270-296 (EvR level: conv, cat+fuse_two_dir, down), :386-408 (decoder: ConvTranspose2d,
 *   cat), :488-503 (ResidualBlock), :659-678 + :719-726 + :755-758 (EvR hidden-state
 *   update trunk), fusion_modules.py:300-331 (the 1x1 convs of EGACA),
 *   XXNet_final_attenfusion_arch.py:147-149,215 (heads, pred).
 * Replaces (backward): the autograd conv-backward "input gradient" of the same calls
 *   (SURVEY.md Appendix A.2): dgrad is the same tile run on the transposed/flipped
 *   packed weights (refid_pack_conv_weights with role=REFID_ROLE_DGRAD).
 *
 * mode 0: ordinary conv, kh x kw, stride 1 or 2, zero padding `pad`.
 * mode 1: ConvTranspose2d(k=2,s=2) forward as a 1x1 GEMM with 4*Co columns and a
 *         pixel-shuffle store (column j -> (dy,dx)=(j/Co/2, j/Co%2), channel j%Co);
 *         `cout` is 4*Co, (h,w) the INPUT size, output is (2h,2w).
 * mode 2: input-gradient of conv4x4/stride2/pad1 (`conv_down`): four output-parity
 *         classes, each a 2x2-tap conv over the (h,w) gradient, stored at (2y+py,2x+px).
 * ---------------------------------------------------------------------------------- */
/* Fusions around a POINTWISE conv (algo 3, more than 32 output channels) -- the non-GEMM steps of EGACA
 * (fusion_modules.py:290-333) folded into its 1x1 convs; every member is optional (NULL / 0 = off):
 *   ln_*   LayerNorm2d PROLOGUE (fm:97-108,300,302,323-325): the operand is normalised over its (<= 64, one source)
 *          channels -- mean / biased variance, eps inside the root, times ln_gamma plus ln_beta -- before the GEMM;
 *          ln_out (pitch ld_ln_out) receives the normalised tensor (the training stash conv's weight gradient needs).
 *   se_*   SQUEEZE-EXCITE in the kernel (fm:253-260,309-315): `pool` holds the per-workgroup partial sums of the
 *          depthwise kernel, (n, pool_parts, se_c); every workgroup derives its sample's
 *          s = sigmoid(W2 relu(W1 mean + b1) + b2) and multiplies operand channel k by s[k mod se_c] as it is loaded
 *          ([xi*s | xe*s] is never materialised; xs_out optionally receives it for conv3's weight gradient);
 *          se_m / se_z1 / se_s (n,se_c) / (n,se_c/2) / (n,se_c) optionally receive the vectors SE's backward needs.
 *          hw = pixels per sample, must be a multiple of 128.
 *   res2   second residual tensor added with `res` (y = ev + img + beta*conv3(.), fm:319).
 *   out2   receives GELU_erf(out) (fm:327-329); `out` keeps the pre-activation for the backward pass. */
typedef struct refid_pw_extras {
    const float* ln_gamma; const float* ln_beta; float ln_eps; float* ln_out; int ld_ln_out;
    const float* pool; int pool_parts; float inv_hw; int hw; int se_c;
    const float* se_w1; const float* se_b1; const float* se_w2; const float* se_b2;
    float* se_m; float* se_z1; float* se_s;
    float* xs_out; int ld_xs_out;
    const float* res2; int ld_res2;
    float* out2; int ld_out2;
} refid_pw_extras;

typedef struct refid_conv_desc {
    const float* in_a;  const float* in_b;      /* NHWC sources                         */
    int ld_a, ld_b;                             /* pixel pitch (floats)                 */
    int c_a, c_b;                               /* channels taken from each source      */
    const float* w_packed;                      /* see refid_pack_conv_weights          */
    const float* bias;                          /* [cout] or NULL                       */
    float* out;          int ld_out;
    const float* res;    int ld_res;            /* NULL = none                          */
    const float* mask;   int ld_mask;           /* NULL = none                          */
    int n, h, w;                                /* input batch / height / width         */
    int ho, wo;                                 /* GEMM pixel grid (output h,w; mode 1/2: = h,w) */
    int cout;                                   /* GEMM columns computed by this call   */
    int cout_pad;                               /* rows per tap in w_packed             */
    int co_base;                                /* first row of w_packed / bias used    */
    int kh, kw, stride, pad;
    int mode;
    float slope_pre, slope_post, slope_mask;
    int algo;                                   /* 0 = direct implicit GEMM; 1 = Winograd F(2x2,3x3)
                                                   (3x3, stride 1, mode 0 only; w_packed must come
                                                   from the REFID_ROLE_WINO_* packings);
                                                   3 = register-operand pointwise tile (1x1, stride 1,
                                                   mode 0; w_packed = REFID_ROLE_FWD/DGRAD packing with
                                                   kc = 8) -- and mode 1, ConvTranspose2d(2,2) as the 1x1
                                                   GEMM over its 4 Co columns with a pixel-shuffle store
                                                   (REFID_ROLE_CONVT packing, kc = 8, bn = 32; bias / res /
                                                   slopes only; Co a multiple of 4, 4 Co > 32) -- and the 2x2
                                                   stride-2 conv (ConvTranspose2d's input gradient: non-
                                                   overlapping patches) as one GEMM with K = 4 c_a: a pixel's
                                                   patch is two contiguous runs of 2 c_a floats (dense pixels:
                                                   ld_a == c_a, one source; REFID_ROLE_CONVT_DGRAD_PW packing;
                                                   out2 / add2 supported);
                                                   2 = direct tile with bf16 MFMA operands (fp32
                                                   accumulate/epilogue/tensors; w_packed from
                                                   refid_pack_conv_weights_bf16 with kc doubled);
                                                   4 = direct 3x3 / stride-1 tile with SPLIT fp32 operands on
                                                   the bf16 matrix cores (mode 0; w_packed from
                                                   refid_pack_conv_weights_split with the same number of
                                                   planes): every fp32 operand is the exact sum of three bf16
                                                   numbers, six bf16 MFMAs give the fp32 product to one
                                                   rounding (see `mfma_terms`);
                                                   5 = Winograd F(2x2,3x3) with the transform-domain products on the bf16
                                                   matrix cores, six bf16 MFMAs per fp32 product on exactly split
                                                   operands (3x3, stride 1, mode 0; a 64-channel workgroup tile above 32 output
                                                   channels, a 32-channel one up to 32, thin outputs included --, two
                                                   sources: c_a a multiple of 16; w_packed from
                                                   refid_pack_conv_weights_wino6): the fp32 Winograd tile's result to
                                                   fp32 rounding at 2.67x fewer matrix-pipe cycles -- or, with
                                                   mfma_terms = 3, THREE fp16 products on two-plane operands (22 bits;
                                                   w_packed from refid_pack_conv_weights_wino3h): half the MFMAs again  */
    int split_k;                                /* small problems (few output tiles, long K): split K over the grid into
                                                   `ws` partial sums + a finishing pass.  0 = never; 1 = decided by the
                                                   per-sample geometry (a sample's bits do not depend on the batch
                                                   size); 2 = decided by the total grid size (best at 1-2 samples per
                                                   GPU).  Used by algo 1 / 5 and by algo 0's 4x4/s2 tiles (mode 0 and 2).  */
    int wino_tile;                              /* algo 5: 0 = the library chooses between the 64- and the 32-output-channel
                                                   workgroup tile; 1 = the 64-channel tile wherever cout > 32; 4 = the
                                                   32-channel tile everywhere (an A/B switch: measured 9-35 % slower on the
                                                   64-channel layers).  Same results whichever runs.  (2 / 3 selected tiles
                                                   that were removed with ABI 9 -- a persistent and a wide one, both measured
                                                   slower -- and are ignored.)                                          */
    const refid_pw_extras* pw;                  /* algo 3 only: fusions around the pointwise conv, or NULL               */
    int mfma_terms;                             /* algo 3: 0 = fp32 MFMA products; 6 = six bf16 products on exactly split
                                                   operands (w_packed from refid_pack_conv_weights_split with kh = kw = 1,
                                                   three planes; channel counts multiples of 16, more than 32 outputs).
                                                   algo 4: 0 / 6 = three bf16 planes per operand, six products
                                                   (error <= 2^-23 per product: fp32 class, closer to the fp64 result
                                                   than the fp32 Winograd tile); 3 = two planes, three products
                                                   (2^-16 per product: explicit opt-in, still finer than TF32); 19 (16 + 3;
                                                   4x4 stride 2 forward / input gradient only) = three fp16 products on
                                                   two-plane operands scaled by exact powers of two (w_packed from
                                                   refid_pack_conv_weights_split_f16): ~2^-22 per product, the fp32 class
                                                   at half of 6's MFMAs.
                                                   algo 5: 0 / 6 = six bf16 products on three-plane operands; 3 = three
                                                   fp16 products on two-plane operands h = rne16(v), l = rne16(v - h)
                                                   (~2^-22 per product, below the fp32 accumulation's own error at
                                                   K >= 144; both operands travel scaled by exact powers of two -- U per
                                                   packing, V per Winograd tile and online along K -- so any finite fp32
                                                   magnitude keeps the same relative error)                         */
    float* ws;  size_t ws_bytes;                /* caller's scratch, >= refid_conv_workspace_bytes(d) bytes, 16-byte
                                                   aligned, private to this stream until the call's work has run;
                                                   NULL: never split                                            */
    const float* add2; int ld_add2;             /* second output (algo 0 / 1 / 2 / 4 / 5; NULL = off): out2 = out + add2, same shape */
    float* out2;       int ld_out2;             /* as `out` -- the skip sum the reference forms right after this conv
                                                   (XXNet_final_attenfusion_arch.py:16-17,199-203,211) or, in BPTT, the sum of
                                                   this input gradient with another branch's, written by the producing tile
                                                   instead of a separate add kernel.  `out` is still written.            */
    int mask_mode;                              /* 0: the activation-derivative mask above.  1 (algo 3 only): multiply by
                                                   GELU_erf'(mask) instead -- the input gradient of conv5 leaves the tile as
                                                   the gradient of conv4's output (fusion_modules.py:327-329 backward)   */
} refid_conv_desc;

/* Scratch bytes refid_conv2d(d) wants in d->ws (0 = none); depends on the geometry fields and split_k only. */
size_t refid_conv_workspace_bytes(const refid_conv_desc* d);
int refid_conv2d(const refid_conv_desc* d, void* stream);

/* Channel-chunk width (KC) and row padding (BN) the conv tile for this geometry wants
 * its packed weights in. */
int refid_conv_kc(int kh, int kw, int stride, int mode);
/* "Cfg<...>" template signature of the tile refid_conv2d launches for this geometry (matches the
 * kernel name rocprofv3 reports); used by bench.py to attribute time to the dominant kernel. */
const char* refid_conv_tile_name(int kh, int kw, int stride, int mode, int cout);
int refid_conv_bn(int kh, int kw, int stride, int mode, int cout);

/* ------------------------------------------------------------------------------------
 * Weight gradient + bias gradient of the same convolutions (autograd's conv-backward
 * "weight gradient"):
 *   dW[o][i][ky][kx] (+)= sum_{n,y,x} g[n,y,x,o] * src[n, y*s+ky-pad, x*s+kx-pad, i]
 *   db[o]            (+)= sum_{n,y,x} g[n,y,x,o]
 * written in the reference's own parameter layout (OIHW for Conv2d; for
 * ConvTranspose2d pass g := the layer input, src := the output gradient, which yields
 * IOHW).  Two stages: partial products per pixel split into `slabs`, then a reduction
 * that ACCUMULATES into dw/db (gradients of weights shared over the T steps add up,
 * SURVEY.md A.2 first row).
 * ---------------------------------------------------------------------------------- */
#define REFID_WGRAD_MAX_GROUPS 24
typedef struct refid_wgrad_desc {
    const float* g;      int ld_g;   int c_o;   /* output-gradient, (n,ho,wo,c_o)       */
    const float* in_a;   const float* in_b;     /* conv input sources, (n,h,w,c_a|c_b)  */
    int ld_a, ld_b;
    int c_a, c_b;
    float* dw;                                  /* (c_o, c_a+c_b, kh, kw) accumulated   */
    float* db;                                  /* (c_o) accumulated, or NULL           */
    float* slabs;                               /* workspace, refid_wgrad_workspace_bytes */
    int n, h, w, ho, wo;
    int kh, kw, stride, pad;
    int i_base, i_total;                        /* dw second-dim offset / full size (slices) */
    int o_real;                                 /* rows of dw/db actually written (<= c_o); the
                                                   rest of g's channels is padding        */
    int algo;                                   /* 0 = direct; 1 = Winograd F(2x2,3x3) (3x3 stride 1);
                                                   2 = direct with bf16 MFMA operands (3x3 stride 1 pad 1, more than 32
                                                   output and input channels; fp32 accumulation / slabs / dw; slab
                                                   geometry and phases identical to algo 0);
                                                   (3, 4, 6: experiments of ABI <= 8 -- six bf16 products, LDS-DMA staging,
                                                   F(3x3,4x4) -- that did not beat algo 1 / 5: rejected with a message);
                                                   5 = Winograd over 2x4 tiles of g (3x3 stride 1): F(3,2) down the rows,
                                                   F(3,4) along them -- 24 instead of 32 fp32 MFMAs per 8 pixels, packed
                                                   transforms (wgrad_wino24.hip; slabs [split][24][o][i]; pitches / channel
                                                   counts multiples of 4 floats, 16-byte aligned tensors, c_a % 32 == 0 for two
                                                   sources); deviation from the float64 gradient 3e-6 .. 6e-6 of scale;
                                                   7 = conv_down (4x4 stride 2 pad 1, even input size): algo 5's kernel on the four
                                                   parity phases of the input (a stride-2 conv is four 2x2-tap stride-1 convs on
                                                   them): 12 instead of 16 fp32 MFMA-units per output pixel; slabs
                                                   [split][phase][24][o][i];
                                                   8 = the 2x2 stride-2 pad-0 weight gradient over NON-overlapping patches
                                                   (ConvTranspose2d(2,2) with the roles swapped, see above) as one streaming 1x1
                                                   weight gradient (wgrad_pws.hip): K = (dy, dx, c) over the even / odd rows of
                                                   the source, which must have dense pixels (ld_a == c_a; one source, no db,
                                                   c_o >= 64 and a multiple of 32, c_a a multiple of 16, wo a multiple of 32);
                                                   the reduction permutes the columns into dw's [o][c][dy][dx] layout  */
    int phase;                                  /* 0 = partial products + reduction in one call;
                                                   weights shared over the T recurrent steps can instead
                                                   keep accumulating in their own `slabs`:
                                                   1 = partial products, OVERWRITE slabs (first step),
                                                   2 = partial products, ADD into slabs (later steps),
                                                   3 = reduction of the slabs into dw/db only (after BPTT);
                                                   4 = as 3, but only the streaming first stage runs now: the element-wise
                                                   stage is QUEUED and refid_wgrad_finish_flush issues every queued stage as
                                                   one launch per kernel family (a step's ~130 small reductions are
                                                   independent of each other; the caller must not touch `slabs`, dw or db
                                                   between the call and the flush; two queued calls on the same dw block
                                                   flush the queue in between);
                                                   the slab geometry depends on (c_o, i_total), not on c_a/c_b */
    int groups;                                 /* phases 0-2 (not the thin / 1x1 register tiles): 2..REFID_WGRAD_MAX_GROUPS = this launch also adds the
                                                   partial products of groups-1 MORE time steps of the same convolution
                                                   (weights are shared over the T steps; same geometry, pitches and
                                                   source split): one pass over the slabs instead of one per step.
                                                   0 / 1 = just (g, in_a, in_b).                                        */
    const float* g_more[REFID_WGRAD_MAX_GROUPS - 1];
    const float* in_a_more[REFID_WGRAD_MAX_GROUPS - 1];
    const float* in_b_more[REFID_WGRAD_MAX_GROUPS - 1];
} refid_wgrad_desc;

size_t refid_wgrad_workspace_bytes(const refid_wgrad_desc* d);
int refid_conv2d_wgrad(const refid_wgrad_desc* d, void* stream);
/* Issues the element-wise reduction stages queued by phase-4 calls (of this host thread) on `stream`: one launch per kernel
 * family and 40 jobs (autograd's accumulation of the T steps' weight gradients, twoImage_event_recurrent_model.py:303; the
 * reference has no counterpart -- its 2T per-step gradients are added by autograd one by one).  No queued job: no launch. */
int refid_wgrad_finish_flush(void* stream);

/* ------------------------------------------------------------------------------------
 * Weight packing: reference layouts (Conv2d OIHW, ConvTranspose2d IOHW) -> the tile's
 * [chunk][tap][row][KC] layout, rows/chunks zero padded.
 *   role FWD   : rows = O, k = I, taps in order
 *   role DGRAD : rows = I, k = O, taps flipped (stride-1 convs), or 2x2/s2 (convT input
 *                gradient), or the four parity classes of conv4x4/s2 (mode 2)
 *   role CONVT : ConvTranspose2d forward (mode 1): rows = (dy,dx,Co), k = Ci
 * ---------------------------------------------------------------------------------- */
enum { REFID_ROLE_FWD = 0, REFID_ROLE_DGRAD = 1, REFID_ROLE_CONVT = 2, REFID_ROLE_CONVT_DGRAD = 3,
       REFID_ROLE_DOWN_DGRAD = 4,
       /* Winograd-domain weights U = G g G^T, 16 "taps" (algo 1; kc = 8): */
       REFID_ROLE_WINO_FWD = 5, REFID_ROLE_WINO_DGRAD = 6,
       /* ConvTranspose2d(2,2) input gradient as ONE GEMM over 2x2 patches (refid_conv2d algo 3 on a 2x2 stride-2 conv): rows = Ci,
          k = (dy, dx, co) = 4 Co columns, one "tap"; kc = 8, bn = 32: */
       REFID_ROLE_CONVT_DGRAD_PW = 7 };
size_t refid_packed_weight_floats(int role, int o, int i, int kh, int kw, int kc, int bn);
int refid_pack_conv_weights(const float* w, float* packed, int role, int o, int i, int kh, int kw,
                            int kc, int bn, void* stream);

/* Same as refid_pack_conv_weights with a per-OUTPUT-channel factor folded in (FWD/DGRAD
 * roles).  Used to fold EGACA's beta / gamma (fusion_modules.py:287-288,319,331:
 * `y = ev + img + x*beta`, `return y + x*gamma`) into conv3 / conv5 so the scaled branch
 * and the residual add run in the conv tile's epilogue. */
int refid_pack_conv_weights_scaled(const float* w, const float* oscale, float* packed, int role, int o,
                                   int i, int kh, int kw, int kc, int bn, void* stream);
/* bf16 copy of the same packed layout (RNE), kc = 2 * refid_conv_kc(...); oscale may be NULL. */
int refid_pack_conv_weights_bf16(const float* w, const float* oscale, void* packed_bf16, int role, int o, int i,
                                 int kh, int kw, int kc, int bn, void* stream);
/* Split-bf16 packing for refid_conv2d algo 4: every weight (times oscale[row] when given) is written as `planes` bf16
 * numbers h = rne(v), m = rne(v - h), l = v - h - m (planes = 3: exact; 2: h, m; 1: h), in 8-channel sub-chunks:
 *   3x3 (REFID_ROLE_FWD / REFID_ROLE_DGRAD):     [chunk8][plane][tap 0..9][rows padded to bn][8]   (tap 9 = zeros)
 *   4x4 stride 2, REFID_ROLE_FWD (conv_down):     [chunk8][in-block row sy][in-block column sx][plane][block tap (ty,tx)]
 *                                                 [rows][8] = W[row][k][2ty+sy][2tx+sx]
 *   4x4 stride 2, REFID_ROLE_DOWN_DGRAD:          [parity class][chunk8][plane][tap (ta,tb)][rows][8]
 *   1x1 (REFID_ROLE_FWD / REFID_ROLE_DGRAD, refid_conv2d algo 3 with mfma_terms 6, bn = 32):
 *                                                 [chunk16][plane][rows padded to bn][16], a row's 16 channels stored as the
 *                                                 two K halves of the pointwise tile's lanes: slot 8h + t = channel 4h + t
 *                                                 (t < 4) or 8 + 4h + (t - 4)
 * refid_packed_weight_split_bytes gives the buffer size. */
size_t refid_packed_weight_split_bytes(int role, int o, int i, int kh, int kw, int bn, int planes);
/* The 3x3 / 4x4 layouts above with TWO fp16 planes h = rne16(w 2^eW), l = rne16(w 2^eW - h) behind a 64-byte header (int eW at
 * byte 0: max |w| 2^eW in [2^12, 2^13), found by a reduction over the tensor that the call launches first), for refid_conv2d
 * algo 4 with mfma_terms = 19.  `packed` must be 16-byte aligned. */
size_t refid_packed_weight_split_f16_bytes(int role, int o, int i, int kh, int kw, int bn);
int refid_pack_conv_weights_split_f16(const float* w, const float* oscale, void* packed, int role, int o, int i,
                                      int kh, int kw, int bn, void* stream);
int refid_pack_conv_weights_split(const float* w, const float* oscale, void* packed, int role, int o, int i,
                                  int kh, int kw, int bn, int planes, void* stream);
/* Winograd-domain weights U = G g G^T (times oscale[row] when given) for refid_conv2d algo 5: computed in fp32 like
 * REFID_ROLE_WINO_FWD / REFID_ROLE_WINO_DGRAD, then written as three bf16 planes that sum to U exactly:
 *   [chunk of 16 input channels][xi = 0..15][plane][rows padded to bn = 64][16]  (bf16). */
size_t refid_packed_weight_wino6_bytes(int role, int o, int i, int bn);
int refid_pack_conv_weights_wino6(const float* w, const float* oscale, void* packed, int role, int o, int i, int bn,
                                  void* stream);
/* The same U for refid_conv2d algo 5 with mfma_terms = 3: a 64-byte header (int eU at byte 0: the packing's power-of-two
 * scale, max |U| 2^eU in [2^12, 2^15), found by a reduction over the tensor that the call launches first) followed by
 *   [chunk of 16 input channels][xi = 0..15][plane h / l][rows padded to bn = 64][16]  (fp16),
 * h = rne16(U 2^eU), l = rne16(U 2^eU - h).  `packed` must be 16-byte aligned. */
size_t refid_packed_weight_wino3h_bytes(int role, int o, int i, int bn);
int refid_pack_conv_weights_wino3h(const float* w, const float* oscale, void* packed, int role, int o, int i, int bn,
                                   void* stream);
/* All packings of a model in ONE launch.  The caller builds a table of refid_pack_entry_bytes()-sized records in host
 * memory with refid_pack_entry_fill (kind 0: refid_pack_conv_weights[_scaled / _bf16] -- `planes` = 1 selects bf16 output;
 * 1: refid_pack_conv_weights_split, 3x3 / 4x4; 2: the same, 1x1; 3: refid_pack_conv_weights_wino6; 4: dst[e] = w[e] *
 * oscale[e] for e < o (refid_mul_vec); 5: refid_pack_conv_weights_wino3h; 6: refid_pack_conv_weights_split_f16), copies it to
 * device memory once, and calls
 * refid_pack_batch whenever the weights have changed -- after refid_pack_batch_prepass on the same stream when the table holds
 * kind-5 / kind-6 records (their scale exponents: one workgroup per record, a no-op for the other kinds).  `blk0` = the sum of the values returned for the records before this one (each call returns its record's
 * workgroup count >= 1, -1 on error -- the same argument checks as the one-by-one entry points); nblocks = the sum over
 * all records.  refid_pack_table_check walks a finished HOST table (every record filled, first blocks consecutive from 0:
 * the kernel finds a workgroup's record by binary search over them) and returns that sum, -1 on error.  Same bits as the
 * one-by-one calls. */
size_t refid_pack_entry_bytes(void);
int refid_pack_table_check(const void* table_host, int n);
int refid_pack_entry_fill(void* entry_host, int kind, const float* w, const float* oscale, void* dst, int role, int o, int i,
                          int kh, int kw, int kc, int bn, int planes, int blk0);
int refid_pack_batch_prepass(const void* table_dev, int n, void* stream);
int refid_pack_batch(const void* table_dev, int n, int nblocks, void* stream);
int refid_mul_vec(const float* a, const float* b, float* out, int n, void* stream);
/* After BPTT, turn the gradient of the FOLDED conv (scale[r]*W[r,:], scale[r]*b[r]) of THIS backward pass
 * (gw_folded, gb_folded: private buffers, zero before BPTT) into gradients of (W, b, scale) and ACCUMULATE them:
 *   dscale[r] += <W[r,:],Gf[r,:]> + b[r]*gbf[r];  gw[r,:] += scale[r]*Gf[r,:];  gb[r] += scale[r]*gbf[r].
 * (gw/gb may already hold gradients of earlier backward passes: gradient accumulation stays exact.) */
int refid_fold_back(const float* w, const float* b, const float* scale, const float* gw_folded,
                    const float* gb_folded, float* gw, float* gb, float* dscale, int rows, int k, void* stream);

/* ------------------------------------------------------------------------------------
 * EGACA non-GEMM pieces (fusion_modules.py:290-333) and LayerNorm2d (fusion_modules.py:97-134).
 * c in {16,32,64,128}.  Parameter gradients ACCUMULATE into dw/db (deterministic: per-workgroup partial rows in the
 * caller's scratch, finished in index order -- no floating-point atomics).
 * ---------------------------------------------------------------------------------- */
int refid_layernorm2d_fwd(const float* x, int ld_x, const float* w, const float* b, float* out,
                          int ld_out, long long npix, int c, float eps, void* stream);
/* LayerNormFunction.backward, fusion_modules.py:110-122; gx = (res ? res : 0) + dL/dx.  res may be
 * NULL or alias gx (in-place accumulate); pass a distinct gx when another stream still reads res. */
/* parts: scratch of refid_layernorm2d_bwd_parts(npix, c) * 2c floats (per-workgroup partial sums of dw / db, added in a
 * fixed order: the parameter gradients are deterministic, no floating-point atomics). */
int refid_layernorm2d_bwd_parts(long long npix, int c);
int refid_layernorm2d_bwd(const float* g, int ld_g, const float* x, int ld_x, const float* w, float* gx,
                          int ld_gx, const float* res, int ld_res, float* dw, float* db, float* parts, long long npix,
                          int c, float eps, void* stream);
/* ------------------------------------------------------------------------------------
 * SingleMultiConnectEVHINet non-GEMM pieces (SURVEY.md 8f row 4;
 * archs/single_multiconnect_evhinet_arch.py:233-237 HIN + LeakyReLU, archs/arch_util.py:421-426 FAC_bias).
 * Deterministic (two-stage reductions, no atomics).
 * ---------------------------------------------------------------------------------- */
int refid_hin_parts(int hw);                 /* partial-sum rows the scratch buffers must hold per sample */
/* out = LeakyReLU( [ InstanceNorm(x[:, :ch]) * gamma + beta | x[:, ch:] ] ); biased variance, eps inside the
 * root; stats (n,2,ch) receives mean / rstd; parts = scratch (n, refid_hin_parts(hw), 2, ch).  ch = 0: plain
 * LeakyReLU.  ch must be 4 * 2^k. */
int refid_hin_lrelu_fwd(const float* x, int ld_x, const float* gamma, const float* beta, float* out, int ld_out,
                        float* stats, float* parts, int n, int hw, int c, int ch, float eps, float slope,
                        void* stream);
/* gx = d/dx of the above given g = dL/dout; dgamma/dbeta ACCUMULATE; sums = scratch (n,2,ch). */
int refid_hin_lrelu_bwd(const float* g, int ld_g, const float* out, int ld_out, const float* x, int ld_x,
                        const float* gamma, const float* stats, float* gx, int ld_gx, float* dgamma,
                        float* dbeta, float* sums, float* parts, int n, int hw, int c, int ch, float slope,
                        void* stream);
/* FAC_bias: out = feat * filt[:, :c] + filt[:, c:2c]  and its backward (gfilt = [g*feat | g]). */
int refid_fac_fwd(const float* feat, int ld_feat, const float* filt, int ld_filt, float* out, int ld_out,
                  long long npix, int c, void* stream);
int refid_fac_bwd(const float* g, int ld_g, const float* feat, int ld_feat, const float* filt, int ld_filt,
                  float* gfeat, int ld_gfeat, float* gfilt, int ld_gfilt, long long npix, int c, void* stream);

/* pre = dwconv3x3(in)+b ; act = GELU(pre) ; pool[n][part][c] = per-workgroup partial sums of act
 * (fm:304-309 + se_1's AdaptiveAvgPool2d, fm:253-254; pool may be NULL; parts =
 * refid_dwconv_pool_parts(h,w,c); deterministic, no atomics).  w is the reference (c,1,3,3) tensor. */
int refid_dwconv_pool_parts(int h, int wd, int c);
int refid_dwconv3x3_gelu_fwd(const float* in, int ld_in, const float* w, const float* b, float* pre,
                             float* act, float* pool, int n, int h, int wd, int c, void* stream);
/* gd = gradient w.r.t. `pre`; gin = input gradient; dw/db accumulate.  parts: scratch of
 * n * refid_dwconv3x3_bwd_parts(h,wd,c) * 10c floats (deterministic two-stage reduction; required). */
int refid_dwconv3x3_bwd_parts(int h, int wd, int c);
int refid_dwconv3x3_bwd(const float* gd, const float* in, int ld_in, const float* w, float* gin,
                        float* dw, float* db, float* parts, int n, int h, int wd, int c, void* stream);
/* se_1 (fm:253-260): m = (sum_parts pool)*inv_hw ; z1 = relu(W1 m + b1) ; s = sigmoid(W2 z1 + b2) */
int refid_se_fwd(const float* pool, int n_parts, float inv_hw, const float* w1, const float* b1, const float* w2,
                 const float* b2, float* m, float* z1, float* s, int n, int c, void* stream);
/* scratch: n * (c + c/2) floats; the samples' contributions to dw1 / db1 / dw2 / db2 are added in sample order */
int refid_se_bwd(const float* gs, const float* s, const float* z1, const float* m, const float* w1,
                 const float* w2, float* gm, float* dw1, float* db1, float* dw2, float* db2, float* scratch, int n,
                 int c, void* stream);
/* out[n,p,0:c] = xi*s[n] ; out[n,p,c:2c] = xe*s[n]   (fm:312-315, no cat temporary) */
int refid_scale_cat(const float* xi, const float* xe, const float* s, float* out, int n, int hw, int c,
                    void* stream);
/* gs[n][c] = sum_p gxs[n,p,c]*xi + gxs[n,p,c+C]*xe  (dL/ds of the two products) */
/* parts: scratch of n * refid_egaca_gs_reduce_parts(hw, c) * c floats (fixed-order two-stage reduction) */
int refid_egaca_gs_reduce_parts(int hw, int c);
int refid_egaca_gs_reduce(const float* gxs, const float* xi, const float* xe, float* gs, float* parts, int n, int hw,
                          int c, void* stream);
/* gdwe = (gxs_e*s + gm*inv_hw) * GELU'(dwe) ; gxi (+)= gxs_i*s */
int refid_egaca_bwd_elem(const float* gxs, const float* s, const float* gm, float inv_hw, const float* dwe,
                         float* gdwe, float* gxi, int accumulate_xi, int n, int hw, int c, void* stream);
int refid_gelu_fwd(const float* in, float* out, long long count, void* stream);
int refid_gelu_bwd(const float* g, const float* in, float* out, long long count, void* stream);
/* db[c] += sum_p g[p][c]   (bias gradient where the wgrad call does not carry it); parts: scratch of
 * refid_colsum_parts(npix, c) * c floats (fixed-order two-stage reduction) */
int refid_colsum_parts(long long npix, int c);
int refid_colsum(const float* g, int ld_g, float* db, float* parts, long long npix, int c, void* stream);

/* ------------------------------------------------------------------------------------
 * Train-step tail (twoImage_event_recurrent_model.py:273-310; losses/losses.py:28-30,143-173).
 * ---------------------------------------------------------------------------------- */
/* Every scalar reduction of this section and of the validation tail below is TWO-STAGE and deterministic (no floating-point
 * atomics: a logged loss / PSNR / SSIM does not depend on workgroup arrival order): per-workgroup partial sums go to the
 * caller's `parts` scratch (refid_*_parts(...) doubles), a second launch adds them in index order. */
/* *loss_sum = sum sqrt((pred-gt)^2+eps); grad = (pred-gt)/sqrt(.)*grad_scale (grad may be NULL);
 * parts: refid_charbonnier_parts(count) doubles. */
int refid_charbonnier_parts(long long count);
int refid_charbonnier(const float* pred, const float* gt, float* grad, double* loss_sum, double* parts, long long count,
                      float eps, float grad_scale, void* stream);
/* PSNRLoss (losses.py:95-120, toY = False; image_event_restoration_model.py uses it for the HINet network):
 * *loss = weight * 10/ln10 * mean_b log(mse_b + 1e-8), mse_b over the per_sample elements of sample b;
 * grad (may be NULL) = d loss / d pred; sq = n_samples doubles (receives the per-sample squared errors);
 * parts: refid_psnr_loss_parts(n_samples, per_sample) doubles. */
int refid_psnr_loss_parts(int n_samples, long long per_sample);
int refid_psnr_loss(const float* pred, const float* gt, float* grad, double* sq, double* loss, double* parts, int n_samples,
                    long long per_sample, float weight, void* stream);
/* out[0] = sum g^2 (deterministic two-stage reduction: replicas of a data-parallel job must agree bit for
 * bit); `out` must hold REFID_SQNORM_WORDS doubles (out[1..] is scratch). */
#define REFID_SQNORM_WORDS 2049
int refid_grad_sqnorm(const float* g, double* out, long long count, void* stream);
/* clip_grad_norm_(max_norm) (max_norm <= 0: off) + AdamW step over flat arenas; gradients are
 * pre-multiplied by grad_scale (1/world_size after a SUM all-reduce). */
int refid_clip_adamw(float* p, const float* g, float* m, float* v, const double* sqnorm, float max_norm,
                     float grad_scale, float lr, float beta1, float beta2, float eps, float weight_decay,
                     int step, long long count, void* stream);

/* The same step with the per-iteration scalars in DEVICE memory: hyper[0] = lr, hyper[1] = 1 - beta1^t,
 * hyper[2] = sqrt(1 - beta2^t).  A train step captured in a hipGraph is replayed with different lr / t every
 * iteration; kernel arguments are frozen at capture time, device memory is not. */
int refid_clip_adamw_dev(float* p, const float* g, float* m, float* v, const double* sqnorm, float max_norm,
                         float grad_scale, const float* hyper, float beta1, float beta2, float eps,
                         float weight_decay, long long count, void* stream);

/* ------------------------------------------------------------------------------------
 * Layout / elementwise helpers on the boundary.
 * ---------------------------------------------------------------------------------- */
/* NCHW (n,c,h,w) -> NHWC (n,h,w,c_pad), channels >= c zero filled.  Replaces the
 * einops.rearrange calls at XXNet_final_attenfusion_arch.py:140-143 (layout change only). */
int refid_nchw_to_nhwc(const float* src, long long src_batch_stride, float* dst, int n, int c, int h, int w,
                       int c_pad, void* stream);
/* The same conversion of sum_t src[:, t]: src is (n, t_count, c, h, w) with the given batch / time strides (floats).
 * Gradient of a tensor every time step reads (`head` in `pred(z_t + head)`, XXNet_final_attenfusion_arch.py:215):
 * autograd sums the T per-step gradients. */
int refid_nchw_tsum_to_nhwc(const float* src, long long src_batch_stride, long long t_stride, int t_count, float* dst,
                            int n, int c, int h, int w, int c_pad, void* stream);
/* NHWC (n,h,w,ld) first c channels -> NCHW (n,c,h,w) with an output batch stride (so a
 * (B,T,3,H,W) stack is written in place: XXNet_final_attenfusion_arch.py:218). */
int refid_nhwc_to_nchw(const float* src, int ld, float* dst, long long dst_batch_stride,
                       int n, int c, int h, int w, void* stream);
/* The two conversions for a whole (B, T, C, H, W) stack in ONE launch, TIME-MAJOR on the NHWC side: NHWC sample t nb + b
 * <-> the stack's (b, t) block at b b_stride + t t_stride floats.  The event stack in (XXNet_final_attenfusion_arch.py:149,
 * 172-176: the recurrent loops read e[:, t]), the output stack out and its gradient back in (arch:218). */
int refid_nchw_to_nhwc_tb(const float* src, long long b_stride, long long t_stride, float* dst, int nb, int nt, int c, int h,
                          int w, int c_pad, void* stream);
int refid_nhwc_to_nchw_tb(const float* src, int ld, float* dst, long long b_stride, long long t_stride, int nb, int nt, int c,
                          int h, int w, void* stream);
/* Deferred per-channel parameter-gradient sums.  refid_layernorm2d_bwd, refid_dwconv3x3_bwd and refid_colsum each end with a
 * small launch that adds their per-workgroup partial rows into dw / db (autograd's accumulation of the T steps' gradients,
 * twoImage_event_recurrent_model.py:303).  refid_rows_sum_defer(1) makes these calls (of this host thread) QUEUE that sum
 * instead; refid_rows_sum_flush issues the queued sums grouped by destination, in call order within a destination (the bits of
 * the one-by-one launches), as one launch per ~160 sums, and turns deferral off.  Between the two the caller keeps every `parts`
 * buffer alive and the destinations untouched.  refid_rows_sum_defer also drops whatever an aborted pass left queued. */
int refid_rows_sum_defer(int on);
int refid_rows_sum_flush(void* stream);
/* out = a + b (skip sums: XXNet_final_attenfusion_arch.py:16-17,199-203,211,215;
 * recurrent_sub_modules.py:278). count = number of floats, multiple of 4. */
int refid_add(const float* a, const float* b, float* out, long long count, void* stream);
/* out = in[0] + ... + in[n-1] in index order (n <= REFID_SUM_MAX host pointers to device tensors of `count` floats; out may
 * alias in[0]).  The sums autograd accumulates over the T steps that share a tensor (x_blocks, head, the final backward
 * states: SURVEY.md A.2 "residual / skip adds") in one launch after BPTT. */
#define REFID_SUM_MAX 48
int refid_sum_n(const float* const* in, int n, float* out, long long count, void* stream);
/* out = (acc ? out : 0) + g * (y > 0 ? 1 : slope): activation derivative. */
int refid_act_bwd(const float* g, const float* y, float* out, float slope, int accumulate,
                  long long count, void* stream);

/* ------------------------------------------------------------------------------------
 * Callers either side of the path.
 * ---------------------------------------------------------------------------------- */
/* events_to_voxel_grid (basicsr/data/event_util.py:6-66): voxel (num_bins,height,width) is zeroed,
 * then every event adds pol*(1-dt) to bin floor(t') and pol*dt to the next, t' = (bins-1)(t-first)/dT;
 * polarity 0 counts as -1.  fp32 atomics: summation order differs from np.add.at (tolerance). */
int refid_events_to_voxel(const double* ts, const int* xs, const int* ys, const float* ps, long long n_events,
                          int num_bins, int width, int height, double first_stamp, double last_stamp,
                          float* voxel, void* stream);
/* tensor2img quantisation (utils/img_util.py:90-117: clamp [0,1], x255, round) fused with the squared
 * error of calculate_psnr (metrics/psnr_ssim.py:48-63): sq[f] = sum (q(a)-q(b))^2 per frame, float64;
 * parts: refid_sqerr_u8_parts(...) doubles (refid_ssim3d_u8: refid_ssim3d_u8_parts(...)). */
int refid_sqerr_u8_parts(int n_frames, long long frame_elems);
int refid_sqerr_u8(const float* a, const float* b, int n_frames, long long frame_elems, double* sq, double* parts,
                   void* stream);
/* calculate_ssim -> _ssim_3d (metrics/psnr_ssim.py:135-182,225-303): separable 11^3 Gaussian (sigma 1.5,
 * replicate padding) over (H, W, C=3) of the uint8-quantised frames, fp32; sum_out[f] = sum of the ssim
 * map of frame f (divide by 3*h*w for the mean).  a, b: (n_frames, 3, h, w) in [0,1]. */
int refid_ssim3d_u8_parts(int n_frames, int h, int w);
int refid_ssim3d_u8(const float* a, const float* b, int n_frames, int h, int w, double* sum_out, double* parts,
                    void* stream);
/* grids_inverse (twoImage_event_recurrent_model.py:252-268): acc[:, i0:i0+th, j0:j0+tw] += tile,
 * cnt += 1; then acc /= cnt. */
int refid_tile_add(const float* tile, float* acc, float* cnt, int c, int th, int tw, int h, int w, int i0, int j0,
                   void* stream);
int refid_tile_normalize(float* acc, const float* cnt, int c, int h, int w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* REFID_HIP_H */
