// Kernel-side argument block shared by the conv tiles (direct implicit GEMM and Winograd).
#pragma once

struct ConvKArgs {
    const float* inA; const float* inB;
    int ldA, ldB, Ca, Ctot;
    const float* w; const float* bias;
    float* out; int ldO;
    const float* res; int ldR;
    const float* mask; int ldM;
    int N, H, W, Ho, Wo;
    int Cout, CoutPad, coBase;
    int pad, nchunks, tilesX, tilesY;
    float slopePre, slopePost, slopeMask;
    long long wClsStride;
    int vecOK;                 // out/res/mask/bias allow 16-byte channel-quad accesses
    int ncot;                  // output-channel tiles (Winograd tile's XCD-aware work mapping)
    int bf16;                  // bf16 MFMA operands (packed weights are bf16), fp32 everything else
    int ksplit = 1;            // Winograd tile: K (input-channel chunk) ranges per output tile, grid.y (small grids)
    long long wsStride = 0;    // floats between the partial outputs of two K ranges
    int gridTiles = 0;         // persistent Winograd tile: number of virtual blocks (XCD-aware tile enumeration)
    // second output: out2 = out + add2 (the skip sums the reference builds right after a conv -- arch:16-17,199-203,211 --
    // and their backward counterparts leave with the producing tile instead of a separate add kernel); NULL = off
    const float* add2 = nullptr; float* out2 = nullptr; int ldA2 = 0, ldO2 = 0;
    int maskMode = 0;          // 0: out *= (mask > 0 ? 1 : slopeMask);  1 (pointwise tile): out *= GELU'(mask)
    // pointwise tile only: ConvTranspose2d(2,2) forward (refid_conv_desc.mode 1) -- GEMM column j = (q, co) of pixel (n, y, x)
    // is channel co of output pixel (n, 2y + q/2, 2x + q%2); Cout = 4 Co; the bias has Co entries
    int shuffle = 0;
    // pointwise tile only, patch form (a 2x2 stride-2 conv as one GEMM): source pixel p sits at
    // (p / patchW) * patchRow + (p % patchW) * ld floats instead of p * ld (both sources); 0 = dense
    int patchW = 0, patchRow = 0;
};


// conv_wino.hip
// ws / ws_bytes: caller-provided split-K workspace (refid_wino3x3_workspace_bytes; NULL = never split)
size_t refid_wino3x3_workspace_bytes(const ConvKArgs& a, int split_mode);
int refid_launch_wino3x3(const ConvKArgs& a, float* ws, size_t ws_bytes, int split_mode, int tile_hint, hipStream_t st);
int refid_launch_splitk_finish(const ConvKArgs& f, const float* ws, int ldW, long long npix, hipStream_t st);
// conv_wino6.hip: Winograd F(2x2,3x3) with six bf16 products per fp32 product (algo 5)
bool refid_wino6_eligible(const ConvKArgs& a);
size_t refid_wino6_workspace_bytes(const ConvKArgs& a, int split_mode);
// terms: 0 / 6 = six bf16 products (three planes per operand), 3 = three fp16 products (two planes, scaled operands)
int refid_launch_wino6(const ConvKArgs& a, float* ws, size_t ws_bytes, int split_mode, int tile_hint, int terms, hipStream_t st);
// conv_split.hip: direct 3x3 tile with split-bf16 operands (algo 4); terms = 6 (fp32-class products) or 3
bool refid_split3x3_eligible(const ConvKArgs& a);
// split_mode (refid_conv_desc.split_k): 1 = tile choice from the per-sample geometry (batch-independent bits)
int refid_launch_split3x3(const ConvKArgs& a, int terms, int mode, int cus, int split_mode, hipStream_t st);
// conv_pw.hip
// Fusions around a pointwise conv (refid_pw_extras in refid_hip.h): EGACA's LayerNorm2d prologue, the squeeze-excite
// vector computed in the kernel and applied to the operand, second residual, GELU second output.
struct PwExtra {
    const float* lnG = nullptr; const float* lnB = nullptr; float lnEps = 0.f; float* lnOut = nullptr; int ldLn = 0;
    const float* pool = nullptr; int poolParts = 0; float invHW = 0.f; int hw = 0; int seC = 0;
    const float* seW1 = nullptr; const float* seB1 = nullptr; const float* seW2 = nullptr; const float* seB2 = nullptr;
    float* seM = nullptr; float* seZ1 = nullptr; float* seS = nullptr;
    float* xsOut = nullptr; int ldXs = 0;
    const float* res2 = nullptr; int ldR2 = 0;
    float* out2 = nullptr; int ldO2 = 0;
};
// terms: 0 = fp32 MFMA products, 6 = six bf16 products on exactly split operands (weights from refid_pack_conv_weights_split, 1x1)
int refid_launch_pointwise(const ConvKArgs& a, hipStream_t st, const PwExtra* ex = nullptr, int terms = 0);
