/* C ABI of libvitpose_hip.so: the gfx950 (MI355X) kernels behind the ViTPose hot path.
 *
 * The reference (MiraPurkrabek/ViTPose) has no native code: every op on the path is a
 * stock PyTorch/ATen call.  Each entry point below therefore cites the reference
 * *Python* op it replaces (file:line relative to the reference tree).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; nothing is allocated,
 *     freed or synchronised here (safe under HIP graph capture);
 *   - `stream` is a hipStream_t passed as void*; NULL = the default stream;
 *   - return 0 on success, VP_ERR_ARG (-1) for a rejected argument, VP_ERR_UNSUPPORTED (-2)
 *     for a shape outside the compiled set, or a positive hipError_t from the launch;
 *   - "bf16" is the 16-bit brain-float bit pattern (torch.bfloat16), "f32" IEEE float;
 *   - token-major activations: row m = b * tokens + n, columns = channels.
 */
#ifndef VITPOSE_HIP_H
#define VITPOSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP__
typedef void* vp_stream_t;
#else
#include <hip/hip_runtime.h>
typedef hipStream_t vp_stream_t;
#endif

#define VP_ABI_VERSION 4
int vp_abi_version(void);
/* gfx target the library was compiled for, e.g. "gfx950". */
const char* vp_build_arch(void);

/* ------------------------------------------------------------------ GEMM --------
 * C[i][j] = sum_k X[i][k] * Y[j][k], bf16 operands, f32 accumulation on MFMA.
 * Replaces nn.Linear forward/backward on the path: Attention.qkv / .proj
 * (mmpose/models/backbones/vit.py:93,96,101,112), Mlp.fc1 / .fc2 (vit.py:66-74), and, via
 * im2col / sub-pixel views, PatchEmbed.proj (vit.py:157,161) and the head's
 * ConvTranspose2d / Conv2d (mmpose/models/heads/topdown_heatmap_simple_head.py:132-139,323-332).
 *   x_kmajor / y_kmajor = 0: operand stored P[r*ld + k];  1: stored P[k*ld + r].
 */
enum {
    VP_EPI_BF16 = 0,       /* out(bf16) = acc*colscale + bias                                   */
    VP_EPI_GELU = 1,       /* out2(bf16, optional) = u = acc + bias; out(bf16) = gelu_erf(u)    */
    VP_EPI_RESID_F32 = 2,  /* out(f32) = addend + rowscale[i/rows_per_scale]*(acc+bias); addend = resid row i,
                              else pos row i%pos_rows (resid wins when both are given), else 0 */
    VP_EPI_DGELU = 3,      /* out(bf16) = acc * gelu_erf'(aux(bf16))                            */
    VP_EPI_SPLITK_F32 = 4, /* K slice z of split_k writes its partial sums to out + z*split_stride (f32) */
    VP_EPI_F32 = 5,        /* out(f32) = acc + bias                                             */
    VP_EPI_SCALE_RELU = 6, /* out(bf16) = relu(acc*colscale + bias)                             */
    VP_EPI_DRELU = 7,      /* out(bf16) = acc * (aux(bf16) > 0)                                 */
    VP_EPI_PAIR = 8        /* split3 only: out = hi / lo pair array of act(acc*colscale + bias), act = pair_act */
};

typedef struct vp_gemm_args {
    int M, N, K;
    const void* x; int ldx; int x_kmajor;
    const void* y; int ldy; int y_kmajor;
    int epilogue;
    void* out; int ldo;          /* ldo also strides resid / pos / out2 */
    void* out2;
    const float* bias;           /* [N] or NULL */
    const float* colscale;       /* [N] or NULL */
    const float* resid;          /* [M][ldo] f32 or NULL */
    const float* rowscale;       /* [ceil(M/rows_per_scale)] or NULL (DropPath per-sample scale) */
    int rows_per_scale;
    const void* aux; int ldaux;  /* [M][ldaux] bf16 for DGELU / DRELU */
    const float* pos;            /* [pos_rows][ldo] f32 or NULL */
    int pos_rows;
    int split_k;                 /* SPLITK_F32 only: number of K slices (grid.y), <= K/64 */
    long split_stride;           /* SPLITK_F32 only: elements between the slices' slabs (>= M*ldo) */
    /* optional output row remap (sub-pixel phases of a transposed conv): when out_row_div > 0,
     * row i of C is stored at row (i / div) * hi + (i % div) * lo + off of out / out2. */
    int out_row_div, out_row_hi, out_row_lo, out_row_off;
    /* optional scratch (caller-owned, >= 32 MiB covers every shape): when given, the last partial
     * round of tiles is cut along K so that it fills the chip, and a fix-up kernel applies the
     * epilogue.  NULL = plain launch.  Launches sharing one scratch must be stream-ordered. */
    float* tail_ws; size_t tail_ws_bytes;
    int tail_force;              /* 1: always split the tail round when tail_ws is given (default: heuristic) */
    int tile_mode;               /* 0: choose by shape; 1: 128x128 tiles (2 workgroups / CU); 2: 256x256 tiles (1 / CU);
                                    3: 192x192 tiles (1 / CU, three LDS stages); 4: 192x128 tiles, 2 workgroups / CU
                                    (k-contiguous x, no K split: VP_ERR_UNSUPPORTED otherwise); 5: 192x192 tiles, one
                                    persistent workgroup per CU that writes output tile i while it accumulates tile i + 1
                                    (k-contiguous x, bf16 outputs VP_EPI_BF16 / GELU / DGELU, K >= 640, M % 192 == 0,
                                    no K split / row remap / colscale: VP_ERR_UNSUPPORTED otherwise); 6: 192x128 tiles of
                                    the three-stage kernel of mode 3 (1 / CU; k-contiguous x, no K split: VP_ERR_UNSUPPORTED
                                    otherwise): N = 1024 at 6144 rows is exactly 256 such tiles; 7: 192x192 tiles, the two
                                    wave groups split every K tile (96x96 per wave; the halves are summed through LDS once per
                                    tile): long reductions; epilogues VP_EPI_BF16 / RESID_F32 / F32, K >= 128, no K split:
                                    VP_ERR_UNSUPPORTED otherwise */
    /* optional per-image routing of the Y operand: the mixture-of-experts FFN of ViTPose+ (MoEMlp.forward,
     * mmpose/models/backbones/vit_moe.py:98-115: the last `part_features` output channels of fc2 come from the
     * expert Linear selected by the image's dataset index).  Rows are images of exactly 192 tokens (M % 192 == 0);
     * image b uses expert route_expert[b].  Y rows (y_kmajor = 0: output columns, the forward product;
     * y_kmajor = 1: reduction rows, the data gradient) at or beyond route_split are read route_delta[expert]
     * ELEMENTS further on than the shared matrix would hold them (a multiple of 8), and bias columns >= route_split
     * route_bias_delta[expert] floats further on.  All three arrays live on the device; the expert indices are read
     * by the kernel and cannot be validated by the launcher: every route_expert[b] must index route_delta (the Python
     * host checks them against num_expert before each batch).  Uses the 192 x 192 tile
     * schedule; route_split % 48 == 0, k-contiguous x, no K split. */
    const int* route_expert; const long* route_delta; const long* route_bias_delta; int route_split;
    /* optional (gather16_C > 0): the Y operand is the reduction-major 16-tap gather of `y`, never materialised --
     * the weight gradient of ConvTranspose2d(k=4, s=2, p=1) (autograd of topdown_heatmap_simple_head.py:323-332):
     * y = dout [B*2H*2W][C] bf16 (H, W = gather16_H / _W = the layer's INPUT grid, C = gather16_C its output
     * channels), Y[(b,iy,ix)][(ky*4+kx)*C + c] = dout[b][2iy-1+ky][2ix-1+kx][c] (0 outside; gather16_zero: >= 128
     * bytes of zeros on the device), K = B*H*W, N = 16*C, x_kmajor = y_kmajor = 1, VP_EPI_SPLITK_F32, C % 128 == 0. */
    int gather16_H, gather16_W, gather16_C; const void* gather16_zero;
    /* split3 = 1: fp32-accurate product on the bf16 matrix cores (the reference computes in fp32: nn.Linear /
     * conv of vit.py and topdown_heatmap_simple_head.py on CPU tensors).  x and y are hi / lo PAIR arrays: a row
     * of K values is stored as K / 32 groups of 64 bf16, hi(v[32 j .. 32 j + 31]) then lo(the same values), with
     * hi = bf16(v), lo = bf16(v - hi); ldx / ldy count pair elements (>= 2 K).  Every K step accumulates
     * hi*hi + hi*lo + lo*hi in ONE accumulator set.  K % 32 == 0, k-contiguous operands, epilogue VP_EPI_F32,
     * VP_EPI_RESID_F32 or VP_EPI_PAIR (out = pair array [M][ldo >= 2 N], N % 32 == 0, of act(acc*colscale + bias);
     * pair_act 0 none, 1 exact-erf GELU, 2 ReLU).  route_* are honoured with VP_EPI_RESID_F32 (route_delta in pair
     * elements). */
    int split3; int pair_act;
    /* optional LayerNorm fold (evaluation only: the weights are fixed).  Block.forward (vit.py:137-140) feeds
     * norm1(x) to attn.qkv and norm2(x) to mlp.fc1; with W' = W diag(gamma),
     *   LN(x) W^T + b = rstd (x W'^T - mean s) + c,   s[n] = sum_k W'[n][k],   c = W beta + b,
     * so the Linear can read the raw residual stream and the LayerNorm launch (and its f32 re-read of the stream)
     * disappears.  PRODUCER (ln_stats_out != NULL, VP_EPI_RESID_F32, tile_mode 2 / 3 / 6): besides `out` it writes
     * the bf16 copy of the stream to out2 [M][ldo] and, per row and column tile, the sum and the sum of squares of
     * the tile's columns to ln_stats_out [M][ln_slots][2] (ln_slots = vp_gemm_ln_slots(N, tile_mode)).  CONSUMER
     * (ln_stats != NULL, VP_EPI_BF16 / VP_EPI_GELU, any tile_mode but 5): x = that bf16 copy, y = W' (bf16),
     * colscale = s, bias = c, K = the normalised width; mean / rstd of a row come from its ln_slots pairs (summed in
     * slot order: deterministic, independent of the batch), variance = E[x^2] - mean^2 in f32, eps = ln_eps. */
    float* ln_stats_out; const float* ln_stats; int ln_slots; float ln_eps;
} vp_gemm_args;

int vp_gemm_bf16(const vp_gemm_args* args, vp_stream_t stream);
/* column tiles (= statistics pairs per row) of a folded-LayerNorm producer on tile_mode 2 / 3 / 6; 0 otherwise */
int vp_gemm_ln_slots(int N, int tile_mode);
/* The folded weights of one Linear behind a LayerNorm (see vp_gemm_args.ln_stats): W [N][K] f32, gamma / beta [K],
 * bias [N] or NULL -> Wg [N][K] bf16 = bf16(W[n][k] gamma[k]), s [N] = sum_k Wg[n][k] (of the ROUNDED weights: the
 * mean term then cancels exactly what the matrix cores accumulate), c [N] = sum_k W[n][k] beta[k] + bias[n]. */
int vp_ln_fold_weights(const float* W, const float* gamma, const float* beta, const float* bias, void* Wg, float* s,
                       float* c, int N, int K, vp_stream_t stream);
/* out[i] (+)= sum_z slab[z*stride + i], i < n: the deterministic reduction of the split-K slabs
 * (weight gradients; replaces autograd's dW accumulation of nn.Linear / conv layers). */
int vp_splitk_reduce(const float* slab, int nsplit, long stride, size_t n, float* out, int accumulate,
                     vp_stream_t stream);

/* dtype tags for the layout/cast helpers */
enum { VP_F32 = 0, VP_BF16 = 1 };

/* ------------------------------------------------------------- LayerNorm --------
 * nn.LayerNorm(E, eps=1e-6) over the channel axis of token-major x [M][E] f32
 * (Block.norm1 / norm2, vit.py:125,133,138-139; ViT.last_norm, vit.py:242,328).
 * fwd: y (bf16 and/or f32) = (x - mean) * rstd * gamma + beta; mean/rstd [M] saved if non-NULL.
 * bwd: dx = dres + dLN(dy); dxb (bf16) = rowscale[m / rows_per_scale] * dx;
 *      dgamma/dbeta/dcolsum [E] are ACCUMULATED (f32 atomics); dcolsum = column sum of dxb
 *      (the bias gradient of the Linear whose output feeds this residual stream). */
int vp_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* y_f32,
                     float* mean, float* rstd, int M, int E, float eps, vp_stream_t stream);
int vp_layernorm_bwd(const void* dy_bf16, const float* x, const float* mean, const float* rstd,
                     const float* gamma, const float* dres, float* dx, void* dxb_bf16,
                     const float* rowscale, int rows_per_scale, float* dgamma, float* dbeta,
                     float* dcolsum, int M, int E, float* workspace, size_t workspace_floats,
                     vp_stream_t stream);
/* workspace: caller-owned scratch for the per-workgroup partial column sums, at least 3*E floats
 * (3*E*512 floats lets the kernel use its full grid).
 * Deferred column sums: with dgamma == dbeta == dcolsum == NULL the call leaves its partial sums in `workspace`
 * ([blocks][3][E], blocks = vp_layernorm_bwd_blocks(M, E, workspace_floats)) and vp_layernorm_bwd_finalize adds
 * the sums of several such workspaces to their targets in ONE launch (a training step has 2 * depth + 1 LayerNorm
 * backward passes whose parameter gradients nobody reads before the optimizer or the gradient exchange).
 * `table_dev` lives in device memory; NULL targets of an entry are skipped. */
typedef struct vp_ln_partial {
    const float* partial; float* dgamma; float* dbeta; float* dcolsum; int blocks;
    int overwrite;   /* 0: the sums are added to the targets; 1: they replace what the targets hold */
} vp_ln_partial;   /* 40 bytes */
int vp_layernorm_bwd_blocks(int M, int E, size_t workspace_floats);
int vp_layernorm_bwd_finalize(const vp_ln_partial* table_dev, int count, int E, vp_stream_t stream);
/* vp_layernorm_bwd without a workspace argument.  No counterpart in the reference: it is for callers that cannot
 * allocate device memory (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).
 * The per-workgroup partial column sums go to a file-scope __device__ array of the library
 * (VP_LN_STATIC_MAX_BLOCKS * 3 * VP_LN_STATIC_MAX_E floats = 7.5 MiB, held by every GPU the library runs on until
 * the process ends); the call allocates and frees nothing.  Results are bit-identical to vp_layernorm_bwd called
 * with a workspace of that many floats.  E > VP_LN_STATIC_MAX_E returns VP_ERR_UNSUPPORTED; at least one of
 * dgamma / dbeta / dcolsum must be given (the sums cannot be deferred: every call on a GPU shares that GPU's
 * array, so calls that overlap on one GPU -- two streams, two threads -- race; stream-ordered calls are safe).
 * The first call on each device looks the array's device address up (hipGetSymbolAddress): make it outside
 * graph capture. */
#define VP_LN_STATIC_MAX_E 1280
#define VP_LN_STATIC_MAX_BLOCKS 512
int vp_layernorm_bwd_static(const void* dy_bf16, const float* x, const float* mean, const float* rstd,
                            const float* gamma, const float* dres, float* dx, void* dxb_bf16,
                            const float* rowscale, int rows_per_scale, float* dgamma, float* dbeta,
                            float* dcolsum, int M, int E, vp_stream_t stream);

/* ------------------------------------------------------------- Attention --------
 * Attention.forward core, vit.py:102-111: softmax(q k^T * scale) v over all `tokens`
 * keys of one image, per head.  qkv = packed [B*tokens][3*H*head_dim] bf16 exactly as
 * the qkv Linear emits it (q | k | v, head-major inside each); out = [B*tokens][H*head_dim]
 * bf16 (the layout `.transpose(1, 2).reshape(B, N, -1)` produces); lse [B*H][tokens] f32 =
 * log-sum-exp of the scaled scores (saved for backward).  tokens must be 192, head_dim 32 or 64.
 * bwd writes dqkv in the packed layout (every element written once). */
int vp_attention_fwd(const void* qkv, void* out, float* lse, int B, int H, int head_dim, int tokens,
                     float scale, vp_stream_t stream);
int vp_attention_bwd(const void* qkv, const void* out_fwd, const void* dout, const float* lse, void* dqkv,
                     int B, int H, int head_dim, int tokens, float scale, vp_stream_t stream);

/* ---------------------------------------------------- layout / data movement ---- */
/* PatchEmbed.proj as a GEMM (vit.py:157,161): rows of the NCHW f32 image gathered into
 * col [B*Hp*Wp][C*patch*patch] bf16, k = (c, ky, kx) = the flattened conv weight order. */
int vp_patch_im2col(const float* img, void* col_bf16, int B, int C, int H, int W, int patch, int stride,
                    int pad, vp_stream_t stream);
/* out[i0][i1][i2][i3] (contiguous) = (cast) in[i0*s0 + i1*s1 + i2*s2 + i3*s3]: weight re-layouts,
 * NCHW <-> token-major permutes (vit.py:164,330). */
int vp_strided_cast(const void* in, int in_dtype, void* out, int out_dtype, const int dims[4],
                    const long strides[4], vp_stream_t stream);
/* out[N] += column sums of a bf16 [M][N] matrix (bias gradients). */
int vp_colsum_bf16(const void* a, float* out, int M, int N, int lda, vp_stream_t stream);
/* out[n] = sum_b x[b][n] (pos_embed gradient, vit.py:320). */
int vp_batch_sum_f32(const float* x, float* out, int B, size_t n, vp_stream_t stream);
/* ConvTranspose2d(k=4, s=2, p=1) (topdown_heatmap_simple_head.py:323-332) as 4 sub-pixel GEMMs:
 * col[(b,y,x)][t*C + c] = in[b][y+dy][x+dx][c] for output parity (py, px), channels-last bf16. */
int vp_deconv_phase_im2col(const void* in, void* col, int B, int H, int W, int C, int py, int px,
                           vp_stream_t stream);
/* its backward: g[(b,iy,ix)][(ky*4+kx)*C + c] = dout[b][2iy-1+ky][2ix-1+kx][c] (H, W = input grid). */
int vp_deconv_out_gather16(const void* dout, void* g, int B, int H, int W, int C, vp_stream_t stream);
int vp_cast_f32_bf16(const float* x, void* y, size_t n, vp_stream_t stream);

/* ------------------------------------------------------------- BatchNorm --------
 * nn.BatchNorm2d(256) + ReLU of the deconv stack (topdown_heatmap_simple_head.py:333-334) on
 * channels-last bf16 z [M][C].  stats: sums[0:C] += sum z, sums[C:2C] += sum z^2.
 * finalize: training (sums != NULL) -> batch statistics, running stats updated (momentum,
 * unbiased var); eval (sums == NULL) -> running statistics.  scale/shift fold BN into y = z*scale+shift. */
int vp_bn_stats(const void* z, float* sums, int M, int C, vp_stream_t stream);
int vp_bn_finalize(const float* sums, int M, int C, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float momentum, float eps, float* scale,
                   float* shift, float* save_mean, float* save_invstd, vp_stream_t stream);
int vp_bn_apply_relu(const void* z, void* a, const float* scale, const float* shift, int M, int C,
                     vp_stream_t stream);
/* g = gradient wrt the BN output with the ReLU mask applied; dgamma/dbeta ACCUMULATE. */
int vp_bn_bwd(const void* g, const void* z, const float* mean, const float* invstd, const float* gamma,
              float* dgamma, float* dbeta, void* dz, int M, int C, vp_stream_t stream);
/* The same with statistics that are constants of the backward pass (nn.BatchNorm2d in eval mode normalises with its
 * running mean / variance, torch/nn/modules/batchnorm.py as used by topdown_heatmap_simple_head.py:333): mean / invstd =
 * the running statistics (vp_bn_finalize with sums == NULL saves them), dz = g * gamma * invstd. */
int vp_bn_bwd_eval(const void* g, const void* z, const float* mean, const float* invstd, const float* gamma,
              float* dgamma, float* dbeta, void* dz, int M, int C, vp_stream_t stream);

/* ------------------------------------------------------- simple decoder --------
 * TopdownHeatmapSimpleHead with upsample > 0 and/or final_conv_kernel = 3
 * (topdown_heatmap_simple_head.py:278-287 `_transform_inputs`: ReLU + bilinear resize; :132-139 final
 * Conv2d).  The channel contraction runs first on the low-resolution grid as a GEMM,
 * G[(b,j)][t*kpad + k] = sum_c W[k][c][tap t] * relu(feat)[b][j][c]; these kernels do the rest:
 *   fwd: out[b][k][y][x] (NCHW f32) = bias[k] + sum_t Up(G_t)[y + ty - p][x + tx - p][k], p = (ksize-1)/2,
 *        zero outside the up-sampled grid; yi/yw [Ho][2], xi/xw [Wo][2] = the two source rows / columns
 *        and weights of every output coordinate (F.interpolate bilinear; built by the host).
 *   bwd: dG (bf16 [B*Hg*Wg][ldg], columns >= ksize^2*kpad zeroed) = the transpose of fwd applied to dout;
 *        tyi/tyw [Hg][Ly], txi/txw [Wg][Lx] list, per source row / column, the output coordinates that
 *        interpolate from it and their weights (index -1 = empty slot).
 * kpad (the per-tap column count of G, >= K) must be 20, 32 or 64. */
int vp_upconv_fwd(const float* G, int ldg, const float* bias, float* out, const int* yi, const float* yw,
                  const int* xi, const float* xw, int B, int K, int kpad, int Hg, int Wg, int Ho, int Wo,
                  int ksize, vp_stream_t stream);
int vp_upconv_bwd(const float* dout, void* dG_bf16, int ldg, const int* tyi, const float* tyw, const int* txi,
                  const float* txw, int Ly, int Lx, int B, int K, int kpad, int Hg, int Wg, int Ho, int Wo,
                  int ksize, vp_stream_t stream);
/* out[k] += sum_b sum_p x[b][k][p] (bias gradient of a conv on NCHW f32 maps). */
int vp_plane_sum_f32(const float* x, float* out, int B, int K, int HW, vp_stream_t stream);
/* y = max(x, 0) on n bf16 elements (F.relu of `_transform_inputs`, topdown_heatmap_simple_head.py:282). */
int vp_relu_bf16(const void* x, void* y, size_t n, vp_stream_t stream);
/* NCHW f32 [B][C][HW] -> channels-last bf16 [B*HW][ld] (columns >= C zeroed): the entry bridge of a
 * head fed by a foreign NCHW backbone, and of NCHW heat-map gradients (vit.py:330 is the inverse). */
int vp_nchw_to_cl_bf16(const float* src, void* dst_bf16, int B, int C, int HW, int ld, vp_stream_t stream);

/* ConvTranspose2d(k=4, s=2, p=1, bias=False) forward (topdown_heatmap_simple_head.py:306-337, the deconv layers)
 * on channels-last bf16 in ONE launch: the four output parities are four implicit GEMMs whose X operand (the
 * sub-pixel im2col of `in`) is gathered on the fly, never materialised.  in [B*H*W][C]; wph [4][Cout][4*C] = the
 * phase matrices wph[py*2+px][co][(ty*2+tx)*C + ci] = W[ci][co][(1-py)+2ty][(1-px)+2tx]; out [B*2H*2W][Cout] bf16.
 * epilogue VP_EPI_BF16 (training: raw conv output for the batch statistics) or VP_EPI_SCALE_RELU
 * (eval: out = relu(acc * colscale + shift), BatchNorm folded).  zero_page: >= 128 bytes of zeros on the device
 * (source of the taps that fall outside the image).  C % 64 == 0, Cout % 8 == 0. */
int vp_deconv_fwd_bf16(const void* in_bf16, const void* wph_bf16, void* out_bf16, int B, int H, int W, int C, int Cout,
                       int epilogue, const float* colscale, const float* shift, const void* zero_page,
                       vp_stream_t stream);

/* The data gradient of the same layer in ONE launch, the 16-tap neighbourhood of the output-grid gradient gathered
 * on the fly (autograd of nn.ConvTranspose2d, topdown_heatmap_simple_head.py:323-332):
 *   din[(b,iy,ix)][ci] = sum_{ky,kx,co} dout[b][2iy-1+ky][2ix-1+kx][co] * wd[ci][(ky*4+kx)*C + co],
 * dout [B*2H*2W][C] bf16 (H, W = the layer's INPUT grid, C = its output channels), wd [Cin][16*C] bf16 (the layout
 * vp_deconv_weight_prep writes), din [B*H*W][Cin] bf16.  relu_aux (optional, [B*H*W][Cin] bf16): the gradient is
 * masked where relu_aux <= 0 (the ReLU in front of the layer).  C % 64 == 0, Cin % 8 == 0. */
int vp_deconv_dgrad_bf16(const void* dout_bf16, const void* wd_bf16, void* din_bf16, int B, int H, int W, int C, int Cin,
                         const void* relu_aux_bf16, const void* zero_page, vp_stream_t stream);

/* Both compute layouts of a ConvTranspose2d(k=4) weight W [Ci][Co][4][4] (f32 master) in one pass: the phase
 * matrices wph [4][Co][4*Ci] of vp_deconv_fwd_bf16 and the 16-tap matrix wd [Ci][16*Co] the data gradient reads
 * (wd[ci][(ky*4+kx)*Co + co] = W[ci][co][ky][kx]); both bf16.  Replaces five vp_strided_cast launches per layer. */
int vp_deconv_weight_prep(const float* w, void* wph_bf16, void* wd_bf16, int Ci, int Co, vp_stream_t stream);
/* nn.Conv2d(C, *, 3, stride 1, padding 1) of the head's optional conv layers and 3 x 3 final layer after a deconv stack
 * (topdown_heatmap_simple_head.py:112-138) as a GEMM on channels-last maps: col [B*H*W][9*C] bf16,
 * col[(b,y,x)][(3 ky + kx) * C + c] = in[b][y+ky-1][x+kx-1][c] (zero outside), to be multiplied with the weight re-laid
 * as [Cout][(3 ky + kx) * C + c]; and its transpose (the data gradient): din[(b,y,x)][c] = sum over the taps of dcol,
 * masked where relu_aux <= 0 when relu_aux is given (the ReLU in front of the layer).  C % 8 == 0. */
int vp_im2col3x3_bf16(const void* in_bf16, void* col_bf16, int B, int H, int W, int C, vp_stream_t stream);
int vp_col2im3x3_bf16(const void* dcol_bf16, void* din_bf16, const void* relu_aux_bf16, int B, int H, int W, int C,
                      vp_stream_t stream);
/* input_transform='resize_concat' (topdown_heatmap_simple_head.py:289-298: every selected feature map bilinearly resized
 * to the first one's size, then concatenated along the channels): one NCHW f32 map [B][C][Hi][Wi] resized to Ho x Wo into
 * the columns [coff, coff + C) of the channels-last bf16 matrix dst [B*Ho*Wo][ld].  yi/yw [Ho][2], xi/xw [Wo][2] are
 * the source index pairs and weights of F.interpolate(mode='bilinear', align_corners) per output coordinate.  The
 * backward entry gathers the gradient g [B*Ho*Wo][ld] back: tyi/tyw [Hi][Ly], txi/txw [Wi][Lx] list, per source
 * coordinate, the output coordinates that read it and their weights (index -1 = unused).  C, coff, ld multiples of 8. */
int vp_resize_to_cl_bf16(const float* src, void* dst_bf16, const int* yi, const float* yw, const int* xi, const float* xw,
                         int B, int C, int Hi, int Wi, int Ho, int Wo, int ld, int coff, vp_stream_t stream);
int vp_resize_to_cl_bwd(const void* g_bf16, float* dx, const int* tyi, const float* tyw, const int* txi, const float* txw,
                        int Ly, int Lx, int B, int C, int Hi, int Wi, int Ho, int Wo, int ld, int coff, vp_stream_t stream);
/* The other two kernel sizes the head accepts (topdown_heatmap_base_head.py:105-120: kernel 3 -> padding 1,
 * output_padding 1; kernel 2 -> padding 0): w [n][k][k] f32 (n = Ci * Co) embedded in w4 [n][4][4] f32, the 4 x 4
 * kernel with padding 1 that computes the same transposed convolution (k = 3: taps 0..2, k = 2: taps 1..2 of each
 * axis; zeros elsewhere).  Every 4 x 4 entry point then serves the layer. */
int vp_deconv_weight_pad4(const float* w, float* w4, size_t n, int ksize, vp_stream_t stream);

/* Grouped weight gradients: several independent dW_i [M_i][N_i] = x_i^T y_i (x_i [K][M_i], y_i [K][N_i], bf16,
 * reduction-major, f32 out, no K split) with one common reduction extent K (= tokens) in ONE launch, so that the
 * tiles of all problems together fill the chip in whole rounds (the Linear weight gradients of several Blocks,
 * vit.py:93-95,60-62 under autograd; none of them alone has enough tiles).
 *   vp_wgrad_group_table: host-side; fills tiles_n / ntiles / tile0 of `problems` (host memory, <= 64 entries)
 *     for the tile size (tile_mode 2: 256, 3 and 7: 192) and returns the grid size in *total_tiles.  The caller copies
 *     the filled array to device memory.
 *   vp_wgrad_grouped_bf16: the launch; `table_dev` is that device copy. */
typedef struct vp_group_entry {
    const void* x; const void* y; float* out;
    int ldx, ldy, ldo, M, N;
    int tiles_n, ntiles, tile0;   /* filled by vp_wgrad_group_table */
    float* rowsum;                /* optional [M]: rowsum[m] = sum_k x[k][m], the bias gradient of the same Linear */
    /* vp_wgrad_grouped_kmap_bf16 only (else NULL): the K rows are images of tiles_per_image * 64 rows each, and
     * this problem reduces over the images whose bit is set in image_mask (ceil(images / 64) 64-bit words in DEVICE
     * memory, so a recorded launch follows per-step routing; no bit set = zeros are written). */
    const unsigned long long* image_mask; int tiles_per_image;
} vp_group_entry;                 /* 80 bytes */
int vp_wgrad_group_table(vp_group_entry* problems_host, int count, int tile_mode, int* total_tiles);
int vp_wgrad_grouped_bf16(const vp_group_entry* table_dev, int count, int total_tiles, int K, int tile_mode,
                          vp_stream_t stream);
/* The same launch (192 x 192 tiles) for tables in which some problems carry an image mask: the expert weight
 * gradients of the mixture-of-experts FFN (vit_moe.py:92-96,108-111 under autograd: expert e only receives
 * gradient from the images routed to it).  K = rows of the operands (upper bound of every list). */
int vp_wgrad_grouped_kmap_bf16(const vp_group_entry* table_dev, int count, int total_tiles, int K,
                               vp_stream_t stream);

/* --------------------------------------------------------------- Input side ------
 * vp_warp_affine_normalize: TopDownAffine's cv2.warpAffine(img, trans, image_size, INTER_LINEAR)
 *   (datasets/pipelines/top_down_transform.py:312-363) + ToTensor + NormalizeTensor (shared_transform.py:21-64)
 *   for a batch of 8-bit RGB images of different sizes: `imgs` = the images packed back to back (HWC bytes),
 *   offsets [B] (bytes), hw [B][2] = (height, width), inv_matrices [B][6] f64 = the INVERTED 2x3 transforms
 *   (destination -> source, what cv2 computes from `trans`), out [B][3][Ho][Wo] f32.  Restates OpenCV's
 *   fixed-point bilinear scheme (1/32 px coordinates, 15-bit weights, constant border 0).
 * vp_generate_target: TopDownGenerateTarget (top_down_transform.py:366-740) for encoding 0 = 'UDP' with
 *   'GaussianHeatmap' targets (_udp_generate_target, :541-661), 1 = 'MSRA', 2 = 'MSRA' with unbiased_encoding
 *   (_msra_generate_target, :418-492).  joints [B][K][3] (x, y in input-image pixels after the affine),
 *   visible [B][K] (joints_3d_visible[:, 0]), joint_weights [K] or NULL (use_different_joint_weights),
 *   target [B][K][H][W], target_weight [B][K]. */
int vp_warp_affine_normalize(const unsigned char* imgs, const long* offsets, const int* hw, const double* inv_matrices,
                             float* out, int B, int Ho, int Wo, const float mean[3], const float std[3],
                             vp_stream_t stream);
int vp_generate_target(const float* joints, const float* visible, const float* joint_weights, float* target,
                       float* target_weight, int B, int K, int H, int W, int img_w, int img_h, float sigma,
                       int encoding, vp_stream_t stream);

/* ------------------------------------------------------------- Test-time decode ----
 * vp_flip_back_avg: out = (heat + flip_back(heat_flipped)) * 0.5 — flip_back (core/post_processing/
 *   post_transforms.py:54-95: swap the left/right joint channels, mirror the width axis; `perm[k]` = the
 *   channel that lands in k) with the optional one-pixel shift (topdown_heatmap_simple_head.py:259-261) and
 *   the flip-test average of TopDown.forward_test (top_down.py:181-189).
 * vp_decode_heatmaps: keypoints_from_heatmaps (core/evaluation/top_down_eval.py:474-622) up to the map back
 *   to image coordinates: per map the first arg-max (_get_max_preds, :63-95; (-1, -1) when the maximum is
 *   <= 0), then mode 0: nothing; 1: the +-0.25 shift of post_process='default' (:598-610); 2: DARK-UDP
 *   (post_dark_udp, :335-396): cv2.GaussianBlur(ksize = ntaps, sigma 0; separable f32, BORDER_REFLECT_101) at
 *   the 7 stencil points only, clip to [0.001, 50], log, Newton step with the f64 2 x 2 inverse.  `taps_host`
 *   (HOST pointer, copied into the kernel arguments) = the 1-D blur kernel.  preds [maps][2] (x, y) in
 *   heat-map pixels, maxvals [maps].  An invalid joint reads the stencil clamped at the map corner (the
 *   reference's flat index wraps into a neighbouring map there). */
#define VP_DECODE_MAX_KERNEL 31
int vp_flip_back_avg(const float* heat, const float* heat_flipped, const int* perm, int shift, float* out,
                     int N, int K, int H, int W, vp_stream_t stream);
int vp_decode_heatmaps(const float* heat, float* preds, float* maxvals, int maps, int H, int W, int mode,
                       const float* taps_host, int ntaps, vp_stream_t stream);
/* The remaining branches of keypoints_from_heatmaps, which blur whole maps (one workgroup per joint, maps in LDS):
 *   mode 3: post_process='unbiased' (top_down_eval.py:591-597): arg-max of the map, then _gaussian_blur (:399-438:
 *     cv2.GaussianBlur inside a zero border of (ntaps - 1) / 2, rescaled so that the maximum is restored),
 *     log(max(., 1e-10)) and the second-order Taylor step of _taylor (:298-332) at the arg-max;
 *   mode 4: post_process='megvii' (:563-564, :598-612, :619-620): _gaussian_blur first, arg-max of the modulated map,
 *     +-0.25 shift and +0.5, maxvals = max / 255 + 0.5;
 *   mode 5: UDP with target_type='CombinedTarget' (:571-585): heat = [joints][3][H][W] (heat map, x offset,
 *     y offset per joint); the heat map is blurred with taps2 (2 * kernel + 1 taps), the offset maps with taps
 *     (kernel taps), BORDER_REFLECT_101; preds = arg-max of the blurred heat map + valid_radius * blurred offsets
 *     there (valid_radius = valid_radius_factor * H).  An invalid joint (maximum <= 0) reads the offsets at the
 *     map corner (the reference's flat index goes negative there and wraps into another map).
 * taps_host / taps2_host: HOST pointers to the 1-D blur kernels (copied into the kernel arguments).
 * preds [joints][2] (x, y) in heat-map pixels, maxvals [joints]. */
int vp_decode_blurred(const float* heat, float* preds, float* maxvals, int joints, int H, int W, int mode,
                      const float* taps_host, int ntaps, const float* taps2_host, int ntaps2, float valid_radius,
                      vp_stream_t stream);

/* ----------------------------------------------------- fp32-accurate path ----------
 * Evaluation at the reference's fp32 accuracy (the 1e-3 max-abs heat-map tolerance on arbitrary weights; the
 * reference computes every op in fp32).  Every GEMM operand is a hi / lo PAIR array (see vp_gemm_args.split3):
 * rows of K values stored as K / 32 groups of 64 bf16, hi(v[32 j .. 32 j + 31]) then lo(the same values),
 * hi = bf16(v), lo = bf16(v - hi); a Linear / conv product is ONE vp_gemm_bf16 launch with split3 = 1
 * (hi*hi + hi*lo + lo*hi per K step into one f32 accumulator) whose epilogue is f32 (bias, exact GELU, BatchNorm
 * scale/shift + ReLU, DropPath-free residual) and writes either f32 or the next product's pair operand.
 *   vp_split_pair_f32: pair array (2 n bf16) of a flat f32 buffer of n values, n % 32 == 0 (weights: tensors that
 *     start on multiples of 32 elements with rows of a multiple of 32 values keep their row structure);
 *   vp_layernorm_fwd_pair: vp_layernorm_fwd whose output is the pair array [M][2 E] of LN(x) (relu = 1: of
 *     max(LN(x), 0), the F.relu of the simple decoder, topdown_heatmap_simple_head.py:282), optionally also f32;
 *   vp_attention_split: Attention.forward core (vit.py:102-111) on packed f32 qkv [B*tokens][3 E]: Q, K, V and the
 *     softmax probabilities are split the same way, f32 softmax, output = pair array [B*tokens][2 E];
 *   vp_patch_im2col_pair: vp_patch_im2col with the hi / lo split of the pixels, pair rows of 2 * C*patch*patch;
 *   vp_deconv_fwd_split: vp_deconv_fwd_bf16 on pair arrays: in [B*H*W][2 C], wph [4][Cout][8 C] (the pair array of
 *     the f32 phase matrices), out [B*2H*2W][2 Cout] = pair of relu(acc * colscale + shift) (eval-mode BatchNorm
 *     folded, topdown_heatmap_simple_head.py:323-334).  C % 32 == 0, Cout % 32 == 0. */
int vp_split_pair_f32(const float* x, void* pair_bf16, size_t n, vp_stream_t stream);
int vp_layernorm_fwd_pair(const float* x, const float* gamma, const float* beta, void* y_pair, float* y_f32,
                          int relu, int M, int E, float eps, vp_stream_t stream);
int vp_attention_split(const float* qkv, void* out_pair, int B, int H, int head_dim, int tokens, float scale,
                       vp_stream_t stream);
int vp_patch_im2col_pair(const float* img, void* col_pair, int B, int C, int H, int W, int patch, int stride, int pad,
                         vp_stream_t stream);
int vp_deconv_fwd_split(const void* in_pair, const void* wph_pair, void* out_pair, int B, int H, int W, int C, int Cout,
                        const float* colscale, const float* shift, const void* zero_page, vp_stream_t stream);

/* ------------------------------------------------------------------ Loss --------
 * JointsMSELoss.forward (mmpose/models/losses/mse_loss.py:24-44) fused with its gradient and the
 * channels-last -> NCHW copy of the heatmaps.  heat [B*HW][ldh] f32 (first K columns valid);
 * heat_nchw [B][K][HW] f32 (optional); target NCHW f32 and tw [B][K] (NULL = no loss);
 * *loss += loss value; dheat [B*HW][ldh] bf16 = dL/dheat (optional). */
int vp_heatmap_loss(const float* heat, int ldh, float* heat_nchw, const float* target, const float* tw,
                    float* loss, void* dheat, int B, int K, int HW, float loss_weight, vp_stream_t stream);
/* The same loss on NCHW f32 heatmaps (the layout TopdownHeatmapSimpleHead.forward returns):
 * *loss += value; dpred (optional, NCHW f32) = dL/dpred.  tw [B][K] or NULL (use_target_weight=False). */
int vp_mse_loss_nchw(const float* pred, const float* target, const float* tw, float* loss, float* dpred,
                     int B, int K, int HW, float loss_weight, vp_stream_t stream);
/* _get_max_preds (mmpose/core/evaluation/top_down_eval.py:63-95): first arg-max of each row. */
int vp_argmax_rows(const float* x, int rows, int L, int* idx, float* val, vp_stream_t stream);

/* ------------------------------------------------------------- Optimizer --------
 * Gradient clipping (mmcv OptimizerHook grad_clip max_norm, configs/.../ViTPose_base_coco_256x192.py:30)
 * and torch.optim.AdamW over flat f32 buffers; per-chunk lr scale / weight decay carry the
 * layer-decay groups of mmcv_custom/layer_decay_optimizer_constructor.py:39-78.
 * *out += sum x^2.  adamw: grads are multiplied by grad_scale * min(1, max_norm / (sqrt(*gnorm_sq) *
 * grad_scale + 1e-6)) on the fly; p_bf16 (optional) receives the bf16 copy of the new weights. */
int vp_sumsq_f32(const float* x, size_t n, float* out, vp_stream_t stream);
/* The same sum with a fixed summation order (per-workgroup partial sums in `partials`, npartials floats of caller-owned
 * scratch, then one workgroup adds them): bit-identical on every data-parallel replica that holds the same gradients,
 * which the atomics of vp_sumsq_f32 do not guarantee. */
int vp_sumsq_f32_det(const float* x, size_t n, float* out, float* partials, int npartials, vp_stream_t stream);
/* sum x^2 over the chunks of the optimizer's table only ({start, length} in elements, lengths multiples of 4): the
 * norm torch's clip_grad_norm_ takes over the parameters that have a gradient -- frozen parameters (no chunk) stay out
 * (mmcv OptimizerHook.clip_grads filters on requires_grad).  Fixed summation order; *out is ASSIGNED. */
int vp_sumsq_chunks_det(const float* x, const long* chunk_start, const int* chunk_len, int nchunks, float* out,
                        float* partials, int npartials, vp_stream_t stream);
/* DropPath.forward (vit.py:48-59 -> timm drop_path): dp[i][...] = floor(keep_i + U[0,1)) / keep_i per sample,
 * keep_i = 1 - rates[i] (rates on the device; rate 0 -> exactly 1: Block 0 and Blocks in eval mode), `per_block`
 * values per Block.  U comes from Philox-4x32-10 keyed by (seed, counter): one counter value per training step. */
int vp_drop_path_scales(float* dp, const float* rates, int depth, int per_block, unsigned long long seed,
                        unsigned long long counter, vp_stream_t stream);
/* x (bf16, n % 8 == 0) *= *scale_dev: seeds the backward pass with the incoming dL/dloss (autograd hands
 * TopDown.forward_train's loss a device scalar). */
int vp_scale_bf16(void* x, size_t n, const float* scale_dev, vp_stream_t stream);
/* Backward of `self.last_norm = nn.Identity()` (mmpose/models/backbones/vit.py:257, ViT(last_norm=False)): dx (f32
 * [M][E]) = dy (bf16), dxb (bf16) = dy * rowscale[row / rows_per_scale] (the last Block's DropPath scale, or NULL). */
int vp_identity_norm_bwd(const void* dy, const float* rowscale, int rows_per_scale, float* dx, void* dxb, int M, int E,
                         vp_stream_t stream);
/* x (f32, n % 4 == 0) *= *scale_dev (the simple decoder's heat-map gradient is f32 NCHW). */
int vp_scale_f32(float* x, size_t n, const float* scale_dev, vp_stream_t stream);
/* out (bf16, n elements, n % 8 == 0) = sum of `count` <= 8 bf16 arrays (host array of device pointers), added in f32:
 * the heads of TopDownMoE share one feature map (mmpose/models/detectors/top_down_moe.py:177-197), whose gradient is the
 * sum of the heads' (autograd's accumulation into `features`). */
int vp_sum_bf16(const void* const* srcs_host, int count, void* out, size_t n, vp_stream_t stream);
/* out[b][0:n] = source[b] == which ? x[b][0:n] : 0 -- `target * (img_sources == i)` and the same for target_weight
 * (top_down_moe.py:181-196); source: int32 [B] on the device. */
int vp_mask_rows_f32(const float* x, const int* source, int which, float* out, int B, size_t n, vp_stream_t stream);
/* hipMemsetAsync(p, 0, bytes) on `stream` (accumulating reductions start from zero). */
int vp_zero(void* p, size_t bytes, vp_stream_t stream);
/* Up to VP_ZERO_MAX_REGIONS small f32 regions zeroed by one launch (BatchNorm sums and parameter gradients, the
 * loss scalar ...).  ptrs_host / floats_host are HOST arrays (copied into the kernel arguments). */
#define VP_ZERO_MAX_REGIONS 16
int vp_zero_regions(float* const* ptrs_host, const int* floats_host, int count, vp_stream_t stream);
/* out[t][e] = pos_embed[1 + t][e] + pos_embed[0][e] (vit.py:320: x + pos_embed[:, 1:] + pos_embed[:, :1]), the
 * [T][E] table the patch-embedding epilogue adds; recomputed after every weight update. */
int vp_pos_combine(const float* pos_embed, float* out, int T, int E, vp_stream_t stream);
/* Gradient of pos_embed [1][1 + T][E] from the gradient dx [B][T][E] of the embedded tokens (vit.py:320:
 * x + pos_embed[:, 1:] + pos_embed[:, :1]): rows 1 .. T = sum over images, row 0 = sum of those rows.  Overwrites. */
int vp_pos_embed_grad(const float* dx, float* gpos, int B, int T, int E, vp_stream_t stream);
int vp_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, const long* chunk_start,
                  const int* chunk_len, const float* chunk_lr_scale, const float* chunk_wd, int nchunks,
                  float lr, float beta1, float beta2, float eps, int step, const float* gnorm_sq,
                  float max_norm, float grad_scale, vp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VITPOSE_HIP_H */
