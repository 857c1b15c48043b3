/*
 * rfuse.h -- C ABI of librfuse_hip.so: the MI355X (gfx950) kernels behind the refinement-inference hot path of
 * RetrievalFuse.
 *
 * The reference (nihalsid/retrieval-fuse) is 100 % Python on PyTorch; it has NO FFI of its own.  Its boundary for
 * this path is the Python class surface in model/__init__.py:6-61.  This header is therefore the build-defined
 * native boundary underneath the drop-in Python classes in retrieval-fuse_amd/model/: every entry point below names
 * the reference code whose arithmetic it replaces (file:line under /root/reference).  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.  All tensor pointers are DEVICE pointers to contiguous
 *     float32 (unless said otherwise) in the reference's own layouts: activations NCDHW, conv weights OIDHW,
 *     Linear weights [out][in].
 *   - No allocation, no global state, stream-ordered on `stream` (a hipStream_t passed as void*).  Scratch is
 *     caller-provided (`ws`, sized by the matching *_ws_bytes function).
 *   - Return 0 on success; <0 on error (RF_E_*).  rf_last_error() gives a thread-local message.
 *   - Volumes are cubic with power-of-two edge (1..128), as every shipped config of the reference is.
 */
#ifndef RFUSE_H
#define RFUSE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RF_OK 0
#define RF_E_INVALID (-1)      /* bad argument (null pointer, non power-of-two edge, ...) */
#define RF_E_UNSUPPORTED (-2)  /* shape outside what the kernels implement */
#define RF_E_LAUNCH (-3)       /* HIP reported a launch error */
#define RF_E_WORKSPACE (-4)    /* caller-provided scratch too small */

int rf_abi_version(void);
const char* rf_last_error(void);

/* ------------------------------------------------------------------------------------------ U-Net primitives */

/* Re-lay a 3x3x3 conv weight OIDHW [cout][cin][27] as the MFMA B-operand image [27][cin4][cout16]
 * (cin4 = cin rounded up to 4, cout16 = cout rounded up to 16, zero filled).  One-off per weight update.
 * Replaces nothing arithmetic; feeds rf_conv3d_k3_gn_relu.  Weights: model/unet.py:15-16,53 (bias=False). */
int rf_conv3_pack_weight(const float* w_oidhw, int cout, int cin, float* w_packed, void* stream);
size_t rf_conv3_packed_floats(int cout, int cin);

/* GroupNorm statistics of the conv INPUT folded into a per-(sample, channel) affine triple, gn_affine [n][c0+c1][4] floats
 * = (center, scale, shift, 0):
 *   y = (x - center) * scale + shift  ==  GroupNorm(G, C, eps, affine)(x)          model/unet.py:54-66 ('g' before 'c')
 * with center = fl32(mean), scale = fl32(gamma*rstd), shift = fl32(beta - (mean - center)*gamma*rstd), mean / rstd in float64.
 * (The centred form keeps the rounding error at eps*|y|; x*scale' + shift' would carry eps*|mean*rstd*gamma|, which on a
 * near-constant input -- a truncation-saturated TSDF patch -- is 100x larger.)
 * The input is the virtual tensor cat(src0[n][c0][edge^3], nearest_upsample_x2(src1[n][c1][(edge/2)^3])) along
 * channels -- Decoder.forward's interpolate + concat, model/unet.py:297-308,354-360 -- with c0 or c1 possibly 0.
 * groups collapses to 1 when (c0+c1) < groups (model/unet.py:62-63) -- done by the CALLER; here groups divides c0+c1.
 * Biased variance, float64 accumulation. */
int rf_gn_stats(const float* src0, int c0, const float* src1, int c1, int n, int edge,
                const float* gamma, const float* beta, int groups, float eps,
                float* gn_affine, void* ws, size_t ws_bytes, void* stream);
size_t rf_gn_stats_ws_bytes(int n, int groups);

/* out[n][cout][edge^3] = ReLU( conv3d_k3_pad1( GN(cat(src0, up2(src1))) ) ), no bias.
 * One SingleConv of order 'gcr' (model/unet.py:19-76,79-100) including the decoder's upsample+concat read
 * (model/unet.py:297-308) when c1 > 0.  Zero padding applies to the NORMALISED tensor.  fp32 MFMA
 * (v_mfma_f32_16x16x4_f32), exact fp32 FMA chains.  w_packed from rf_conv3_pack_weight. */
int rf_conv3d_k3_gn_relu(const float* src0, int c0, const float* src1, int c1, int n, int edge,
                         const float* gn_affine, const float* w_packed, int cout,
                         float* out, void* stream);

/* Same convolution, additionally emitting the GroupNorm statistics of its (ReLU'd) output for the NEXT layer:
 * stats [n][cout][tiles] pairs of float64 (sum, sum of squares) per workgroup tile, tiles = rf_conv3d_stats_tiles(...)
 * (0 = this shape takes a path without fused statistics).  Deterministic (fixed reduction order, no atomics). */
int rf_conv3d_k3_gn_relu_stats(const float* src0, int c0, const float* src1, int c1, int n, int edge,
                               const float* gn_affine, const float* w_packed, int cout,
                               float* out, double* stats, void* stream);
int rf_conv3d_stats_tiles(int c0, int c1, int n, int edge, int cout);

/* rf_gn_stats's result from producer-side statistics instead of re-reading the tensors: stats0 [n][c0][tiles0],
 * stats1 [n][c1][tiles1] (the low-res source; its sums count 8x).  edge = full resolution.  model/unet.py:54-66. */
int rf_gn_from_stats(const double* stats0, int c0, int tiles0, const double* stats1, int c1, int tiles1, int n, int edge,
                     const float* gamma, const float* beta, int groups, float eps, float* gn_affine, void* stream);

/* Same contract on the plain VALU path (one thread per output); the kernels' own cross-check and the path for
 * 1^3 volumes.  Takes the ORIGINAL OIDHW weight. */
int rf_conv3d_k3_gn_relu_direct(const float* src0, int c0, const float* src1, int c1, int n, int edge,
                                const float* gn_affine, const float* w_oidhw, int cout,
                                float* out, void* stream);

/* MaxPool3d(kernel 2, stride 2): Encoder.forward, model/unet.py:237,249-251.  x [n][c][edge^3] -> [n][c][(edge/2)^3] */
int rf_maxpool3d_2(const float* x, int n, int c, int edge, float* out, void* stream);
/* ... also emitting stats [n*c][rf_maxpool_stats_tiles(edge)] (sum, sum of squares) of the pooled output */
int rf_maxpool3d_2_stats(const float* x, int n, int c, int edge, float* out, double* stats, void* stream);
int rf_maxpool_stats_tiles(int edge);

/* out[n][0][v] = (tanh(sum_c w[c] x[n][c][v] + b) + post_add) * post_mul
 * Conv3d(nf,1,1)+Tanh of Superresolution08FinalDecoder (model/refinement.py:54-55); with post_add=1,
 * post_mul=trunc/2 also network_pred_to_df (trainer/train_refinement.py:242-243). */
int rf_conv1x1_tanh(const float* x, int n, int c, size_t voxels, const float* w, const float* b,
                    float post_add, float post_mul, float* out, void* stream);

/* Valid (no padding) strided Conv3d + bias + LeakyReLU(slope): the layers of the conv patch encoders
 * (Patch08 model/retrieval.py:140-147, PCPatch48 :221-234, Patch32 :8-19; slope 0.2).  x [n][cin][s^3], w OIDHW
 * [cout][cin][k^3], out [n][cout][so^3] with so = (s-k)/stride + 1.  Plain fp32 FMAs (direct form). */
int rf_conv3d_valid_leaky(const float* x, int n, int cin, int s, const float* w, const float* bias, int cout, int k,
                          int stride, float slope, float* out, void* stream);

/* The same layer on the matrix cores (fp32 MFMA implicit GEMM, K index = tap*cin + ci, operands gathered from cache, no
 * LDS): w_packed = rf_convv_pack_weight image [k^3*cin -> 4][cout -> 16].  Used by the conv patch encoders. */
int rf_conv3d_valid_leaky_mfma(const float* x, int n, int cin, int s, const float* w_packed, const float* bias, int cout, int k,
                               int stride, float slope, float* out, void* stream);
int rf_convv_pack_weight(const float* w_oidhw, int cout, int cin, int k, float* w_packed, void* stream);
size_t rf_convv_packed_floats(int cout, int cin, int k);

/* The same layer as an LDS-staged implicit GEMM for the LARGE layers (output edge >= 8: the first four layers of PCPatch48 /
 * Patch32, model/retrieval.py:221-228, 8-15): whole output rows per workgroup, the input tile of <= 2 channels staged in LDS,
 * K walked in groups of four taps of one channel.  rf_conv3d_valid_lds_supported: 1 when this form takes the shape, else use
 * rf_conv3d_valid_leaky_mfma.  w_packed: rf_convv_lds_pack_weight image [cin][ceil(k^3/4)][4][cout -> 16]. */
int rf_conv3d_valid_lds_supported(int n, int cin, int s, int cout, int k, int stride);
int rf_conv3d_valid_leaky_lds(const float* x, int n, int cin, int s, const float* w_packed, const float* bias, int cout, int k,
                              int stride, float slope, float* out, void* stream);
int rf_convv_lds_pack_weight(const float* w_oidhw, int cout, int cin, int k, float* w_packed, void* stream);
size_t rf_convv_lds_packed_floats(int cout, int cin, int k);

/* The same layer on the F16 matrix cores by operand splitting (x = h + l/2^11, three f16 MFMAs per product block, fp32
 * accumulation; closer to float64 than the fp32 MFMA chain): the large layers with cin a multiple of 4 (PCPatch48's 12 -> 24 k3
 * @44^3, 24 -> 48 k3 s2 @42^3, 48 -> 48 k3 s2 @20^3, model/retrieval.py:222-228; Patch32's, :9-15).  K is walked in pieces of
 * (tap, 4 channels); the tile / chunk plan depends on (cin, s, cout, k, stride) only and the weight image is packed for it:
 * rf_convv_split_packed_bytes is 0 when the form does not take the layer. */
int rf_conv3d_valid_split_supported(int n, int cin, int s, int cout, int k, int stride);
int rf_conv3d_valid_leaky_split(const float* x, int n, int cin, int s, const void* w_packed, const float* bias, int cout, int k,
                                int stride, float slope, float* out, void* stream);
/* ... with the activations BETWEEN two such layers in split form (no GroupNorm sits between the encoders' convs, so the producer can always
 * write what the consumer would otherwise make of every value it stages -- with its halo, three times): a tensor [n][c][s^3] in split form is
 * [n][c/4][h | l][s^3] 8-byte slots (4 channels of a voxel as f16: h = f16(x/16), l = f16((x/16 - h) * 2^11)), the same number of bytes as fp32.
 * in_split / out_split select the form of x / out (out_split needs cout in multiples of 4).  Same results bit for bit. */
int rf_conv3d_valid_leaky_split_ex(const void* x, int in_split, int n, int cin, int s, const void* w_packed, const float* bias, int cout, int k,
                                   int stride, float slope, void* out, int out_split, void* stream);
int rf_convv_split_pack_weight(const float* w_oidhw, int cout, int cin, int k, int s, int stride, void* w_packed, void* stream);
size_t rf_convv_split_packed_bytes(int cout, int cin, int k, int s, int stride);
/* ... and as a PERSISTENT kernel for the layers the encoders evaluate once on the whole padded chunk instead of per window (PCPatch48's 12 -> 24 k3 @140^3,
 * model/retrieval.py:222): weights and tables LDS-resident, 512-voxel tiles double buffered, split form in and out (x and out as rf_conv3d_valid_leaky_split_ex
 * with in_split = out_split = 1), the epilogue from registers.  Same operands and K order as rf_conv3d_valid_leaky_split_ex on another MFMA shape: equal to it within the last
 * bits of an fp32 sum; its own weight image.  The instantiation built: 12 input channels, 17..24 couts in fours, k = 3, stride 1, even edges 64..254. */
int rf_conv3d_valid_split_pg_supported(int n, int cin, int s, int cout, int k, int stride);
int rf_conv3d_valid_leaky_split_pg(const void* x, int n, int cin, int s, const void* w_packed, const float* bias, int cout, int k, int stride,
                                   float slope, void* out, void* stream);
int rf_convv_split_pg_pack_weight(const float* w_oidhw, int cout, int cin, int k, int s, int stride, void* w_packed, void* stream);
size_t rf_convv_split_pg_packed_bytes(int cout, int cin, int k, int s, int stride);

/* The same layer on the packed-fp32 VALU for the FIRST layers of the patch encoders (stride 1; 1 -> 8/12 with k = 3/5, 1 -> 16,
 * 8 -> 16 and 12 -> 24 with k = 3): couts of 12 / 24 waste a quarter of the 16-wide MFMA tiles, the vector unit has the same fp32
 * peak and no padding.  w_t: the weight as [cin][k^3][cout] (OIDHW permuted to (1,2,3,4,0)): the kernel reads whole cout vectors
 * through the scalar cache.  rf_conv3d_valid_valu_supported: 1 when this form takes the shape. */
int rf_conv3d_valid_valu_supported(int n, int cin, int s, int cout, int k, int stride);
int rf_conv3d_valid_leaky_valu(const float* x, int n, int cin, int s, const float* w_t, const float* bias, int cout, int k,
                               int stride, float slope, float* out, void* stream);
int rf_conv3d_valid_leaky_valu_ex(const float* x, int n, int cin, int s, const float* w_t, const float* bias, int cout, int k,
                                  int stride, float slope, void* out, int out_split, void* stream);

/* rf_conv3d_k3_gn_relu with the encoder's MaxPool3d(2) (model/unet.py:230-253) fused into the epilogue: additionally
 * writes pool_out [n][cout][(edge/2)^3] = maxpool2(out) and, when pool_stats is non-NULL, its (sum, sum of squares)
 * [n][cout][rf_conv3d_stats_tiles(...)][2] for rf_gn_from_stats.  out == NULL: only the pooled tensor is written (an
 * encoder level whose full-resolution output nobody reads: UNet3D with remove_n_final_layers, model/unet.py:500-507);
 * stats must then be NULL.  Only shapes with rf_conv3d_pool_supported(...) == 1 (the 8^3-box tiling; whole 4^3 volumes on
 * the position-major kernel). */
int rf_conv3d_pool_supported(int c0, int c1, int n, int edge, int cout);
int rf_conv3d_k3_gn_relu_pool(const float* src0, int c0, const float* src1, int c1, int n, int edge,
                              const float* gn_affine, const float* w_packed, int cout,
                              float* out, double* stats, float* pool_out, double* pool_stats, void* stream);

/* Decoder form of rf_conv3d_k3_gn_relu (model/unet.py:297-308: nearest x2 upsample of the low-res source, concat after
 * the skip source, then SingleConv 'gcr'): same inputs and result, but the c1 upsampled channels are convolved in LOW
 * resolution -- per output parity the 27 taps collapse to 2x2x2 taps with pre-summed weights (8/27 of the multiply-adds
 * for those channels).  Weight image: rf_conv3_up_pack_weight(OIDHW weight, cout, c0, c1) -> rf_conv3_up_packed_floats
 * floats (pre-sums in float64, rounded once).  stats: optional [n][cout][rf_conv3d_up_stats_tiles(...)][2] float64
 * (sum, sum of squares) of the output for rf_gn_from_stats, or NULL.  rf_conv3d_up_supported: 1 when this kernel takes
 * the shape (needs c1 > 0, edge 4 or a multiple of 8, and enough boxes to fill the chip), else use rf_conv3d_k3_gn_relu. */
size_t rf_conv3_up_packed_floats(int cout, int c0, int c1);
int rf_conv3_up_pack_weight(const float* w_oidhw, int cout, int c0, int c1, float* w_packed, void* stream);
int rf_conv3d_up_supported(int c0, int c1, int n, int edge, int cout);
int rf_conv3d_up_stats_tiles(int c0, int c1, int n, int edge, int cout);
/* which tiling rf_conv3d_up_k3_gn_relu uses for a shape (0 parity-split boxes, 1 position-major 4^3): they issue different numbers of multiply-adds (zero-padding taps left out), for reporting only */
int rf_conv3d_up_variant(int c0, int c1, int n, int edge, int cout);
int rf_conv3d_up_k3_gn_relu(const float* src0, int c0, const float* src1, int c1, int n, int edge,
                            const float* gn_affine, const float* w_packed, int cout, float* out, double* stats, void* stream);

/* The same decoder-form convolution on the F16 matrix cores by OPERAND SPLITTING (csrc/conv3d_up_split.hip): every fp32 operand
 * is carried as two f16 numbers h = f16(x), l = f16((x - h) * 2^11); a*b ~ ah*bh + (ah*bl + al*bh) / 2^11 with exact f16 x f16
 * products, fp32 accumulation in two separate accumulators, combined once.  Same inputs, outputs, statistics layout (one tile per
 * sample) and arithmetic contract as rf_conv3d_up_k3_gn_relu -- measured rounding error against float64 is half that of the
 * fp32 MFMA chain (tools/micro/split_probe.hip) at three 16-cycle MFMAs per 8192 multiply-adds instead of eight 32-cycle ones.
 * Takes whole 8^3 samples (edge == 8, n >= 256), c0 and c1 in multiples of 8, c1 <= 64, 33..64 couts, or whole 4^3 samples
 * (edge == 4, n >= 1024, c0 and c1 in multiples of 8, any cout) (rf_conv3d_up_split_supported).  Weight image: rf_conv3_up_split_pack_weight -> rf_conv3_up_split_packed_bytes bytes
 * (f16 fragment order, pre-sums in float64 and split from the float64 value). */
size_t rf_conv3_up_split_packed_bytes(int cout, int c0, int c1);
int rf_conv3_up_split_pack_weight(const float* w_oidhw, int cout, int c0, int c1, void* w_packed, void* stream);
int rf_conv3d_up_split_supported(int c0, int c1, int n, int edge, int cout);
/* statistics tiles per (sample, cout) of rf_conv3d_up_split_k3_gn_relu's `stats` [n][cout][tiles]: 1 for whole-sample workgroups, (edge/8)^3 for the
 * box-tiled form (c0 == 0, edge >= 16: DecoderNoJoining's first conv, model/unet.py:311-322) */
int rf_conv3d_up_split_stats_tiles(int c0, int c1, int n, int edge, int cout);
int rf_conv3d_up_split_k3_gn_relu(const float* src0, int c0, const float* src1, int c1, int n, int edge,
                                  const float* gn_affine, const void* w_packed, int cout, float* out, double* stats, void* stream);

/* rf_conv3d_k3_gn_relu / _stats / _pool on the F16 matrix cores by operand splitting (csrc/conv3d_split.hip; arithmetic as
 * rf_conv3d_up_split_k3_gn_relu above), 8^3 output boxes, one full-resolution source.  out / stats / pool_out / pool_stats as in
 * rf_conv3d_k3_gn_relu_pool (pool_out NULL: no pooling; out NULL: pooled tensor only); statistics tiles per sample: (edge/8)^3 =
 * rf_conv3d_stats_tiles of the same shape.  Takes cin >= 8 (channel counts between multiples of 8 are padded with zero slots and
 * taken when at least 3/4 of the slots are real: 12, 20, 28, 42 ...), up to 32 couts, edge >= 8, and enough boxes; also whole 4^3
 * samples (n >= 1024, cin in eights, any cout, no fused max-pool: pool_out must be NULL) (rf_conv3d_split_supported).  Weight image: rf_conv3_split_pack_weight -> rf_conv3_split_packed_bytes bytes. */
size_t rf_conv3_split_packed_bytes(int cout, int cin);
int rf_conv3_split_pack_weight(const float* w_oidhw, int cout, int cin, void* w_packed, void* stream);
int rf_conv3d_split_supported(int c0, int c1, int n, int edge, int cout);
int rf_conv3d_split_k3_gn_relu(const float* src, int cin, int n, int edge, const float* gn_affine, const void* w_packed, int cout,
                               float* out, double* stats, float* pool_out, double* pool_stats, void* stream);

/* The same layer on whole 2^3 volumes as ONE dense GEMM on the F16 matrix cores (csrc/conv3d_e2_split.hip): in a 2^3 volume every input voxel is a
 * neighbour of every output voxel, so out[n][(co, v)] = GN(x)[n][(ci, u)] . B[(ci, u)][(co, v)] with B[(ci, u)][(co, v)] = W[co][ci][tap(u - v)] -- x and
 * out ARE those row-major matrices.  Split operands like rf_conv3d_split_k3_gn_relu.  edge = 2 (cin a multiple of 4) or edge = 1 (only the centre tap
 * touches data: K = cin, N = cout), cin >= 8, n >= 16; the weight image is packed for one edge; stats [n][cout][1][2] float64 (sum, sum of squares) or
 * null.  Deepest levels of the U-Nets: reference model/unet.py:125-144 on 2^3 / 1^3. */
size_t rf_conv3_e2_split_packed_bytes(int cout, int cin, int edge);
int rf_conv3_e2_split_pack_weight(const float* w_oidhw, int cout, int cin, int edge, void* w_packed, void* stream);
int rf_conv3d_e2_split_supported(int cin, int n, int edge, int cout);
int rf_conv3d_e2_split_k3_gn_relu(const float* src, int cin, int n, int edge, const float* gn_affine, const void* w_packed, int cout,
                                  float* out, double* stats, void* stream);

/* Activations in split form end to end ("pre-split" tensors).  Layout of C channels on edge^3 voxels: [n][cg = ceil(C/8)][plane h | plane l]
 * [voxel][8 halves] -- per (sample, 8-channel group) the LDS image of the split box kernel in global memory (rf_split_act_bytes bytes, as
 * many as the fp32 tensor), ALREADY normalised by the consumer's GroupNorm, scaled by 2^-4 and split into f16 pairs, so that the consumer
 * stages it with copies instead of re-normalising and re-splitting every input voxel (1.95x with the halo).
 *   rf_conv3d_cin1_presplit       first conv of a level-0 DoubleConv (model/unet.py:125-144: GroupNorm(1) -> Conv3d(1, 8, 3, pad 1) -> ReLU) on
 *                                 whole 16^3 samples (in_gamma / in_beta [1]: the first GroupNorm, statistics taken in the kernel), emitting the SECOND
 *                                 conv's input: statistics of the 8 output channels over the sample,
 *                                 that layer's GroupNorm(next_groups, 8, eps; next_gamma / next_beta [8]) applied, split, written.
 *                                 Conv in the tap order of rf_conv3d_k3_gn_relu's first-layer kernel; GroupNorm arithmetic = gn_affine.
 *   rf_conv3d_split_pre_k3_relu   rf_conv3d_split_k3_gn_relu on such an input (no affine table); outputs / statistics / fused max-pool alike. */
size_t rf_split_act_bytes(int n, int c, int edge);
int rf_conv3d_cin1_presplit_supported(int n, int edge, int cout, int next_groups);
int rf_conv3d_cin1_presplit(const float* src, int n, int edge, const float* in_gamma, const float* in_beta, float in_eps, const float* w_packed, int cout,
                            const float* next_gamma, const float* next_beta, int next_groups, float eps, void* out_presplit, void* stream);

/* The decoder form on whole 8^3 samples (rf_conv3d_up_split_k3_gn_relu's shapes at edge 8) with its output handed to the next SingleConv
 * pre-split: the workgroup holds the sample, so it takes the output's statistics, applies the NEXT layer's GroupNorm (next_gamma / next_beta
 * [cout], next_groups, eps), splits and writes rf_split_act_bytes(n, cout, 8) bytes for rf_conv3d_split_pre_k3_relu; the fp32 output is not
 * written.  StepDownDoubleConv / DoubleConv of a decoder: reference model/unet.py:125-159. */
/* rf_conv3d_split_k3_gn_relu with the final decoder's head (Conv3d(nf, 1, 1) + bias -> tanh -> (pred + post_add) * post_mul: reference
 * model/refinement.py:48-61, trainer/train_refinement.py:242-243) fused into its epilogue: out1 [n][1][edge^3]; the arithmetic of rf_conv1x1_tanh on the
 * conv's output, bit for bit, without the cout-channel tensor ever reaching memory. */
int rf_conv3d_split_pointwise_supported(int cin, int n, int edge, int cout);
int rf_conv3d_split_k3_gn_relu_pointwise_tanh(const float* src, int cin, int n, int edge, const float* gn_affine, const void* w_packed, int cout,
                                              const float* pw_w, const float* pw_b, float post_add, float post_mul, float* out1, void* stream);

/* ... and the box form on whole 8^3 samples with up to 16 couts (an encoder level's first conv, model/unet.py:125-144): the same hand-over. */
int rf_conv3d_split_presplit_supported(int cin, int n, int edge, int cout, int next_groups);
int rf_conv3d_split_presplit(const float* src, int cin, int n, int edge, const float* gn_affine, const void* w_packed, int cout, const float* next_gamma,
                             const float* next_beta, int next_groups, float eps, void* out_presplit, double* stats, void* stream);
int rf_conv3d_up_split_presplit_supported(int c0, int c1, int n, int edge, int cout, int next_groups);
int rf_conv3d_up_split_presplit(const float* src0, int c0, const float* src1, int c1, int n, int edge, const float* gn_affine, const void* w_packed,
                                int cout, const float* next_gamma, const float* next_beta, int next_groups, float eps, void* out_presplit,
                                double* stats, void* stream);
int rf_conv3d_split_pre_supported(int cin, int n, int edge, int cout);
/* tiles per (sample, cout) of the statistics rf_conv3d_split_pre_k3_relu writes ([n][cout][tiles] (sum, sum of squares)): one per 8^3 box, or ONE per
 * sample where the persistent z-column form takes the layer (csrc/conv3d_split_zc.hip: 8 -> <= 16 channels on 16^3 samples, the second conv of the
 * retrieval backbone's level 0, model/unet.py:125-144 -- a workgroup walks whole samples and sums their boxes itself) */
int rf_conv3d_split_pre_stats_tiles(int cin, int n, int edge, int cout);
int rf_conv3d_split_pre_k3_relu(const void* src_presplit, int cin, int n, int edge, const void* w_packed, int cout,
                                float* out, double* stats, float* pool_out, double* pool_stats, void* stream);
/* The decoder pair of the retrieval backbone's last stage (reference model/refinement.py:64-73, model/unet.py:149-159) with the hand-over in PARITY-MAJOR slot
 * order: per (sample, 8-channel group, h | l) the 512 voxel slots of an 8^3 sample at ((z & 1) 4 + (y & 1) 2 + (x & 1)) 64 + (z >> 1) 16 + (y >> 1) 4 + (x >> 1)
 * instead of z 64 + y 8 + x.  rf_conv3d_up_split_presplit_pm is the persistent producer (csrc/conv3d_up_split.hip: k_conv3_up_split_pp; one workgroup per CU
 * walks samples, a wave owns one output parity and leaves its slots from registers in 256-byte runs); same values and statistics as
 * rf_conv3d_up_split_presplit up to the summation order.  rf_conv3d_split_pre_pm_k3_relu = rf_conv3d_split_pre_k3_relu on such a tensor (persistent multi-chunk
 * form: n >= 2048 samples, cin >= 16 in eights, <= 32 couts). */
int rf_conv3d_up_split_presplit_pm_supported(int c0, int c1, int n, int edge, int cout, int next_groups);
int rf_conv3d_up_split_presplit_pm(const float* src0, int c0, const float* src1, int c1, int n, int edge, const float* gn_affine, const void* w_packed,
                                   int cout, const float* next_gamma, const float* next_beta, int next_groups, float eps, void* out_presplit_pm,
                                   double* stats, void* stream);
int rf_conv3d_split_pre_pm_supported(int cin, int n, int edge, int cout);
int rf_conv3d_split_pre_pm_k3_relu(const void* src_presplit_pm, int cin, int n, int edge, const void* w_packed, int cout,
                                   float* out, double* stats, float* pool_out, double* pool_stats, void* stream);

/* ------------------------------------------------------------------------------- backward (training slice, N4) */

/* rf_conv3d_k3_gn_relu with the ReLU optional (relu = 0: plain GroupNorm + conv): the DATA-GRADIENT convolution of the
 * backward pass is this kernel on dz with the transposed, tap-flipped weight and an identity affine. */
int rf_conv3d_k3_gn(const float* src0, int c0, const float* src1, int c1, int n, int edge,
                    const float* gn_affine, const float* w_packed, int cout, int relu, float* out, void* stream);
/* out = dy where y > 0 else 0 (count floats, a multiple of 4): the ReLU of SingleConv 'gcr' (model/unet.py:45) backwards */
int rf_relu_backward(const float* dy, const float* y, size_t count, float* out, void* stream);
/* The data-gradient conv on the F16 matrix cores (split operands need their input inside the f16 pair's range, a gradient can be anywhere):
 *   rf_relu_backward_amax   rf_relu_backward that also leaves max |out| as the maximum over amax_slots[rf_relu_backward_amax_slots()] (device
 *                           floats, one per workgroup, all of them written: no atomics, no zeroing);
 *   rf_dgrad_scale_affine   from those slots: the identity GroupNorm affine with scale s = the power of two that puts the maximum into [512, 1024)
 *                           (rows x (0, s, 0, 0), the gn_affine layout) and scales = (s, 1 / s) -- exact scaling, no host sync;
 *   rf_conv3d_split_k3_gn   rf_conv3d_split_k3_gn_relu as a plain operator: any cout (16 or 32 per workgroup), ReLU optional, no statistics.
 * d xn = rf_conv3d_split_k3_gn(dz, affine of rf_dgrad_scale_affine, W^T with flipped taps, relu = 0) / s; the GroupNorm backward that follows takes
 * 1 / s out (rf_gn_backward's dxn_inv_scale = scales + 1). */
int rf_relu_backward_amax_slots(void);
int rf_relu_backward_amax(const float* dy, const float* y, size_t count, float* out, float* amax_slots, void* stream);
int rf_dgrad_scale_affine(const float* amax_slots, int rows, float* affine, float* scales, void* stream);
int rf_conv3d_split_k3_gn_supported(int cin, int n, int edge, int cout);
int rf_conv3d_split_k3_gn(const float* src, int cin, int n, int edge, const float* gn_affine, const void* w_packed, int cout, int relu,
                          float* out, void* stream);
/* The pooling / upsampling ops of the training graph on [n][c][edge^3] tensors:
 *   rf_maxpool3d_2_backward   MaxPool3d(2) backward (model/unet.py:159): dy [n][c][(edge/2)^3] goes to the FIRST maximum of each 2x2x2 cell of x in
 *                             (z, y, x) order (torch's choice), zeros elsewhere -> dx [n][c][edge^3];
 *   rf_upsample3d_2           nearest x2 upsample (model/unet.py:297-308) lo [n][c][edge_lo^3] -> hi [n][c][(2 edge_lo)^3];
 *   rf_sumpool3d_2            its backward: the sum over each 2x2x2 cell in (z, y, x) order. */
int rf_maxpool3d_2_backward(const float* x, const float* dy, int n, int c, int edge, float* dx, void* stream);
int rf_upsample3d_2(const float* lo, int n, int c, int edge_lo, float* hi, void* stream);
int rf_sumpool3d_2(const float* hi, int n, int c, int edge, float* lo, void* stream);
/* GroupNorm backward (model/unet.py:54-66; torch.nn.GroupNorm semantics, biased variance): x, dxn [n][c][edge^3], gamma [c] ->
 * dx [n][c][edge^3], dgamma [c], dbeta [c] (float64 sums in a fixed order, rounded once).  dxn_inv_scale: null, or a device float when d xn arrives
 * multiplied by 1 / *dxn_inv_scale (rf_dgrad_scale_affine's power of two): the three outputs are multiplied by it (they are linear in d xn). */
int rf_gn_backward(const float* x, const float* dxn, int n, int c, int edge, const float* gamma, int groups, float eps, const float* dxn_inv_scale,
                   float* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream);
size_t rf_gn_backward_ws_bytes(int n, int c, int edge);
/* Weight gradient of the 3x3x3 conv: dw[co][ci][tap] = sum_{n,v} dz[n][co][v] * GN(x)[n][ci][v + tap - 1] (zero padded), fp32 MFMA
 * with K = voxels; edge a power of two >= 4 (8^3 boxes; whole 4^3 samples eight at a time).  gn_affine as in the forward. */
int rf_conv3d_k3_wgrad(const float* x, int cin, int n, int edge, const float* gn_affine, const float* dz, int cout, float* dw, void* ws,
                       size_t ws_bytes, void* stream);
size_t rf_conv3d_k3_wgrad_ws_bytes(int cin, int cout, int n, int edge);
/* The same weight gradient on the F16 matrix cores by operand splitting (csrc/conv3d_wgrad_split.hip): k = 32 voxels = four x-rows of an 8^3 box per
 * MFMA, dz scaled by the power of two in scales = (s, 1 / s) of rf_dgrad_scale_affine (|dz| * s inside the f16 range) and GroupNorm(x) / 16 as f16
 * pairs, exact products, fp32 accumulation per workgroup, float64 across workgroups.  cin >= 6 (at least 3/4 of the next multiple of 8), cout >= 8,
 * edge a power of two >= 8; the caller keeps GroupNorm outputs inside the split forms' range (|xn| <= 65504 * 16, as for rf_conv3d_split_k3_gn_relu). */
int rf_conv3d_k3_wgrad_split_supported(int cin, int cout, int n, int edge);
size_t rf_conv3d_k3_wgrad_split_ws_bytes(int cin, int cout, int n, int edge);
int rf_conv3d_k3_wgrad_split(const float* x, int cin, int n, int edge, const float* gn_affine, const float* dz, int cout, const float* scales,
                             float* dw, void* ws, size_t ws_bytes, void* stream);

/* --------------------------------------------------------------------------------------------- fold / unfold */

/* Unfold3D.forward (model/attention.py:186-188): x [b][c][s^3] -> rows [(b*r^3)][c][e^3], r = s/e, row = ((b*r+px)*r+py)*r+pz */
int rf_unfold3d(const float* x, int b, int c, int s, int e, float* rows, void* stream);
/* Fold3D.forward (model/attention.py:170-176): exact inverse of rf_unfold3d */
int rf_fold3d(const float* rows, int b, int c, int s, int e, float* x, void* stream);

/* ------------------------------------------------------------------------------------------------ Linear/MLP */

/* nn.Linear weight [nout][nin] -> MFMA B-operand image [nin4][nout16] (zero padded). */
int rf_linear_pack_weight(const float* w, int nout, int nin, float* w_packed, void* stream);
size_t rf_linear_packed_floats(int nout, int nin);

#define RF_ACT_NONE 0
#define RF_ACT_RELU 1
#define RF_ACT_LEAKY 2
/* y[rows][nout] = act(x[rows][nin] . W^T + bias); act = none | ReLU | LeakyReLU(slope).
 * AttentionFeatureEncoder layers (model/attention.py:36-42, slope 0.01), Patch04 layers (model/retrieval.py:68-78),
 * final_layer of the conv patch encoders (model/retrieval.py:149).  fp32 MFMA. */
int rf_linear(const float* x, int rows, int nin, const float* w_packed, const float* bias, int nout,
              int act, float slope, float* y, void* stream);

/* Weight gradient of a Linear layer: dw[M][N] = sum_k a[k][M] * b[k][N] with a = dL/d(pre-activation) [K rows][M = nout] and
 * b = the layer's input [K rows][N = nin] (training slice; the reference trains these layers in trainer/train_refinement.py:108-116).
 * Split-K fp32 MFMA, slices summed in float64 in a fixed order (deterministic).  ws: rf_linear_wgrad_ws_bytes(K, M, N). */
int rf_linear_wgrad(const float* a, const float* b, int K, int M, int N, float* dw, void* ws, size_t ws_bytes, void* stream);
size_t rf_linear_wgrad_ws_bytes(int K, int M, int N);

/* x[rows][dim] /= max(||x||_2, eps) in place: F.normalize (util/retrieval.py:66, model/attention.py:92-93) */
int rf_l2_normalize_rows(float* x, int rows, int dim, float eps, void* stream);

/* -------------------------------------------------------------------------------------------------- attention */

#define RF_ATTN_SOFTMAX 0      /* attn_retrieval_mode False: softmax(sharpness * scores), model/attention.py:105-107 */
#define RF_ATTN_GUMBEL_HARD 1  /* attn_retrieval_mode True: gumbel_softmax(25*scores, tau=1, hard=True), :100-103 */
/* AttentionBlock.forward after the theta/phi encoders (model/attention.py:92-112), normalize=True, g=o=Identity,
 * blend_mode=True:
 *   xf[b][f], pf[b][k][f] raw encoder outputs (f = 32), L2-normalised here (eps 1e-12);
 *   scores[b][k] = <xf, pf[k]>; switch = relu(max_k scores); weights by mode (noise[b][k] = Gumbel samples, mode 1);
 *   out[b][d] = x[b][d]*(1-switch) + (sum_k weights[k]*p[b][k][d])*switch,  d = c*e^3 values per row.
 * Optional debug outputs (may be NULL): scores_out[b][k], weights_out[b][k]. */
int rf_attn_fuse(const float* x, const float* p, const float* xf, const float* pf, const float* noise,
                 int b, int k, int d, int f, int mode, float sharpness,
                 float* out, float* scores_out, float* weights_out, void* stream);

/* PatchedAttentionBlock's regroup (model/attention.py:148-152): folded retrieved features [bk][c][s^3] (bk = b*K)
 * -> attention rows p[(b*r^3)][K][c][e^3].  src_layout 0 = NCDHW volumes; 1 = patch-major output of the retrieval
 * backbone [(b*K*q^3)][c][t^3] (q = s/t patches per edge, the layout Fold3D(q,t,c) would consume,
 * trainer/train_refinement.py:37,112) so the fold is never materialised. */
int rf_attn_gather_retrieved(const float* src, int src_layout, int b, int k, int c, int s, int e, int t,
                             float* p_rows, void* stream);

/* Volume-domain route of PatchedAttentionBlock.forward (model/attention.py:141-157) for attention patch extent e = 2,
 * hidden width 128, feature width 32 (every shipped config): the unfolded rows are never materialised.
 *
 * rf_attn_mlp_pack: the 4 Linear layers of one AttentionFeatureEncoder (model/attention.py:36-42; w1 [128][n_in],
 *   w2, w3 [128][128], w4 [32][128], nn.Linear layout) -> one MFMA operand image of rf_attn_mlp_packed_floats(n_in) floats.
 * rf_attn_mlp_rows:   out[rows][32] = encoder(x[rows][n_in])                       (LeakyReLU 0.01 between layers)
 * rf_attn_mlp_volume: the same encoder applied to every 2^3 attention patch of b*kv feature volumes read in place:
 *   src = [(b*kv*q^3)][c][t^3] patch-major (q = s/t; t == s: plain NCDHW volumes [b*kv][c][s^3]), row feature order
 *   (c, e0, e1, e2) as Unfold3D gives; out[((bb*r^3 + prow)*kv + k)][32], r = s/2, prow = (p0*r + p1)*r + p2.
 * rf_attn_weights: per row: L2-normalise xf[rows][f], pf[rows][k][f]; scores; switch = relu(max_k scores); weights by
 *   mode (as rf_attn_fuse) -> weights[rows][k], switches[rows], optional scores_out[rows][k].
 * rf_attn_blend: out[b][c][s^3] = x*(1-switch[row]) + (sum_k weights[row][k]*retrieved_k)*switch[row], row = the
 *   attention patch of the voxel; `retrieved` in the same patch-major / volume layout as rf_attn_mlp_volume's src.   */
/* Split-operand form of the same encoder (F16 matrix cores, every fp32 operand as two f16 pieces, exact products, hi / lo fp32
 * accumulators: csrc/attention_fused.hip): rf_attn_mlp_split_pack writes the f16 fragment image of the four weight matrices
 * (rf_attn_mlp_split_packed_floats floats); rf_attn_mlp_split_rows / _volume take it next to the fp32 image (biases are read from
 * that one) and return the same rows, closer to float64 than the fp32 MFMA form. */
size_t rf_attn_mlp_split_packed_floats(int n_in);
int rf_attn_mlp_split_pack(const float* w1, const float* w2, const float* w3, const float* w4, int n_in, float* packed_split, void* stream);
int rf_attn_mlp_split_rows(const float* x, int rows, int n_in, const float* packed, const float* packed_split, float* out, void* stream);
int rf_attn_mlp_split_volume(const float* src, int b, int kv, int c, int s, int t, const float* packed, const float* packed_split,
                             float* out, void* stream);
size_t rf_attn_mlp_packed_floats(int n_in);
int rf_attn_mlp_pack(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                     const float* w4, const float* b4, int n_in, float* packed, void* stream);
int rf_attn_mlp_rows(const float* x, int rows, int n_in, const float* packed, float* out, void* stream);
int rf_attn_mlp_volume(const float* src, int b, int kv, int c, int s, int t, const float* packed, float* out, void* stream);
int rf_attn_weights(const float* xf, const float* pf, const float* noise, int rows, int k, int f, int mode, float sharpness,
                    float* weights, float* switches, float* scores_out, void* stream);
/* Gumbel-hard weights with the noise of gumbel_softmax (model/attention.py:100-103) drawn INSIDE the kernel: Philox4x32-10 keyed by
 * rng_state[0] (seed), counter (row, draw, rng_state[1] = offset); rng_state = 3 x uint64 on the device {seed, offset, 0}, the offset advances by one
 * per call (also under graph replay).  noise_out (optional, [rows][k]) receives the noise used: rf_attn_weights on it reproduces the weights. */
int rf_attn_weights_sampled(const float* xf, const float* pf, int rows, int k, int f, float sharpness, void* rng_state,
                            float* weights, float* switches, float* scores_out, float* noise_out, void* stream);
int rf_attn_blend(const float* x, const float* retrieved, int b, int k, int c, int s, int t, const float* weights,
                  const float* switches, float* out, void* stream);

/* ----------------------------------------------------------------------------------------- retrieval (online) */

/* Query windows of the retrieval dataset: pad raw input chunk [b][s^3] by `ctx` with pad_value, cut (s/ps)^3 windows
 * of edge ps+2ctx at stride ps (dataset/scene.py:61,152-160), normalise (x-mean)/std
 * (dataset/patched_scene_dataset.py:127).  out [(b*(s/ps)^3)][w^3]. */
int rf_query_windows(const float* raw, int b, int s, int ps, int ctx, float pad_value, float mean, float stddev,
                     float* out, void* stream);

/* Windows of a feature grid: grid [n][c][g^3] -> out [(n np^3)][c][w^3], window (p0, p1, p2) starts at step * (p0, p1, p2); windows in
 * (sample, p0, p1, p2) order = the order of rf_query_windows.  Used by the fully-convolutional evaluation of the conv patch encoders (the
 * reference evaluates model/retrieval.py:4-28,217-243 on every window of dataset/scene.py:152-160 separately; valid convolutions commute
 * with the window cut as long as the window origins stay on the layers' sampling lattice). */
int rf_gather_windows(const float* grid, int n, int c, int g, int w, int step, int np, float* out, void* stream);
int rf_gather_windows_split(const void* grid, int n, int c, int g, int w, int step, int np, void* out, void* stream);      /* grid / out in the valid convs' split form */

/* Database embedding image for the scans: emb [n][dim] row-major (dim = 64) -> rf_db_packed_floats(n, dim) floats holding
 * the blocked view [ceil(n/64)][dim][64] of the VALU scan, the chunk-permuted row view [n32][64] of the MFMA-filtered scans and
 * its per-row half norms [n32], the rows rounded to f16 [n32][64] and their half norms [n32] for the f16 filter (n32 = n rounded up
 * to 32; padding rows can never be returned).
 * DB rows: util/retrieval.py:32,39-45 (columns 7..70). */
int rf_db_pack_embeddings(const float* emb, int64_t n, int dim, float* packed, void* stream);
size_t rf_db_packed_floats(int64_t n, int dim);

/* Exact squared-L2 top-k2 of q[nq][dim] against one DB shard (packed image of `n` rows whose global row ids start
 * at row_base): stands where the reference calls FLANN nn_index(feats, 2K) (util/retrieval.py:92).
 * dist = sum_d (q_d - x_d)^2 in fp32 (four fixed FMA chains, the same bits on every code path), ascending, ties -> lower
 * global row id.  algo: 0 = pick by shard size, 1 = VALU scan (every pair evaluated exactly), 2 / 3 = MFMA-filtered scan
 * (matrix-core dot products -- 2: fp32 inputs, 3: inputs rounded to f16 with a wider margin -- discard the pairs that provably
 * cannot enter a list, the rest is evaluated exactly in fp32): all return the same bits.
 * out_dist [nq][k2] float32, out_idx [nq][k2] int64 (global ids); missing candidates (n < k2): dist=+inf, idx=-1. */
int rf_l2_topk(const float* q, int nq, int dim, const float* db_packed, int64_t n, int64_t row_base, int k2, int algo,
               float* out_dist, int64_t* out_idx, void* ws, size_t ws_bytes, void* stream);
size_t rf_l2_topk_ws_bytes(int nq, int64_t n, int k2);
/* The same search emitting packed 64-bit keys  (float bits of dist) << 32 | global row id  (all ones = no candidate):
 * unsigned order of the keys = (dist, row id) order, so ONE all-gather of out_keys [nq][k2] exchanges a shard's candidates
 *  and rf_topk_merge_keys reduces in_keys [parts][nq][k2] to the global top-k2. */
int rf_l2_topk_keys(const float* q, int nq, int dim, const float* db_packed, int64_t n, int64_t row_base, int k2, int algo,
                    uint64_t* out_keys, void* ws, size_t ws_bytes, void* stream);
int rf_topk_merge_keys(const uint64_t* in_keys, int parts, int nq, int k2, float* out_dist, int64_t* out_idx, void* stream);

/* Merge `parts` candidate lists per query (e.g. the all-gathered per-shard top-k2): in_dist/in_idx [parts][nq][k2]
 * -> out [nq][k2] ascending by (dist, idx).  New (the reference searches one index); defines the sharded-DB merge. */
int rf_topk_merge(const float* in_dist, const int64_t* in_idx, int parts, int nq, int k2,
                  float* out_dist, int64_t* out_idx, void* stream);

/* flann_knn_worker's post-processing (util/retrieval.py:93-100): look up db_meta[idx] = (scene, x0,x1,y0,y1,z0,z1),
 * stably move neighbours whose scene == query_scene[q] (>=0) to the back, keep the first K.
 * db_meta [n_total][7] int32.  out_meta [nq][K][7] int32, out_dist [nq][K], out_idx [nq][K].
 * query_keep [nq] uint8 or NULL: 0 marks a query patch the dataset's occupancy filter dropped
 * (dataset/patched_scene_dataset.py:28-32): it gets K "no neighbour" entries (scene -1, idx -1, dist +inf), which
 * rf_gather_patches turns into the truncation fill -- the patch keeps the trunc init of util/retrieval.py:148,151. */
int rf_demote_same_scene(const float* dist, const int64_t* idx, int nq, int k2, const int32_t* db_meta,
                         const int32_t* query_scene, const uint8_t* query_keep, int K, int32_t* out_meta, float* out_dist,
                         int64_t* out_idx, void* stream);

/* create_retrieval_from_mapping (util/retrieval.py:145-164, no_overlap) fused with the retrieval normalisation
 * (dataset/patched_scene_dataset.py:130-133) and Unfold3D(16,1) (trainer/train_refinement.py:34,112):
 * for chunk c, neighbour k, slot p: copy db_volumes[scene][x0:x1,y0:y1,z0:z1] (16^3) * trunc_ratio, scene<0 -> trunc
 * fill; then (v-mean)/std.  meta [chunks*64][K][7] (slot-major as the mapping).
 * layout 0: out [chunks][K][64^3] composed volumes; layout 1: out [(chunks*K*64)][16^3] unfolded patch rows. */
int rf_gather_patches(const float* db_volumes, int64_t n_scenes, const int32_t* meta, int chunks, int K,
                      float trunc_fill, float trunc_ratio, float mean, float stddev, int layout,
                      float* out, void* stream);

/* The same gather from a float16 voxel store [n_scenes][64^3] -- the precision the reference itself holds scenes in (dataset/scene.py:61 np.float16 on
 * disk, :71 in memory; `get_scene_target` widens, util/retrieval.py:158) -- widened on the fly: the same bits as rf_gather_patches on the widened store
 * for half the bytes read (PatchDatabase(half_store=True): 1 M patches = 8.2 GB per GPU instead of 16.4 GB). */
int rf_gather_patches_f16(const void* db_volumes_f16, int64_t n_scenes, const int32_t* meta, int chunks, int K,
                          float trunc_fill, float trunc_ratio, float mean, float stddev, int layout,
                          float* out, void* stream);
/* create_retrieval_from_mapping for an OVERLAPPING patch grid of one scene (util/retrieval.py:145-164 with dataset.no_overlap False; :156: a patch overwrites its box of
 * retrieval k only while the mean of the distances stored there is above its own).  mapping [P][K][8] float32 rows (scene index, X0, X1, Y0, Y1, Z0, Z1, distance) of the
 * P patches patch_from_scene_lookup holds, in its order; boxes [P][6] int32 their unpadded target boxes (xx0, xx1, yy0, yy1, zz0, zz1) in the scene; db_volumes
 * [n_scenes][64^3] float32 (half = 0) or float16 (half = 1); out, dist_ws [K][sx][sy][sz] float32 (dist_ws: workspace).  Sequential over the patches by construction. */
int rf_compose_overlap(const void* db_volumes, int half, int64_t n_scenes, const float* mapping, const int32_t* boxes, int P, int K, int sx, int sy, int sz,
                       float trunc_fill, float trunc_ratio, float* out, float* dist_ws, void* stream);

/* out[m][width] = src[idx[m]][width] (width % 4 == 0): row gather used to fetch cached, query-independent retrieval
 * backbone features of database patches (an optional serving mode; the reference recomputes them,
 * trainer/train_refinement.py:112).  idx[m] < 0 ("no neighbour") or >= n_src selects row n_src-1, where the database keeps
 * its all-trunc sentinel patch (util/retrieval.py:21-26,45). */
int rf_gather_rows(const float* src, int64_t n_src, const int64_t* idx, int64_t m, int width, float* out, void* stream);

/* ------------------------------------------------------------------------------- pre-split hand-over across the level-0 max-pool */

/* The retrieval backbone's level 0 -> level 1 (reference model/unet.py:230-253: DoubleConv, MaxPool3d(2), DoubleConv) without an fp32 consumer in between:
 * rf_conv3d_split_pre_k3_relu_pool_presplit = rf_conv3d_split_pre_k3_relu (pooled output only) whose persistent workgroups also write the pooled tensor
 * pre-split for the NEXT level's first conv (that layer's GroupNorm applied from the sample's own statistics: next_gamma / next_beta [cout], next_groups, eps);
 * rf_conv3d_split_pre_presplit = that first conv, pre-split in and pre-split out (whole 8^3 samples).  No rf_gn_from_stats launch, no conversion in either
 * consumer.  pool_scratch: rf_conv3d_split_pre_pool_presplit_scratch_floats(cout) floats of workspace -- a slot per persistent workgroup for the pooled fp32
 * values of the sample in flight (read back through L2 once the sample's statistics are known); NOT a pooled tensor: nothing else consumes one. */
size_t rf_conv3d_split_pre_pool_presplit_scratch_floats(int cout);
int rf_conv3d_split_pre_pool_presplit_supported(int cin, int n, int edge, int cout, int next_groups);
int rf_conv3d_split_pre_k3_relu_pool_presplit(const void* src_presplit, int cin, int n, int edge, const void* w_packed, int cout, float* pool_scratch,
                                              double* pool_stats, const float* next_gamma, const float* next_beta, int next_groups, float eps,
                                              void* out_presplit, void* stream);
int rf_conv3d_split_pre_presplit_supported(int cin, int n, int edge, int cout, int next_groups);
int rf_conv3d_split_pre_presplit(const void* src_presplit, int cin, int n, int edge, const void* w_packed, int cout, const float* next_gamma,
                                 const float* next_beta, int next_groups, float eps, void* out_presplit, double* stats, void* stream);

/* ------------------------------------------------------------------------------- channel-interleaved hand-over (the final decoder's 64^3 pair) */

/* "ch8": an fp32 activation tensor stored [n][c / 8][edge^3][8 channels] instead of NCDHW.  The final decoder's conv pair (reference model/refinement.py:48-61:
 * DecoderNoJoining(16, 16) on 64^3, then Conv3d(16, 1, 1) + tanh) cannot hand over pre-split -- a 64^3 sample is 512 boxes, no workgroup has the statistics
 * the second GroupNorm needs -- but the second conv stages 8 channels of a voxel at a time, and from an NCDHW tensor that is eight 4-byte gathers in 40-byte
 * runs per voxel (bound by the address path, not by HBM).  The producer therefore writes its output ch8 (256 contiguous bytes per box row), the consumer
 * loads two 16-byte pieces per voxel.  Values and statistics are those of rf_conv3d_up_split_k3_gn_relu / rf_conv3d_split_k3_gn_relu_pointwise_tanh. */
int rf_conv3d_up_split_ch8_supported(int c0, int c1, int n, int edge, int cout);
int rf_conv3d_up_split_k3_gn_relu_ch8(const float* src0, int c0, const float* src1, int c1, int n, int edge, const float* gn_affine,
                                      const void* w_packed, int cout, float* out_ch8, double* stats, void* stream);
int rf_conv3d_split_pointwise_ch8_supported(int cin, int n, int edge, int cout);
int rf_conv3d_split_k3_gn_relu_pointwise_tanh_ch8(const float* src_ch8, int cin, int n, int edge, const float* gn_affine, const void* w_packed, int cout,
                                                  const float* pw_w, const float* pw_b, float post_add, float post_mul, float* out1, void* stream);

/* ------------------------------------------------------------------------------- mesh export */

/* Marching cubes on the device: the last step of the reference's inference loop (util/visualization.py:34-37 visualize_sdf_as_mesh:
 * marching_cubes(sdf, 0.75) -> vertices, triangles -> .obj; trainer/train_refinement.py:170-173).  The reference's `marching_cubes` package is third-party
 * and not pinned: PARITY UNPINNED -- contract = the construction itself (rfuse/mesh.py, csrc/mesh.hip).  sdf [x][y][z] fp32, inside = value < level.
 * Two passes around exclusive scans the caller makes (torch.cumsum):
 *   rf_mc_classify: cube_ntri [x*y*z] int32 triangles of the cube whose corner 0 is that voxel; edge_flag [x*y*z*3] int32: the level crosses the grid
 *                   edge that starts at that voxel along axis 0 / 1 / 2.  tri_count int8 [256]: triangles per corner configuration.
 *   rf_mc_emit:     verts [V][3] fp32 (voxel-index coordinates, one per crossed grid edge at edge_off), tris [T][3] int32 (vertex indices, cube by cube at
 *                   cube_off).  tri_table int8 [256][16]: cube-edge numbers (axis * 4 + u + 2 v), -1 terminated (rfuse/mesh.py:build_tables). */
int rf_mc_classify(const float* sdf, int x, int y, int z, float level, const signed char* tri_count, int* cube_ntri, int* edge_flag, void* stream);
int rf_mc_emit(const float* sdf, int x, int y, int z, float level, const signed char* tri_table, const long long* cube_off, const long long* edge_off,
               const int* cube_ntri, const int* edge_flag, float* verts, int* tris, void* stream);

/* Scene recomposition on the device (reference dataset/patched_scene_dataset.py:160-186 combine_predictions after trainer/train_refinement.py:160-167's
 * `.cpu().half()`): m chunks of a refined batch df [b][64^3] fp32 -> float16 rounding (round_half != 0) -> float64 -> pasted into `flat` at element
 * offsets dst[i] (chunk sel[i] of the batch) with canvas strides sx, sy (elements; z contiguous; offsets and strides even: 16-byte stores). */
int rf_paste_chunks(const float* df, int b, const int* sel, const long long* dst, int m, long long sx, long long sy, int round_half, double* flat,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RFUSE_H */
