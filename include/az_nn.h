/*
 * az_nn.h - C ABI of the fused "glue" kernels of the leaf evaluator (part of libaz_mcts.so).
 *
 * These do not replace a reference FFI entry point: the reference's evaluator is a PyTorch
 * module (src/environments/Connect4/Network.py) and stays one here.  They replace, inside our
 * inference twin of that module (alphazero-al_amd/src/fast_net.py), the chains of small
 * PyTorch kernels between its GEMM-shaped operations.  All tensors are DEVICE pointers to
 * contiguous bf16 data in token layout (batch, 42, channels) unless stated; `stream` is a
 * hipStream_t; every function only enqueues work and returns 0, or 1 on a bad argument.
 *
 * Compact batches: the four kernels of the forward pass (embed, conv_block, attn_block, heads)
 * take `batch_dev`, a device pointer to an int64 (or NULL): when given, only the first
 * min(batch, *batch_dev) samples are processed - `batch` sizes the launch, the device decides
 * the work (the leaves that missed the transposition table, az_mcts.h).  `gather[b]` (embed)
 * is the row of the feature tensor that compact sample b shows, `scatter[b]` (heads) the row
 * of the mask and of the three output arrays it belongs to; NULL = identity.
 */
#ifndef AZ_NN_H
#define AZ_NN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* tokens[b, cell, :] = pos[cell, :] + own(b, cell) * emb_own + opp(b, cell) * emb_opp
 * (Network.py:226-239).  features: float32 (batch, 3, 6, 7) relative planes; embed_dim 32. */
int az_nn_embed(const float *features, const void *emb_own, const void *emb_opp, const void *pos,
                void *tokens, int64_t batch, int embed_dim, const int32_t *gather, const int64_t *batch_dev,
                void *stream);
/* GroupNorm(num_groups=1) over each sample's 42*channels values + per-channel affine
 * (Network.py:38,44); channels 64. */
int az_nn_groupnorm1(const void *x, const void *gamma, const void *beta, void *y, int64_t batch,
                     int channels, float eps, void *stream);
/* y = residual + silu(x + bias[channel]) - convolution bias, SiLU and skip connection in one
 * pass (Network.py:44-48).  bias (channels values, channel = fastest dimension of x) and
 * residual may be NULL; n_elements and channels multiples of 8. */
int az_nn_silu_add(const void *x, const void *bias, int channels, const void *residual, void *y,
                   int64_t n_elements, void *stream);
/* One whole convolution block as a single MFMA kernel (nn_conv.hip):
 *   y = [x +] silu(conv3x3([GroupNorm1(x) * gamma + beta]) + bias)        (Network.py:27-48,166-170)
 * x (batch, 42, c_in), y (batch, 42, 64); weight_ohwi = the (64, c_in, 3, 3) weight stored
 * output-major with the input channel fastest, i.e. (64, 3, 3, c_in) contiguous
 * (torch channels_last memory of the OIHW tensor).  gamma/beta NULL = no normalisation.
 * Supported: c_in 64 with normalisation (residual 0/1), c_in 32 without either (the stem). */
int az_nn_conv_block(const void *x, int c_in, const void *weight_ohwi, const void *bias, const void *gamma,
                     const void *beta, int residual, void *y, int64_t batch, float eps, const int64_t *batch_dev,
                     void *stream);
/* The residual block (c_in 64, normalised, residual) in its second form (nn_conv2.hip): one wavefront per SIMD on
 * 32x32x16 MFMAs, GroupNorm folded into weights and epilogue.  The caller prepares, once per set of weights
 * (alphazero-al_amd/src/fast_net.py `fold_block`):
 *   weight_folded_ohwi  bf16 (64, 3, 3, 64): weight * gamma[c_in], rounded to bf16 once
 *   t1  float32 (9, 64): for border class k = 3 * rowclass + colclass (0 first / 1 inner / 2 last row or column)
 *       the sum of the FOLDED (rounded) weights over the taps that fall inside the board and over c_in
 *   t2_scaled  float32 (9, 64): log2(e) * (bias + the same sum of weight * beta)
 * so that  conv(W, pad(GN(x)))[o, cell] + bias = rstd * (conv(Wf, pad(x)) - mean * t1[class(cell)][o]) + t2[class][o].
 * Same result as az_nn_conv_block up to where the one bf16 rounding sits (weights instead of normalised activations). */
int az_nn_conv_block2(const void *x, const void *weight_folded_ohwi, const float *t1, const float *t2_scaled, void *y,
                      int64_t batch, float eps, const int64_t *batch_dev, void *stream);

/* The stem with the embedding fused in: az_nn_embed + az_nn_conv_block(c_in 32) as one kernel
 * that builds its tokens from the feature planes (no (batch, 42, 32) tensor in HBM). */
int az_nn_stem_embed(const float *features, const void *emb_own, const void *emb_opp, const void *pos,
                     const void *weight_ohwi, const void *bias, void *y, int64_t batch, const int32_t *gather,
                     const int64_t *batch_dev, void *stream);
/* The same with the planes built from leaf POSITIONS in HBM instead of a feature tensor: per row two
 * bitboards (bit = 7 * column + height, Connect4.h:15-29; player +1 / player -1), the side to move and the
 * symmetry id the leaf is shown under (1 = columns mirrored, Connect4.h:249-262) - what selection leaves
 * behind for every leaf.  Saves the 504-byte feature row per leaf and the kernel that writes it. */
typedef struct az_nn_positions {
    const uint64_t *bb_p1, *bb_p2;
    const int32_t *turn, *sym;
} az_nn_positions;
int az_nn_stem_embed_positions(const az_nn_positions *positions, const void *emb_own, const void *emb_opp, const void *pos,
                               const void *weight_ohwi, const void *bias, void *y, int64_t batch, const int32_t *gather,
                               const int64_t *batch_dev, void *stream);
/* The stem from FOLDED tables (nn_stem.hip): the tokens are own * e_own + opp * e_opp + pos with own / opp in {0, 1} and the
 * convolution is linear, so the layer is a K = 18 GEMM on a 0/1 operand built from the planes / bitboards plus a per-cell
 * constant (same inputs, same output as az_nn_stem_embed[_positions], fp32-accurate sums instead of bf16 tokens):
 *   w_frag  bf16 [2][4][64][8]: the table  T[o][k], k = 2 * tap + plane (tap = 3 * ky + kx; plane 0 own, 1 opponent; k >= 18
 *           zero),  T[o][2 tap] = sum_c W[o][c][tap] e_own[c],  T[o][2 tap + 1] likewise with e_opp, split into a bf16 high
 *           part [0] and a bf16 low part [1] (T - high), each in MFMA fragment order: [channel tile i][lane][j] =
 *           part[32 (i / 2) + 8 (r / 4) + 4 (i % 2) + r % 4][8 (lane >> 4) + j] with r = lane & 15 (the rows of two
 *           neighbouring tiles that a lane ends up with are eight consecutive channels);
 *   pmap    float32 [48][68]: rows 0..41, columns 0..63 = conv3x3(pos map, W)[o, cell] + bias[o] (zero padding), the rest 0.
 * Both come from alphazero-al_amd/src/fast_net.py fold_stem. */
int az_nn_stem_folded(const float *features, const void *w_frag, const float *pmap, void *y, int64_t batch,
                      const int32_t *gather, const int64_t *batch_dev, void *stream);
int az_nn_stem_folded_positions(const az_nn_positions *positions, const void *w_frag, const float *pmap, void *y,
                                int64_t batch, const int32_t *gather, const int64_t *batch_dev, void *stream);
/* The whole gated attention block as a single MFMA kernel (nn_attn.hip):
 *   y = x + o_proj(sigmoid(gate) * softmax(qnorm(Q) knorm(K)^T / 4) V),  [Q|K|V|gate] = qkvg(RMSNorm(x))
 * (Network.py:51-93).  x, y (batch, 42, 64); qkvg_w (196, 64) row-major [out][in] with rows
 * 0-63 Q, 64-127 K, 128-191 V, 192-195 gate; o_w (64, 64) [out][in]; 4 heads of 16. */
int az_nn_attn_block(const void *x, const void *prenorm_w, const void *qkvg_w, const void *q_norm_w,
                     const void *k_norm_w, const void *o_w, void *y, int64_t batch, float eps, const int64_t *batch_dev,
                     void *stream);
/* timing experiments on az_nn_conv_block: bit 0 skips its MFMA phase, bit 1 its epilogue and
 * stores, bit 4 records per-wavefront cycle totals of its phases (az_nn_conv_profile: 8 values
 * per wavefront - P1, barrier, MFMA + epilogue, staging wait, barrier, store - for the first
 * n / 8 wavefronts of the last launch). */
int az_nn_debug(int flags);
int az_nn_conv_profile(unsigned long long *out, int n);
/* nn.RMSNorm over the last dimension of 64 */
int az_nn_rmsnorm64(const void *x, const void *w, void *y, int64_t rows, float eps, void *stream);
/* qkvg (batch*42, row_len) with row_len 196 or 200 (3*64 q|k|v, 4 gate logits, optional zero
 * pad) -> q, k, v (batch, 4, 42, 16) with per-head RMSNorm on q and k, and sigmoid(gate)
 * (batch*42, 4) (Network.py:66-71,80). */
int az_nn_qkv_prep(const void *qkvg, int row_len, const void *q_norm_w, const void *k_norm_w, void *q,
                   void *k, void *v, void *gate_sigmoid, int64_t batch, float eps, void *stream);
/* out[tok, h*16+d] = attn[b, h, t, d] * gate_sigmoid[tok, h] (Network.py:80-82) */
int az_nn_attn_post(const void *attn, const void *gate_sigmoid, void *out, int64_t batch, void *stream);
/* policy-head pooling and value-head mean of one pass over the final tokens (batch, 42, 64):
 * col (batch, 7, 64) = softmax-over-rows weighted sum of the RMS-normalised tokens of each
 * column, mean (batch, 64) = plain token mean (Network.py:107-113,135). */
int az_nn_heads_prep(const void *tokens, const void *p_norm_w, const void *p_gate_w, float p_gate_b,
                     void *col, void *mean, int64_t batch, float eps, void *stream);

/* Both output heads as one kernel (nn_heads.hip): final tokens (batch, 42, 64) ->
 *   probs (batch, 7) f32       softmax of the column policy head, illegal columns (mask byte 0,
 *                              mask (batch, 7) uint8 or NULL) filled with -1e9 before the softmax
 *   wdl (batch, 3) f32         softmax of the value head, RELATIVE order [draw, win, loss]
 *   moves_left (batch) f32     aux_scale * sigmoid(aux head)
 * i.e. exactly the three arrays `predict` returns (Network.py:96-141, 267-288) and
 * az_mcts_dev_backprop consumes.  Weights are bf16 device arrays with the reference's
 * state-dict shapes ([out][in] row-major linears); the four scalars are host floats. */
typedef struct az_nn_heads_weights {
    const void *p_norm, *p_gate_w, *p_fc_w, *p_fc_b, *p_out_w;          /* policy_head.{norm,row_gate,fc,out} */
    const void *d_pool_norm, *d_pool_w, *d_pool_b, *d_norm, *d_fc_w, *d_fc_b, *d_out_norm;
    const void *d_val_w, *d_val_b, *d_aux_w;                             /* dual_head.{value_out,aux_out} */
    float p_gate_b, p_out_b, d_aux_b, aux_scale;
} az_nn_heads_weights;
int az_nn_heads(const void *tokens, const az_nn_heads_weights *w, const uint8_t *mask, float *probs, float *wdl,
                float *moves_left, int64_t batch, float eps, const int32_t *scatter, const int64_t *batch_dev,
                void *stream);

/* The whole evaluator as one call (nn_model.hip): az_nn_stem_embed, n_blocks x az_nn_conv_block
 * (64 -> 64, normalised, residual), az_nn_attn_block, az_nn_heads on `stream`, issued from native
 * code - what alphazero-al_amd/src/fast_net.py does per call from Python.  The object keeps the
 * POINTERS given here (the caller keeps the arrays alive and unchanged) and is immutable, so it
 * may be used from several host threads / streams at once; each call brings its own `scratch`
 * (device memory, az_nn_model_scratch_bytes(batch) bytes: two activation tensors).
 * rows / n_rows: the compact form described at the top (both NULL = every row 0..batch-1). */
#define AZ_NN_MAX_BLOCKS 8
typedef struct az_nn_model_weights {
    const void *emb_own, *emb_opp, *pos;                    /* as az_nn_stem_embed */
    const void *stem_w, *stem_b;
    int32_t n_blocks;                                       /* residual blocks (reference: 3) */
    const void *block_w[AZ_NN_MAX_BLOCKS], *block_b[AZ_NN_MAX_BLOCKS];          /* as az_nn_conv_block */
    const void *block_gamma[AZ_NN_MAX_BLOCKS], *block_beta[AZ_NN_MAX_BLOCKS];
    const void *pre_w, *qkvg_w, *qn_w, *kn_w, *o_w;         /* as az_nn_attn_block */
    az_nn_heads_weights heads;                              /* as az_nn_heads */
    float eps;
    const void *stem_frag;                                  /* as az_nn_stem_folded; both NULL: the stem runs az_nn_stem_embed */
    const float *stem_pmap;
} az_nn_model_weights;
typedef struct az_nn_model az_nn_model;
int az_nn_model_create(const az_nn_model_weights *w, az_nn_model **out);
/* The integer-hash evaluator as a model object (game: 0 Connect4, 1 Othello - AZ_GAME_*): a pure
 * function of the position shown by the feature planes, bit-identical to tests/scenarios.py
 * `hash_eval` / `ot_hash_eval` and to src/hash_eval.py.  Not a network: it exists so that the whole
 * native loop (az_mcts_dev_search) can be compared bit for bit with the CPU oracle, and so that the
 * tree kernels can be timed with no evaluator to speak of.  Needs no scratch. */
int az_nn_model_create_hash(int game, az_nn_model **out);
#define AZ_NN_KIND_CONNECT4_CNN  0
#define AZ_NN_KIND_HASH_CONNECT4 1
#define AZ_NN_KIND_HASH_OTHELLO  2
int az_nn_model_kind(const az_nn_model *m);
void az_nn_model_destroy(az_nn_model *m);
uint64_t az_nn_model_scratch_bytes(const az_nn_model *m, int64_t batch);
int az_nn_model_forward(const az_nn_model *m, const float *features, const uint8_t *mask, float *probs,
                        float *wdl, float *moves_left, int64_t batch, const int32_t *rows,
                        const int64_t *n_rows, void *scratch, uint64_t scratch_bytes, void *stream);
/* az_nn_model_forward with the leaves given as positions (az_nn_positions above; Othello: bit = 8 * row + col,
 * symmetry ids of Othello.h:45) instead of feature planes: what az_mcts_dev_search feeds the evaluator. */
int az_nn_model_forward_positions(const az_nn_model *m, const az_nn_positions *positions, const uint8_t *mask, float *probs,
                                  float *wdl, float *moves_left, int64_t batch, const int32_t *rows,
                                  const int64_t *n_rows, void *scratch, uint64_t scratch_bytes, void *stream);
/* Kernel timing inside az_nn_model_forward (bench.py's roofline of the evaluator, measured on the
 * launches of the timed region instead of a synthetic one): enable = n >= 1 puts a HIP event pair
 * around the FIRST residual convolution block of every n-th forward call (process-wide, at most
 * 4096 pairs between reads; not while the stream is capturing); az_nn_model_profile_read
 * synchronises the device and returns the summed milliseconds and the number of launches summed. */
int az_nn_model_profile(int enable);
int az_nn_model_profile_read(double *out_ms, int64_t *out_launches);
/* The same for every kernel kind of the forward pass: the stem, the first residual block, the attention
 * block, the heads (an event pair around each on every n-th call): summed milliseconds and launches
 * per kind, in the order of AZ_NN_PROFILE_*.  Reading empties the rings (either reader). */
#define AZ_NN_PROFILE_STEM  0
#define AZ_NN_PROFILE_CONV  1
#define AZ_NN_PROFILE_ATTN  2
#define AZ_NN_PROFILE_HEADS 3
int az_nn_model_profile_read_kernels(double out_ms[4], int64_t out_launches[4]);

/* One 3x3 convolution layer of the reference's Othello network (Othello/Network.py:22-66,129-139:
 * 256 output channels on 10x10 / 8x8 maps) as an implicit-GEMM MFMA kernel (nn_othello.hip):
 *   y = silu( post_scale * conv3x3( zero_pad( [pre_scale * x + pre_shift] ) ) + post_shift [+ residual] )
 * x (batch, h_in, h_in, c_in) and y, residual (batch, h_out, h_out, 256) are NHWC bf16, h_out =
 * h_in + 2 * pad - 2.  pre_* (c_in floats, both or neither): the BatchNorm in front of the
 * convolution; post_* (256 floats, required: ones / zeros for none): the one behind it.
 * w_packed: the (256, c_in, 3, 3) weight as bf16 in fragment order [tap][c_in / 32][channel tile 16]
 * [lane 64 = 16 * k group + channel][8 input channels] (fast_othello.pack_conv_weight).
 * Supported (c_in, h_in, pad): (32, 8, 2), (256, 10, 1) with or without pre / residual, (256, 10, 0),
 * (256, 8, 1); apply_silu must be 1.  batch_dev (or NULL): device count of a compact batch, as above.
 * Returns 0, 1 on an unsupported combination. */
int az_nn_othello_conv(const void *x, const void *w_packed, const float *pre_scale, const float *pre_shift,
                       const float *post_scale, const float *post_shift, const void *residual, void *y,
                       int64_t batch, int c_in, int h_in, int pad, int apply_silu, const int64_t *batch_dev, void *stream);

/* The dual head's 8-channel bottleneck of the same network (Othello/Network.py:81-83): 3x3, no padding,
 * 256 -> 8 channels on the 10x10 map, BatchNorm, SiLU.  x (batch, 10, 10, 256) NHWC bf16 -> y (batch, 8, 8, 8)
 * NHWC bf16.  w_packed16: the (8, 256, 3, 3) weight padded with zeros to 16 output channels, packed like
 * az_nn_othello_conv's with ONE channel tile; post_*16: 16 floats each (entries 8..15 unused). */
int az_nn_othello_conv_narrow(const void *x, const void *w_packed16, const float *post_scale16,
                              const float *post_shift16, void *y, int64_t batch, const int64_t *batch_dev, void *stream);

/* The thin ends of the same network as kernels (nn_othello_heads.hip), so that the whole Othello evaluator can
 * be one native object.  az_nn_othello_embed: leaf positions (bit = 8 * row + col, symmetry ids of
 * Othello.h:45) + action masks (rows x 65, the symmetrised frame) -> tokens (batch, 8, 8, 32) NHWC bf16 through a
 * (64 cells x 4 kinds, 32) bf16 table: kind 0 own stone, 1 opponent stone, 2 empty and legal, 3 empty and illegal
 * (Othello/Network.py:201-211).  az_nn_othello_heads: the policy stem's output (batch, 8, 8, 256) and the
 * bottleneck (batch, 8, 8, 8), both NHWC bf16 -> probs (rows, 65), relative wdl (rows, 3), score utility (rows):
 * what `predict` returns (Othello/Network.py:40-104, 229-261).  gather / scatter / batch_dev: compact batches. */
typedef struct az_nn_othello_heads_weights {
    const void  *board_w;                 /* policy_head.board_out.weight, 256 bf16 */
    const float *pass_norm_w, *pass_fc_w; /* policy_head.pass_norm.weight, pass_fc.weight: 256 floats each */
    const float *v_conv_w;                /* dual_head.value_out[0].weight as (72, 8): [ci*9 + ky*3 + kx][co] */
    const float *v_bn_s, *v_bn_b;         /* its BatchNorm as scale / shift, 8 each */
    const float *v_fc_w, *v_fc_b;         /* value_out[5]: (3, 72) row-major, (3) */
    const float *a_fc_wt, *a_fc_b;        /* aux_out[1].weight TRANSPOSED to (512 in, 512 out), bias (512) */
    const float *a_norm_w, *a_out_w;      /* aux_out[2].weight (512), aux_out[5].weight (512) */
    float board_b, pass_fc_b, a_out_b;
    float aux_to_score;                   /* aux_target_offset / score_scale (64 / 8) */
    float eps;                            /* the RMSNorms' epsilon (1e-5) */
    const void *a_fc_w16;                 /* optional: aux_out[1].weight as bf16 (512 out, 512 in) with the INPUT index in the
                                           * bottleneck's NHWC order (8 * cell + channel) - the layer then runs on the matrix
                                           * cores, 16 samples per workgroup (k_oth_heads16); NULL: fp32 from a_fc_wt */
} az_nn_othello_heads_weights;
int az_nn_othello_embed(const az_nn_positions *positions, const uint8_t *mask, const void *embed_table, void *tokens,
                        int64_t batch, const int32_t *gather, const int64_t *batch_dev, void *stream);
int az_nn_othello_heads(const void *policy_map, const void *bottleneck, const az_nn_othello_heads_weights *w, float *probs,
                        float *wdl, float *utility, int64_t batch, const int32_t *scatter, const int64_t *batch_dev, void *stream);

/* The whole Othello evaluator as one az_nn_model (kind AZ_NN_KIND_OTHELLO_CNN) for az_mcts_dev_search /
 * az_nn_model_forward_positions: embedding, the body's convolutions in order - conv[0] the stem (32 -> 256,
 * 8x8, pad 2), then pairs (conv1, conv2 with the block's input as residual), then the last body convolution;
 * conv[n_body], conv[n_body + 1] the policy stem (10x10 pad 0, 8x8 pad 1) - the bottleneck convolution, the heads. */
#define AZ_NN_OTHELLO_MAX_CONVS 16
typedef struct az_nn_othello_conv_layer {
    const void *w_packed;
    const float *pre_scale, *pre_shift, *post_scale, *post_shift;
    int32_t residual, c_in, h_in, pad;
} az_nn_othello_conv_layer;
typedef struct az_nn_othello_weights {
    const void *embed_table;
    int32_t n_body, n_convs;                                   /* n_convs = n_body + 2 */
    az_nn_othello_conv_layer conv[AZ_NN_OTHELLO_MAX_CONVS];
    const void *dual_w16;                                      /* as az_nn_othello_conv_narrow */
    const float *dual_scale16, *dual_shift16;
    az_nn_othello_heads_weights heads;
} az_nn_othello_weights;
#define AZ_NN_KIND_OTHELLO_CNN 3
int az_nn_model_create_othello(const az_nn_othello_weights *w, az_nn_model **out);

#ifdef __cplusplus
}
#endif
#endif
