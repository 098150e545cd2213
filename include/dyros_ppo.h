/* dyros_ppo.h -- C-ABI of the fused minibatch update of the on-GPU PPO consumer.
 *
 * What it replaces: the body of `calc_gradients` of the reference's learner for this task,
 *   learning/rl_games_custom/a2c_continuous_seperate.py:108-193 -- forward of the two separate MLPs under autocast, the losses of
 *   learning/rl_games_custom/common_losses.py:4-26 and models_dyros.py:59-62 (neglogp), `scaler.scale(loss).backward()`,
 *   `scaler.unscale_` of both optimisers, `clip_grad_norm_` of the actor's parameters, both `scaler.step`s and `scaler.update()`
 *   (cfg/train/DyrosDynamicWalkPPO.yaml: mlp units [256, 256], relu, mixed_precision, separate_opt, grad_norm 0.5, e_clip 0.2).
 * With torch's autograd that is ~190 kernel launches for a 4096 x 487 minibatch whose arithmetic is 10 GFLOP: launch-bound
 * (0.98 ms per update inside a hipGraph on an MI355X).  Two forms here, same arithmetic types (fp16 operands, fp32 accumulation; fp32 losses,
 * masters and moments; dynamic loss scale):
 *   four launches, the products on the matrix cores (0.05 ms per update):
 *     dwp_mlp         observations -> fp16, the three layers of both nets, loss and output gradient, the two input-gradient products, relu
 *                     masks, all bias gradients (v_mfma_f32_16x16x32_f16; weights read in fragment order: dwp_retile, kept by dwp_adam)
 *     dwp_wgrad       the three weight gradients of both nets (fp32 accumulators).  DEVIATION from autocast, towards more bits: the weight
 *                     gradients stay fp32 sums of fp16 products, where a backward under autocast (and the seventeen-launch form) rounds them to
 *                     fp16 once more and overflows at 65 504.  Consequences: found_inf fires on an overflow of the fp16 output / activation
 *                     gradients only, so with gradients near the fp16 range GradScaler backs off later than the reference would, and the
 *                     actor's clip norm is taken from unrounded gradients (differences of relative size 2^-11 per entry).
 *     dwp_grad_stats  bias gradients from their buckets, sum of squares of the actor's unscaled gradients (clip_grad_norm_), inf / nan flags
 *                     of both nets (unscale_)
 *     dwp_adam        unscale, clip (actor), Adam step on the fp32 master parameters unless the net's flag is set, fp16 copies for the
 *                     next forward (what autocast's weight cast produces)
 *     dwp_finish      moves the loss scale as GradScaler.update does, counts the steps, publishes the logged means, clears accumulators
 *                     (dwp_adam_finish: the last two in one launch -- what the four-launch form calls)
 *   seventeen launches, the eight products as library calls (torch.bmm: hipBLASLt / rocBLAS; actor and critic as one batched GEMM per layer
 *   and direction), with between them:
 *     dwp_stage_obs   fp32 observations of minibatch i -> the fp16 input matrix (autocast's cast of the Linear input)
 *     dwp_bias_relu   bias + relu of a hidden layer, in place on the batched product
 *     dwp_loss        the heads' biases, then from the two heads' outputs: neglogp, PPO ratio, the clipped surrogate, the value loss, the logged
 *                     bound loss, clip fraction and KL; d loss / d outputs times the loss scale as fp16; the heads' bias gradients
 *     dwp_relu_bwd    d relu in place on a hidden layer's gradient + that layer's bias gradient
 *   and the same dwp_grad_stats / dwp_adam / dwp_finish.
 * Sharded over the GPUs of a node (one process each) the four-launch form is mlp | wgrad | dwp_grad_bucket | ONE all-reduce of 1.61 MB (RCCL) |
 * grad_stats | adam_finish: the gradients are averaged while still scaled, before unscale_ / clip / step, as the reference's Horovod
 * optimizer.synchronize() does (a2c_continuous_seperate.py:171-180).
 * All pointers are device pointers; every function enqueues on `stream` and returns 0, or -1 with dwp_last_error() set.
 *
 * Parameter layout (fp32 masters `p`, fp16 copies `p16`, Adam moments `m`, `v`: the same layout; IN = 487 padded to INP = 512 -- a 974-byte row
 * is not even word aligned and costs the first layer's GEMMs half their speed --, HID = 256, OUTP = 16):
 *   W1 [2][HID][INP] | W2 [2][HID][HID] | W3 [2][OUTP][HID] | b1 [2][HID] | b2 [2][HID] | b3 [2][OUTP]
 * index 0 of the leading dimension is the actor, 1 the critic; the heads are padded to OUTP = 16 rows (actor: 13 action means,
 * critic: 1 value; the other rows are zero and stay zero: their gradients are zero).  Weight gradients arrive as fp16 in `g16`
 * (the weight part of the layout: what a backward under autocast produces), bias gradients as fp32 sums in `gb`
 * (b1 | b2 | b3), both still multiplied by the loss scale. */
#ifndef DYROS_PPO_H
#define DYROS_PPO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DWP_ABI_VERSION 8
#define DWP_IN    487   /* observation words (DyrosDynamicWalk.yaml numObservations)        */
#define DWP_INP   512   /* ... padded: rows of the input matrix and of W1 (zero columns), so that the GEMMs see aligned rows */
#define DWP_HID   256   /* cfg/train/DyrosDynamicWalkPPO.yaml:27 units [256, 256]            */
#define DWP_OUTP  16    /* rows of the padded heads                                          */
#define DWP_ACT   13    /* action means (the actor's head)                                   */

/* accumulators and state of one update, float words of `state` (device memory, zero-initialised by the caller once) */
#define DWP_S_ALOSS      0   /* sums over the minibatch: surrogate loss                      */
#define DWP_S_CLOSS      1   /*   value loss                                                 */
#define DWP_S_BLOSS      2   /*   bound loss                                                 */
#define DWP_S_CLIPPED    3   /*   samples with |ratio - 1| > e_clip                          */
#define DWP_S_KL         4   /*   policy KL                                                  */
#define DWP_S_NORM2      5   /* sum of squares of the actor's unscaled gradients             */
#define DWP_S_FOUND_INF  6   /* [2] unscale_'s found_inf of the actor / the critic           */
#define DWP_S_SCALE      8   /* GradScaler: the loss scale (initialise: 65536)               */
#define DWP_S_GROWTH     9   /*   growth tracker                                             */
#define DWP_S_STEP       10  /* [2] Adam step counts of the actor / the critic               */
#define DWP_S_LR         12  /* [2] learning rates (the caller's schedule writes them)       */
#define DWP_S_MB         14  /* index of the minibatch the next update takes (as a float)    */
#define DWP_S_G16        15  /* non-zero: dwp_grad_stats / dwp_adam round dwp_wgrad's summed WEIGHT gradients through fp16 first -- what a backward under
                              * autocast hands unscale_ (inf beyond 65 504, so found_inf and the loss scale move as the reference's would); 0: the
                              * fp32 sums as they are (default: more bits, found_inf from the fp16 output / activation gradients only)             */
#define DWP_S_OUT        16  /* [8] published by dwp_finish: a_loss, c_loss, b_loss, clip fraction, kl, grad norm, scale, skipped */
#define DWP_S_WORDS      32

int dwp_abi_version(void);
const char *dwp_last_error(void);

/* x16 [B][INP] (fp16) = obs[(mb * B + i)][k] for k < IN, 0 in the padding; mb = (int)state[DWP_S_MB] */
int dwp_stage_obs(const float *obs, const float *state, int32_t B, uint16_t *x16, void *stream);

/* h16 [2][B][HID] = relu(h16 + b16[net][.]) in place: bias and activation of a hidden Linear behind the bare batched product (the GEMM's
 * fp16 output, then the bias in fp32 and one more rounding: at most one fp16 ulp from a fused epilogue) */
int dwp_bias_relu(uint16_t *h16, const uint16_t *b16, int32_t B, void *stream);

/* out16 [2][B][OUTP] fp16: the heads' bare products on entry, their outputs on exit (b3_16 [2][OUTP] added: row i of [0] the action
 * means, word 0 of row i of [1] the value).  The per-sample inputs are the
 * epoch's flat arrays (row mb * B + i is used): act [.][ACT], old_nlp [.], old_mu [.][ACT], adv [.], ret [.].  logstd [ACT]: the
 * fixed log sigma.  dout16 [2][B][OUTP] = scale * d loss / d out16, loss = mean(surrogate) + 0.5 * critic_coef * mean((ret - v)^2)
 * (entropy and bound loss have coefficient 0 in this configuration and are only logged).  Adds the heads' bias gradients to
 * gb[2 * HID * 2 ..] and the logged sums to state. */
int dwp_loss(uint16_t *out16, const uint16_t *b3_16, const float *act, const float *old_nlp, const float *old_mu, const float *adv, const float *ret,
             const float *logstd, float *state, float *gb, int32_t B, float e_clip, float critic_coef, uint16_t *dout16, void *stream);

/* dh16 [2][B][HID] *= (h16 > 0), and gb_layer [2][HID] += column sums of the result (fp32) */
int dwp_relu_bwd(const uint16_t *h16, uint16_t *dh16, float *gb_layer, int32_t B, void *stream);

#define DWP_PARTS 2048  /* words of `part`: [0,256) sums of squares, [256,512) inf / nan flags, [512,520) scale, steps, learning rates, the G16 switch;
                         * dwp_stats_adam_finish only: [641] the update's number, [642] set once a wait timed out, [1024,1536) the blocks' 64-bit shares */
#define DWP_P16F_WORDS 548864   /* halves of p16f, the weights once more in the order dwp_mlp's matrix instructions take them (csrc/dw_ppo.hip frag_pos) */
#define DWP_P32F_WORDS 401408   /* floats of p32f, the fp32 weights in the order dwp_policy's matrix instructions take them (csrc/dw_ppo.hip frag32_pos) */
#define DWP_WGRAD_SLABS 4    /* dwp_wgrad splits the samples into this many slabs: g32 is [DWP_WGRAD_SLABS][weights] partial gradients */
#define DWP_PBUF_WORDS 544   /* words of a row of dwp_mlp's accumulators */
#define DWP_PBUF_BUCKETS 32  /* rows per net: pbuf is [DWP_PBUF_BUCKETS][2][DWP_PBUF_WORDS] floats, zero-initialised by the caller once */
#define DWP_ROLL_TERMS_MAX 64  /* logged reward columns dwp_rollout_post can reduce (15 on the plane, 15 + terrain types with a curriculum) */
/* part[0 .. 256) = partial sums over the actor's parameters of (g / scale)^2; state[FOUND_INF + net] = 1 where a gradient of
 * that net is not finite (also per block in part[256 .. 512): bit `net`); part[512 ..] = state's SCALE, STEP[2], LR[2] as they are now
 * (what dwp_adam_finish's blocks read instead of `state`).  pbuf (or NULL): dwp_mlp's accumulators: the bias gradients are their sums over the buckets (cleared here)
 * and are left in gb for dwp_adam (without it gb holds them already: dwp_loss / dwp_relu_bwd).
 * g32 (or NULL): the weight gradients are the sums over dwp_wgrad's partial gradients [DWP_WGRAD_SLABS][weights] instead of g16;
 * g32_slabs: DWP_WGRAD_SLABS for that, or 1: g32 is ONE array of weight gradients (dwp_grad_bucket's bucket after the ranks' all-reduce;
 * gb then points at its bias part and pbuf is NULL) */
int dwp_grad_stats(const uint16_t *g16, float *gb, float *state, float *part, float *pbuf, const float *g32, int32_t g32_slabs, void *stream);

/* Sharded training, one process per GPU: this rank's still-scaled gradient as ONE contiguous bucket [weights | biases] (the parameter order
 * of p: 401 408 + 1 056 floats = 1.61 MB) times inv_world = 1 / world, so that ONE all-reduce (sum) of the bucket over the ranks leaves the
 * average in it -- Horovod's optimizer.synchronize() of the reference, which runs before unscale_ / clip_grad_norm_ / step
 * (learning/rl_games_custom/a2c_continuous_seperate.py:171-180, a2c_common_dyros.py:980-981).  Weights: the sum of dwp_wgrad's slabs in
 * dwp_grad_stats' order; biases: the sums over pbuf's buckets, which are cleared as dwp_grad_stats would.  The update then goes on with
 * dwp_grad_stats(NULL, bucket + weights, state, part, NULL, bucket, 1) and dwp_adam_finish(..., gb = bucket + weights, g32 = bucket, 1, ...).
 * With world a power of two the scaling is exact, so a rank alone (world 1) computes the bits of the unsharded update. */
int dwp_grad_bucket(const float *g32, float *pbuf, float *bucket, float inv_world, void *stream);

/* the Adam step of torch.optim.Adam(fused, capturable; betas (0.9, 0.999), eps 1e-8, no weight decay) behind GradScaler.step, with
 * clip_grad_norm_(actor, max_norm) applied to the actor's unscaled gradients first (norm^2 = the sum of `part`, published in
 * state[NORM2]).  p16f (or NULL): the fragment-order fp16 copy of the weights that dwp_mlp reads (DWP_P16F_WORDS halves, zero-initialised
 * by the caller and filled once with dwp_retile).  g32 (or NULL): as dwp_grad_stats.
 * p32f (or NULL): the fp32 fragment-order copy of the weights that dwp_policy reads (DWP_P32F_WORDS floats; filled once with dwp_retile32) */
int dwp_adam(float *p, uint16_t *p16, float *m, float *v, const uint16_t *g16, const float *gb, float *state, const float *part, float max_norm,
             uint16_t *p16f, const float *g32, int32_t g32_slabs, float *p32f, void *stream);

/* dwp_adam (weight gradients from g32) and dwp_finish in one launch: every block takes the loss scale, step counts, learning rates and
 * flags from `part` as dwp_grad_stats left them, block 0 does dwp_finish's work on `state` meanwhile.  gb is not cleared (with pbuf
 * dwp_grad_stats overwrites it) */
int dwp_adam_finish(float *p, uint16_t *p16, float *m, float *v, const float *gb, float *state, const float *part, float max_norm, uint16_t *p16f,
                    const float *g32, int32_t g32_slabs, float *p32f, int32_t B, int32_t num_minibatches, int32_t growth_interval, float *pbuf, void *stream);

/* dwp_grad_stats and dwp_adam_finish in ONE launch (ABI 8): every Adam block keeps its eight parameters' gradients in registers, publishes its share of
 * the actor's norm and its inf / nan flags as one tagged 64-bit word, and every block waits for all 197 shares before the first parameter moves (198
 * blocks of 256 threads: co-resident whenever the launch has the device to itself).  The wait is bounded: a thread that does not see its share within
 * 32 768 polls takes both nets as non-finite (its block skips), sets part[642] and goes on, so the grid always drains; DWP_S_OUT[7] then reads 2 and the
 * caller should treat the update as failed (part[642] stays set).  part: as dwp_grad_stats, plus words [640, 643) and [1024, 1536) that the caller
 * zero-initialises with the rest and never touches.  pbuf_bias: dwp_mlp's accumulators if the bias gradients are still in their buckets (the plain form:
 * = pbuf), or NULL if gb holds them already (the sharded form: gb = bucket + weights).  Same arithmetic as the two launches except the ORDER of the
 * norm's partial sums (per Adam block here): norm2 and the actor's clipped step may differ from theirs in the last place; the critic's step, the flags
 * and the scaler are the same bits. */
int dwp_stats_adam_finish(float *p, uint16_t *p16, float *m, float *v, float *gb, float *state, float *part, float max_norm, uint16_t *p16f, const float *g32,
                          int32_t g32_slabs, float *p32f, int32_t B, int32_t num_minibatches, int32_t growth_interval, float *pbuf, float *pbuf_bias, void *stream);

/* GradScaler.update (growth 2.0 every growth_interval clean updates, backoff 0.5), step counts, logged means (divided by B),
 * accumulators and gb cleared, minibatch index advanced modulo num_minibatches.  pbuf (or NULL): dwp_mlp's accumulators, whose logged-sum
 * words are added to the logged sums first (and cleared) */
int dwp_finish(float *state, float *gb, int32_t B, int32_t num_minibatches, int32_t growth_interval, float *pbuf, void *stream);

/* dwp_stage_obs + the three layers of both nets + dwp_loss + the two input-gradient products with their relu masks and all bias
 * gradients, in ONE launch on the matrix cores (v_mfma_f32_16x16x32_f16): a workgroup of EIGHT wavefronts takes 32 samples through one net
 * (every product split eight ways by columns: two waves on every SIMD of the CU), activations
 * in LDS, weights from the fragment-order fp16 copy (resident in L2; every request of a wave is one contiguous KB).  What is left of an update after it: the three weight-gradient GEMMs
 * (dout' h2, dz2' h1, dz1' x16: library calls), dwp_grad_stats, dwp_adam, dwp_finish.  Buffers as the other entry points name them;
 * pbuf: the accumulators of the bias gradients and the logged sums (DWP_PBUF_*: a wave adds into the row of its bucket), read and cleared
 * by dwp_grad_stats and dwp_finish.  B: a multiple of 32. */
/* GAE of one rollout, `discount_values` of learning/rl_games_custom/a2c_common_dyros.py:485-500 (a Python loop of H steps there): advs [H][N] from
 * fdones [N] (the dones after the last step), last_values [N], mb_fdones [H][N], mb_values [H][N], mb_rewards [H][N]; per element the same fp32
 * operations in the same order.  (Note the reference's indexing: step t uses mb_fdones[t + 1] and mb_values[t + 1], the last step fdones and
 * last_values.) */
int dwp_gae(const float *fdones, const float *last_values, const float *mb_fdones, const float *mb_values, const float *mb_rewards, float gamma, float tau, int32_t H,
            int32_t N, float *advs, void *stream);

/* The rollout's bookkeeping around the env step, `play_steps` of learning/rl_games_custom/a2c_common_dyros.py:629-703, in two launches (the torch
 * form is ~30 small kernels per step).  n: device int64, the step's row of the rollout buffers (the caller advances it).
 * dwp_rollout_pre: a = mu + exp(logstd) * noise (noise: the caller's standard-normal draws [N][ACT]); row n of mb_obs [H][N][num_obs], mb_act, mb_mu
 *   [H][N][ACT], mb_nlp (neglogp of a, models_dyros.py:59-62), mb_val, mb_done [H][N]; act [N][ACT] = clamp(a, -1, 1) for the env.
 *   env_major_steps = H > 0: mb_obs is [N][H][num_obs] instead -- the env-major flat batch the update reads (swap_and_flatten01,
 *   a2c_common_dyros.py:1080, done while the rollout runs: no 4 GB transpose per epoch at 16384 envs).  obs_half != 0 (with env_major_steps):
 *   mb_obs points to HALVES [N][H][DWP_INP], zero-initialised by the caller once: the observations as the update's first layer takes them
 *   (autocast's cast of the Linear input), for DwpMlp.obs16.
 * dwp_rollout_post: mb_rew[n] = rew * reward_scale (+ gamma * value * time_outs: the bootstrap of :656-659; time_outs NULL = off); terms[c] += mean over
 *   the envs of stacked[.][c], c < num_terms <= DWP_ROLL_TERMS_MAX (terms NULL = off); g_dones = float(done_buf); g_obs = new_obs (skipped when they are one buffer).
 * H: the rows of the rollout buffers.  The row counter n is device memory that a replayed graph advances; a step whose n is outside [0, H)
 *   -- a caller that did not rewind it -- writes NO row of any buffer (the env still gets its clipped action, g_dones / g_obs are still kept),
 *   so a forgotten rewind costs recorded samples, never memory outside the buffers. */
int dwp_rollout_pre(const float *mu, const float *value, const float *noise, const float *obs, const float *dones, const float *logstd, const int64_t *n, int32_t N,
                    int32_t num_obs, float *mb_obs, float *mb_act, float *mb_mu, float *mb_nlp, float *mb_val, float *mb_done, float *act, int32_t env_major_steps,
                    int32_t obs_half, int32_t H, void *stream);
int dwp_rollout_post(const float *rew, const float *value, const int64_t *time_outs, const float *stacked, int32_t stacked_cols, const int64_t *done_buf, const float *new_obs,
                     const int64_t *n, int32_t N, int32_t num_obs, float reward_scale, float gamma, float *mb_rew, float *terms, int32_t num_terms, float *g_dones,
                     float *g_obs, int32_t H, void *stream);

/* The rollout's policy forward, get_action_values of the reference (fp32: no autocast there): mu [N][ACT] and value [N] of both nets for obs [N][IN], on
 * v_mfma_f32_16x16x4_f32 (the library's fp32 GEMMs take 0.25 ms of a 0.39 ms rollout step at 16384 envs).  p: the fp32 masters (biases), p32f: the weights
 * in operand order (dwp_retile32 once, then kept by dwp_adam).  Any N >= 1. */
int dwp_policy(const float *obs, const float *p, const float *p32f, int32_t N, float *mu, float *value, void *stream);
int dwp_retile32(const float *p, float *p32f, void *stream);

/* p16f from p16 (all weights; after construction or after loading parameters) */
int dwp_retile(const uint16_t *p16, uint16_t *p16f, void *stream);

typedef struct DwpMlp {
    const float *obs, *state, *act, *old_nlp, *old_mu, *adv, *ret, *logstd;
    const uint16_t *obs16;               /* used when obs is NULL: the batch's observations as fp16 rows [batch][DWP_INP], columns 487.. zero */
    const uint16_t *p16, *p16t;          /* p16t: the fragment-order copy (p16f of dwp_adam / dwp_retile) */
    float *pbuf;
    uint16_t *x16, *h1, *h2, *out16, *dout16, *dz2, *dz1;
    uint16_t *xf, *h1f, *h2f, *doutf, *dz2f, *dz1f;          /* (or all NULL) the same once more as operands of dwp_wgrad: xf [B * INP], h1f .. dz1f [2][B * HID], doutf [2][B * OUTP] */
    int32_t B;
    float e_clip, critic_coef;
} DwpMlp;
int dwp_mlp(const DwpMlp *a, void *stream);
int dwp_sizeof_mlp(void);          /* sizeof(DwpMlp) as the library was built: a binding checks its own mirror of the struct against it */

/* the three weight gradients of both nets from dwp_mlp's operand-order copies, on the matrix cores: g32 [DWP_WGRAD_SLABS][weights], fp32, the
 * parameter layout's weight part once per slab of samples -- every word is written by exactly one wave (no atomics, nothing to clear); the
 * gradient is the sum over the slabs, formed by dwp_grad_stats / dwp_adam as they read */
int dwp_wgrad(const uint16_t *xf, const uint16_t *h1f, const uint16_t *h2f, const uint16_t *doutf, const uint16_t *dz2f, const uint16_t *dz1f, const float *state,
              float *g32, int32_t B, void *stream);

#ifdef __cplusplus
}
#endif
#endif
