// dw_ppo.hip -- the five kernels between the GEMMs of one PPO minibatch update (include/dyros_ppo.h; reference:
// learning/rl_games_custom/a2c_continuous_seperate.py:108-193, common_losses.py:4-26, models_dyros.py:59-62).  gfx950; fp16 storage as
// _Float16, all arithmetic in fp32.  These are HBM-trivial kernels (the largest touches 12 MB): what they buy is launch count -- a
// replayed hipGraph node costs ~5 us on an MI355X whatever it does, and torch's autograd needs ~190 of them per update.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>

#include "../../include/dyros_ppo.h"

namespace {

char g_err[256] = "";
int fail_hip(const char *who, hipError_t e) { snprintf(g_err, sizeof(g_err), "%s: %s", who, hipGetErrorString(e)); return -1; }
int fail(const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg); return -1; }

constexpr int IN = DWP_IN, INP = DWP_INP, HID = DWP_HID, OUTP = DWP_OUTP, ACT = DWP_ACT;
constexpr int NW1 = 2 * HID * INP, NW2 = 2 * HID * HID, NW3 = 2 * OUTP * HID, NWT = NW1 + NW2 + NW3;
constexpr int NB1 = 2 * HID, NB2 = 2 * HID, NB3 = 2 * OUTP, NBT = NB1 + NB2 + NB3, NP = NWT + NBT;

// which net (0 actor, 1 critic) owns element i of the parameter layout
__device__ __forceinline__ int net_of(int i) {
    if (i < NW1) return i >= NW1 / 2;
    i -= NW1;
    if (i < NW2) return i >= NW2 / 2;
    i -= NW2;
    if (i < NW3) return i >= NW3 / 2;
    i -= NW3;
    if (i < NB1) return i >= NB1 / 2;
    i -= NB1;
    if (i < NB2) return i >= NB2 / 2;
    i -= NB2;
    return i >= NB3 / 2;
}
__device__ __forceinline__ float wave_sum(float x) {
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

__global__ __launch_bounds__(256) void k_stage_obs(const float *__restrict__ obs, const float *__restrict__ state, int B, _Float16 *__restrict__ x16) {
    // one thread per pair of output words: consecutive lanes read consecutive floats of a row (rows are 487 words: no wider aligned load exists)
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const size_t base = (size_t)(int)state[DWP_S_MB] * B * IN;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)B * (INP / 2)) return;
    const size_t row = t / (INP / 2);
    const int c0 = (int)(t % (INP / 2)) * 2;
    h2 v;
    v[0] = c0 < IN ? (_Float16)obs[base + row * IN + c0] : (_Float16)0.0f;
    v[1] = c0 + 1 < IN ? (_Float16)obs[base + row * IN + c0 + 1] : (_Float16)0.0f;
    *reinterpret_cast<h2 *>(x16 + row * INP + c0) = v;
}

// 16 lanes per sample (lane k < 13: action k; lane 0 also the value), 64 samples per block (few blocks: every block ends in 19 atomic adds
// on the same 19 words, and same-address atomics are served one after the other).  Forward (fp32, as the ops autocast keeps
// in fp32 see it): the heads' biases are added here (out16 arrives as the bare product and leaves as the Linear's fp16 output),
// neglogp of the stored action under (mu, sigma), ratio = exp(old - new), surrogate = max(-A ratio, -A clamp(ratio, 1 - e, 1 + e)),
// value loss (ret - v)^2, and the logged terms.  Backward: d/d mu_k = [unclipped branch active] * A * ratio * (-(a_k - mu_k) / sigma_k^2) / B
// (torch.maximum gives both branches half the gradient on a tie, and inside the clip range the branches are the same function: the
// sum is the whole gradient); d/d v = critic_coef * (v - ret) / B (0.5 * critic_coef * the mean's 2 (v - ret) / B).
__device__ __forceinline__ float sum16(float x) {
    for (int o = 8; o >= 1; o >>= 1) x += __shfl_xor(x, o, 16);
    return x;
}
__global__ __launch_bounds__(1024) void k_loss(_Float16 *__restrict__ out16, const _Float16 *__restrict__ b3, const float *__restrict__ act,
                                              const float *__restrict__ old_nlp, const float *__restrict__ old_mu, const float *__restrict__ adv,
                                              const float *__restrict__ ret, const float *__restrict__ logstd, float *__restrict__ state, float *__restrict__ gb,
                                              int B, float e_clip, float critic_coef, _Float16 *__restrict__ dout16) {
    __shared__ float red[16][5 + OUTP + 1];
    const int k = threadIdx.x & 15, i = blockIdx.x * 64 + (threadIdx.x >> 4);
    const bool on = i < B, ak = k < ACT;
    const int r = on ? i : 0;
    const size_t row = (size_t)(int)state[DWP_S_MB] * B + r;
    const float scale = state[DWP_S_SCALE], invB = 1.0f / (float)B;
    // the two heads' outputs with their biases (rounded to fp16 as the Linear's epilogue would), written back
    const _Float16 mu16 = (_Float16)((float)out16[(size_t)r * OUTP + k] + (ak ? (float)b3[k] : 0.0f));
    const _Float16 v16 = (_Float16)((float)out16[((size_t)B + r) * OUTP] + (float)b3[OUTP]);
    if (on) { out16[(size_t)r * OUTP + k] = ak ? mu16 : (_Float16)0.0f; if (k == 0) out16[((size_t)B + r) * OUTP] = v16; }
    const float mu = (float)mu16, a = ak ? act[row * ACT + k] : 0.0f, ls = ak ? logstd[k] : 0.0f, om = ak ? old_mu[row * ACT + k] : 0.0f;
    const float sg = expf(ls), z = (a - mu) / sg, s2 = sg * sg;
    const float hi = fminf(mu - 1.1f, 0.0f), lo = fminf(-mu + 1.1f, 0.0f);          // a2c_continuous_seperate.py:233-241 as written there
    const float sq = sum16(ak ? z * z : 0.0f), lsum = sum16(ls), bl = sum16(ak ? lo * lo + hi * hi : 0.0f);
    const float kl = sum16(ak ? logf(sg / sg + 1e-5f) + (s2 + (om - mu) * (om - mu)) / (2.0f * (s2 + 1e-5f)) - 0.5f : 0.0f);
    const float nlp = 0.5f * sq + 0.5f * 1.8378770664093453f * (float)ACT + lsum;          // log(2 pi)
    const float A = adv[row], ratio = expf(old_nlp[row] - nlp);
    const float rc = fminf(fmaxf(ratio, 1.0f - e_clip), 1.0f + e_clip);
    const float s1 = -A * ratio, sc = -A * rc;
    const float al = fmaxf(s1, sc);
    const bool inside = ratio >= 1.0f - e_clip && ratio <= 1.0f + e_clip;          // (clamp passes the gradient on its closed range)
    // gradient through the first branch: all of it if it is the larger, half on a tie; through the second: the same function of
    // ratio inside the range (the other half on a tie, all of it if it is the larger), nothing outside
    float w = s1 > sc ? 1.0f : (s1 == sc ? 0.5f : 0.0f);
    if (inside) w += sc > s1 ? 1.0f : (s1 == sc ? 0.5f : 0.0f);
    const float dnlp = A * ratio * w;          // d al / d nlp
    const float v = (float)v16, rt = ret[row];
    const float cl = (rt - v) * (rt - v);
    const _Float16 dmu16 = (_Float16)(on && ak ? scale * invB * dnlp * (-(a - mu) / s2) : 0.0f);
    const _Float16 dv16 = (_Float16)(on && k == 0 ? scale * invB * critic_coef * (v - rt) : 0.0f);
    if (on) { dout16[(size_t)r * OUTP + k] = dmu16; dout16[((size_t)B + r) * OUTP + k] = dv16; }
    // sums over the block's 16 samples: five logged terms (lane 0 of a sample carries them), the 13 + 1 head bias gradients (from the
    // rounded fp16 gradients, as a sum over grad_output rows is)
    const bool first = on && k == 0;
    float st[5] = {first ? al : 0.0f, first ? cl : 0.0f, first ? bl : 0.0f, first && fabsf(ratio - 1.0f) > e_clip ? 1.0f : 0.0f, first ? kl : 0.0f};
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int q = 0; q < 5; ++q) { const float s = wave_sum(st[q]); if (lane == 0) red[wv][q] = s; }
    float gm = (float)dmu16, gv = (float)dv16;
    gm += __shfl_xor(gm, 16, 64); gm += __shfl_xor(gm, 32, 64);          // the wave's four samples, per column
    gv = wave_sum(gv);
    if (lane < OUTP) red[wv][5 + lane] = gm;
    if (lane == 0) red[wv][5 + OUTP] = gv;
    __syncthreads();
    if (threadIdx.x < 5 + OUTP + 1) {
        const int q = threadIdx.x;
        float s = 0.0f;
        for (int w_ = 0; w_ < 16; ++w_) s += red[w_][q];
        float *gb3 = gb + NB1 + NB2;
        if (q < 5) atomicAdd(&state[q], s);
        else if (q < 5 + ACT) atomicAdd(&gb3[q - 5], s);
        else if (q == 5 + OUTP) atomicAdd(&gb3[OUTP], s);
    }
}

// h16 [2][B][HID] = relu(h16 + b16[net]) in place: the bias and activation of a hidden Linear behind the bare batched product
__global__ __launch_bounds__(256) void k_bias_relu(_Float16 *__restrict__ h16, const _Float16 *__restrict__ b16, int B) {
    const size_t n = (size_t)2 * B * HID, i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    const int net = i >= (size_t)B * HID, col = (int)(i % HID);
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    h8 x = *reinterpret_cast<const h8 *>(h16 + i);
    const h8 b = *reinterpret_cast<const h8 *>(b16 + net * HID + col);
    for (int q = 0; q < 8; ++q) { const _Float16 y = (_Float16)((float)x[q] + (float)b[q]); x[q] = (float)y > 0.0f ? y : (_Float16)0.0f; }
    *reinterpret_cast<h8 *>(h16 + i) = x;
}

// block = 32 column groups of 8 (one 16-byte load each) x 8 row lanes; RB_ROWS rows per block; column sums over the row lanes in LDS
constexpr int RB_ROWS = 64;
__global__ __launch_bounds__(256) void k_relu_bwd(const _Float16 *__restrict__ h16, _Float16 *__restrict__ dh16, float *__restrict__ gb_layer, int B) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    __shared__ float red[8][HID];
    const int net = blockIdx.y, cg = threadIdx.x & 31, rl = threadIdx.x >> 5, r0 = blockIdx.x * RB_ROWS;
    float sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = r0 + rl; r < r0 + RB_ROWS && r < B; r += 8) {
        const size_t idx = ((size_t)net * B + r) * HID + cg * 8;
        const h8 h = *reinterpret_cast<const h8 *>(h16 + idx);
        h8 d = *reinterpret_cast<const h8 *>(dh16 + idx);
        for (int q = 0; q < 8; ++q) { if (!((float)h[q] > 0.0f)) d[q] = (_Float16)0.0f; sum[q] += (float)d[q]; }
        *reinterpret_cast<h8 *>(dh16 + idx) = d;
    }
    for (int q = 0; q < 8; ++q) red[rl][cg * 8 + q] = sum[q];
    __syncthreads();
    const int col = threadIdx.x;
    float s_ = 0.0f;
    for (int q = 0; q < 8; ++q) s_ += red[q][col];
    atomicAdd(&gb_layer[net * HID + col], s_);
}

__device__ __forceinline__ float scaled_grad(const _Float16 *g16, const float *gb, int i) { return i < NWT ? (float)g16[i] : gb[i - NWT]; }

constexpr int GS_BLOCKS = 256;          // partial sums of squares, one per block, in `part`; dwp_adam's blocks add them up (no atomics: 256
                                        // adds on one word are served one after the other and were most of this kernel's 12 us)
__global__ __launch_bounds__(256) void k_grad_stats(const _Float16 *__restrict__ g16, const float *__restrict__ gb, float *__restrict__ state, float *__restrict__ part) {
    __shared__ float red[4];
    const float inv = 1.0f / state[DWP_S_SCALE];
    float sq = 0.0f;
    int bad0 = 0, bad1 = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < NP; i += GS_BLOCKS * 256) {
        const float g = scaled_grad(g16, gb, i);
        const int net = net_of(i);
        if (!isfinite(g)) { if (net) bad1 = 1; else bad0 = 1; }
        const float u = g * inv;
        if (net == 0) sq += u * u;
    }
    sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    if (bad0) state[DWP_S_FOUND_INF] = 1.0f;
    if (bad1) state[DWP_S_FOUND_INF + 1] = 1.0f;
}

__global__ __launch_bounds__(256) void k_adam(float *__restrict__ p, _Float16 *__restrict__ p16, float *__restrict__ m, float *__restrict__ v,
                                              const _Float16 *__restrict__ g16, const float *__restrict__ gb, float *__restrict__ state, const float *__restrict__ part,
                                              float max_norm) {
    __shared__ float red[4];
    static_assert(GS_BLOCKS == 256, "one partial per thread");
    {   // the actor's gradient norm from dwp_grad_stats' partial sums (every block adds them up the same way; block 0 publishes it)
        const float s = wave_sum(part[threadIdx.x]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
    }
    const float norm2 = red[0] + red[1] + red[2] + red[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) state[DWP_S_NORM2] = norm2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NP) return;
    const int net = net_of(i);
    if (state[DWP_S_FOUND_INF + net] != 0.0f) return;          // GradScaler.step: this optimiser's step is skipped
    float g = scaled_grad(g16, gb, i) * (1.0f / state[DWP_S_SCALE]);
    if (net == 0) {
        const float coef = max_norm / (sqrtf(norm2) + 1e-6f);          // torch.nn.utils.clip_grad_norm_
        g *= fminf(coef, 1.0f);
    }
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    const float step = state[DWP_S_STEP + net] + 1.0f, lr = state[DWP_S_LR + net];
    const float mi = m[i] + (g - m[i]) * (1.0f - b1);          // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = b2 * v[i] + (1.0f - b2) * g * g;
    m[i] = mi; v[i] = vi;
    const float bc1 = 1.0f - powf(b1, step), bc2 = 1.0f - powf(b2, step);
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    const float pi = p[i] - (lr / bc1) * (mi / denom);
    p[i] = pi;
    p16[i] = (_Float16)pi;
}

__global__ __launch_bounds__(256) void k_finish(float *__restrict__ state, float *__restrict__ gb, int B, int nmb, int growth_interval) {
    for (int i = threadIdx.x; i < NBT; i += 256) gb[i] = 0.0f;
    if (threadIdx.x != 0) return;
    const float invB = 1.0f / (float)B;
    const bool f0 = state[DWP_S_FOUND_INF] != 0.0f, f1 = state[DWP_S_FOUND_INF + 1] != 0.0f;
    float *o = state + DWP_S_OUT;
    o[0] = state[DWP_S_ALOSS] * invB; o[1] = state[DWP_S_CLOSS] * invB; o[2] = state[DWP_S_BLOSS] * invB; o[3] = state[DWP_S_CLIPPED] * invB;
    o[4] = state[DWP_S_KL] * invB; o[5] = sqrtf(state[DWP_S_NORM2]); o[6] = state[DWP_S_SCALE]; o[7] = (f0 || f1) ? 1.0f : 0.0f;
    // torch.amp.GradScaler.update (_amp_update_scale_): backoff 0.5 on any inf, growth 2.0 after growth_interval clean updates
    if (f0 || f1) { state[DWP_S_SCALE] *= 0.5f; state[DWP_S_GROWTH] = 0.0f; }
    else {
        const float t = state[DWP_S_GROWTH] + 1.0f;
        if ((int)t == growth_interval) { state[DWP_S_SCALE] *= 2.0f; state[DWP_S_GROWTH] = 0.0f; }
        else state[DWP_S_GROWTH] = t;
    }
    if (!f0) state[DWP_S_STEP] += 1.0f;
    if (!f1) state[DWP_S_STEP + 1] += 1.0f;
    for (int k = 0; k < 8; ++k) state[k] = 0.0f;
    const int mb = (int)state[DWP_S_MB] + 1;
    state[DWP_S_MB] = (float)(mb >= nmb ? 0 : mb);
}

int done(const char *who) {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail_hip(who, e);
}

}  // namespace

extern "C" {

int dwp_abi_version(void) { return DWP_ABI_VERSION; }
const char *dwp_last_error(void) { return g_err; }

int dwp_stage_obs(const float *obs, const float *state, int32_t B, uint16_t *x16, void *stream) {
    if (!obs || !state || !x16 || B < 1) return fail("dwp_stage_obs: bad argument");
    const size_t n = (size_t)B * (INP / 2);
    hipLaunchKernelGGL(k_stage_obs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, obs, state, B, (_Float16 *)x16);
    return done("dwp_stage_obs");
}

int dwp_loss(uint16_t *out16, const uint16_t *b3_16, const float *act, const float *old_nlp, const float *old_mu, const float *adv, const float *ret,
             const float *logstd, float *state, float *gb, int32_t B, float e_clip, float critic_coef, uint16_t *dout16, void *stream) {
    if (!out16 || !b3_16 || !act || !old_nlp || !old_mu || !adv || !ret || !logstd || !state || !gb || !dout16 || B < 1) return fail("dwp_loss: bad argument");
    hipLaunchKernelGGL(k_loss, dim3((B + 63) / 64), dim3(1024), 0, (hipStream_t)stream, (_Float16 *)out16, (const _Float16 *)b3_16, act, old_nlp, old_mu, adv, ret, logstd,
                       state, gb, B, e_clip, critic_coef, (_Float16 *)dout16);
    return done("dwp_loss");
}

int dwp_bias_relu(uint16_t *h16, const uint16_t *b16, int32_t B, void *stream) {
    if (!h16 || !b16 || B < 1) return fail("dwp_bias_relu: bad argument");
    const size_t n = (size_t)2 * B * HID;
    hipLaunchKernelGGL(k_bias_relu, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (_Float16 *)h16, (const _Float16 *)b16, B);
    return done("dwp_bias_relu");
}

int dwp_relu_bwd(const uint16_t *h16, uint16_t *dh16, float *gb_layer, int32_t B, void *stream) {
    if (!h16 || !dh16 || !gb_layer || B < 1) return fail("dwp_relu_bwd: bad argument");
    hipLaunchKernelGGL(k_relu_bwd, dim3((B + RB_ROWS - 1) / RB_ROWS, 2), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)h16, (_Float16 *)dh16, gb_layer, B);
    return done("dwp_relu_bwd");
}

int dwp_grad_stats(const uint16_t *g16, const float *gb, float *state, float *part, void *stream) {
    if (!g16 || !gb || !state || !part) return fail("dwp_grad_stats: bad argument");
    hipLaunchKernelGGL(k_grad_stats, dim3(GS_BLOCKS), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)g16, gb, state, part);
    return done("dwp_grad_stats");
}

int dwp_adam(float *p, uint16_t *p16, float *m, float *v, const uint16_t *g16, const float *gb, float *state, const float *part, float max_norm, void *stream) {
    if (!p || !p16 || !m || !v || !g16 || !gb || !state || !part) return fail("dwp_adam: bad argument");
    hipLaunchKernelGGL(k_adam, dim3((NP + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, (_Float16 *)p16, m, v, (const _Float16 *)g16, gb, state, part, max_norm);
    return done("dwp_adam");
}

int dwp_finish(float *state, float *gb, int32_t B, int32_t num_minibatches, int32_t growth_interval, void *stream) {
    if (!state || !gb || B < 1 || num_minibatches < 1 || growth_interval < 1) return fail("dwp_finish: bad argument");
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(256), 0, (hipStream_t)stream, state, gb, B, num_minibatches, growth_interval);
    return done("dwp_finish");
}

}  // extern "C"
