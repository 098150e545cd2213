// dw_ppo.hip -- one PPO minibatch update of the DyrosDynamicWalk configuration (include/dyros_ppo.h; reference:
// learning/rl_games_custom/a2c_continuous_seperate.py:108-193, common_losses.py:4-26, models_dyros.py:59-62).  gfx950; fp16 storage as
// _Float16, products on v_mfma_f32_16x16x32_f16 with fp32 accumulation, everything else in fp32.  Two forms: k_mlp + k_wgrad (the whole
// forward / backward in two launches on the matrix cores) or the small kernels that sit between library GEMMs; both end in k_grad_stats,
// k_adam, k_finish (the first form: k_adam<true>, which finishes the update too).  The update is launch- and latency-bound, not bandwidth-bound (10 GFLOP, 30 MB): a replayed hipGraph node costs 4-5 us
// on an MI355X whatever it does, and torch's autograd needs ~190 of them.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>

#include "../../include/dyros_ppo.h"

namespace {

char g_err[256] = "";
int fail_hip(const char *who, hipError_t e) { snprintf(g_err, sizeof(g_err), "%s: %s", who, hipGetErrorString(e)); return -1; }
int fail(const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg); return -1; }

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int IN = DWP_IN, INP = DWP_INP, HID = DWP_HID, OUTP = DWP_OUTP, ACT = DWP_ACT;
constexpr int NW1 = 2 * HID * INP, NW2 = 2 * HID * HID, NW3 = 2 * OUTP * HID, NWT = NW1 + NW2 + NW3;
constexpr int NB1 = 2 * HID, NB2 = 2 * HID, NB3 = 2 * OUTP, NBT = NB1 + NB2 + NB3, NP = NWT + NBT;
constexpr int PBK = DWP_PBUF_BUCKETS, PBW = DWP_PBUF_WORDS, PB_B1 = 0, PB_B2 = HID, PB_B3 = 2 * HID, PB_ST = 2 * HID + OUTP;          // a row of dwp_mlp's partial sums (k_mlp)
static_assert(PB_ST + 5 <= PBW, "row of partial sums");

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
// dwp_mlp reads its weights in FRAGMENT order: the eight halves lane l of a wave feeds to v_mfma_f32_16x16x32_f16 as B for k-step kk and
// column tile nt -- W[row 16 nt + (l & 15)][k = 32 kk + 8 (l >> 4) + j] -- are 16 consecutive bytes at ((kk * NT + nt) * 64 + l) * 16, so a
// wave's request is ONE contiguous KB and successive requests walk through memory.  (Read from the row-major copy a request touched 16
// rows a KB apart -- 16 half-used cache lines on 4 of a die's 16 channels -- and the kernel waited for L2 70 % of its time.)
// Layout of p16f (halves): W1 | W2 | W3 as forward operands (row = output, k = input), then W2 | W3 as input-gradient operands (row = input,
// k = output; the head's k padded 16 -> 32 with zeros).  dwp_adam writes every weight to its one or two places; dwp_retile fills the lot.
constexpr int F_W1 = 0, F_W2 = F_W1 + NW1, F_W3 = F_W2 + NW2, F_W2T = F_W3 + NW3, F_W3T = F_W2T + NW2, F_END = F_W3T + 2 * HID * 32;
static_assert(F_END == DWP_P16F_WORDS, "include/dyros_ppo.h");
__device__ __forceinline__ int frag_pos(int nt_count, int row, int k) { return ((((k >> 5) * nt_count + (row >> 4)) * 64 + ((k & 31) >> 3) * 16 + (row & 15)) << 3) + (k & 7); }
__device__ __forceinline__ void write_frags(_Float16 *__restrict__ p16f, int i, _Float16 v) {          // i: index of a WEIGHT in the parameter layout
    if (i < NW1) { const int net = i / (HID * INP), o = (i / INP) % HID, k = i % INP; p16f[F_W1 + net * HID * INP + frag_pos(HID / 16, o, k)] = v; return; }
    i -= NW1;
    if (i < NW2) {
        const int net = i / (HID * HID), o = (i / HID) % HID, k = i % HID;
        p16f[F_W2 + net * HID * HID + frag_pos(HID / 16, o, k)] = v;
        p16f[F_W2T + net * HID * HID + frag_pos(HID / 16, k, o)] = v;
        return;
    }
    i -= NW2;
    const int net = i / (OUTP * HID), o = (i / HID) % OUTP, k = i % HID;
    p16f[F_W3 + net * OUTP * HID + frag_pos(1, o, k)] = v;
    p16f[F_W3T + net * HID * 32 + frag_pos(HID / 16, k, o)] = v;
}
__global__ __launch_bounds__(256) void k_retile(const _Float16 *__restrict__ p16, _Float16 *__restrict__ p16f) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < NWT) write_frags(p16f, i, p16[i]);
}

// The same for the rollout's fp32 policy forward (dwp_policy, v_mfma_f32_16x16x4_f32: one A and one B word per lane and instruction, lane l:
// A[row l & 15][k = l >> 4], B[k = l >> 4][col l & 15]).  Four k-steps are one 16-byte request: the words of k = 16 kg + 4 j + (l >> 4),
// j = 0 .. 3, of row 16 nt + (l & 15) sit at ((kg * NT + nt) * 64 + l) * 4 + j.  p32f (floats): W1 | W2 | W3 as forward operands.
constexpr int G_W1 = 0, G_W2 = G_W1 + NW1, G_W3 = G_W2 + NW2, G_END = G_W3 + NW3;
static_assert(G_END == DWP_P32F_WORDS, "include/dyros_ppo.h");
__device__ __forceinline__ int frag32_pos(int nt_count, int row, int k) { return ((((k >> 4) * nt_count + (row >> 4)) * 64 + (k & 3) * 16 + (row & 15)) << 2) + ((k >> 2) & 3); }
__device__ __forceinline__ void write_frag32(float *__restrict__ p32f, int i, float v) {          // i: index of a WEIGHT in the parameter layout
    if (i < NW1) { const int net = i / (HID * INP), o = (i / INP) % HID, k = i % INP; p32f[G_W1 + net * HID * INP + frag32_pos(HID / 16, o, k)] = v; return; }
    i -= NW1;
    if (i < NW2) { const int net = i / (HID * HID), o = (i / HID) % HID, k = i % HID; p32f[G_W2 + net * HID * HID + frag32_pos(HID / 16, o, k)] = v; return; }
    i -= NW2;
    const int net = i / (OUTP * HID), o = (i / HID) % OUTP, k = i % HID;
    p32f[G_W3 + net * OUTP * HID + frag32_pos(1, o, k)] = v;
}
__global__ __launch_bounds__(256) void k_retile32(const float *__restrict__ p, float *__restrict__ p32f) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < NWT) write_frag32(p32f, i, p[i]);
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
    h8 x = *reinterpret_cast<const h8 *>(h16 + i);
    const h8 b = *reinterpret_cast<const h8 *>(b16 + net * HID + col);
    for (int q = 0; q < 8; ++q) { const _Float16 y = (_Float16)((float)x[q] + (float)b[q]); x[q] = (float)y > 0.0f ? y : (_Float16)0.0f; }
    *reinterpret_cast<h8 *>(h16 + i) = x;
}

// block = 32 column groups of 8 (one 16-byte load each) x 8 row lanes; RB_ROWS rows per block; column sums over the row lanes in LDS
constexpr int RB_ROWS = 64;
__global__ __launch_bounds__(256) void k_relu_bwd(const _Float16 *__restrict__ h16, _Float16 *__restrict__ dh16, float *__restrict__ gb_layer, int B) {
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

// a (still scaled) gradient: weights from the fp16 buffer of the library GEMMs, or -- g32 given -- the sum of dwp_wgrad's partial products
// over its slabs of samples (g32 [DWP_WGRAD_SLABS][weights], every word written by exactly one wave per update: no atomics, nothing to
// clear); biases from gb
constexpr int WG_SLABS = DWP_WGRAD_SLABS;
template <int SLABS>
__device__ __forceinline__ float scaled_grad(const _Float16 *g16, const float *g32, const float *gb, int i) {
    if (i >= NWT) return gb[i - NWT];
    if (!g32) return (float)g16[i];
    float s = g32[i];
#pragma unroll
    for (int k = 1; k < SLABS; ++k) s += g32[(size_t)k * NWT + i];
    return s;
}

constexpr int PART_FLAGS = 256, PART_SNAP = 512;          // `part` [DWP_PARTS]: sums of squares | inf / nan flags (bit per net) per block | scale, steps, learning rates
static_assert(DWP_PARTS >= PART_SNAP + 8, "part buffer");
constexpr int GS_BLOCKS = 256;          // partial sums of squares, one per block, in `part`; dwp_adam's blocks add them up (no atomics: 256
                                        // adds on one word are served one after the other and were most of this kernel's 12 us)
// SLABS: the slabs of g32 -- DWP_WGRAD_SLABS as dwp_wgrad leaves them, or 1: the ranks' averaged gradient in dwp_grad_bucket's bucket
template <int SLABS>
__global__ __launch_bounds__(256) void k_grad_stats(const _Float16 *__restrict__ g16, float *__restrict__ gb, float *__restrict__ state, float *__restrict__ part,
                                                    float *__restrict__ pbuf, const float *__restrict__ g32) {
    __shared__ float red[4];
    __shared__ int redf[4];
    const float inv = 1.0f / state[DWP_S_SCALE];
    const bool g16r = g32 && state[DWP_S_G16] != 0.0f;          // the summed weight gradients through fp16 first (autocast's rounding and its overflow at 65 504)
    float sq = 0.0f;
    int bad0 = 0, bad1 = 0;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (g32) {
        // weights: 16-byte pieces of dwp_wgrad's slabs, a thread's pieces all requested before the first is used (the kernel is one memory
        // latency long, not one per piece: 6.5 -> 4 us)
        constexpr int NQ = NWT / 4, PER = (NQ + GS_BLOCKS * 256 - 1) / (GS_BLOCKS * 256);
        static_assert(NWT % 4 == 0 && NW1 % 8 == 0 && NW2 % 8 == 0 && NW3 % 8 == 0, "a piece of four never crosses a net or a tensor");
        f4 a[PER][SLABS];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int q = t + u * GS_BLOCKS * 256, qc = q < NQ ? q : 0;
#pragma unroll
            for (int k = 0; k < SLABS; ++k) a[u][k] = reinterpret_cast<const f4 *>(g32 + (size_t)k * NWT)[qc];
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int q = t + u * GS_BLOCKS * 256;
            if (q < NQ) {
                f4 s4 = a[u][0];
#pragma unroll
                for (int k = 1; k < SLABS; ++k) s4 += a[u][k];          // (the same order as scaled_grad / dwp_adam: the sums are the same numbers)
                const int net = net_of(4 * q);
                float s2 = 0.0f;
                int bad = 0;
#pragma unroll
                for (int c = 0; c < 4; ++c) { const float g = g16r ? (float)(_Float16)s4[c] : s4[c]; bad |= !isfinite(g); const float w = g * inv; s2 += w * w; }
                if (net == 0) sq += s2;
                if (bad) { if (net) bad1 = 1; else bad0 = 1; }
            }
        }
    }
    for (int i = g32 ? NWT + t : t; i < NP; i += GS_BLOCKS * 256) {
        const int net = net_of(i);
        float g;
        if (i >= NWT && pbuf) {
            // a bias gradient of the dwp_mlp path: the column's sum over the buckets (cleared for the next update), left in gb for dwp_adam
            const int q = i - NWT;
            const int col = q < NB1 ? PB_B1 + q % HID : (q < NB1 + NB2 ? PB_B2 + (q - NB1) % HID : PB_B3 + (q - NB1 - NB2) % OUTP);
            float *pc = pbuf + (size_t)net * PBW + col;
            float s0 = 0.0f;
#pragma unroll
            for (int w = 0; w < PBK; ++w) { s0 += pc[(size_t)w * 2 * PBW]; }
#pragma unroll
            for (int w = 0; w < PBK; ++w) pc[(size_t)w * 2 * PBW] = 0.0f;
            g = s0;
            gb[q] = g;
        } else g = scaled_grad<SLABS>(g16, g32, gb, i);
        if (!isfinite(g)) { if (net) bad1 = 1; else bad0 = 1; }
        const float u = g * inv;
        if (net == 0) sq += u * u;
    }
    sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    // per block, for the Adam kernel that also finishes the update (k_adam<true>: its blocks read nothing of `state`, which its block 0
    // rewrites): the flags next to the partial sum, and -- block 0 -- the words of `state` an Adam block needs
    const unsigned long long w0 = __ballot(bad0), w1 = __ballot(bad1);
    if ((threadIdx.x & 63) == 0) redf[threadIdx.x >> 6] = (w0 ? 1 : 0) | (w1 ? 2 : 0);
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
        part[PART_FLAGS + blockIdx.x] = (float)(redf[0] | redf[1] | redf[2] | redf[3]);
        if (blockIdx.x == 0) {
            part[PART_SNAP + 0] = state[DWP_S_SCALE];
            part[PART_SNAP + 1] = state[DWP_S_STEP]; part[PART_SNAP + 2] = state[DWP_S_STEP + 1];
            part[PART_SNAP + 3] = state[DWP_S_LR]; part[PART_SNAP + 4] = state[DWP_S_LR + 1];
            part[PART_SNAP + 5] = state[DWP_S_G16];
        }
    }
    if (bad0) state[DWP_S_FOUND_INF] = 1.0f;
    if (bad1) state[DWP_S_FOUND_INF + 1] = 1.0f;
}

// Sharded training (one process per GPU): this rank's (still scaled) gradient as ONE contiguous bucket for ONE all-reduce -- weights: the
// sum of dwp_wgrad's slabs, in dwp_grad_stats' order; biases: the sums over dwp_mlp's buckets (cleared here, as dwp_grad_stats would) --
// times 1 / world, so that the ranks' SUM is the average the reference's Horovod optimizer.synchronize() forms
// (learning/rl_games_custom/a2c_continuous_seperate.py:171-173) before unscale_ / clip / step.  An inf or nan of any rank survives the
// sum, so every rank's dwp_grad_stats finds it and every rank skips alike.
__global__ __launch_bounds__(256) void k_grad_bucket(const float *__restrict__ g32, float *__restrict__ pbuf, float *__restrict__ bucket, float inv_world) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    constexpr int NQ = NWT / 4;
    if (t < NQ) {
        f4 a[WG_SLABS];
#pragma unroll
        for (int k = 0; k < WG_SLABS; ++k) a[k] = reinterpret_cast<const f4 *>(g32 + (size_t)k * NWT)[t];
        f4 s4 = a[0];
#pragma unroll
        for (int k = 1; k < WG_SLABS; ++k) s4 += a[k];
        reinterpret_cast<f4 *>(bucket)[t] = s4 * inv_world;
    } else if (t - NQ < NBT) {
        const int q = t - NQ, net = net_of(NWT + q);
        const int col = q < NB1 ? PB_B1 + q % HID : (q < NB1 + NB2 ? PB_B2 + (q - NB1) % HID : PB_B3 + (q - NB1 - NB2) % OUTP);
        float *pc = pbuf + (size_t)net * PBW + col;
        float s0 = 0.0f;
#pragma unroll
        for (int w = 0; w < PBK; ++w) s0 += pc[(size_t)w * 2 * PBW];
#pragma unroll
        for (int w = 0; w < PBK; ++w) pc[(size_t)w * 2 * PBW] = 0.0f;
        bucket[NWT + q] = s0 * inv_world;
    }
}

// the end of an update, by ONE block of 256 threads (all of them arrive): logged means, GradScaler.update, step counts, minibatch index
__device__ __forceinline__ void finish_update(float *__restrict__ state, int B, int nmb, int growth_interval, float *__restrict__ pbuf) {
    // Two round trips to memory in all: every word is requested before the first is used (a thread that walks the 64 bucket rows, or reads a
    // word of `state` behind a store to `state`, pays one round trip per word: that was 4 of dwp_adam_finish's 10 us).
    float t5[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (threadIdx.x >= 64) return;          // (wave 0 does it)
    if (pbuf) {          // dwp_mlp's logged sums: over the buckets and the two nets (lane = bucket row), cleared for the next update
        static_assert(2 * PBK == 64, "one bucket row per lane of wave 0");
        float *row = pbuf + (size_t)threadIdx.x * PBW + PB_ST;
#pragma unroll
        for (int q = 0; q < 5; ++q) t5[q] = row[q];
#pragma unroll
        for (int q = 0; q < 5; ++q) row[q] = 0.0f;
    }
    float sv[5], s_f0 = 0.0f, s_f1 = 0.0f, s_norm2 = 0.0f, s_scale = 0.0f, s_growth = 0.0f, s_st0 = 0.0f, s_st1 = 0.0f, s_mb = 0.0f;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 5; ++q) sv[q] = state[q];
        s_f0 = state[DWP_S_FOUND_INF]; s_f1 = state[DWP_S_FOUND_INF + 1]; s_norm2 = state[DWP_S_NORM2]; s_scale = state[DWP_S_SCALE];
        s_growth = state[DWP_S_GROWTH]; s_st0 = state[DWP_S_STEP]; s_st1 = state[DWP_S_STEP + 1]; s_mb = state[DWP_S_MB];
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) t5[q] = wave_sum(t5[q]);
    if (threadIdx.x != 0) return;
    static_assert(DWP_S_ALOSS == 0 && DWP_S_CLOSS == 1 && DWP_S_BLOSS == 2 && DWP_S_CLIPPED == 3 && DWP_S_KL == 4, "the logged sums are state[0 .. 4]");
    const float invB = 1.0f / (float)B;
    const bool f0 = s_f0 != 0.0f, f1 = s_f1 != 0.0f;
    float *o = state + DWP_S_OUT;
#pragma unroll
    for (int q = 0; q < 5; ++q) o[q] = (sv[q] + t5[q]) * invB;
    o[5] = sqrtf(s_norm2); o[6] = s_scale; o[7] = (f0 || f1) ? 1.0f : 0.0f;
    // torch.amp.GradScaler.update (_amp_update_scale_): backoff 0.5 on any inf, growth 2.0 after growth_interval clean updates
    if (f0 || f1) { state[DWP_S_SCALE] = s_scale * 0.5f; state[DWP_S_GROWTH] = 0.0f; }
    else {
        const float t = s_growth + 1.0f;
        if ((int)t == growth_interval) { state[DWP_S_SCALE] = s_scale * 2.0f; state[DWP_S_GROWTH] = 0.0f; }
        else state[DWP_S_GROWTH] = t;
    }
    if (!f0) state[DWP_S_STEP] = s_st0 + 1.0f;
    if (!f1) state[DWP_S_STEP + 1] = s_st1 + 1.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) state[k] = 0.0f;
    const int mb = (int)s_mb + 1;
    state[DWP_S_MB] = (float)(mb >= nmb ? 0 : mb);
}
__global__ __launch_bounds__(256) void k_finish(float *__restrict__ state, float *__restrict__ gb, int B, int nmb, int growth_interval, float *__restrict__ pbuf) {
    for (int i = threadIdx.x; i < NBT; i += 256) gb[i] = 0.0f;
    finish_update(state, B, nmb, growth_interval, pbuf);
}

// Eight consecutive parameters per thread: a run of eight never crosses a row, a net or a tensor (every row length is a multiple of
// 8), and it is exactly one fragment of the forward operand order -- so the master, the moments, the row-major fp16 copy and the forward
// fragment copy move as 16- / 32-byte pieces; only the input-gradient copies of W2 / W3 (k = the OUTPUT index) are eight scattered halves.
static_assert(NW1 % 8 == 0 && NW2 % 8 == 0 && NW3 % 8 == 0 && NB1 % 8 == 0 && NB2 % 8 == 0 && NB3 % 8 == 0 && INP % 8 == 0 && HID % 8 == 0, "runs of eight");
// the Adam step of eight consecutive parameters held in registers (gs: their still-scaled gradients) and every copy of the weights
__device__ __forceinline__ void adam_apply(float *__restrict__ p, _Float16 *__restrict__ p16, float *__restrict__ m, float *__restrict__ v, _Float16 *__restrict__ p16f,
                                           float *__restrict__ p32f, int i0, int net, f4 (&pv)[2], f4 (&mv)[2], f4 (&vv)[2], const float (&gs)[8], float scale, float step, float lr,
                                           float norm2, float max_norm) {
    const float inv = 1.0f / scale;
    const float coef = net == 0 ? fminf(max_norm / (sqrtf(norm2) + 1e-6f), 1.0f) : 1.0f;          // torch.nn.utils.clip_grad_norm_ (the actor only)
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    const float bc1 = 1.0f - powf(b1, step), sq2 = sqrtf(1.0f - powf(b2, step)), ss = lr / bc1;
    h8 ph;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float g = gs[q] * inv * coef;
        const float mq = mv[q >> 2][q & 3], vq = vv[q >> 2][q & 3];
        const float mi = mq + (g - mq) * (1.0f - b1);          // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = b2 * vq + (1.0f - b2) * g * g;
        const float denom = sqrtf(vi) / sq2 + eps;
        float pi = pv[q >> 2][q & 3] - ss * (mi / denom);
        mv[q >> 2][q & 3] = mi; vv[q >> 2][q & 3] = vi; pv[q >> 2][q & 3] = pi;
        // the fp16 copies are the STORED fp32 value rounded (what a cast of the master gives, e.g. after a checkpoint is loaded): without the
        // opaque touch the compiler rounds the multiply-add once, straight to fp16 (v_fma_mixlo_f16), and ties fall the other way
        asm volatile("" : "+v"(pi));
        ph[q] = (_Float16)pi;
    }
    reinterpret_cast<f4 *>(p + i0)[0] = pv[0]; reinterpret_cast<f4 *>(p + i0)[1] = pv[1];
    reinterpret_cast<f4 *>(m + i0)[0] = mv[0]; reinterpret_cast<f4 *>(m + i0)[1] = mv[1];
    reinterpret_cast<f4 *>(v + i0)[0] = vv[0]; reinterpret_cast<f4 *>(v + i0)[1] = vv[1];
    *reinterpret_cast<h8 *>(p16 + i0) = ph;
    if (p32f && i0 < NWT) {
#pragma unroll
        for (int q = 0; q < 8; ++q) write_frag32(p32f, i0 + q, pv[q >> 2][q & 3]);          // (the rollout's fp32 policy reads the masters in ITS operand order)
    }
    if (p16f && i0 < NWT) {
        // forward operand order: my eight are one fragment (frag_pos of the first, k & 7 = 0); input-gradient order: one half each
        int i = i0;
        if (i < NW1) { const int nn = i / (HID * INP), o = (i / INP) % HID, k = i % INP; *reinterpret_cast<h8 *>(p16f + F_W1 + nn * HID * INP + frag_pos(HID / 16, o, k)) = ph; }
        else if ((i -= NW1) < NW2) {
            const int nn = i / (HID * HID), o = (i / HID) % HID, k = i % HID;
            *reinterpret_cast<h8 *>(p16f + F_W2 + nn * HID * HID + frag_pos(HID / 16, o, k)) = ph;
#pragma unroll
            for (int q = 0; q < 8; ++q) p16f[F_W2T + nn * HID * HID + frag_pos(HID / 16, k + q, o)] = ph[q];
        } else {
            i -= NW2;
            const int nn = i / (OUTP * HID), o = (i / HID) % OUTP, k = i % HID;
            *reinterpret_cast<h8 *>(p16f + F_W3 + nn * OUTP * HID + frag_pos(1, o, k)) = ph;
#pragma unroll
            for (int q = 0; q < 8; ++q) p16f[F_W3T + nn * HID * 32 + frag_pos(HID / 16, k + q, o)] = ph[q];
        }
    }
}

// FIN: the launch also finishes the update (dwp_adam_finish: one graph node less).  Its blocks then read the loss scale, step counts,
// learning rates and inf / nan flags from `part`, where dwp_grad_stats left them, because block 0 rewrites `state` while the others run.
struct FinArgs { int B, nmb, growth_interval; float *pbuf; };
template <bool FIN, int SLABS>
__global__ __launch_bounds__(256) void k_adam(float *__restrict__ p, _Float16 *__restrict__ p16, float *__restrict__ m, float *__restrict__ v,
                                              const _Float16 *__restrict__ g16, const float *__restrict__ gb, float *__restrict__ state, const float *__restrict__ part,
                                              float max_norm, _Float16 *__restrict__ p16f, const float *__restrict__ g32, float *__restrict__ p32f, FinArgs fin) {
    __shared__ float red[4];
    __shared__ int redf[4];
    static_assert(GS_BLOCKS == 256, "one partial per thread");
    // this thread's parameters, moments and gradients are requested first, the partial sums behind them: one memory latency for both
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * 8, ic = i0 < NP ? i0 : 0;
    f4 pv[2] = {reinterpret_cast<const f4 *>(p + ic)[0], reinterpret_cast<const f4 *>(p + ic)[1]};
    f4 mv[2] = {reinterpret_cast<const f4 *>(m + ic)[0], reinterpret_cast<const f4 *>(m + ic)[1]};
    f4 vv[2] = {reinterpret_cast<const f4 *>(v + ic)[0], reinterpret_cast<const f4 *>(v + ic)[1]};
    float gs[8];
    if (ic >= NWT) {
#pragma unroll
        for (int q = 0; q < 8; ++q) gs[q] = gb[ic - NWT + q];
    } else if (g32) {
        f4 a[SLABS][2];
#pragma unroll
        for (int k = 0; k < SLABS; ++k) { a[k][0] = reinterpret_cast<const f4 *>(g32 + (size_t)k * NWT + ic)[0]; a[k][1] = reinterpret_cast<const f4 *>(g32 + (size_t)k * NWT + ic)[1]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) gs[q] = 0.0f;
#pragma unroll
        for (int k = 0; k < SLABS; ++k) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { gs[q] += a[k][0][q]; gs[4 + q] += a[k][1][q]; }
        }
    } else {
        const h8 gh = *reinterpret_cast<const h8 *>(g16 + ic);
#pragma unroll
        for (int q = 0; q < 8; ++q) gs[q] = (float)gh[q];
    }
    if (g32 && ic < NWT && (FIN ? part[PART_SNAP + 5] : state[DWP_S_G16]) != 0.0f) {          // (as dwp_grad_stats saw them)
#pragma unroll
        for (int q = 0; q < 8; ++q) gs[q] = (float)(_Float16)gs[q];
    }
    const float my_part = part[threadIdx.x];
    const int fl = FIN ? (int)part[PART_FLAGS + threadIdx.x] : 0;
    {   // the actor's gradient norm from dwp_grad_stats' partial sums (every block adds them up the same way; block 0 publishes it)
        const float s = wave_sum(my_part);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        if (FIN) {
            const unsigned long long w0 = __ballot(fl & 1), w1 = __ballot(fl & 2);
            if ((threadIdx.x & 63) == 0) redf[threadIdx.x >> 6] = (w0 ? 1 : 0) | (w1 ? 2 : 0);
        }
        __syncthreads();
    }
    const float norm2 = red[0] + red[1] + red[2] + red[3];
    const int net = net_of(ic);
    bool skip;
    float scale, step, lr;
    if (FIN) {
        skip = (((redf[0] | redf[1] | redf[2] | redf[3]) >> net) & 1) != 0;
        scale = part[PART_SNAP]; step = part[PART_SNAP + 1 + net] + 1.0f; lr = part[PART_SNAP + 3 + net];
    } else {
        skip = state[DWP_S_FOUND_INF + net] != 0.0f;
        scale = state[DWP_S_SCALE]; step = state[DWP_S_STEP + net] + 1.0f; lr = state[DWP_S_LR + net];
    }
    if (FIN) {
        // the finishing block is an EXTRA one (the last): three dependent round trips to memory (partials, logged sums, state) that run beside
        // the Adam blocks' own instead of behind block 0's (10 -> ? us)
        if (blockIdx.x == gridDim.x - 1) {
            if (threadIdx.x == 0) state[DWP_S_NORM2] = norm2;
            finish_update(state, fin.B, fin.nmb, fin.growth_interval, fin.pbuf);
            return;
        }
    } else if (blockIdx.x == 0 && threadIdx.x == 0) state[DWP_S_NORM2] = norm2;
    if (i0 >= NP) return;
    if (skip) return;          // GradScaler.step: this optimiser's step is skipped
    adam_apply(p, p16, m, v, p16f, p32f, i0, net, pv, mv, vv, gs, scale, step, lr, norm2, max_norm);
}

// ------------------------------------------------------------------------------------------------ dwp_mlp: forward, loss, input gradients
// Three layers on v_mfma_f32_16x16x32_f16 per net (blockIdx.y: 0 actor, 1 critic): A = activation rows from LDS, B = weight fragments
// straight from the fragment-order fp16 copy in L2 (the 1 MB of weights is resident there), bias + relu in the epilogue, the loss on the
// head's accumulator tiles (their 16 columns are the 16 lanes of a DPP row: per-sample sums are row reductions), then the two
// input-gradient products with the relu masks and the bias gradients in their epilogues.  Writes every buffer the weight gradients read
// (x16, h1, h2, dout, dz2, dz1), row-major and in dwp_wgrad's operand order.
// Lane maps (cdna_hip_programming.md section 3; checked by tests/test_ppo_gpu.py against torch matmul on the same operands):
//   A[row l & 15][k = 8 (l >> 4) + j], B[k = 8 (l >> 4) + j][col l & 15], j = 0..7;  C/D[row 4 (l >> 4) + r][col l & 15], r = 0..3.
// A workgroup = 4 waves takes MT = 32 samples (MR = 2 row tiles of 16) through ONE net.  The waves share the activation images in LDS and
// split every product by COLUMNS: wave w owns column tiles 4 w .. 4 w + 3 of the 16 (so a weight fragment is fetched by one wave of the
// workgroup and used for MR products), writes its columns of the layer's output image, and a workgroup barrier separates the layers.
// (One wave doing all 16 column tiles of 16 or 32 rows took 43-47 us whatever the tiling: a lone wave per SIMD runs its GEMM steps,
// epilogues and copies one after the other; split four ways each SIMD of the CU has a wave and a quarter of the work.)
constexpr int MR = 2, MT = 16 * MR, XS = INP + 8, HS = HID + 8, NTL = HID / 16;          // LDS row strides in halves (+16 B)
// dwp_mlp's workgroup: MWPB waves, two per SIMD -- one hides the other's LDS, L2 and transcendental latencies (four waves, one per SIMD: 19 -> ? us)
constexpr int MWPB = 8, MNTW = NTL / MWPB, MRD = 6;          // MRD: depth of the ring of weight-fragment requests

// acc[mr][t] = As[16 mr .. +15][K] . W[16 (nt0 + t) .. +15][K]' for NT column tiles from nt0 (NTA = column tiles of the whole matrix); W in
// fragment order.  A lone wave per SIMD has nothing to hide an L2 round trip behind but its own products: the B fragments are requested
// RD - 1 k-steps ahead into a ring of RD register sets (the loop is unrolled: every index is static), and scheduling barriers keep the
// requests where they are written -- the machine scheduler otherwise sinks every request to just before its product (fewer live
// registers, and the ring is gone: measured 71 % of the wave's cycles waiting at vmcnt(1) / vmcnt(2)).
template <int K, int NTA, int NT, int RD>
__device__ __forceinline__ void ring_fill(const _Float16 *__restrict__ W, int nt0, h8 (&b)[RD][NT], int lane) {          // the first RD - 1 k-steps' requests
    const h8 *wl = reinterpret_cast<const h8 *>(W) + nt0 * 64 + lane;          // (W: the matrix in fragment order, frag_pos)
#pragma unroll
    for (int pk = 0; pk < RD - 1 && pk < K / 32; ++pk)
#pragma unroll
        for (int t = 0; t < NT; ++t) b[pk][t] = wl[(pk * NTA + t) * 64];
    __builtin_amdgcn_sched_barrier(0);
}
template <int K, int AS, int NTA, int NT, int RD>
__device__ __forceinline__ void mfma_go(const _Float16 *As, const _Float16 *__restrict__ W, int nt0, f4 (&acc)[MR][NT], h8 (&b)[RD][NT], int lane) {
    constexpr int KS = K / 32;
    const int ar = lane & 15, ak = 8 * (lane >> 4);
    const h8 *wl = reinterpret_cast<const h8 *>(W) + nt0 * 64 + lane;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[mr][t] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        if (kk + RD - 1 < KS) {
#pragma unroll
            for (int t = 0; t < NT; ++t) b[(kk + RD - 1) % RD][t] = wl[((kk + RD - 1) * NTA + t) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        h8 a[MR];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) a[mr] = *reinterpret_cast<const h8 *>(As + (16 * mr + ar) * AS + 32 * kk + ak);
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[mr][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mr], b[kk % RD][t], acc[mr][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int K, int AS, int NTA, int NT, int RD>
__device__ __forceinline__ void mfma_rows(const _Float16 *As, const _Float16 *__restrict__ W, int nt0, f4 (&acc)[MR][NT], int lane) {
    h8 b[RD][NT];
    ring_fill<K, NTA, NT, RD>(W, nt0, b, lane);
    mfma_go<K, AS, NTA, NT, RD>(As, W, nt0, acc, b, lane);
}
// the workgroup's MT rows of an LDS image [MT][stride] to global rows of `cols` halves, 16 bytes per thread and request
template <int COLS, int STRIDE>
__device__ __forceinline__ void rows_out(const _Float16 *Ls, _Float16 *__restrict__ g, int tid) {
#pragma unroll
    for (int t = tid; t < MT * (COLS / 8); t += 64 * MWPB) {
        const int r = t / (COLS / 8), c = (t % (COLS / 8)) * 8;
        *reinterpret_cast<h8 *>(g + (size_t)r * COLS + c) = *reinterpret_cast<const h8 *>(Ls + r * STRIDE + c);
    }
}
// The workgroup's MT = 32 rows of an LDS image as operands of the weight-gradient products, whose k index is the SAMPLE: 32 samples are
// one k-step, and for feature tile t lane l of the consuming wave wants samples 8 (l >> 4) .. + 7 of feature 16 t + (l & 15) as its eight
// halves (A operand if the features are the product's rows, B if its columns: the same packing).  Block (k-step) kb of a buffer with
// FEAT features: (kb * FEAT / 16 + t) * 64 + l, 16 bytes each -- a consumer's request is one contiguous KB.
template <int FEAT, int STRIDE>
__device__ __forceinline__ void frags_out(const _Float16 *Ls, _Float16 *__restrict__ dstblock, int tid) {
    for (int it = tid; it < (FEAT / 16) * 64; it += 64 * MWPB) {
        const int t = it >> 6, l = it & 63;
        const _Float16 *src = Ls + (8 * (l >> 4)) * STRIDE + 16 * t + (l & 15);
        h8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[j * STRIDE];
        reinterpret_cast<h8 *>(dstblock)[it] = v;
    }
}
// The same operand-order rows straight from a wave's C tiles (no LDS round trip): the lane of group g holds samples 16 mr + 4 g + r (r = 0 .. 3)
// of its column; the operand lane for those samples is group 2 mr + (g >> 1) of the same column and wants eight -- the four of group g
// (even) followed by the four of group g + 1.  One exchange with the lane 16 up, then the even groups store 16 bytes per row tile.
__device__ __forceinline__ void frag_store_tile(const _Float16 (&z)[MR][4], _Float16 *__restrict__ dstblock, int feat_tiles, int nt, int cr, int g) {
    (void)feat_tiles;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 lo = {z[mr][0], z[mr][1]}, hi = {z[mr][2], z[mr][3]};
        const int plo = __shfl_down(__builtin_bit_cast(int, lo), 16, 64), phi = __shfl_down(__builtin_bit_cast(int, hi), 16, 64);
        if ((g & 1) == 0) {
            const h2 qlo = __builtin_bit_cast(h2, plo), qhi = __builtin_bit_cast(h2, phi);
            const h8 v = {z[mr][0], z[mr][1], z[mr][2], z[mr][3], qlo[0], qlo[1], qhi[0], qhi[1]};
            reinterpret_cast<h8 *>(dstblock)[nt * 64 + (2 * mr + (g >> 1)) * 16 + cr] = v;
        }
    }
}
// bias + relu of my column tiles of a hidden layer from the accumulator tiles into the layer's LDS image
__device__ __forceinline__ void hidden_out(const f4 (&acc)[MR][MNTW], const float (&bia)[MNTW], _Float16 *Hs, _Float16 *__restrict__ fdst, int nt0, int cr, int g) {
#pragma unroll
    for (int t = 0; t < MNTW; ++t) {
        _Float16 z[MR][4];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const _Float16 h = (_Float16)(acc[mr][t][r] + bia[t]);
                z[mr][r] = (float)h > 0.0f ? h : (_Float16)0.0f;
                Hs[(16 * mr + 4 * g + r) * HS + 16 * (nt0 + t) + cr] = z[mr][r];
            }
        if (fdst) frag_store_tile(z, fdst, NTL, nt0 + t, cr, g);
    }
}
// relu mask of my column tiles of a hidden layer's gradient (Hs: that layer's activations) into Zs, their bias gradient into the accumulators
__device__ __forceinline__ void masked_out(const f4 (&acc)[MR][MNTW], const _Float16 *Hs, _Float16 *Zs, _Float16 *__restrict__ fdst, float *pcol, int nt0, int cr,
                                           int g, int lane) {
#pragma unroll
    for (int t = 0; t < MNTW; ++t) {
        float cs = 0.0f;
        _Float16 z[MR][4];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int idx = (16 * mr + 4 * g + r) * HS + 16 * (nt0 + t) + cr;
                z[mr][r] = (float)Hs[idx] > 0.0f ? (_Float16)acc[mr][t][r] : (_Float16)0.0f;
                Zs[idx] = z[mr][r];
                cs += (float)z[mr][r];
            }
        if (fdst) frag_store_tile(z, fdst, NTL, nt0 + t, cr, g);
        cs += __shfl_xor(cs, 16, 64); cs += __shfl_xor(cs, 32, 64);
        if (lane < 16) atomicAdd(&pcol[16 * (nt0 + t) + lane], cs);
    }
}

// Bias gradients and logged sums of dwp_mlp: every workgroup adds its samples' column sums into ONE OF 32 rows of accumulators (its
// bucket: workgroup index mod 32) -- [0, 256) the first hidden layer's bias gradient, [256, 512) the second's, [512, 528) the head's,
// [528, 533) the logged sums -- so that an accumulator word sees 4 adds per update instead of 128.  dwp_grad_stats adds the buckets up
// into gb and clears them, dwp_finish does the same for the logged sums.
struct MlpArgs {
    const float *obs, *state, *act, *old_nlp, *old_mu, *adv, *ret, *logstd;
    const _Float16 *obs16;          // (or null) the batch's observations as fp16 rows of INP (zero padding), instead of obs
    const _Float16 *p16, *p16t;
    float *pbuf;
    _Float16 *x16, *h1, *h2, *out16, *dout16, *dz2, *dz1;
    _Float16 *xf, *h1f, *h2f, *doutf, *dz2f, *dz1f;          // (or all null) the same once more in dwp_wgrad's operand order
    int B;
    float e_clip, critic_coef;
};

#if defined(DWP_STAMPS)          // (profiling builds only: cycle stamps of workgroup 0's wave 0 into the tail of pbuf's first row)
#define MLP_STAMP(n) do { if ((n) <= 10 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) A.pbuf[PB_ST + 5 + (n)] = (float)(unsigned)(__builtin_readcyclecounter() & 0xffffff); } while (0)
#else
#define MLP_STAMP(n) do { } while (0)
#endif
__global__ __launch_bounds__(64 * MWPB) void k_mlp(const MlpArgs A) {
    __shared__ _Float16 Xs[MT * XS];          // the input rows; after the first layer: the masked gradient of the second hidden layer
    __shared__ _Float16 H1s[MT * HS], H2s[MT * HS], Ds[MT * HS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, net = blockIdx.y, r0 = blockIdx.x * MT, B = A.B;
    const int nt0 = MNTW * wv;          // my column tiles: nt0 .. nt0 + MNTW - 1
    const int mb = (int)A.state[DWP_S_MB];
    const int cr = lane & 15, g = lane >> 4;          // my column in a C tile, my group of four rows
    float *prow = A.pbuf + ((size_t)(blockIdx.x & (PBK - 1)) * 2 + net) * PBW;          // (my bucket's row of accumulators)
    const _Float16 *W1 = A.p16t + F_W1 + (size_t)net * HID * INP, *W2 = A.p16t + F_W2 + (size_t)net * HID * HID, *W3 = A.p16t + F_W3 + (size_t)net * OUTP * HID;
    const _Float16 *b1 = A.p16 + NWT + net * HID, *b2 = A.p16 + NWT + NB1 + net * HID, *b3 = A.p16 + NWT + NB1 + NB2 + net * OUTP;
    const _Float16 *W2T = A.p16t + F_W2T + (size_t)net * HID * HID, *W3T = A.p16t + F_W3T + (size_t)net * HID * 32;
    // (a product's first weight fragments are requested before the work that precedes it -- the staging, the previous layer's epilogue --
    //  so that their L2 round trip passes under it: the weights do not depend on anything computed here)
    MLP_STAMP(0);
    h8 ring[MRD][MNTW];
    ring_fill<INP, NTL, MNTW, MRD>(W1, nt0, ring, lane);
    // ---- the input rows: fp32 observations -> fp16, zero padding (autocast's cast of the Linear input) ----
    if (A.obs16) {
        // the batch holds them as fp16 rows already (dwp_rollout_pre wrote them so): 16-byte pieces, eight per thread
        constexpr int PPT = MT * INP / 8 / (64 * MWPB);
        static_assert(MT * INP / 8 == PPT * 64 * MWPB, "whole pieces per thread");
        const h8 *src = reinterpret_cast<const h8 *>(A.obs16 + ((size_t)mb * B + r0) * INP);
        h8 v[PPT];
#pragma unroll
        for (int u = 0; u < PPT; ++u) v[u] = src[tid + 64 * MWPB * u];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < PPT; ++u) { const int pc = tid + 64 * MWPB * u, r = pc >> 6, c = pc & 63; *reinterpret_cast<h8 *>(&Xs[r * XS + 8 * c]) = v[u]; }
        __syncthreads();
        if (net == 0) { if (A.x16) rows_out<INP, XS>(Xs, A.x16 + (size_t)r0 * INP, tid); if (A.xf) frags_out<INP, XS>(Xs, A.xf + (size_t)blockIdx.x * INP * 32, tid); }
    } else {
        const float *src = A.obs + ((size_t)mb * B + r0) * IN;
        // thread t takes column t (+ a multiple of the workgroup's size) of every row (consecutive threads: consecutive floats of a row; no index
        // arithmetic per word): all of a thread's requests go out before it converts the first word -- one memory latency for the block
        constexpr int CPT = INP / (64 * MWPB);
        static_assert(INP == CPT * 64 * MWPB, "whole column slots per thread");
        float v[CPT][MT];
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int c = tid + 64 * MWPB * u, cl = c < IN ? c : 0;          // (the padding columns re-read column 0: no branch per request)
#pragma unroll
            for (int r = 0; r < MT; ++r) v[u][r] = src[(size_t)r * IN + cl];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int c = tid + 64 * MWPB * u;
#pragma unroll
            for (int r = 0; r < MT; ++r) Xs[r * XS + c] = c < IN ? (_Float16)v[u][r] : (_Float16)0.0f;
        }
        __syncthreads();
        if (net == 0) { if (A.x16) rows_out<INP, XS>(Xs, A.x16 + (size_t)r0 * INP, tid); if (A.xf) frags_out<INP, XS>(Xs, A.xf + (size_t)blockIdx.x * INP * 32, tid); }
    }
    MLP_STAMP(1);
    f4 acc[MR][MNTW];
    float bia[MNTW];          // (a layer's biases, requested before its products: their latency passes under the GEMM)
    // ---- hidden layer 1 ----
#pragma unroll
    for (int t = 0; t < MNTW; ++t) bia[t] = (float)b1[16 * (nt0 + t) + cr];
    mfma_go<INP, XS, NTL, MNTW, MRD>(Xs, W1, nt0, acc, ring, lane);
    MLP_STAMP(2);
    ring_fill<HID, NTL, MNTW, MRD>(W2, nt0, ring, lane);
    const size_t fblk = ((size_t)net * gridDim.x + blockIdx.x) * HID * 32;          // (my block of the per-net operand-order buffers)
    hidden_out(acc, bia, H1s, A.h1f ? A.h1f + fblk : nullptr, nt0, cr, g);
    __syncthreads();
    if (A.h1) rows_out<HID, HS>(H1s, A.h1 + ((size_t)net * B + r0) * HID, tid);
    // ---- hidden layer 2 ----
#pragma unroll
    for (int t = 0; t < MNTW; ++t) bia[t] = (float)b2[16 * (nt0 + t) + cr];
    MLP_STAMP(3);
    mfma_go<HID, HS, NTL, MNTW, MRD>(H1s, W2, nt0, acc, ring, lane);
    MLP_STAMP(4);
    hidden_out(acc, bia, H2s, A.h2f ? A.h2f + fblk : nullptr, nt0, cr, g);
    __syncthreads();
    if (A.h2) rows_out<HID, HS>(H2s, A.h2 + ((size_t)net * B + r0) * HID, tid);
    // ---- the head (16 padded outputs = one column tile; every wave forms it -- 16 products) and the loss on its accumulator tiles: column =
    //      lane of a DPP row; wave w takes accumulator row w of every lane group, i.e. samples 4 g + w of each row tile ----
    // (what the loss needs from memory is requested before the head's products; so are my head weights of the product after it)
    const bool ak = cr < ACT;
    const float scale = A.state[DWP_S_SCALE], invB = 1.0f / (float)B;
    const float bias = (float)b3[cr], ls = ak ? A.logstd[cr] : 0.0f;
    // the loss's rows: the MR * 4 accumulator rows of a lane group go round the MWPB waves -- wave w takes row w & 3 of LM row tiles from mr0
    constexpr int LM = MR * 4 / MWPB;
    static_assert(LM >= 1 && LM * MWPB == MR * 4, "the head's rows split evenly over the waves");
    const int wr = wv & 3, mr0 = (wv >> 2) * LM;
    float in_a[LM], in_om[LM], in_adv[LM], in_nlp[LM], in_ret[LM];
#pragma unroll
    for (int i = 0; i < LM; ++i) {
        const size_t srow = (size_t)mb * B + r0 + 16 * (mr0 + i) + 4 * g + wr;
        in_a[i] = net == 0 && ak ? A.act[srow * ACT + cr] : 0.0f; in_om[i] = net == 0 && ak ? A.old_mu[srow * ACT + cr] : 0.0f;
        in_adv[i] = A.adv[srow]; in_nlp[i] = A.old_nlp[srow]; in_ret[i] = A.ret[srow];
    }
    h8 w3t[MNTW];
#pragma unroll
    for (int t = 0; t < MNTW; ++t) w3t[t] = reinterpret_cast<const h8 *>(W3T)[(nt0 + t) * 64 + lane];          // (k 16 .. 31: the zero padding)
    f4 o4[MR][1];
    MLP_STAMP(5);
    mfma_rows<HID, HS, 1, 1, 8>(H2s, W3, 0, o4, lane);
    MLP_STAMP(6);
    {
        float st[5] = {0, 0, 0, 0, 0}, gsum = 0.0f;
        // (the per-sample sums of both row tiles are reduced TOGETHER over the 16 lanes of a DPP row -- eight independent chains of four
        //  exchanges instead of one after the other: a lone wave waits out every exchange)
        _Float16 o16[LM];
        float red[LM][4];
        const float sg = expf(ls), s2 = sg * sg;
#pragma unroll
        for (int mr = 0; mr < LM; ++mr) {
            f4 ot = o4[0][0];
#pragma unroll
            for (int m = 1; m < MR; ++m) if (mr0 + mr == m) ot = o4[m][0];
            const float oacc = wr == 0 ? ot[0] : (wr == 1 ? ot[1] : (wr == 2 ? ot[2] : ot[3]));
            o16[mr] = (net == 0 ? ak : cr == 0) ? (_Float16)(oacc + bias) : (_Float16)0.0f;
            const float mu = (float)o16[mr], z = (in_a[mr] - mu) / sg, om = in_om[mr];
            const float hi = fminf(mu - 1.1f, 0.0f), lo = fminf(-mu + 1.1f, 0.0f);
            red[mr][0] = ak ? z * z : 0.0f; red[mr][1] = ls; red[mr][2] = ak ? lo * lo + hi * hi : 0.0f;
            red[mr][3] = ak ? logf(sg / sg + 1e-5f) + (s2 + (om - mu) * (om - mu)) / (2.0f * (s2 + 1e-5f)) - 0.5f : 0.0f;
        }
        if (net == 0) {
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1)
#pragma unroll
                for (int mr = 0; mr < LM; ++mr)
#pragma unroll
                    for (int q = 0; q < 4; ++q) red[mr][q] += __shfl_xor(red[mr][q], o, 16);
        }
#pragma unroll
        for (int mr = 0; mr < LM; ++mr) {
            const int row = 16 * (mr0 + mr) + 4 * g + wr;
            A.out16[((size_t)net * B + r0 + row) * OUTP + cr] = o16[mr];
            _Float16 d16 = (_Float16)0.0f;
            if (net == 0) {
                const float mu = (float)o16[mr], a = in_a[mr];
                const float sq = red[mr][0], lsum = red[mr][1], bl = red[mr][2], kl = red[mr][3];
                const float nlp = 0.5f * sq + 0.5f * 1.8378770664093453f * (float)ACT + lsum;
                const float Ad = in_adv[mr], ratio = expf(in_nlp[mr] - nlp);
                const float rc = fminf(fmaxf(ratio, 1.0f - A.e_clip), 1.0f + A.e_clip);
                const float s1 = -Ad * ratio, sc = -Ad * rc;
                const bool inside = ratio >= 1.0f - A.e_clip && ratio <= 1.0f + A.e_clip;
                float w = s1 > sc ? 1.0f : (s1 == sc ? 0.5f : 0.0f);
                if (inside) w += sc > s1 ? 1.0f : (s1 == sc ? 0.5f : 0.0f);
                d16 = (_Float16)(ak ? scale * invB * (Ad * ratio * w) * (-(a - mu) / s2) : 0.0f);
                if (cr == 0) { st[0] += fmaxf(s1, sc); st[2] += bl; st[3] += fabsf(ratio - 1.0f) > A.e_clip ? 1.0f : 0.0f; st[4] += kl; }
            } else {
                const float v = (float)o16[mr], rt = in_ret[mr];
                d16 = (_Float16)(cr == 0 ? scale * invB * A.critic_coef * (v - rt) : 0.0f);
                if (cr == 0) st[1] += (rt - v) * (rt - v);
            }
            A.dout16[((size_t)net * B + r0 + row) * OUTP + cr] = d16;
            Ds[row * HS + cr] = d16;
            Ds[row * HS + 16 + cr] = (_Float16)0.0f;          // (k 16 .. 31 of the head's input-gradient product: zero)
            gsum += (float)d16;
        }
        // the head's bias gradient (column sums over my rows) and the logged sums, into the workgroup's row of accumulators: six values
        // reduced over the wave together
        float w6[6] = {st[0], st[1], st[2], st[3], st[4], 0.0f};
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1)
#pragma unroll
            for (int q = 0; q < 5; ++q) w6[q] += __shfl_xor(w6[q], o, 64);
        gsum += __shfl_xor(gsum, 16, 64); gsum += __shfl_xor(gsum, 32, 64);
        if (lane < OUTP && gsum != 0.0f) atomicAdd(&prow[PB_B3 + lane], gsum);
        if (lane < 5 && w6[lane == 0 ? 0 : (lane == 1 ? 1 : (lane == 2 ? 2 : (lane == 3 ? 3 : 4)))] != 0.0f) {
            const float t = lane == 0 ? w6[0] : (lane == 1 ? w6[1] : (lane == 2 ? w6[2] : (lane == 3 ? w6[3] : w6[4])));
            atomicAdd(&prow[PB_ST + lane], t);
        }
    }
    __syncthreads();
    if (A.doutf) frags_out<OUTP, HS>(Ds, A.doutf + ((size_t)net * gridDim.x + blockIdx.x) * OUTP * 32, tid);
    MLP_STAMP(7);
    // ---- gradient of the second hidden layer: dOut [rows x 16 (+16 zeros)] . W3 [16 x 256], relu mask, bias gradient ----
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        const h8 a = *reinterpret_cast<const h8 *>(Ds + (16 * mr + cr) * HS + 8 * g);          // (lanes with k = 16 .. 31: A is zero there)
#pragma unroll
        for (int t = 0; t < MNTW; ++t) acc[mr][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, w3t[t], (f4){0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
    }
    MLP_STAMP(8);
    ring_fill<HID, NTL, MNTW, MRD>(W2T, nt0, ring, lane);
    _Float16 *Z2 = Xs;          // [MT][HS]
    masked_out(acc, H2s, Z2, A.dz2f ? A.dz2f + fblk : nullptr, prow + PB_B2, nt0, cr, g, lane);
    __syncthreads();
    if (A.dz2) rows_out<HID, HS>(Z2, A.dz2 + ((size_t)net * B + r0) * HID, tid);
    // ---- gradient of the first hidden layer: dz2 [rows x 256] . W2 [256 x 256], relu mask, bias gradient ----
    MLP_STAMP(9);
    mfma_go<HID, HS, NTL, MNTW, MRD>(Z2, W2T, nt0, acc, ring, lane);
    MLP_STAMP(10);
    _Float16 *Z1 = Ds;          // (every wave has read its rows of dOut from it: the barrier above)
    if (A.dz1) {          // (the row-major copy: tests and the library-GEMM weight gradients; the operand-order copy needs no LDS image)
        masked_out(acc, H1s, Z1, A.dz1f ? A.dz1f + fblk : nullptr, prow + PB_B1, nt0, cr, g, lane);
        __syncthreads();
        rows_out<HID, HS>(Z1, A.dz1 + ((size_t)net * B + r0) * HID, tid);
    } else masked_out(acc, H1s, Z1, A.dz1f ? A.dz1f + fblk : nullptr, prow + PB_B1, nt0, cr, g, lane);
    MLP_STAMP(11);
}

// ------------------------------------------------------------------------------------------------ dwp_wgrad: the three weight gradients
// G[o][i] = sum over the samples of dz[s][o] act[s][i], per net, for the three layers, from dwp_mlp's operand-order copies (frags_out): a
// wave takes a block of MTB x NTB output tiles and a slab of the samples (the k index: one k-step = 32 samples = one workgroup of
// dwp_mlp), and stores its partial tiles in ITS SLAB's copy of the gradient (the readers add the slabs up).  Ring of requests and
// scheduling barriers as in mfma_rows.
template <int MTB, int NTB, int FA, int FB>          // FA / FB: features of the A / B operand buffers
__device__ __forceinline__ void wgrad_block(const _Float16 *__restrict__ Af, const _Float16 *__restrict__ Bf, int mt0, int nt0, int kb0, int kb1, float *__restrict__ G,
                                            int ld, int lane) {
    constexpr int RD = 5;
    f4 acc[MTB][NTB];
    h8 fa[RD][MTB], fb[RD][NTB];
    const h8 *pa = reinterpret_cast<const h8 *>(Af) + mt0 * 64 + lane, *pb = reinterpret_cast<const h8 *>(Bf) + nt0 * 64 + lane;
#pragma unroll
    for (int m = 0; m < MTB; ++m)
#pragma unroll
        for (int n = 0; n < NTB; ++n) acc[m][n] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
    auto request = [&](int slot, int kb) {
#pragma unroll
        for (int m = 0; m < MTB; ++m) fa[slot][m] = pa[((size_t)kb * (FA / 16) + m) * 64];
#pragma unroll
        for (int n = 0; n < NTB; ++n) fb[slot][n] = pb[((size_t)kb * (FB / 16) + n) * 64];
    };
    // (the slab has a multiple of RD k-steps or not: the ring is indexed by a counter that is static inside an unrolled group of RD)
    const int nk = kb1 - kb0;
#pragma unroll
    for (int i = 0; i < RD - 1; ++i) if (nk > i) request(i, kb0 + i);
    for (int k0 = 0; k0 < nk; k0 += RD) {
#pragma unroll
        for (int u = 0; u < RD; ++u) {
            const int k = k0 + u;
            if (k < nk) {
                if (k + RD - 1 < nk) request((u + RD - 1) % RD, kb0 + k + RD - 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MTB; ++m)
#pragma unroll
                    for (int n = 0; n < NTB; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[u][m], fb[u][n], acc[m][n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const int cr = lane & 15, g = lane >> 4;
#pragma unroll
    for (int m = 0; m < MTB; ++m)
#pragma unroll
        for (int n = 0; n < NTB; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) G[(size_t)(16 * (mt0 + m) + 4 * g + r) * ld + 16 * (nt0 + n) + cr] = acc[m][n][r];
}
struct WgradArgs { const _Float16 *xf, *h1f, *h2f, *doutf, *dz2f, *dz1f; const float *state; float *g32; int nkb; };
// tasks per net and slab: layer 1: 4 x 8 blocks of 4 x 4 tiles (256 x 512); layer 2: 4 x 4 blocks of 4 x 4 tiles (256 x 256); the heads: 4 blocks
// of 1 x 4 tiles (16 x 256)
constexpr int WG_T1 = 32, WG_T2 = 16, WG_T3 = 4, WG_TASKS = WG_T1 + WG_T2 + WG_T3;
__global__ __launch_bounds__(64) void k_wgrad(const WgradArgs A) {
    // blocks b and b + 8 land on one XCD (round-robin placement: for speed only): block b works on (net, slab) = b % 8, so that an XCD's waves
    // all read ONE net's fragments of ONE slab of samples -- 3 MB, which its 4 MB L2 then serves -- instead of every XCD pulling all 16 MB
    static_assert(2 * WG_SLABS == 8, "one (net, slab) pair per XCD");
    const int lane = threadIdx.x, task = blockIdx.x >> 3, net = blockIdx.x & 1, slab = (blockIdx.x >> 1) & (WG_SLABS - 1);
    const int per = (A.nkb + WG_SLABS - 1) / WG_SLABS, kb0 = slab * per < A.nkb ? slab * per : A.nkb, kb1 = kb0 + per < A.nkb ? kb0 + per : A.nkb;
    float *G = A.g32 + (size_t)slab * NWT;          // (a slab without samples -- tiny minibatches -- stores zeros)
    const size_t nb = (size_t)net * A.nkb;          // (net's first block in the per-net operand buffers)
    if (task < WG_T1) {
        wgrad_block<4, 4, HID, INP>(A.dz1f + nb * HID * 32, A.xf, 4 * (task >> 3), 4 * (task & 7), kb0, kb1, G + (size_t)net * HID * INP, INP, lane);
    } else if (task < WG_T1 + WG_T2) {
        const int t = task - WG_T1;
        wgrad_block<4, 4, HID, HID>(A.dz2f + nb * HID * 32, A.h1f + nb * HID * 32, 4 * (t >> 2), 4 * (t & 3), kb0, kb1, G + NW1 + (size_t)net * HID * HID, HID, lane);
    } else {
        const int t = task - WG_T1 - WG_T2;
        wgrad_block<1, 4, OUTP, HID>(A.doutf + nb * OUTP * 32, A.h2f + nb * HID * 32, 0, 4 * t, kb0, kb1, G + NW1 + NW2 + (size_t)net * OUTP * HID, HID, lane);
    }
}

// ------------------------------------------------------------------------------------------------ dwp_gae
// Generalised advantage estimation of one rollout (learning/rl_games_custom/a2c_common_dyros.py:485-500 `discount_values`): one thread per
// env walks its H steps backwards; tensors are [H][N] (step-major), so a wave's 64 envs read 256 consecutive bytes per step.  The reference
// is a Python loop of H iterations with eight torch kernels each.  Same arithmetic, same order per element (fp32).
__global__ __launch_bounds__(256) void k_gae(const float *__restrict__ fdones, const float *__restrict__ last_values, const float *__restrict__ mb_fdones,
                                             const float *__restrict__ mb_values, const float *__restrict__ mb_rewards, float gamma, float gamma_tau, int H, int N,
                                             float *__restrict__ advs) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= N) return;
    float lastgaelam = 0.0f, nextnonterminal = 1.0f - fdones[e], nextvalues = last_values[e];
    for (int t = H - 1; t >= 0; --t) {
        const size_t i = (size_t)t * N + e;
        const float val = mb_values[i];
        const float delta = mb_rewards[i] + gamma * nextvalues * nextnonterminal - val;
        lastgaelam = delta + gamma_tau * nextnonterminal * lastgaelam;          // (gamma * tau: a Python float product in the reference, formed on the host)
        advs[i] = lastgaelam;
        nextnonterminal = 1.0f - mb_fdones[i];
        nextvalues = val;
    }
}

// ------------------------------------------------------------------------------------------------ statistics + Adam + finish in ONE launch
// dwp_grad_stats and dwp_adam_finish read the same gradients twice and are a launch apart only because clip_grad_norm_ needs the norm over
// ALL of the actor's gradients before the first parameter moves.  Here every Adam block holds its eight parameters' gradients in registers
// and PUBLISHES its share of the norm and its inf / nan flags; every thread t of every block then waits for block t's share, and the block
// sums the 197 shares (always the same way) and steps its parameters.  The grid barrier is in the data:
//   * a share is ONE 64-bit word {partial sum, tag}, tag = the update's number (+ 1, 30 bits) | flags << 30, written with one agent-scope
//     store (an sc1 access: it lands at the memory side of the eight XCDs' L2s) and polled with agent-scope loads -- no counter, no
//     read-modify-write, no fence: a share is complete the moment its tag reads right.  What was measured on the way (profiles/r06_ppo_tail_forms.txt):
//     an arrival counter under agent-scope release / acquire fences costs +120 us per update (every block writes back and invalidates the whole L2);
//     the counter with waitcnt-only ordering, in one level or two, makes the launch exactly as long as the two it replaces (a store's
//     acknowledgement, one or two serialised atomics, a poll and the partials' read are five round trips to the memory side);
//   * 197 + 1 blocks of 256 threads on 256 CUs are co-resident whenever the launch has the device to itself (the update's launches are a chain
//     on one stream), which is what waiting on each other needs; the wait is nevertheless BOUNDED: a thread that does not see its share within
//     32 768 polls (tens of ms) takes the inf flags of both nets as set (its block skips its step), leaves a mark in part[642] and goes on, so the
//     grid always drains; the finishing block then backs the loss scale off and publishes DWP_S_OUT[7] = 2.  (Blocks that saw every share before
//     another gave up have stepped: a timed-out update is a fault to report, not a state to train on -- hence the mark.)
//   * the update's number lives in part[641] (as unsigned), advanced by the finishing block; nothing is reset between updates.
// Every block reads what it needs of `state` BEFORE it publishes (s_waitcnt vmcnt(0) in every thread), and the finishing block rewrites `state`
// only after it has seen every share.  The norm's partial sums are per Adam block here and per dwp_grad_stats block there: the same terms in
// another order, so norm2 -- and through the clip coefficient the actor's step -- may differ from the two-launch form in the last place.
constexpr int SA_BLOCKS = (NP / 8 + 255) / 256;          // Adam blocks; one more finishes the update
static_assert(SA_BLOCKS <= 256, "one share per thread");
constexpr int PART_BAR = 640;          // `part` word 641 (as unsigned): the update's number; 642: non-zero once a wait timed out
constexpr int PART_SHARES = 1024;          // `part` words [1024, 1024 + 2 x 256): the blocks' shares, 64 bits each
static_assert(DWP_PARTS >= PART_SHARES + 2 * 256 && PART_SHARES % 2 == 0, "part buffer");
template <int SLABS>
__global__ __launch_bounds__(256) void k_stats_adam(float *__restrict__ p, _Float16 *__restrict__ p16, float *__restrict__ m, float *__restrict__ v, float *__restrict__ gb,
                                                    float *__restrict__ state, float *__restrict__ part, float max_norm, _Float16 *__restrict__ p16f,
                                                    const float *__restrict__ g32, float *__restrict__ p32f, FinArgs fin, float *__restrict__ pbuf) {
    __shared__ float red[4];
    __shared__ int redf[4];
    __shared__ __attribute__((aligned(16))) float bsum[NBT];
    __shared__ f4 btail[8][PBK];
    unsigned *bar = reinterpret_cast<unsigned *>(part + PART_BAR);
    unsigned long long *shares = reinterpret_cast<unsigned long long *>(part + PART_SHARES);
    const bool finisher = blockIdx.x == SA_BLOCKS;
    // what every block needs of `state`, read BEFORE the barrier: the finishing block rewrites these words behind it
    const float scale = state[DWP_S_SCALE], g16r = state[DWP_S_G16];
    const float st0 = state[DWP_S_STEP], st1 = state[DWP_S_STEP + 1], lr0 = state[DWP_S_LR], lr1 = state[DWP_S_LR + 1];
    const unsigned gen = __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * 8, ic = (!finisher && i0 < NP) ? i0 : 0;
    const int net = net_of(ic);
    f4 pv[2] = {reinterpret_cast<const f4 *>(p + ic)[0], reinterpret_cast<const f4 *>(p + ic)[1]};
    f4 mv[2] = {reinterpret_cast<const f4 *>(m + ic)[0], reinterpret_cast<const f4 *>(m + ic)[1]};
    f4 vv[2] = {reinterpret_cast<const f4 *>(v + ic)[0], reinterpret_cast<const f4 *>(v + ic)[1]};
    float gs[8];
    if (blockIdx.x == SA_BLOCKS - 1 && pbuf) {
        // Bias gradients of the dwp_mlp path (all 1056 belong to this block's first 132 threads): the columns' sums over the 32 buckets, cleared
        // for the next update and left in gb.  By the WHOLE block, four columns per thread and every bucket row requested at once -- a thread
        // that walks its own eight columns pays eight round trips to memory one after the other while 197 blocks wait at the barrier (+25 us).
        // The last eight quads: one bucket row per thread, added up in bucket order from LDS.
        static_assert(NBT % 4 == 0 && NBT / 4 == 256 + 8 && PBK == 32 && PBW % 4 == 0 && SA_BLOCKS * 2048 >= NP && (SA_BLOCKS - 1) * 2048 == NWT, "the bias block");
        auto quad = [&](int j) -> float * {
            const int qq = 4 * j;
            const int col = qq < NB1 ? PB_B1 + qq % HID : (qq < NB1 + NB2 ? PB_B2 + (qq - NB1) % HID : PB_B3 + (qq - NB1 - NB2) % OUTP);
            return pbuf + (size_t)net_of(NWT + qq) * PBW + col;
        };
        float *pc = quad(threadIdx.x), *pt = quad(256 + (threadIdx.x >> 5)) + (size_t)(threadIdx.x & 31) * 2 * PBW;
        f4 a[PBK];
#pragma unroll
        for (int w = 0; w < PBK; ++w) a[w] = *reinterpret_cast<const f4 *>(pc + (size_t)w * 2 * PBW);
        btail[threadIdx.x >> 5][threadIdx.x & 31] = *reinterpret_cast<const f4 *>(pt);
        f4 s4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int w = 0; w < PBK; ++w) s4 += a[w];
        const f4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int w = 0; w < PBK; ++w) *reinterpret_cast<f4 *>(pc + (size_t)w * 2 * PBW) = z4;
        *reinterpret_cast<f4 *>(pt) = z4;
        *reinterpret_cast<f4 *>(bsum + 4 * threadIdx.x) = s4;
        *reinterpret_cast<f4 *>(gb + 4 * threadIdx.x) = s4;
        __syncthreads();
        if (threadIdx.x < 8) {
            f4 t4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int w = 0; w < PBK; ++w) t4 += btail[threadIdx.x][w];
            *reinterpret_cast<f4 *>(bsum + 4 * (256 + threadIdx.x)) = t4;
            *reinterpret_cast<f4 *>(gb + 4 * (256 + threadIdx.x)) = t4;
        }
        __syncthreads();
    }
    if (ic >= NWT) {
#pragma unroll
        for (int q = 0; q < 8; ++q) gs[q] = pbuf ? bsum[ic - NWT + q] : gb[ic - NWT + q];
    } else {
        f4 a[SLABS][2];
#pragma unroll
        for (int k = 0; k < SLABS; ++k) { a[k][0] = reinterpret_cast<const f4 *>(g32 + (size_t)k * NWT + ic)[0]; a[k][1] = reinterpret_cast<const f4 *>(g32 + (size_t)k * NWT + ic)[1]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) gs[q] = 0.0f;
#pragma unroll
        for (int k = 0; k < SLABS; ++k) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { gs[q] += a[k][0][q]; gs[4 + q] += a[k][1][q]; }
        }
        if (g16r != 0.0f) {
#pragma unroll
            for (int q = 0; q < 8; ++q) gs[q] = (float)(_Float16)gs[q];
        }
    }
    // ---- this block's share of the statistics
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (every word of `state` this thread asked for has arrived: see above)
    {
        const float inv = 1.0f / scale;
        float sq = 0.0f;
        int bad = 0;
        if (!finisher && i0 < NP) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { bad |= !isfinite(gs[q]); const float w = gs[q] * inv; sq += w * w; }
        }
        if (net != 0) sq = 0.0f;
        const float s = wave_sum(sq);
        const unsigned long long w0 = __ballot(bad && net == 0), w1 = __ballot(bad && net != 0);
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; redf[threadIdx.x >> 6] = (w0 ? 1 : 0) | (w1 ? 2 : 0); }
        __syncthreads();
        if (threadIdx.x == 0 && !finisher) {
            const unsigned tag = ((gen + 1u) & 0x3fffffffu) | ((unsigned)(redf[0] | redf[1] | redf[2] | redf[3]) << 30);
            const unsigned long long share = ((unsigned long long)tag << 32) | __float_as_uint(red[0] + red[1] + red[2] + red[3]);
            __hip_atomic_store(shares + blockIdx.x, share, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- the norm and the flags of the whole gradient: thread t waits for block t's share
    float my_part = 0.0f;
    int fl = 0;
    if (threadIdx.x < SA_BLOCKS) {
        const unsigned want = (gen + 1u) & 0x3fffffffu;
        int spins = 0;
        for (;;) {
            const unsigned long long w = __hip_atomic_load(shares + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned tag = (unsigned)(w >> 32);
            if ((tag & 0x3fffffffu) == want) { my_part = __uint_as_float((unsigned)w); fl = (int)(tag >> 30); break; }
            if (++spins > (1 << 15)) { fl = 4 | 3; break; }          // (never reached when the blocks are co-resident)
            __builtin_amdgcn_s_sleep(4);
        }
        if (fl & 4) __hip_atomic_store(part + PART_BAR + 2, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    {
        const float s = wave_sum(my_part);
        const unsigned long long w0 = __ballot(fl & 1), w1 = __ballot(fl & 2), w2 = __ballot(fl & 4);
        __syncthreads();          // (red / redf are reused)
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; redf[threadIdx.x >> 6] = (w0 ? 1 : 0) | (w1 ? 2 : 0) | (w2 ? 4 : 0); }
        __syncthreads();
    }
    const float norm2 = red[0] + red[1] + red[2] + red[3];
    const int flags = redf[0] | redf[1] | redf[2] | redf[3];
    const bool met = !(flags & 4);
    if (finisher) {
        if (threadIdx.x == 0) {
            state[DWP_S_NORM2] = norm2;
            if (flags & 1) state[DWP_S_FOUND_INF] = 1.0f;
            if (flags & 2) state[DWP_S_FOUND_INF + 1] = 1.0f;
            __hip_atomic_store(bar + 1, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        finish_update(state, fin.B, fin.nmb, fin.growth_interval, fin.pbuf);
        if (!met && threadIdx.x == 0) state[DWP_S_OUT + 7] = 2.0f;          // (the barrier timed out: that is why the update was skipped)
        return;
    }
    if (i0 >= NP) return;
    if ((flags >> net) & 1) return;          // GradScaler.step: this optimiser's step is skipped
    adam_apply(p, p16, m, v, p16f, p32f, i0, net, pv, mv, vv, gs, scale, (net ? st1 : st0) + 1.0f, net ? lr1 : lr0, norm2, max_norm);
}

// ------------------------------------------------------------------------------------------------ the rollout's bookkeeping around env.step
// What play_steps does between the policy's forward and the env step, and after it (learning/rl_games_custom/a2c_common_dyros.py:629-703):
// sample the action, its neglogp, the step's row of every rollout buffer; then the shaped reward with the time-out bootstrap, the logged
// reward terms, the new dones and observations.  ~30 torch kernels per step otherwise, for a few KB of arithmetic and two 32 MB copies.
struct RollPre { const float *mu, *value, *noise, *obs, *dones, *logstd; const long long *n; float *mb_obs, *mb_act, *mb_mu, *mb_nlp, *mb_val, *mb_done, *act; int N, nobs, env_major_steps, obs_half, H; };
__global__ __launch_bounds__(256) void k_roll_pre(const RollPre A) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, n = (size_t)*A.n, N = A.N;
    const size_t nf4 = N * A.nobs / 4;          // (N * nobs is a multiple of 4: checked by the launcher)
    // the row counter lives on the device (a replayed graph advances it): a caller that forgot to rewind it must not write past the H rows of
    // the rollout buffers.  Such a step records nothing and still hands the env its clipped action (checked on the host: RolloutRecorder.rows()).
    const bool in_rows = (long long)*A.n >= 0 && n < (size_t)A.H;
    if (!in_rows) {
        if (i < N * ACT) { const int k = (int)(i % ACT); A.act[i] = fminf(fmaxf(A.mu[i] + expf(A.logstd[k]) * A.noise[i], -1.0f), 1.0f); }
        return;
    }
    if (A.obs_half) {
        // mb_obs: halves [N][H][INP] -- what the update's first layer reads (autocast's cast of the Linear input, done here once instead of in
        // every one of the five passes over the batch); a thread = eight consecutive words of a row -> one 16-byte piece (the row's last
        // piece ends in the zero padding, which nobody else writes)
        const size_t ppr = (size_t)(A.nobs + 7) / 8;          // pieces per row
        if (i < N * ppr) {
            const size_t e = i / ppr, c = i - e * ppr;
            const float *src = A.obs + e * A.nobs + 8 * c;
            h8 v;
            if (8 * c + 8 <= (size_t)A.nobs) {
                // (two 16-byte requests on a 4-byte boundary: a wave reads 2 KB in one piece instead of eight strided words per lane)
                typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                const f4u lo = *reinterpret_cast<const f4u *>(src), hi = *reinterpret_cast<const f4u *>(src + 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q] = (_Float16)lo[q]; v[4 + q] = (_Float16)hi[q]; }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = 8 * c + q < (size_t)A.nobs ? (_Float16)src[q] : (_Float16)0.0f;
            }
            *reinterpret_cast<h8 *>(reinterpret_cast<_Float16 *>(A.mb_obs) + (e * (size_t)A.env_major_steps + n) * INP + 8 * c) = v;
        }
    } else if (A.env_major_steps) {
        // mb_obs [N][H][nobs], the flat batch of the update (swap_and_flatten01 done as the rollout goes): four words of one row per thread
        // (a row of 487 words starts on a 4-byte boundary only)
        if (i < nf4) {
            const f4 v = reinterpret_cast<const f4 *>(A.obs)[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t w = 4 * i + q, e = w / (size_t)A.nobs, k = w - e * A.nobs;
                A.mb_obs[(e * (size_t)A.env_major_steps + n) * A.nobs + k] = v[q];
            }
        }
    } else if (i < nf4) reinterpret_cast<f4 *>(A.mb_obs + n * N * A.nobs)[i] = reinterpret_cast<const f4 *>(A.obs)[i];
    if (i < N * ACT) {
        const int k = (int)(i % ACT);
        const float m_ = A.mu[i], a = m_ + expf(A.logstd[k]) * A.noise[i];          // mu + sigma * randn
        A.mb_act[n * N * ACT + i] = a; A.mb_mu[n * N * ACT + i] = m_;
        A.act[i] = fminf(fmaxf(a, -1.0f), 1.0f);          // clip_actions
    }
    if (i < N) {
        float sq = 0.0f, lsum = 0.0f;
        for (int k = 0; k < ACT; ++k) {
            const float ls = A.logstd[k], sg = expf(ls), m_ = A.mu[i * ACT + k], a = m_ + sg * A.noise[i * ACT + k];
            const float z = (a - m_) / sg;
            sq += z * z; lsum += ls;
        }
        A.mb_nlp[n * N + i] = 0.5f * sq + 0.5f * 1.8378770664093453f * (float)ACT + lsum;          // models_dyros.py:59-62
        A.mb_val[n * N + i] = A.value[i];
        A.mb_done[n * N + i] = A.dones[i];
    }
}
constexpr int ROLL_TERMS_MAX = DWP_ROLL_TERMS_MAX;          // logged reward columns dwp_rollout_post reduces (one thread each)
struct RollPost { const float *rew, *value, *stacked, *new_obs; const long long *time_outs, *done, *n; float *mb_rew, *terms, *g_dones, *g_obs; int N, nobs, nterms, ncols; float scale, gamma; int H; };
__global__ __launch_bounds__(256) void k_roll_post(const RollPost A) {
    __shared__ float red[4][ROLL_TERMS_MAX];
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, n = (size_t)*A.n, N = A.N;
    const size_t nf4 = N * A.nobs / 4;
    const bool in_rows = (long long)*A.n >= 0 && n < (size_t)A.H;          // (as k_roll_pre: a row past the buffers is not written)
    if (A.g_obs != A.new_obs && i < nf4) reinterpret_cast<f4 *>(A.g_obs)[i] = reinterpret_cast<const f4 *>(A.new_obs)[i];
    if (i < N) {
        float r = A.rew[i] * A.scale;
        if (A.time_outs) r = r + A.gamma * A.value[i] * (float)A.time_outs[i];          // value_bootstrap (:656-659)
        if (in_rows) A.mb_rew[n * N + i] = r;
        A.g_dones[i] = (float)A.done[i];
    }
    // the logged reward terms: mean over the envs of the first nterms columns, added to the epoch's sums (blocks that hold envs only).  With a
    // terrain curriculum the env reports 15 + terrain types columns (tasks/dyros_dynamic_walk.py:417-421): up to ROLL_TERMS_MAX of them
    if ((size_t)blockIdx.x * 256 < N && A.terms && in_rows) {
        for (int c = 0; c < A.nterms && c < ROLL_TERMS_MAX; ++c) {
            const float t = wave_sum(i < N ? A.stacked[i * A.ncols + c] : 0.0f);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c] = t;
        }
        __syncthreads();
        if (threadIdx.x < A.nterms && threadIdx.x < ROLL_TERMS_MAX) atomicAdd(&A.terms[threadIdx.x], (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / (float)N);
    }
}

// ------------------------------------------------------------------------------------------------ dwp_policy: the rollout's forward in fp32
// mu [N][13] and value [N] of both nets for N observations, fp32 throughout as the reference's get_action_values (no autocast there), on
// v_mfma_f32_16x16x4_f32: workgroup of four waves = 32 samples through one net, activations in LDS with the k index permuted so that a
// lane's four k-steps are 16 consecutive bytes, weights from the fp32 fragment-order copy, products split by columns, ring of requests.
constexpr int XS32 = INP + 4, HS32 = HID + 4;          // LDS row strides in floats
__device__ __forceinline__ int perm16(int k) { return (k & ~15) + 4 * (k & 3) + ((k >> 2) & 3); }          // position of k inside its group of 16
template <int KG, int AS, int NTA, int NT, int RD>
__device__ __forceinline__ void mfma32_rows(const float *As, const float *__restrict__ W, int nt0, f4 (&acc)[MR][NT], int lane) {
    const int ar = lane & 15, g = lane >> 4;
    const f4 *wl = reinterpret_cast<const f4 *>(W) + nt0 * 64 + lane;
    f4 b[RD][NT];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[mr][t] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int pk = 0; pk < RD - 1 && pk < KG; ++pk)
#pragma unroll
        for (int t = 0; t < NT; ++t) b[pk][t] = wl[(pk * NTA + t) * 64];
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
        if (kg + RD - 1 < KG) {
#pragma unroll
            for (int t = 0; t < NT; ++t) b[(kg + RD - 1) % RD][t] = wl[((kg + RD - 1) * NTA + t) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        f4 a[MR];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) a[mr] = *reinterpret_cast<const f4 *>(As + (16 * mr + ar) * AS + 16 * kg + 4 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[mr][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mr][j], b[kg % RD][t][j], acc[mr][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// dwp_policy's workgroup: PWPB waves, two per SIMD
constexpr int PWPB = 8, PNTW = NTL / PWPB;
__global__ __launch_bounds__(64 * PWPB) void k_policy(const float *__restrict__ obs, const float *__restrict__ p, const float *__restrict__ p32f, int N,
                                                     float *__restrict__ mu, float *__restrict__ value) {
    __shared__ float Xs[MT * XS32];          // the input rows; after the first layer: the second hidden layer
    __shared__ float H1s[MT * HS32];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, net = blockIdx.y, r0 = blockIdx.x * MT;
    const int nt0 = PNTW * wv, cr = lane & 15, g = lane >> 4;
    const float *W1 = p32f + G_W1 + (size_t)net * HID * INP, *W2 = p32f + G_W2 + (size_t)net * HID * HID, *W3 = p32f + G_W3 + (size_t)net * OUTP * HID;
    const float *b1 = p + NWT + net * HID, *b2 = p + NWT + NB1 + net * HID, *b3 = p + NWT + NB1 + NB2 + net * OUTP;
    {
        const float *src = obs + (size_t)r0 * IN;
        constexpr int CPT = INP / (64 * PWPB);
        static_assert(INP == CPT * 64 * PWPB, "whole column slots per thread");
        const int rmax = N - 1 - r0;          // (rows past the end -- N is not a multiple of 32 -- read the last row again; nothing of theirs is stored)
        float v[CPT][MT];
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int c = tid + 64 * PWPB * u, cl = c < IN ? c : 0;
#pragma unroll
            for (int r = 0; r < MT; ++r) { const int rr = r < rmax ? r : rmax; v[u][r] = src[(size_t)rr * IN + cl]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int c = tid + 64 * PWPB * u, q = perm16(c);
#pragma unroll
            for (int r = 0; r < MT; ++r) Xs[r * XS32 + q] = c < IN ? v[u][r] : 0.0f;
        }
        __syncthreads();
    }
    f4 acc[MR][PNTW];
    float bia[PNTW];
#pragma unroll
    for (int t = 0; t < PNTW; ++t) bia[t] = b1[16 * (nt0 + t) + cr];
    mfma32_rows<INP / 16, XS32, NTL, PNTW, 3>(Xs, W1, nt0, acc, lane);
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int t = 0; t < PNTW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) H1s[(16 * mr + 4 * g + r) * HS32 + perm16(16 * (nt0 + t) + cr)] = fmaxf(acc[mr][t][r] + bia[t], 0.0f);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < PNTW; ++t) bia[t] = b2[16 * (nt0 + t) + cr];
    mfma32_rows<HID / 16, HS32, NTL, PNTW, 3>(H1s, W2, nt0, acc, lane);
    float *H2s = Xs;          // [MT][HS32]
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int t = 0; t < PNTW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) H2s[(16 * mr + 4 * g + r) * HS32 + perm16(16 * (nt0 + t) + cr)] = fmaxf(acc[mr][t][r] + bia[t], 0.0f);
    __syncthreads();
    // the head: one column tile (16 padded outputs); waves 0 and 1 take one row tile each
    if (wv < MR) {
        const int ar = lane & 15;
        const f4 *wl = reinterpret_cast<const f4 *>(W3) + lane;
        f4 o = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int kg = 0; kg < HID / 16; ++kg) {
            const f4 bq = wl[kg * 64];
            const f4 aq = *reinterpret_cast<const f4 *>(H2s + (16 * wv + ar) * HS32 + 16 * kg + 4 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) o = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[j], bq[j], o, 0, 0, 0);
        }
        const float bias = b3[cr];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = r0 + 16 * wv + 4 * g + r;
            if (row < N) {
                if (net == 0) { if (cr < ACT) mu[(size_t)row * ACT + cr] = o[r] + bias; }
                else if (cr == 0) value[row] = o[r] + bias;
            }
        }
    }
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

static bool slabs_ok(const float *g32, int32_t g32_slabs) { return !g32 || g32_slabs == WG_SLABS || g32_slabs == 1; }

int dwp_grad_stats(const uint16_t *g16, float *gb, float *state, float *part, float *pbuf, const float *g32, int32_t g32_slabs, void *stream) {
    if ((!g16 && !g32) || !gb || !state || !part || !slabs_ok(g32, g32_slabs)) return fail("dwp_grad_stats: bad argument");
    if (g32 && g32_slabs == 1) hipLaunchKernelGGL(k_grad_stats<1>, dim3(GS_BLOCKS), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)g16, gb, state, part, pbuf, g32);
    else hipLaunchKernelGGL(k_grad_stats<WG_SLABS>, dim3(GS_BLOCKS), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)g16, gb, state, part, pbuf, g32);
    return done("dwp_grad_stats");
}

int dwp_grad_bucket(const float *g32, float *pbuf, float *bucket, float inv_world, void *stream) {
    if (!g32 || !pbuf || !bucket || !(inv_world > 0.0f) || inv_world > 1.0f) return fail("dwp_grad_bucket: bad argument");
    hipLaunchKernelGGL(k_grad_bucket, dim3((NWT / 4 + NBT + 255) / 256), dim3(256), 0, (hipStream_t)stream, g32, pbuf, bucket, inv_world);
    return done("dwp_grad_bucket");
}

int dwp_adam(float *p, uint16_t *p16, float *m, float *v, const uint16_t *g16, const float *gb, float *state, const float *part, float max_norm, uint16_t *p16t,
             const float *g32, int32_t g32_slabs, float *p32f, void *stream) {
    if (!p || !p16 || !m || !v || (!g16 && !g32) || !gb || !state || !part || !slabs_ok(g32, g32_slabs)) return fail("dwp_adam: bad argument");
    if (g32 && g32_slabs == 1)
        hipLaunchKernelGGL((k_adam<false, 1>), dim3((NP / 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, (_Float16 *)p16, m, v, (const _Float16 *)g16, gb, state, part, max_norm,
                           (_Float16 *)p16t, g32, p32f, FinArgs{0, 0, 0, nullptr});
    else
        hipLaunchKernelGGL((k_adam<false, WG_SLABS>), dim3((NP / 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, (_Float16 *)p16, m, v, (const _Float16 *)g16, gb, state, part, max_norm,
                           (_Float16 *)p16t, g32, p32f, FinArgs{0, 0, 0, nullptr});
    return done("dwp_adam");
}

int dwp_adam_finish(float *p, uint16_t *p16, float *m, float *v, const float *gb, float *state, const float *part, float max_norm, uint16_t *p16t, const float *g32,
                    int32_t g32_slabs, float *p32f, int32_t B, int32_t num_minibatches, int32_t growth_interval, float *pbuf, void *stream) {
    if (!p || !p16 || !m || !v || !g32 || !gb || !state || !part || !pbuf || B < 1 || num_minibatches < 1 || growth_interval < 1 || !slabs_ok(g32, g32_slabs))
        return fail("dwp_adam_finish: bad argument");
    if (g32_slabs == 1)
        hipLaunchKernelGGL((k_adam<true, 1>), dim3((NP / 8 + 255) / 256 + 1), dim3(256), 0, (hipStream_t)stream, p, (_Float16 *)p16, m, v, (const _Float16 *)nullptr, gb, state, part, max_norm,
                           (_Float16 *)p16t, g32, p32f, FinArgs{B, num_minibatches, growth_interval, pbuf});
    else
        hipLaunchKernelGGL((k_adam<true, WG_SLABS>), dim3((NP / 8 + 255) / 256 + 1), dim3(256), 0, (hipStream_t)stream, p, (_Float16 *)p16, m, v, (const _Float16 *)nullptr, gb, state, part, max_norm,
                           (_Float16 *)p16t, g32, p32f, FinArgs{B, num_minibatches, growth_interval, pbuf});
    return done("dwp_adam_finish");
}

int dwp_stats_adam_finish(float *p, uint16_t *p16, float *m, float *v, float *gb, float *state, float *part, float max_norm, uint16_t *p16t, const float *g32,
                          int32_t g32_slabs, float *p32f, int32_t B, int32_t num_minibatches, int32_t growth_interval, float *pbuf, float *pbuf_bias, void *stream) {
    if (!p || !p16 || !m || !v || !g32 || !gb || !state || !part || !pbuf || B < 1 || num_minibatches < 1 || growth_interval < 1 || !slabs_ok(g32, g32_slabs))
        return fail("dwp_stats_adam_finish: bad argument");
    if (g32_slabs == 1)
        hipLaunchKernelGGL((k_stats_adam<1>), dim3(SA_BLOCKS + 1), dim3(256), 0, (hipStream_t)stream, p, (_Float16 *)p16, m, v, gb, state, part, max_norm, (_Float16 *)p16t, g32, p32f,
                           FinArgs{B, num_minibatches, growth_interval, pbuf}, pbuf_bias);
    else
        hipLaunchKernelGGL((k_stats_adam<WG_SLABS>), dim3(SA_BLOCKS + 1), dim3(256), 0, (hipStream_t)stream, p, (_Float16 *)p16, m, v, gb, state, part, max_norm, (_Float16 *)p16t, g32, p32f,
                           FinArgs{B, num_minibatches, growth_interval, pbuf}, pbuf_bias);
    return done("dwp_stats_adam_finish");
}

int dwp_wgrad(const uint16_t *xf, const uint16_t *h1f, const uint16_t *h2f, const uint16_t *doutf, const uint16_t *dz2f, const uint16_t *dz1f, const float *state,
              float *g32, int32_t B, void *stream) {
    if (!xf || !h1f || !h2f || !doutf || !dz2f || !dz1f || !state || !g32 || B < 32 || B % 32) return fail("dwp_wgrad: bad argument");
    WgradArgs A{(const _Float16 *)xf, (const _Float16 *)h1f, (const _Float16 *)h2f, (const _Float16 *)doutf, (const _Float16 *)dz2f, (const _Float16 *)dz1f, state, g32, B / 32};
    hipLaunchKernelGGL(k_wgrad, dim3(WG_TASKS * 2 * WG_SLABS), dim3(64), 0, (hipStream_t)stream, A);
    return done("dwp_wgrad");
}

int dwp_finish(float *state, float *gb, int32_t B, int32_t num_minibatches, int32_t growth_interval, float *pbuf, void *stream) {
    if (!state || !gb || B < 1 || num_minibatches < 1 || growth_interval < 1) return fail("dwp_finish: bad argument");
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(256), 0, (hipStream_t)stream, state, gb, B, num_minibatches, growth_interval, pbuf);
    return done("dwp_finish");
}

int dwp_gae(const float *fdones, const float *last_values, const float *mb_fdones, const float *mb_values, const float *mb_rewards, float gamma, float tau, int32_t H,
            int32_t N, float *advs, void *stream) {
    if (!fdones || !last_values || !mb_fdones || !mb_values || !mb_rewards || !advs || H < 1 || N < 1) return fail("dwp_gae: bad argument");
    hipLaunchKernelGGL(k_gae, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, fdones, last_values, mb_fdones, mb_values, mb_rewards, gamma,
                       (float)((double)gamma * (double)tau), H, N, advs);
    return done("dwp_gae");
}

int dwp_rollout_pre(const float *mu, const float *value, const float *noise, const float *obs, const float *dones, const float *logstd, const int64_t *n, int32_t N,
                    int32_t num_obs, float *mb_obs, float *mb_act, float *mb_mu, float *mb_nlp, float *mb_val, float *mb_done, float *act, int32_t env_major_steps,
                    int32_t obs_half, int32_t H, void *stream) {
    if (!mu || !value || !noise || !obs || !dones || !logstd || !n || !mb_obs || !mb_act || !mb_mu || !mb_nlp || !mb_val || !mb_done || !act || N < 1 || num_obs < ACT ||
        env_major_steps < 0 || (obs_half && (env_major_steps < 1 || num_obs > INP)) || H < 1 || (env_major_steps && env_major_steps != H))
        return fail("dwp_rollout_pre: bad argument");
    if (((size_t)N * num_obs) % 4) return fail("dwp_rollout_pre: N * num_obs must be a multiple of 4");
    RollPre A{mu, value, noise, obs, dones, logstd, (const long long *)n, mb_obs, mb_act, mb_mu, mb_nlp, mb_val, mb_done, act, N, num_obs, env_major_steps, obs_half, H};
    hipLaunchKernelGGL(k_roll_pre, dim3((unsigned)(((size_t)N * num_obs / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A);
    return done("dwp_rollout_pre");
}

int dwp_rollout_post(const float *rew, const float *value, const int64_t *time_outs, const float *stacked, int32_t stacked_cols, const int64_t *done_buf, const float *new_obs,
                     const int64_t *n, int32_t N, int32_t num_obs, float reward_scale, float gamma, float *mb_rew, float *terms, int32_t num_terms, float *g_dones,
                     float *g_obs, int32_t H, void *stream) {
    if (!rew || !value || !done_buf || !new_obs || !n || !mb_rew || !g_dones || !g_obs || N < 1 || H < 1 ||
        (terms && (!stacked || num_terms < 1 || num_terms > ROLL_TERMS_MAX || stacked_cols < num_terms)))
        return fail("dwp_rollout_post: bad argument");
    if (((size_t)N * num_obs) % 4) return fail("dwp_rollout_post: N * num_obs must be a multiple of 4");
    RollPost A{rew, value, stacked, new_obs, (const long long *)time_outs, (const long long *)done_buf, (const long long *)n, mb_rew, terms, g_dones, g_obs, N, num_obs,
               num_terms, stacked_cols, reward_scale, gamma, H};
    // (without the observation copy -- the caller's policy reads the env's own buffer -- the launch covers the envs only)
    const size_t work = g_obs != new_obs ? (size_t)N * num_obs / 4 : (size_t)N;
    hipLaunchKernelGGL(k_roll_post, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A);
    return done("dwp_rollout_post");
}

int dwp_retile(const uint16_t *p16, uint16_t *p16f, void *stream) {
    if (!p16 || !p16f) return fail("dwp_retile: bad argument");
    hipLaunchKernelGGL(k_retile, dim3((NWT + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)p16, (_Float16 *)p16f);
    return done("dwp_retile");
}

int dwp_retile32(const float *p, float *p32f, void *stream) {
    if (!p || !p32f) return fail("dwp_retile32: bad argument");
    hipLaunchKernelGGL(k_retile32, dim3((NWT + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, p32f);
    return done("dwp_retile32");
}

int dwp_policy(const float *obs, const float *p, const float *p32f, int32_t N, float *mu, float *value, void *stream) {
    if (!obs || !p || !p32f || !mu || !value || N < 1) return fail("dwp_policy: bad argument");
    hipLaunchKernelGGL(k_policy, dim3((N + MT - 1) / MT, 2), dim3(64 * PWPB), 0, (hipStream_t)stream, obs, p, p32f, N, mu, value);
    return done("dwp_policy");
}

int dwp_sizeof_mlp(void) { return (int)sizeof(DwpMlp); }

int dwp_mlp(const DwpMlp *a, void *stream) {
    if (!a) return fail("dwp_mlp: null argument");
    const void *need[] = {a->obs ? (const void *)a->obs : (const void *)a->obs16, a->state, a->act, a->old_nlp, a->old_mu, a->adv, a->ret, a->logstd, a->p16, a->p16t, a->pbuf, a->out16, a->dout16};
    if (!a->xf && !a->x16) return fail("dwp_mlp: neither row-major nor operand-order outputs");
    for (const void *q : need) if (!q) return fail("dwp_mlp: null pointer in the argument block");
    if (a->B < MT || a->B % MT) return fail("dwp_mlp: the minibatch must be a multiple of 32 samples");
    MlpArgs A;
    A.obs = a->obs; A.obs16 = a->obs ? nullptr : (const _Float16 *)a->obs16; A.state = a->state; A.act = a->act; A.old_nlp = a->old_nlp; A.old_mu = a->old_mu; A.adv = a->adv; A.ret = a->ret; A.logstd = a->logstd;
    A.p16 = (const _Float16 *)a->p16; A.p16t = (const _Float16 *)a->p16t; A.pbuf = a->pbuf;
    A.x16 = (_Float16 *)a->x16; A.h1 = (_Float16 *)a->h1; A.h2 = (_Float16 *)a->h2; A.out16 = (_Float16 *)a->out16; A.dout16 = (_Float16 *)a->dout16;
    A.dz2 = (_Float16 *)a->dz2; A.dz1 = (_Float16 *)a->dz1; A.B = a->B;
    A.xf = (_Float16 *)a->xf; A.h1f = (_Float16 *)a->h1f; A.h2f = (_Float16 *)a->h2f; A.doutf = (_Float16 *)a->doutf; A.dz2f = (_Float16 *)a->dz2f; A.dz1f = (_Float16 *)a->dz1f; A.e_clip = a->e_clip; A.critic_coef = a->critic_coef;
    hipLaunchKernelGGL(k_mlp, dim3(a->B / MT, 2), dim3(64 * MWPB), 0, (hipStream_t)stream, A);
    return done("dwp_mlp");
}

}  // extern "C"
