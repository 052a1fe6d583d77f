// qs_policy_encoder.hip - fused forward pass of the quad-swarm policy encoder on gfx950 matrix cores.
//
// Reference: swarm_rl/models/quad_multi_model.py:250-350 (QuadMultiEncoder: self encoder, neighbour encoder, optional obstacle
// encoder, feed-forward) with the `mean_embed` neighbour encoder (:22-43, QuadNeighborhoodEncoderDeepsets); tanh non-linearity,
// hidden size 256.  It reads the observation rows straight from the stepper's buffer.
// The `attention` neighbour encoder (:46-101) takes two kernels: its `self_obs.repeat(K, 1)` (:84) and `mean.repeat(K, 1)` (:92)
// tile the WHOLE batch, so row (agent a, neighbour k) is paired with the self observation and the mean embedding of agent
// (a*K + k) mod batch - an inter-agent dependence on an intermediate result.  qs_encoder_embed_kernel computes e_i for every row
// (with that self observation) and g = W_m e_mean per agent; qs_encoder_kernel then runs value MLP, score MLP (W_e e_i + g[(a*K+k)
// mod B] + b), softmax over the neighbours and the weighted sum, all lane-local across accumulator tiles.
//
// One workgroup = 8 waves = 16 agents.  Every layer is a transposed GEMM on v_mfma_f32_16x16x32_bf16:
//     C[feature][row] = sum_k W[feature][k] * X[row][k]
//   A operand = weights, packed on the host in fragment order ([M/16][K/32][64 lanes][8 bf16]) and streamed from L2 with one
//               16-byte load per lane;
//   B operand = activations, row-major bf16 in LDS with K contiguous (one ds_read_b128 per lane);
//   C         = 4 consecutive features of one row per lane  ->  bias, tanh, bf16, one 8-byte LDS store.
// Wave w owns features [32w, 32w+32) of a 256-wide layer.  Neighbour rows are ordered neighbour-major (row = k*16 + agent), so
// the row tile index IS the neighbour index and the mean over neighbours is a lane-local sum across accumulator tiles.
// Operand layout verified on hardware by tools/mfma_layout_probe.hip.
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdlib.h>
#include <stdint.h>

#include <string>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

#define ENC_H 256            // hidden size of every MLP (rnn_size = neighbor_hidden_size = obst_hidden_size = 256)
#define ENC_TA 16            // agents per workgroup = one 16-row tile
#define ENC_MAX_NBR 8        // neighbours per agent (6 or 2 in the reference's configurations)
#ifndef ENC_NH
#define ENC_NH (ENC_MAX_NBR / 2)   // neighbour row tiles per pass of the neighbour MLP
#endif
#ifndef ENC_WAVES
// 2 waves per SIMD: the layer chain of one workgroup is latency-bound, a second wave hides part of it (49 -> 40 us at 8192 agents)
#define ENC_WAVES 8
#endif
#ifndef ENC_OCC
#define ENC_OCC 4     // waves per SIMD the register budget is set for: two workgroups per CU (<= 128 VGPRs)
#endif
#define ENC_MT (16 / ENC_WAVES)       // 16-feature tiles of a 256-wide layer per wave
#define ENC_MTF (32 / ENC_WAVES)      // ... of the 512-wide feed-forward layer
#define ENC_XS 40            // row stride (bf16) of the 32-wide input staging rows  (+8 pad: spreads the LDS banks)
#define ENC_YS (ENC_H + 8)   // row stride of a 256-wide activation buffer
#define ENC_CS (3 * ENC_H + 8)

enum { ENC_NBR_MEAN_EMBED = 0, ENC_NBR_ATTENTION = 1, ENC_NBR_MLP = 2, ENC_NBR_NONE = 3, ENC_MODEL_MHA = 4, ENC_MODEL_S2R = 5 };
#define ENC_XW 72   // row stride of the mlp neighbour encoder's input rows (all neighbours of one agent, K padded to 64)
struct EncLayer { const uint16_t *w; const float *b; int32_t M, K; };   // K padded to a multiple of 32, M to a multiple of 16
struct EncParams {
    int32_t self_dim, nbr_dim, num_nbr, obst_dim, obs_dim;
    int32_t nbr_encoder;        // ENC_NBR_*: mean_embed (:22-43), attention (:46-101), mlp (:104-122), no_encoder (:289-291); ENC_MODEL_MHA
    EncLayer s1, s2;            // self encoder        :303-309
    // neighbour embedding :29-34 (input = neighbour obs) / :52-57 (attention: input = [self obs | neighbour obs])
    EncLayer n1, n2, n3;
                                // / :110-117 (mlp: input = all neighbour obs of the agent, three layers)
    EncLayer o1, o2;            // obstacle encoder    :315-322
    EncLayer v1, v2;            // attention: value MLP :60-65
    EncLayer a1e, a1m, a2;      // attention: score MLP :68-75; its first layer split into the e_i half (with the bias) and the e_mean half
    const float *a3w;           // attention: last score layer 256 -> 1, fp32 weight row [256] ...
    float a3b;                  // ... and bias
    int32_t precision;          // 0: bf16 operands; 1: reference precision - every operand as a pair of fp16 (see "Reference precision" below)
    uint16_t *ebuf;             // attention scratch: e_i of every (agent, neighbour) row, bf16 [B*num_nbr, 256]
    float *gbuf;                // attention scratch: W_m e_mean of every agent, fp32 [B, 256]
    EncLayer f;                 // feed forward        :329-332
    // QuadMultiHeadAttentionEncoder (:124-196, nbr_encoder == ENC_MODEL_MHA): n1 / n2 = neighbor_embed_layer on all neighbour
    // columns, o1 / o2 = obstacle_embed_layer; MultiHeadAttention(4, 256, 256, 256) (attention_layer.py:12-56) over the token pair.
    // QuadSingleHeadAttentionEncoder_Sim2Real (:199-248, ENC_MODEL_S2R): one-layer embeddings (s1, n1, o1; the second layers are
    // not read), OneHeadAttention(256) (attention_layer.py:56-97), feed forward 768 -> 256
    EncLayer mq, mk, mv;        // w_qs, w_ks, w_vs: 256 -> heads x 256, no bias (bias pointer not read)
    EncLayer mfc;               // fc: heads x 256 -> 256, no bias
    const float *ln_w, *ln_b;   // LayerNorm(256, eps 1e-6) weight / bias
    // optional linear head on the encoder output, fused into the epilogue: head_out[B, head_dim] = out . head_w^T + head_b
    const float *head_w, *head_b;   // fp32 [head_dim, 512], [head_dim]
    float *head_out;
    int32_t head_dim;               // 0: no head; <= 8
    // optional Gaussian sampling on the head's output (rollout segments): act_out[B, head_dim] = head_out + exp(sample_log_std) * N(0, 1),
    // Philox4x32-10 keyed (seed, *sample_counter + sample_step, agent) - the draws of qs_rollout_pre with that counter value
    uint32_t sample_step;
    const float *sample_log_std;    // fp32 [head_dim]; NULL: no sampling
    float *act_out;                 // fp32 [B, head_dim]
    const uint32_t *sample_counter;
    uint32_t sample_seed_lo, sample_seed_hi;
    // optional trajectory copy (rollout segments): the first kernel of a forward pass also copies the reward and the done flag of every
    // agent of its workgroup - the outputs of the environment step that produced THESE observations - to traj_rew_dst / traj_done_dst
    const float *traj_rew_src;
    float *traj_rew_dst;
    const uint8_t *traj_done_src;
    uint8_t *traj_done_dst;
};

// (rollout segments) reward / done of the previous control step -> trajectory, by the workgroup that owns the agents [a0, a0 + agents)
__device__ __forceinline__ void traj_copy(const EncParams &P, int a0, int agents, int B) {
    if (P.traj_rew_dst != nullptr) {
        const int a = a0 + (int)threadIdx.x;
        if ((int)threadIdx.x < agents && a < B) { P.traj_rew_dst[a] = P.traj_rew_src[a]; P.traj_done_dst[a] = P.traj_done_src[a]; }
    }
}

#ifdef ENC_TIMING   // phase stamps of workgroup 0, wave 0 (tools/enc_quick.py prints them)
__device__ unsigned long long enc_stamps[16];
__device__ unsigned long long enc_wg_times[2 * 8192];   // start / end of every workgroup on the constant 100 MHz clock
#define ENC_STAMP(k) do { if (threadIdx.x == 0) { if (blockIdx.x == 0) enc_stamps[k] = clock64(); \
                                                  if (((k) == 0 || (k) == 9) && blockIdx.x < 8192) enc_wg_times[2 * blockIdx.x + ((k) == 9)] = wall_clock64(); } } while (0)
#else
#define ENC_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ void glue_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
    uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0, h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// action = mean + exp(log_std) * N(0, 1) for component h of agent a: Box-Muller on the Philox group (agent, counter, 0x51, h / 4) - the
// values qs_rollout_pre_kernel draws (one group = two pairs = four components)
__device__ __forceinline__ float sample_action(const EncParams &P, int a, int h, float mean) {
    uint32_t w[4];
    glue_philox((uint32_t)a, *P.sample_counter + P.sample_step, 0x51u, (uint32_t)(h >> 2), P.sample_seed_lo, P.sample_seed_hi, w);
    const int pr = (h >> 1) & 1;
    const float ua = ((float)(w[2 * pr] >> 9) + 0.5f) * (1.0f / 8388608.0f),
        ub = ((float)(w[2 * pr + 1] >> 9) + 0.5f) * (1.0f / 8388608.0f);
    const float r = sqrtf(-2.0f * __logf(ua));
    float sn, cs;
    __sincosf(6.283185307179586f * ub, &sn, &cs);
    return mean + __expf(P.sample_log_std[h]) * r * ((h & 1) ? sn : cs);
}

__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }   // in a scalar register

__device__ __forceinline__ float fast_tanh(float x) {   // 1 - 2 / (exp(2x) + 1); v_exp_f32 + v_rcp_f32
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);   // exp(2x): one multiply, v_exp_f32
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
// The same on two values at once.  A wave issues one VALU instruction every ~5 cycles whatever it is (tools/ubench_valu.hip: v_fma_f32 6.6,
// v_pk_fma_f32 7.0, v_exp_f32 / v_rcp_f32 9-10 ticks per instruction for a wave alone, unchanged with a second wave on the SIMD), so
// the epilogues are bound by their instruction COUNT: the plain half of the tanh as v_pk_* (one issue per two values), and in
// tanh2_bias the bias add and the scale of the exponent in one v_pk_fma_f32:  2^(c (a + b)) with c = 2 log2(e) is 2^(a c + bc), bc = b c.
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define ENC_TANH_C 2.8853900817779268f
__device__ __forceinline__ f32x2 tanh2_of_exponent(f32x2 t) {   // t = 2 log2(e) x
    f32x2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    e = e + (f32x2){1.0f, 1.0f};
    const f32x2 r = {__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
    return __builtin_elementwise_fma(r, (f32x2){-2.0f, -2.0f}, (f32x2){1.0f, 1.0f});
}
__device__ __forceinline__ f32x2 tanh2(f32x2 x) { return tanh2_of_exponent(x * (f32x2){ENC_TANH_C, ENC_TANH_C}); }
__device__ __forceinline__ f32x4 tanh4(const f32x4 &x) {
    const f32x2 lo = tanh2((f32x2){x[0], x[1]}), hi = tanh2((f32x2){x[2], x[3]});
    return (f32x4){lo.x, lo.y, hi.x, hi.y};
}
// tanh(a + b) with bc = {b * ENC_TANH_C}: one fused multiply-add instead of an add and a multiply
__device__ __forceinline__ f32x4 tanh4_bias(const f32x4 &a, const f32x4 &bc) {
    const f32x2 c = {ENC_TANH_C, ENC_TANH_C};
    const f32x2 lo = tanh2_of_exponent(__builtin_elementwise_fma((f32x2){a[0], a[1]}, c, (f32x2){bc[0], bc[1]}));
    const f32x2 hi = tanh2_of_exponent(__builtin_elementwise_fma((f32x2){a[2], a[3]}, c, (f32x2){bc[2], bc[3]}));
    return (f32x4){lo.x, lo.y, hi.x, hi.y};
}

// ------------------------------------------------------------------------------------------------
// Reference precision (qs_enc_params.precision = 1; template parameter SP of the 16-agent kernels).  The reference's modules run in
// fp32 (quad_multi_model.py:250-350); bf16 operands leave the fused features ~1e-2 away from them.  Here every operand of every GEMM -
// weight or activation - is the pair x = h + l / 2048 of fp16 numbers: h = fp16(x) carries 11 significant bits, l = fp16((x - h) * 2048)
// the next 11 (x - h is exact in fp32; the scale keeps l a normal fp16 number whatever the size of x).  A product becomes three
// v_mfma_f32_16x16x32_f16 - h.h into the accumulator, h.l and l.h into a second one that is folded in, times 1 / 2048, behind the K loop;
// the dropped l.l term and the roundings of the l halves are <= 2^-22 relative per product, i.e. fp32-grade.  Weights arrive split from
// the host (two 1 KiB planes per fragment), activations are kept as two planes ENC_SPLANE elements apart in LDS (and in `ebuf`).  The
// matrix cores do three times the work of the bf16 kernels, at a sixteenth of the price of the fp32 MFMA (v_mfma_f32_16x16x4_f32).
// Values below the smallest normal fp16 go into l alone (h = 0: no subnormal operand), values beyond +-65504 (no observation is) clamp.
// ------------------------------------------------------------------------------------------------
#define ENC_SPLANE 39936   // elements between the h and the l plane of an LDS buffer: the largest 16-agent layout (qs_encoder_kernel's)
#define ENC_SPLIT_SCALE 2048.0f
__device__ __forceinline__ void split2(float x, _Float16 &h, _Float16 &l) {
    x = __builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f);
    h = __builtin_fabsf(x) < 6.103515625e-05f ? (_Float16)0.0f : (_Float16)x;
    l = (_Float16)((x - (float)h) * ENC_SPLIT_SCALE);
}
// one activation / four consecutive ones -> LDS (or `ebuf`): bf16, or the two fp16 planes `plane` elements apart
template <bool SP>
__device__ __forceinline__ void put1(uint16_t *p, float v) {
    if constexpr (SP) {
        _Float16 h, l;
        split2(v, h, l);
        p[0] = __builtin_bit_cast(uint16_t, h);
        p[ENC_SPLANE] = __builtin_bit_cast(uint16_t, l);
    } else p[0] = __builtin_bit_cast(uint16_t, (__bf16)v);
}
template <bool SP>
__device__ __forceinline__ void put4(uint16_t *p, const f32x4 &x, int plane = ENC_SPLANE) {
    if constexpr (SP) {
        f16x4 vh, vl;
#pragma unroll
        for (int r = 0; r < 4; ++r) { _Float16 h, l; split2(x[r], h, l); vh[r] = h; vl[r] = l; }
        *(f16x4 *)p = vh;
        *(f16x4 *)(p + plane) = vl;
    } else {
        bf16x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (__bf16)x[r];
        *(bf16x4 *)p = v;
    }
}

// floor(n / d) for small operands (n * d < 2^32) from m = ceil(2^32 / d), computed once per thread: integer division by a
// run-time divisor is ~40 instructions, and the staging loops did two per element
__device__ __forceinline__ uint32_t div_magic(uint32_t d) { return 0xffffffffu / d + 1u; }
__device__ __forceinline__ uint32_t div_by(uint32_t n, uint32_t magic) { return __umulhi(n, magic); }
// j mod B for j < 2^24 (the host bounds batch x neighbours): float estimate of the quotient, one correction step
__device__ __forceinline__ uint32_t mod_batch(uint32_t j, uint32_t B, float invB) {
    const uint32_t q = (uint32_t)((float)j * invB);
    int32_t r = (int32_t)(j - q * B);
    if (r < 0) r += (int32_t)B;
    else if (r >= (int32_t)B) r -= (int32_t)B;
    return (uint32_t)r;
}

// Observation elements through a buffer resource: an invalid element (padding column, row past the batch) gets an out-of-range
// offset and reads as zero, so the staging loops have no branch around their loads and issue them back to back (a conditional
// load per iteration is a branch plus s_waitcnt vmcnt(0): the memory latency once per element instead of once per loop).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t obs_rsrc(const float *obs, int B, int D) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)obs, 0, (uint32_t)B * (uint32_t)D * 4u, 0x00020000);
}
__device__ __forceinline__ float obs_at(__amdgpu_buffer_rsrc_t rs, bool valid, uint32_t index) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, valid ? index * 4u : 0xffffffffu, 0, 0));
}

// acc[mt][nt] (+)= W[features of (wave, mt)] x X[rows of tile nt], K-loop over the whole layer.  NT is a compile-time tile count
// and the steady-state loop has no conditional loads: a run-time bound puts a branch in front of every MFMA and LDS read (a lone
// pair of waves per SIMD pays for each of them) and makes the compiler drain the load counters every iteration.
// Weight fragments come from L2 and are software-pipelined ENC_PD K-steps ahead through a ring of register sets (the slot an
// MFMA group has just consumed is refilled with the fragment of K-step ks + ENC_PD); the activation fragment of a row tile is
// re-read from LDS for K-step ks + 1 as soon as its MFMAs of K-step ks are issued.
#define ENC_PD 4
#define ENC_WLOAD(x) (x)
template <int MT, int NT>
__device__ __forceinline__ void mfma_tile(const bf16x8 (&a)[MT], const bf16x8 &b, f32x4 (&acc)[MT][NT], int nt) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b, acc[mt][nt], 0, 0, 0);
}

// reference precision (see split2): A fragments in two planes per (tile, K-step), B fragments in two LDS planes, three MFMAs per product
template <int MT, int NT>
__device__ __forceinline__ void gemm_tiles_split(const EncLayer &L, int mtile0, const uint16_t *X, int xstride, f32x4 (&acc)[MT][NT]) {
    const int lane = threadIdx.x & 63, ksteps = L.K >> 5;
    const uint16_t *xrow = X + (lane & 15) * xstride + 8 * (lane >> 4);
    const uint32_t voff = lane * 16;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)L.w, 0, L.M * L.K * 4, 0x00020000);
#define ENC_SW(mt, ks, pl) __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, voff, (((mtile0 + (mt)) * ksteps + (ks)) * 2 + (pl)) * 1024, 0))
#define ENC_SX(nt, ks, pl) (*(const f16x8 *)(xrow + (pl) * ENC_SPLANE + (nt) * 16 * xstride + (ks) * 32))
    f32x4 lo[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) lo[mt][nt] = (f32x4){0, 0, 0, 0};
    // one K-step of every tile: h.h for all tiles first, then the two cross terms - consecutive MFMAs never share an accumulator
#define ENC_SSTEP(AH, AL, BH, BL)                                                                                        \
    do {                                                                                                                  \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                                 \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                             \
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16((AH)[mt], (BH)[nt], acc[mt][nt], 0, 0, 0);           \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                                 \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                             \
                lo[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16((AH)[mt], (BL)[nt], lo[mt][nt], 0, 0, 0);             \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                                 \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                             \
                lo[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16((AL)[mt], (BH)[nt], lo[mt][nt], 0, 0, 0);             \
    } while (0)
    f16x8 bh[NT], bl[NT];
    if (ksteps & (ENC_PD - 1)) {   // the 32- and 64-wide input layers
        for (int ks = 0; ks < ksteps; ++ks) {
            f16x8 ah[MT], al[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { ah[mt] = ENC_SW(mt, ks, 0); al[mt] = ENC_SW(mt, ks, 1); }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { bh[nt] = ENC_SX(nt, ks, 0); bl[nt] = ENC_SX(nt, ks, 1); }
            ENC_SSTEP(ah, al, bh, bl);
        }
    } else {
        f16x8 ah[ENC_PD][MT], al[ENC_PD][MT];   // weight ring, ENC_PD K-steps ahead (as in gemm_tiles)
#pragma unroll
        for (int s = 0; s < ENC_PD; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { ah[s][mt] = ENC_SW(mt, s, 0); al[s][mt] = ENC_SW(mt, s, 1); }
        int ks0 = 0;
        for (; ks0 + ENC_PD < ksteps; ks0 += ENC_PD) {
#pragma unroll
            for (int s = 0; s < ENC_PD; ++s) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { bh[nt] = ENC_SX(nt, ks0 + s, 0); bl[nt] = ENC_SX(nt, ks0 + s, 1); }
                ENC_SSTEP(ah[s], al[s], bh, bl);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) { ah[s][mt] = ENC_SW(mt, ks0 + s + ENC_PD, 0); al[s][mt] = ENC_SW(mt, ks0 + s + ENC_PD, 1); }
            }
        }
#pragma unroll
        for (int s = 0; s < ENC_PD; ++s) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { bh[nt] = ENC_SX(nt, ks0 + s, 0); bl[nt] = ENC_SX(nt, ks0 + s, 1); }
            ENC_SSTEP(ah[s], al[s], bh, bl);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mt][nt][r] += lo[mt][nt][r] * (1.0f / ENC_SPLIT_SCALE);
#undef ENC_SSTEP
#undef ENC_SX
#undef ENC_SW
}

template <int MT, int NT, bool SP = false>
__device__ __forceinline__ void gemm_tiles(const EncLayer &L, int mtile0, const uint16_t *X, int xstride, f32x4 (&acc)[MT][NT]) {
    if constexpr (SP) { gemm_tiles_split<MT, NT>(L, mtile0, X, xstride, acc); return; }
    const int lane = threadIdx.x & 63, ksteps = L.K >> 5;
    const uint16_t *xrow = X + (lane & 15) * xstride + 8 * (lane >> 4);
    // fragment address = buffer resource of the layer (scalar registers) + wave-uniform scalar offset (mtile0 is uniform) + one
    // per-lane byte offset shared by every layer: no per-layer 64-bit address pairs in vector registers
    const uint32_t voff = lane * 16;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)L.w, 0, L.M * L.K * 2, 0x00020000);
#define ENC_WFRAG(mt, ks) ENC_WLOAD(__builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, voff, ((mtile0 + (mt)) * ksteps + (ks)) * 1024, 0)))
#define ENC_XFRAG(nt, ks) (*(const bf16x8 *)(xrow + (nt) * 16 * xstride + (ks) * 32))
    if (ksteps & (ENC_PD - 1)) {   // the 32- and 64-wide input layers: one or two K-steps, nothing to pipeline
        for (int ks = 0; ks < ksteps; ++ks) {
            bf16x8 a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = ENC_WFRAG(mt, ks);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) mfma_tile<MT, NT>(a, ENC_XFRAG(nt, ks), acc, nt);
        }
        return;
    }
    bf16x8 a[ENC_PD][MT], b[NT];
#pragma unroll
    for (int s = 0; s < ENC_PD; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[s][mt] = ENC_WFRAG(mt, s);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = ENC_XFRAG(nt, 0);
    int ks0 = 0;
    for (; ks0 + ENC_PD < ksteps; ks0 += ENC_PD) {   // steady state: every load unconditional, so the wait counters stay exact
#pragma unroll
        for (int s = 0; s < ENC_PD; ++s) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                mfma_tile<MT, NT>(a[s], b[nt], acc, nt);
                b[nt] = ENC_XFRAG(nt, ks0 + s + 1);   // this row tile's fragment is consumed: refill it while the other tiles' MFMAs run
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[s][mt] = ENC_WFRAG(mt, ks0 + s + ENC_PD);
        }
    }
#pragma unroll
    for (int s = 0; s < ENC_PD; ++s)   // the last ENC_PD K-steps: their weights are already in flight
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            mfma_tile<MT, NT>(a[s], b[nt], acc, nt);
            if (s + 1 < ENC_PD) b[nt] = ENC_XFRAG(nt, ks0 + s + 1);
        }
}

template <int MT, int NT>
__device__ __forceinline__ void init_bias(const EncLayer &L, int mtile0, f32x4 (&acc)[MT][NT]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const f32x4 bias = *(const f32x4 *)(L.b + (mtile0 + mt) * 16 + (lane >> 4) * 4);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = bias;
    }
}

// tanh, bf16, store: lane holds features f0..f0+3 of row (nt*16 + lane&15)
template <int MT, int NT, bool SP = false>
__device__ __forceinline__ void store_tanh(const f32x4 (&acc)[MT][NT], int mtile0, uint16_t *Y, int ystride, int col0 = 0) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 t;
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = fast_tanh(acc[mt][nt][r]);
            put4<SP>(Y + (nt * 16 + (lane & 15)) * ystride + col0 + (mtile0 + mt) * 16 + (lane >> 4) * 4, t);
        }
}

// one 16-row MLP: Y[:, col0:col0+256] = tanh(L2 tanh(L1 X)), hidden layer through `hid` (one barrier inside)
template <bool SP = false>
__device__ __forceinline__ void mlp2_one_tile(const EncLayer &L1, const EncLayer &L2, int mt0, const uint16_t *X, int xstride,
    uint16_t *hid,
                                              uint16_t *Y, int ystride, int col0) {
    f32x4 acc[ENC_MT][1];
    init_bias<ENC_MT, 1>(L1, mt0, acc);
    gemm_tiles<ENC_MT, 1, SP>(L1, mt0, X, xstride, acc);
    store_tanh<ENC_MT, 1, SP>(acc, mt0, hid, ENC_YS);
    __syncthreads();
    init_bias<ENC_MT, 1>(L2, mt0, acc);
    gemm_tiles<ENC_MT, 1, SP>(L2, mt0, hid, ENC_YS, acc);
    store_tanh<ENC_MT, 1, SP>(acc, mt0, Y, ystride, col0);
}

__device__ __forceinline__ float lane_groups_sum(float v) {   // sum over the 4 lane groups that hold the same row (lane & 15)
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// feed forward: tanh(F [self | neighbourhood | obstacles]) -> out[a][0:512] (fp32)   (:329-332, :349), and optionally a linear
// head on it (Sample Factory's action-parameter or value layer, 512 -> head_dim <= 8) so that a rollout does not have to write
// and re-read the features: per-lane partial dot products, two shuffles over the lane groups, the eight waves through `red`
// (LDS scratch, >= ENC_WAVES * 8 * 16 floats in a buffer nobody reads any more and that is not `cat`).
template <int MTF = ENC_MTF, bool SP = false>   // 16-feature tiles per wave: 512 outputs = 32 tiles; the Sim2Real encoder's 256 = 16 tiles
__device__ __forceinline__ void feed_forward(const EncParams &P, const uint16_t *cat, int a0, int B, float *__restrict__ out, float *red) {
    const int wave = wave_id(), lane = threadIdx.x & 63;
    constexpr int OUT = MTF * ENC_WAVES * 16;
    f32x4 acc[MTF][1];
    const int mf0 = wave * MTF;
    init_bias<MTF, 1>(P.f, mf0, acc);
    gemm_tiles<MTF, 1, SP>(P.f, mf0, cat, ENC_CS, acc);
    ENC_STAMP(8);
    const int ga = a0 + (lane & 15);
    f32x4 v[MTF];
#pragma unroll
    for (int mt = 0; mt < MTF; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[mt][r] = fast_tanh(acc[mt][0][r]);
        if (out && ga < B) *(f32x4 *)(out + (size_t)ga * OUT + (mf0 + mt) * 16 + (lane >> 4) * 4) = v[mt];
    }
    if (P.head_dim > 0) {
        for (int h = 0; h < P.head_dim; ++h) {
            float s = 0.0f;
#pragma unroll
            for (int mt = 0; mt < MTF; ++mt) {
                const f32x4 w = *(const f32x4 *)(P.head_w + h * OUT + (mf0 + mt) * 16 + (lane >> 4) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) s += v[mt][r] * w[r];
            }
            s = lane_groups_sum(s);
            if (lane < 16) red[(wave * 8 + h) * 16 + lane] = s;
        }
        __syncthreads();
        const int tid = threadIdx.x, h = tid >> 4, row = tid & 15;
        if (h < P.head_dim && a0 + row < B) {
            float s = P.head_b[h];
#pragma unroll
            for (int w = 0; w < ENC_WAVES; ++w) s += red[(w * 8 + h) * 16 + row];
            P.head_out[(size_t)(a0 + row) * P.head_dim + h] = s;
            if (P.sample_log_std) P.act_out[(size_t)(a0 + row) * P.head_dim + h] = sample_action(P, a0 + row, h, s);
        }
    }
}

// run CALL(<tile count>) for min(n, LIMIT) tiles; only the counts a pass size of LIMIT can see are instantiated
#define ENC_CASE_NT(k, LIMIT, CALL) case k: if constexpr (k <= (LIMIT)) { CALL(k); } break;
#define ENC_DISPATCH_NT(n, LIMIT, CALL)                                        \
    switch ((n) < (LIMIT) ? (n) : (LIMIT)) {                                   \
        ENC_CASE_NT(1, LIMIT, CALL) ENC_CASE_NT(2, LIMIT, CALL) ENC_CASE_NT(3, LIMIT, CALL) ENC_CASE_NT(4, LIMIT, CALL) \
        ENC_CASE_NT(5, LIMIT, CALL) ENC_CASE_NT(6, LIMIT, CALL) ENC_CASE_NT(7, LIMIT, CALL) ENC_CASE_NT(8, LIMIT, CALL) \
    default: break;                                                            \
    }

// ------------------------------------------------------------------------------------------------
// attention, launch 1: e_i = embedding_mlp([self_obs[(a*K+k) mod B] | neighbour obs (a,k)]) -> ebuf;  g_a = W_m mean_k e_(a,k) -> gbuf
// ------------------------------------------------------------------------------------------------
template <int NTH, bool SP = false>
__device__ __forceinline__ void embed_pass(const EncParams &P, int B, int a0, int t0, bool first, const uint16_t *x_in, uint16_t *buf_a,
    f32x4 (&mean)[ENC_MT]) {
    const int wave = wave_id(), lane = threadIdx.x & 63, mt0 = wave * ENC_MT, NB = P.num_nbr;
    f32x4 acc[ENC_MT][NTH];
    init_bias<ENC_MT, NTH>(P.n1, mt0, acc);
    gemm_tiles<ENC_MT, NTH, SP>(P.n1, mt0, x_in + t0 * ENC_TA * ENC_XS, ENC_XS, acc);
    if (!first) __syncthreads();   // the previous pass is done reading buf_a
    store_tanh<ENC_MT, NTH, SP>(acc, mt0, buf_a, ENC_YS);
    __syncthreads();
    init_bias<ENC_MT, NTH>(P.n2, mt0, acc);
    gemm_tiles<ENC_MT, NTH, SP>(P.n2, mt0, buf_a, ENC_YS, acc);
    const int ga = a0 + (lane & 15);
    constexpr int ES = SP ? 2 * ENC_H : ENC_H;   // ebuf row: bf16 [256], or the two fp16 planes [2][256]
#pragma unroll
    for (int mt = 0; mt < ENC_MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTH; ++nt) {
            f32x4 e;
#pragma unroll
            for (int r = 0; r < 4; ++r) { e[r] = fast_tanh(acc[mt][nt][r]); mean[mt][r] += e[r]; }
            if (ga < B) put4<SP>(P.ebuf + ((size_t)ga * NB + (t0 + nt)) * ES + (mt0 + mt) * 16 + (lane >> 4) * 4, e, ENC_H);
        }
}

template <bool SP>
__device__ __forceinline__ void embed_body(const float *__restrict__ obs, int B, const EncParams &P) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint16_t *x_in = (uint16_t *)smem;                                // [NBR*16][XS]
    uint16_t *buf_a = x_in + ENC_MAX_NBR * ENC_TA * ENC_XS;           // [NH*16][YS]
    uint16_t *emean = buf_a + ENC_NH * ENC_TA * ENC_YS;               // [16][YS]
    const int tid = threadIdx.x, wave = wave_id(), lane = tid & 63, a0 = blockIdx.x * ENC_TA;
    const int NB = P.num_nbr, D = P.obs_dim, mt0 = wave * ENC_MT;
    const float invB = 1.0f / (float)B;
    traj_copy(P, a0, ENC_TA, B);
    const __amdgpu_buffer_rsrc_t ors = obs_rsrc(obs, B, D);
#pragma unroll 4
    for (int idx = tid; idx < NB * ENC_TA * 32; idx += 64 * ENC_WAVES) {
        const int row = idx >> 5, c = idx & 31, k = row >> 4, a = row & 15, ga = a0 + a;
        // self_obs.repeat(K, 1)  (:84)
        const uint32_t i_self = mod_batch((uint32_t)ga * (uint32_t)NB + (uint32_t)k, (uint32_t)B, invB) * (uint32_t)D + c;
        const uint32_t i_nbr = (uint32_t)ga * (uint32_t)D + P.self_dim + k * P.nbr_dim + (c - P.self_dim);
        const float v = obs_at(ors, ga < B && c < P.self_dim + P.nbr_dim, c < P.self_dim ? i_self : i_nbr);
        put1<SP>(x_in + row * ENC_XS + c, v);
    }
    __syncthreads();
    f32x4 mean[ENC_MT];
#pragma unroll
    for (int mt = 0; mt < ENC_MT; ++mt) mean[mt] = (f32x4){0, 0, 0, 0};
    for (int t0 = 0; t0 < NB; t0 += ENC_NH) {
#define ENC_CALL(n) embed_pass<n, SP>(P, B, a0, t0, t0 == 0, x_in, buf_a, mean)
        ENC_DISPATCH_NT(NB - t0, ENC_NH, ENC_CALL)
#undef ENC_CALL
    }
    const float inv = 1.0f / (float)NB;   // e_mean (:90-91), then its half of the score MLP's first layer once per agent
#pragma unroll
    for (int mt = 0; mt < ENC_MT; ++mt) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = mean[mt][r] * inv;
        put4<SP>(emean + (lane & 15) * ENC_YS + (mt0 + mt) * 16 + (lane >> 4) * 4, v);
    }
    __syncthreads();
    f32x4 g[ENC_MT][1];
#pragma unroll
    for (int mt = 0; mt < ENC_MT; ++mt) g[mt][0] = (f32x4){0, 0, 0, 0};
    gemm_tiles<ENC_MT, 1, SP>(P.a1m, mt0, emean, ENC_YS, g);
    const int ga = a0 + (lane & 15);
    if (ga < B) {
#pragma unroll
        for (int mt = 0; mt < ENC_MT; ++mt) *(f32x4 *)(P.gbuf + (size_t)ga * ENC_H + (mt0 + mt) * 16 + (lane >> 4) * 4) = g[mt][0];
    }
}
extern "C" __global__ void __launch_bounds__(64 * ENC_WAVES, ENC_OCC) qs_encoder_embed_kernel(const float *__restrict__ obs, int B,
    EncParams P) {
    embed_body<false>(obs, B, P);
}
extern "C" __global__ void __launch_bounds__(64 * ENC_WAVES, 2) qs_encoder_embed_split_kernel(const float *__restrict__ obs, int B,
    EncParams P) {
    embed_body<true>(obs, B, P);
}

// ------------------------------------------------------------------------------------------------
// attention, launch 2 (:88-101): value MLP, score MLP, softmax over the neighbours, weighted sum; then self / obstacle encoders
// and the feed-forward layer.  One sweep over the neighbour row tiles in groups of ENC_ANH with an online softmax (running
// maximum and denominator per agent, the partial sum rescaled when the maximum moves), so that the h_i of earlier groups do not
// have to be kept: 78 KB of LDS and <= 128 VGPRs, two workgroups per CU.
// ------------------------------------------------------------------------------------------------
#ifndef ENC_ANH
#define ENC_ANH 3
#endif
struct AttnState { f32x4 o[ENC_MT]; float mx, den; };

template <int NTH, bool SP = false>
__device__ __forceinline__ void attn_load_e(const EncParams &P, int B, int a0, int t0, uint16_t *buf_a) {
    // e_i rows in 16-byte chunks, coalesced; 32-bit offsets into a buffer resource (rows past the batch read as zero: out of range)
    constexpr int PL = SP ? 2 : 1;   // reference precision: a row of ebuf is the two fp16 planes [2][256]
    const __amdgpu_buffer_rsrc_t ers = __builtin_amdgcn_make_buffer_rsrc((void *)P.ebuf, 0,
        (uint32_t)B * (uint32_t)P.num_nbr * (ENC_H * 2 * PL), 0x00020000);
    for (int idx = threadIdx.x; idx < NTH * ENC_TA * (ENC_H / 8) * PL; idx += 64 * ENC_WAVES) {
        const int row = idx / (32 * PL), pl = (idx >> 5) & (PL - 1), ch = idx & 31, k = t0 + (row >> 4), ra = a0 + (row & 15);
        const uint32_t off = ra < B ? (((uint32_t)ra * (uint32_t)P.num_nbr + (uint32_t)k) * PL + pl) * (ENC_H * 2) + ch * 16 : 0xffffffffu;
        *(bf16x8 *)(buf_a + pl * ENC_SPLANE + row * ENC_YS + ch * 8) = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ers, off, 0, 0));
    }
}

// One group of NTH neighbour row tiles: scores first (the 256 -> 1 layer is reduced from the accumulators of the layer before it),
// then the values from the same e_i tile, which go straight into the running sum - the h_i are never live together with another
// layer's accumulators.  Four barriers per group.
template <int NTH, bool SP = false>
__device__ __forceinline__ void attn_pass(const EncParams &P, int B, int a0, int t0, uint16_t *buf_a, uint16_t *buf_h, float *s_alpha,
    AttnState &st) {
    const int wave = wave_id(), lane = threadIdx.x & 63, mt0 = wave * ENC_MT;
    const int ga = a0 + (lane & 15);
    attn_load_e<NTH, SP>(P, B, a0, t0, buf_a);
    f32x4 acc[ENC_MT][NTH];
    // score MLP, first layer on [e_i | e_mean.repeat(K, 1)]: W_e e_i + b + g[(a*K + k) mod B]   (:92-94)
    init_bias<ENC_MT, NTH>(P.a1e, mt0, acc);
    {
        const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)P.gbuf, 0, (uint32_t)B * (ENC_H * 4), 0x00020000);
#pragma unroll
        for (int nt = 0; nt < NTH; ++nt) {
            const uint32_t j = mod_batch((uint32_t)ga * (uint32_t)P.num_nbr + (uint32_t)(t0 + nt), (uint32_t)B, 1.0f / (float)B);
            const uint32_t off = ga < B ? j * (ENC_H * 4) + (lane >> 4) * 16 : 0xffffffffu;   // padding rows: out of range, reads zero
#pragma unroll
            for (int mt = 0; mt < ENC_MT; ++mt) {
                const f32x4 gv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(grs, off, (mt0 + mt) * 64, 0));
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mt][nt][r] += gv[r];
            }
        }
    }
    __syncthreads();   // e_i is in buf_a; the previous group's value layers are done with buf_h
    gemm_tiles<ENC_MT, NTH, SP>(P.a1e, mt0, buf_a, ENC_YS, acc);
    store_tanh<ENC_MT, NTH, SP>(acc, mt0, buf_h, ENC_YS);
    __syncthreads();
    init_bias<ENC_MT, NTH>(P.a2, mt0, acc);
    gemm_tiles<ENC_MT, NTH, SP>(P.a2, mt0, buf_h, ENC_YS, acc);
    // last score layer 256 -> 1 straight from the accumulators: per-lane partial dot product with the fp32 weight row, two shuffles
    // over the lane groups, the eight waves' partials through LDS - no activation store, no extra MFMA pass, and e_i stays in buf_a
#pragma unroll
    for (int nt = 0; nt < NTH; ++nt) {
        float sp = 0.0f;
#pragma unroll
        for (int mt = 0; mt < ENC_MT; ++mt) {
            const f32x4 w = *(const f32x4 *)(P.a3w + (mt0 + mt) * 16 + (lane >> 4) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) sp += fast_tanh(acc[mt][nt][r]) * w[r];
        }
        sp = lane_groups_sum(sp);
        if (lane < 16) s_alpha[(wave * ENC_ANH + nt) * 16 + lane] = sp;
    }
    __syncthreads();   // partial scores visible; every wave is done reading buf_h (second score layer)
    // h_i = neighbor_value_mlp(e_i)   (:88)
    init_bias<ENC_MT, NTH>(P.v1, mt0, acc);
    gemm_tiles<ENC_MT, NTH, SP>(P.v1, mt0, buf_a, ENC_YS, acc);
    store_tanh<ENC_MT, NTH, SP>(acc, mt0, buf_h, ENC_YS);
    __syncthreads();
    init_bias<ENC_MT, NTH>(P.v2, mt0, acc);
    gemm_tiles<ENC_MT, NTH, SP>(P.v2, mt0, buf_h, ENC_YS, acc);
    // online softmax over the neighbours of agent (lane & 15)   (:95-100)
    float al[NTH], mx = st.mx;
#pragma unroll
    for (int nt = 0; nt < NTH; ++nt) {
        al[nt] = P.a3b;
#pragma unroll
        for (int w = 0; w < ENC_WAVES; ++w) al[nt] += s_alpha[(w * ENC_ANH + nt) * 16 + (lane & 15)];
        mx = fmaxf(mx, al[nt]);
    }
    const float scale = __expf(st.mx - mx);
    st.den *= scale;
#pragma unroll
    for (int mt = 0; mt < ENC_MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) st.o[mt][r] *= scale;
#pragma unroll
    for (int nt = 0; nt < NTH; ++nt) {
        const float e = __expf(al[nt] - mx);
        st.den += e;
#pragma unroll
        for (int mt = 0; mt < ENC_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) st.o[mt][r] += e * fast_tanh(acc[mt][nt][r]);
    }
    st.mx = mx;
}

template <bool SP>
__device__ __forceinline__ void attn_body(const float *__restrict__ obs, int B, const EncParams &P, float *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint16_t *x_self = (uint16_t *)smem;                              // [16][XS]
    uint16_t *x_obst = x_self + ENC_TA * ENC_XS;                      // [16][XS]
    uint16_t *buf_a = x_obst + ENC_TA * ENC_XS;                       // [ANH*16][YS]  e_i of the group, later the second score layer
    // [ANH*16][YS]  hidden layers; first the self / obstacle MLPs' (one tile)
    uint16_t *buf_h = buf_a + ENC_ANH * ENC_TA * ENC_YS;
    uint16_t *cat = buf_h + ENC_ANH * ENC_TA * ENC_YS;                // [16][CS]: self | neighbourhood | obstacles
    float *s_alpha = (float *)(SP ? (uint16_t *)smem + 2 * ENC_SPLANE : cat + ENC_TA * ENC_CS);   // [8 waves][ANH][16] partial scores of the group
    const int tid = threadIdx.x, wave = wave_id(), lane = tid & 63, a0 = blockIdx.x * ENC_TA;
    const int NB = P.num_nbr, D = P.obs_dim, mt0 = wave * ENC_MT;
    const int col_nbr = ENC_H, col_obst = 2 * ENC_H;

    for (int idx = tid; idx < 2 * ENC_TA * 32; idx += 64 * ENC_WAVES) {   // self and obstacle columns as bf16, zero padded to K = 32
        const int which = idx >> 9, a = (idx >> 5) & 15, c = idx & 31, ga = a0 + a;
        const int dim = which ? P.obst_dim : P.self_dim, col = which ? P.self_dim + P.nbr_dim * NB : 0;
        const float v = obs_at(obs_rsrc(obs, B, D), ga < B && c < dim, (uint32_t)ga * (uint32_t)D + col + c);
        put1<SP>((which ? x_obst : x_self) + a * ENC_XS + c, v);
    }
    __syncthreads();
    mlp2_one_tile<SP>(P.s1, P.s2, mt0, x_self, ENC_XS, buf_h, cat, ENC_CS, 0);
    if (P.obst_dim > 0) {
        __syncthreads();
        mlp2_one_tile<SP>(P.o1, P.o2, mt0, x_obst, ENC_XS, buf_h, cat, ENC_CS, col_obst);
    }
    __syncthreads();

    AttnState st;
    st.mx = -3.0e38f; st.den = 0.0f;
#pragma unroll
    for (int mt = 0; mt < ENC_MT; ++mt) st.o[mt] = (f32x4){0, 0, 0, 0};
    for (int t0 = 0; t0 < NB; t0 += ENC_ANH) {
#define ENC_CALL(n) attn_pass<n, SP>(P, B, a0, t0, buf_a, buf_h, s_alpha, st)
        ENC_DISPATCH_NT(NB - t0, ENC_ANH, ENC_CALL)
#undef ENC_CALL
    }
    const float rden = 1.0f / st.den;
#pragma unroll
    for (int mt = 0; mt < ENC_MT; ++mt) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = st.o[mt][r] * rden;
        put4<SP>(cat + (lane & 15) * ENC_CS + col_nbr + (mt0 + mt) * 16 + (lane >> 4) * 4, v);
    }
    __syncthreads();
    feed_forward<ENC_MTF, SP>(P, cat, a0, B, out, (float *)buf_a);
}
extern "C" __global__ void __launch_bounds__(64 * ENC_WAVES, ENC_OCC) qs_encoder_attn_kernel(const float *__restrict__ obs, int B,
    EncParams P, float *__restrict__ out) {
    attn_body<false>(obs, B, P, out);
}
extern "C" __global__ void __launch_bounds__(64 * ENC_WAVES, 2) qs_encoder_attn_split_kernel(const float *__restrict__ obs, int B,
    EncParams P, float *__restrict__ out) {
    attn_body<true>(obs, B, P, out);
}

// ------------------------------------------------------------------------------------------------
// QuadMultiHeadAttentionEncoder (:124-196, --quads_encoder_type=attention): self / neighbour / obstacle MLPs, 4-head scaled
// dot-product attention over the token pair [neighbour embedding, obstacle embedding] (attention_layer.py:12-56: projections
// without bias, q / sqrt(d_k), softmax over the keys, output projection, residual, LayerNorm eps 1e-6), feed-forward.
// Wave w owns features [128w, 128w+128) of the 1024-wide projections = half of head w/2, in two chunks of 4 feature tiles:
// the 2x2 scores of a head are sums over its features, so they accumulate chunk by chunk and only one chunk of q, k is live;
// lane groups are reduced with two shuffles, the two waves of a head and (for LayerNorm) the eight waves through LDS.
// ------------------------------------------------------------------------------------------------
#define ENC_OS (4 * ENC_H + 8)   // row stride of the concatenated-heads buffer
template <int MT, int NT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[MT][NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0, 0, 0, 0};
}
// 16-row MLP like mlp2_one_tile, but the fp32 result also stays in registers (the attention block's residual)
template <bool SP = false>
__device__ __forceinline__ void mlp2_keep(const EncLayer &L1, const EncLayer &L2, int mt0, const uint16_t *X, int xstride, uint16_t *hid,
    uint16_t *Y,
                                          f32x4 (&keep)[ENC_MT]) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[ENC_MT][1];
    init_bias<ENC_MT, 1>(L1, mt0, acc);
    gemm_tiles<ENC_MT, 1, SP>(L1, mt0, X, xstride, acc);
    store_tanh<ENC_MT, 1, SP>(acc, mt0, hid, ENC_YS);
    __syncthreads();
    init_bias<ENC_MT, 1>(L2, mt0, acc);
    gemm_tiles<ENC_MT, 1, SP>(L2, mt0, hid, ENC_YS, acc);
#pragma unroll
    for (int mt = 0; mt < ENC_MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) keep[mt][r] = fast_tanh(acc[mt][0][r]);
        put4<SP>(Y + (lane & 15) * ENC_YS + (mt0 + mt) * 16 + (lane >> 4) * 4, keep[mt]);
    }
}

// one-layer embedding of 16 rows: Y[:, col..] = tanh(L X); KEEP: the fp32 result also stays in registers (the attention block's residual)
template <bool KEEP, bool SP = false>
__device__ __forceinline__ void mlp1_keep(const EncLayer &L1, int mt0, const uint16_t *X, int xstride, uint16_t *Y, int ystride,
    f32x4 (&keep)[ENC_MT]) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[ENC_MT][1];
    init_bias<ENC_MT, 1>(L1, mt0, acc);
    gemm_tiles<ENC_MT, 1, SP>(L1, mt0, X, xstride, acc);
#pragma unroll
    for (int mt = 0; mt < ENC_MT; ++mt) {
        f32x4 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            t[r] = fast_tanh(acc[mt][0][r]);
            if constexpr (KEEP) keep[mt][r] = t[r];
        }
        put4<SP>(Y + (lane & 15) * ystride + (mt0 + mt) * 16 + (lane >> 4) * 4, t);
    }
}

template <bool S2R>
__device__ __forceinline__ void mha_body(const float *__restrict__ obs, int B, const EncParams &P, float *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint16_t *x_self = (uint16_t *)smem;                              // [16][XS]
    uint16_t *x_obst = x_self + ENC_TA * ENC_XS;                      // [16][XS]
    uint16_t *x_nbr = x_obst + ENC_TA * ENC_XS;                       // [16][XW]  all neighbour columns of the agent, K padded to 64
    uint16_t *hid = x_nbr + ENC_TA * ENC_XW;                          // [16][YS]  hidden layer of the three MLPs
    uint16_t *tok = hid + ENC_TA * ENC_YS;                            // [2][16][YS]  tokens: neighbour embedding, obstacle embedding
    uint16_t *obuf = tok + 2 * ENC_TA * ENC_YS;                       // [2][16][OS]  attention output, heads concatenated
    uint16_t *cat = obuf + 2 * ENC_TA * ENC_OS;                       // [16][CS]: self | token 0 | token 1
    float *red_s = (float *)(cat + ENC_TA * ENC_CS);                  // [4 heads][2 waves][4 (i,j)][16]  partial scores
    float *red_ln = red_s + 4 * 2 * 4 * 16;                           // [8 waves][2 tokens][2 (sum, sum of squares)][16]
    const int tid = threadIdx.x, wave = wave_id(), lane = tid & 63, a0 = blockIdx.x * ENC_TA;
    const int NB = P.num_nbr, D = P.obs_dim, mt0 = wave * ENC_MT, nbw = P.nbr_dim * NB;
    traj_copy(P, a0, ENC_TA, B);

    // columns as bf16, zero padded: [self 32 | obstacle 32 | neighbours 64]
    for (int idx = tid; idx < ENC_TA * 128; idx += 64 * ENC_WAVES) {
        const int a = idx >> 7, c = idx & 127, ga = a0 + a;
        int col = -1;
        uint16_t *dst;
        if (c < 32) { dst = x_self + a * ENC_XS + c; if (c < P.self_dim) col = c; }
        else if (c < 64) { dst = x_obst + a * ENC_XS + (c - 32); if (c - 32 < P.obst_dim) col = P.self_dim + nbw + (c - 32); }
        else { dst = x_nbr + a * ENC_XW + (c - 64); if (c - 64 < nbw) col = P.self_dim + (c - 64); }
        const float v = obs_at(obs_rsrc(obs, B, D), ga < B && col >= 0, (uint32_t)ga * (uint32_t)D + col);
        *dst = __builtin_bit_cast(uint16_t, (__bf16)v);
    }
    __syncthreads();
    f32x4 resid[2][ENC_MT];   // fp32 tokens: features of this wave, rows lane & 15
    if constexpr (S2R) {   // one layer per embedding (:229-240): nothing between them to wait for
        mlp1_keep<false>(P.s1, mt0, x_self, ENC_XS, cat, ENC_CS, resid[0]);
        mlp1_keep<true>(P.n1, mt0, x_nbr, ENC_XW, tok, ENC_YS, resid[0]);
        mlp1_keep<true>(P.o1, mt0, x_obst, ENC_XS, tok + ENC_TA * ENC_YS, ENC_YS, resid[1]);
    } else {
        mlp2_one_tile(P.s1, P.s2, mt0, x_self, ENC_XS, hid, cat, ENC_CS, 0);
        __syncthreads();
        mlp2_keep<>(P.n1, P.n2, mt0, x_nbr, ENC_XW, hid, tok, resid[0]);
        __syncthreads();
        mlp2_keep<>(P.o1, P.o2, mt0, x_obst, ENC_XS, hid, tok + ENC_TA * ENC_YS, resid[1]);
    }
    __syncthreads();

    // ---- scores: s[i][j] = q_i . k_j over the head's 256 features, accumulated over this wave's two chunks ----
    // (one head: wave w owns features [32w, 32w+32) of the 256-wide projections, one chunk of 2 feature tiles)
    constexpr int QT = S2R ? 2 : 4, QC = S2R ? 1 : 2;   // feature tiles per chunk, chunks per wave
    float sc[2][2] = {{0, 0}, {0, 0}};
#pragma unroll 1
    for (int c = 0; c < QC; ++c) {
        f32x4 q[QT][2], k[QT][2];
        zero_acc<QT, 2>(q);
        gemm_tiles<QT, 2>(P.mq, (wave * QC + c) * QT, tok, ENC_YS, q);
        zero_acc<QT, 2>(k);
        gemm_tiles<QT, 2>(P.mk, (wave * QC + c) * QT, tok, ENC_YS, k);
#pragma unroll
        for (int mt = 0; mt < QT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) sc[i][j] += q[mt][i][r] * k[mt][j][r];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float t = lane_groups_sum(sc[i][j]);
            if (lane < 16) red_s[(wave * 4 + i * 2 + j) * 16 + lane] = t;
        }
    __syncthreads();
    float pr[2][2];   // softmax over the keys j of (q_i / sqrt(d_k)) . k_j   (attention_layer.py:118-125)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float t[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float acc_s = 0.0f;
            if constexpr (S2R) {   // one head over all eight waves   (attention_layer.py:83)
#pragma unroll
                for (int w = 0; w < ENC_WAVES; ++w) acc_s += red_s[(w * 4 + i * 2 + j) * 16 + (lane & 15)];
            } else
                acc_s = red_s[((wave & ~1) * 4 + i * 2 + j) * 16 + (lane & 15)] + red_s[((wave | 1) * 4 + i * 2 + j) * 16 + (lane & 15)];
            t[j] = acc_s * (1.0f / 16.0f);
        }
        const float m = fmaxf(t[0], t[1]), e0 = __expf(t[0] - m), e1 = __expf(t[1] - m), rd = 1.0f / (e0 + e1);
        pr[i][0] = e0 * rd;
        pr[i][1] = e1 * rd;
    }
    // ---- o_i = sum_j p_ij v_j -> obuf[i][row][head * 256 + feature] ----
#pragma unroll 1
    for (int c = 0; c < QC; ++c) {
        f32x4 v[QT][2];
        zero_acc<QT, 2>(v);
        gemm_tiles<QT, 2>(P.mv, (wave * QC + c) * QT, tok, ENC_YS, v);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int mt = 0; mt < QT; ++mt) {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (__bf16)(pr[i][0] * v[mt][0][r] + pr[i][1] * v[mt][1][r]);
                *(bf16x4 *)(obuf + (i * ENC_TA + (lane & 15)) * ENC_OS + ((wave * QC + c) * QT + mt) * 16 + (lane >> 4) * 4) = o;
            }
    }
    __syncthreads();
    // ---- fc, residual, LayerNorm -> cat[:, 256 + token * 256 + feature]   (attention_layer.py:47-54) ----
    f32x4 y[ENC_MT][2];
    zero_acc<ENC_MT, 2>(y);
    gemm_tiles<ENC_MT, 2>(P.mfc, mt0, obuf, ENC_OS, y);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int mt = 0; mt < ENC_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                y[mt][i][r] += resid[i][mt][r];
                s1 += y[mt][i][r];
                s2 += y[mt][i][r] * y[mt][i][r];
            }
        s1 = lane_groups_sum(s1);
        s2 = lane_groups_sum(s2);
        if (lane < 16) {
            red_ln[((wave * 2 + i) * 2 + 0) * 16 + lane] = s1;
            red_ln[((wave * 2 + i) * 2 + 1) * 16 + lane] = s2;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int w = 0; w < ENC_WAVES; ++w) {
            s1 += red_ln[((w * 2 + i) * 2 + 0) * 16 + (lane & 15)];
            s2 += red_ln[((w * 2 + i) * 2 + 1) * 16 + (lane & 15)];
        }
        const float mean = s1 * (1.0f / ENC_H), var = fmaxf(s2 * (1.0f / ENC_H) - mean * mean, 0.0f),
            rstd = __builtin_amdgcn_rsqf(var + 1e-6f);
#pragma unroll
        for (int mt = 0; mt < ENC_MT; ++mt) {
            const int f0 = (mt0 + mt) * 16 + (lane >> 4) * 4;
            const f32x4 g = *(const f32x4 *)(P.ln_w + f0), bb = *(const f32x4 *)(P.ln_b + f0);
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (__bf16)((y[mt][i][r] - mean) * rstd * g[r] + bb[r]);
            *(bf16x4 *)(cat + (lane & 15) * ENC_CS + ENC_H * (1 + i) + f0) = o;
        }
    }
    __syncthreads();
    feed_forward<S2R ? ENC_MTF / 2 : ENC_MTF>(P, cat, a0, B, out, (float *)hid);
}
extern "C" __global__ void __launch_bounds__(64 * ENC_WAVES, 2) qs_encoder_mha_kernel(const float *__restrict__ obs, int B, EncParams P,
    float *__restrict__ out) {
    mha_body<false>(obs, B, P, out);
}
extern "C" __global__ void __launch_bounds__(64 * ENC_WAVES, 2) qs_encoder_s2r_kernel(const float *__restrict__ obs, int B, EncParams P,
    float *__restrict__ out) {
    mha_body<true>(obs, B, P, out);
}

// The same block in reference precision (fp16 pairs, see split2).  Two LDS planes leave room for ONE token's concatenated heads, so the
// value projection, the weighted sum and the output projection run per query token (the value GEMM twice); the inputs and the MLPs'
// hidden layer share their space with that buffer (dead before it is written), the reduction scratch sits behind it in the h plane.
#define ENC_MHS_TOK 0
#define ENC_MHS_CAT (2 * ENC_TA * ENC_YS)
#define ENC_MHS_OBUF (ENC_MHS_CAT + ENC_TA * ENC_CS)
#define ENC_MHS_RED (ENC_MHS_OBUF + ENC_TA * ENC_OS)
static_assert(ENC_MHS_RED + 2 * (4 * 2 * 4 * 16 + ENC_WAVES * 2 * 2 * 16) <= ENC_SPLANE, "the multi-head layout fits one plane");
static_assert(2 * ENC_TA * ENC_XS + ENC_TA * ENC_XW + ENC_TA * ENC_YS <= ENC_TA * ENC_OS, "inputs + hidden layer fit under the heads buffer");
template <bool S2R>
__device__ __forceinline__ void mha_body_split(const float *__restrict__ obs, int B, const EncParams &P, float *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint16_t *tok = (uint16_t *)smem + ENC_MHS_TOK;                   // [2][16][YS]  tokens: neighbour embedding, obstacle embedding
    uint16_t *cat = (uint16_t *)smem + ENC_MHS_CAT;                   // [16][CS]: self | token 0 | token 1
    uint16_t *obuf = (uint16_t *)smem + ENC_MHS_OBUF;                 // [16][OS]  attention output of ONE query token, heads concatenated
    uint16_t *x_self = obuf;                                          // [16][XS]   (the inputs and `hid` are dead when obuf is written)
    uint16_t *x_obst = x_self + ENC_TA * ENC_XS;                      // [16][XS]
    uint16_t *x_nbr = x_obst + ENC_TA * ENC_XS;                       // [16][XW]
    uint16_t *hid = x_nbr + ENC_TA * ENC_XW;                          // [16][YS]
    float *red_s = (float *)((uint16_t *)smem + ENC_MHS_RED);         // [4 heads][2 waves][4 (i,j)][16]  partial scores
    float *red_ln = red_s + 4 * 2 * 4 * 16;                           // [8 waves][2 tokens][2 (sum, sum of squares)][16]
    const int tid = threadIdx.x, wave = wave_id(), lane = tid & 63, a0 = blockIdx.x * ENC_TA;
    const int NB = P.num_nbr, D = P.obs_dim, mt0 = wave * ENC_MT, nbw = P.nbr_dim * NB;
    traj_copy(P, a0, ENC_TA, B);
    for (int idx = tid; idx < ENC_TA * 128; idx += 64 * ENC_WAVES) {
        const int a = idx >> 7, c = idx & 127, ga = a0 + a;
        int col = -1;
        uint16_t *dst;
        if (c < 32) { dst = x_self + a * ENC_XS + c; if (c < P.self_dim) col = c; }
        else if (c < 64) { dst = x_obst + a * ENC_XS + (c - 32); if (c - 32 < P.obst_dim) col = P.self_dim + nbw + (c - 32); }
        else { dst = x_nbr + a * ENC_XW + (c - 64); if (c - 64 < nbw) col = P.self_dim + (c - 64); }
        put1<true>(dst, obs_at(obs_rsrc(obs, B, D), ga < B && col >= 0, (uint32_t)ga * (uint32_t)D + col));
    }
    __syncthreads();
    f32x4 resid[2][ENC_MT];
    if constexpr (S2R) {
        mlp1_keep<false, true>(P.s1, mt0, x_self, ENC_XS, cat, ENC_CS, resid[0]);
        mlp1_keep<true, true>(P.n1, mt0, x_nbr, ENC_XW, tok, ENC_YS, resid[0]);
        mlp1_keep<true, true>(P.o1, mt0, x_obst, ENC_XS, tok + ENC_TA * ENC_YS, ENC_YS, resid[1]);
    } else {
        mlp2_one_tile<true>(P.s1, P.s2, mt0, x_self, ENC_XS, hid, cat, ENC_CS, 0);
        __syncthreads();
        mlp2_keep<true>(P.n1, P.n2, mt0, x_nbr, ENC_XW, hid, tok, resid[0]);
        __syncthreads();
        mlp2_keep<true>(P.o1, P.o2, mt0, x_obst, ENC_XS, hid, tok + ENC_TA * ENC_YS, resid[1]);
    }
    __syncthreads();
    constexpr int QT = 2, QC = S2R ? 1 : 4;   // (two feature tiles per chunk: the fp16-pair GEMM holds two weight rings and two accumulator sets)
    float sc[2][2] = {{0, 0}, {0, 0}};
#pragma unroll 1
    for (int c = 0; c < QC; ++c) {
        f32x4 q[QT][2], k[QT][2];
        zero_acc<QT, 2>(q);
        gemm_tiles<QT, 2, true>(P.mq, (wave * QC + c) * QT, tok, ENC_YS, q);
        zero_acc<QT, 2>(k);
        gemm_tiles<QT, 2, true>(P.mk, (wave * QC + c) * QT, tok, ENC_YS, k);
#pragma unroll
        for (int mt = 0; mt < QT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) sc[i][j] += q[mt][i][r] * k[mt][j][r];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float t = lane_groups_sum(sc[i][j]);
            if (lane < 16) red_s[(wave * 4 + i * 2 + j) * 16 + lane] = t;
        }
    __syncthreads();   // (also: every wave is done with the inputs and `hid`, obuf may be written)
    float pr[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float t[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float acc_s = 0.0f;
            if constexpr (S2R) {
#pragma unroll
                for (int w = 0; w < ENC_WAVES; ++w) acc_s += red_s[(w * 4 + i * 2 + j) * 16 + (lane & 15)];
            } else
                acc_s = red_s[((wave & ~1) * 4 + i * 2 + j) * 16 + (lane & 15)] + red_s[((wave | 1) * 4 + i * 2 + j) * 16 + (lane & 15)];
            t[j] = acc_s * (1.0f / 16.0f);
        }
        const float m = fmaxf(t[0], t[1]), e0 = __expf(t[0] - m), e1 = __expf(t[1] - m), rd = 1.0f / (e0 + e1);
        pr[i][0] = e0 * rd;
        pr[i][1] = e1 * rd;
    }
    f32x4 y[ENC_MT][2];
#pragma unroll 1
    for (int i = 0; i < 2; ++i) {   // per query token: o_i = sum_j p_ij v_j -> obuf, then fc + residual -> y[.][i]
        const float p0 = i == 0 ? pr[0][0] : pr[1][0], p1 = i == 0 ? pr[0][1] : pr[1][1];
#pragma unroll 1
        for (int c = 0; c < QC; ++c) {
            f32x4 v[QT][2];
            zero_acc<QT, 2>(v);
            gemm_tiles<QT, 2, true>(P.mv, (wave * QC + c) * QT, tok, ENC_YS, v);
#pragma unroll
            for (int mt = 0; mt < QT; ++mt) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = p0 * v[mt][0][r] + p1 * v[mt][1][r];
                put4<true>(obuf + (lane & 15) * ENC_OS + ((wave * QC + c) * QT + mt) * 16 + (lane >> 4) * 4, o);
            }
        }
        __syncthreads();
        f32x4 yi[ENC_MT][1];
        zero_acc<ENC_MT, 1>(yi);
        gemm_tiles<ENC_MT, 1, true>(P.mfc, mt0, obuf, ENC_OS, yi);
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int mt = 0; mt < ENC_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = yi[mt][0][r] + (i == 0 ? resid[0][mt][r] : resid[1][mt][r]);
                if (i == 0) y[mt][0][r] = t; else y[mt][1][r] = t;
                s1 += t;
                s2 += t * t;
            }
        s1 = lane_groups_sum(s1);
        s2 = lane_groups_sum(s2);
        if (lane < 16) {
            red_ln[((wave * 2 + i) * 2 + 0) * 16 + lane] = s1;
            red_ln[((wave * 2 + i) * 2 + 1) * 16 + lane] = s2;
        }
        __syncthreads();   // every wave has read this token's obuf; the sums are visible
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int w = 0; w < ENC_WAVES; ++w) {
            s1 += red_ln[((w * 2 + i) * 2 + 0) * 16 + (lane & 15)];
            s2 += red_ln[((w * 2 + i) * 2 + 1) * 16 + (lane & 15)];
        }
        const float mean = s1 * (1.0f / ENC_H), var = fmaxf(s2 * (1.0f / ENC_H) - mean * mean, 0.0f), rstd = 1.0f / __builtin_sqrtf(var + 1e-6f);
#pragma unroll
        for (int mt = 0; mt < ENC_MT; ++mt) {
            const int f0 = (mt0 + mt) * 16 + (lane >> 4) * 4;
            const f32x4 g = *(const f32x4 *)(P.ln_w + f0), bb = *(const f32x4 *)(P.ln_b + f0);
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (y[mt][i][r] - mean) * rstd * g[r] + bb[r];
            put4<true>(cat + (lane & 15) * ENC_CS + ENC_H * (1 + i) + f0, o);
        }
    }
    __syncthreads();
    feed_forward<S2R ? ENC_MTF / 2 : ENC_MTF, true>(P, cat, a0, B, out, (float *)hid);
}
extern "C" __global__ void __launch_bounds__(64 * ENC_WAVES, 2) qs_encoder_mha_split_kernel(const float *__restrict__ obs, int B, EncParams P,
    float *__restrict__ out) {
    mha_body_split<false>(obs, B, P, out);
}
extern "C" __global__ void __launch_bounds__(64 * ENC_WAVES, 2) qs_encoder_s2r_split_kernel(const float *__restrict__ obs, int B, EncParams P,
    float *__restrict__ out) {
    mha_body_split<true>(obs, B, P, out);
}


// ------------------------------------------------------------------------------------------------
// mean_embed / mlp / no_encoder: one launch
// ------------------------------------------------------------------------------------------------
template <int NTH, bool SP = false>
__device__ __forceinline__ void mean_pass(const EncParams &P, int t0, const uint16_t *x_nbr, uint16_t *buf_a, f32x4 (&mean)[ENC_MT]) {
    const int wave = wave_id(), mt0 = wave * ENC_MT;
    f32x4 acc[ENC_MT][NTH];
    init_bias<ENC_MT, NTH>(P.n1, mt0, acc);
    gemm_tiles<ENC_MT, NTH, SP>(P.n1, mt0, x_nbr + t0 * ENC_TA * ENC_XS, ENC_XS, acc);
    ENC_STAMP(4);
    if (t0) __syncthreads();   // the previous pass's second layer is done reading buf_a
    store_tanh<ENC_MT, NTH, SP>(acc, mt0, buf_a, ENC_YS);
    __syncthreads();
    ENC_STAMP(5);
    init_bias<ENC_MT, NTH>(P.n2, mt0, acc);
    gemm_tiles<ENC_MT, NTH, SP>(P.n2, mt0, buf_a, ENC_YS, acc);
    ENC_STAMP(6);
    // e_i = tanh(.); the mean over neighbours is a sum over the row tiles (same lane, same register)
#pragma unroll
    for (int mt = 0; mt < ENC_MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTH; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mean[mt][r] += fast_tanh(acc[mt][nt][r]);
}

template <bool SP>
__device__ __forceinline__ void main_body(const float *__restrict__ obs, int B, const EncParams &P, float *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint16_t *x_self = (uint16_t *)smem;                              // [16][XS]
    uint16_t *x_nbr = x_self + ENC_TA * ENC_XS;                       // [NBR*16][XS]
    uint16_t *x_obst = x_nbr + ENC_MAX_NBR * ENC_TA * ENC_XS;         // [16][XS]
    uint16_t *buf_a = x_obst + ENC_TA * ENC_XS;                       // [NH*16][YS]  hidden layer of the neighbour MLP (one pass at a time)
    uint16_t *buf_b = buf_a + ENC_NH * ENC_TA * ENC_YS;               // [16][YS]     hidden layer of the self / obstacle MLPs
    uint16_t *cat = buf_b + ENC_TA * ENC_YS;                          // [16][CS]: self | neighbourhood | obstacles

    const int tid = threadIdx.x, wave = wave_id(), lane = tid & 63, a0 = blockIdx.x * ENC_TA;
    const int NB = P.num_nbr, D = P.obs_dim;
    const int mode = P.nbr_encoder;
    // no_encoder: the neighbour columns are in the row but nothing reads them (:289-291)
    const bool nbr_enc = NB > 0 && mode != ENC_NBR_NONE;
    const int col_nbr = ENC_H, col_obst = ENC_H * (nbr_enc ? 2 : 1);   // column blocks of `cat` in the order of the reference's torch.cat
    const int mt0 = wave * ENC_MT;   // first of this wave's 16-feature tiles of a 256-wide layer

    ENC_STAMP(0);
    traj_copy(P, a0, ENC_TA, B);
    // ---- stage the observation rows as bf16, zero padded to K = 32 ----
    // The 16 rows of the workgroup are one contiguous block of obs: read it coalesced (every load issued before the first use),
    // then scatter each element to its slot of the self / neighbour / obstacle staging rows.
    {
        uint32_t *z = (uint32_t *)x_self;   // x_self, x_nbr, x_obst are contiguous: clear the padding first
        for (int idx = tid; idx < (2 + ENC_MAX_NBR) * ENC_TA * ENC_XS / 2; idx += 64 * ENC_WAVES) {
            z[idx] = 0;
            if constexpr (SP) z[ENC_SPLANE / 2 + idx] = 0;
        }
        // upper bound on elements per thread
        constexpr int PER = (ENC_TA * (32 + 32 * ENC_MAX_NBR + 32) + 64 * ENC_WAVES - 1) / (64 * ENC_WAVES);
        const int total = ENC_TA * D;
        const size_t first = (size_t)a0 * D;
        const __amdgpu_buffer_rsrc_t ors = obs_rsrc(obs, B, D);
        const uint32_t mD = div_magic(D), mN = div_magic(P.nbr_dim > 0 ? P.nbr_dim : 1);
        float v[PER];
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int idx = tid + it * 64 * ENC_WAVES;
            v[it] = obs_at(ors, idx < total, (uint32_t)first + idx);   // rows past the batch: beyond the resource
        }
        __syncthreads();   // zeros are in place
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int idx = tid + it * 64 * ENC_WAVES;
            if (idx < total) {
                const int a = div_by(idx, mD), cidx = idx - a * D;
                uint16_t *dst;
                if (cidx < P.self_dim) dst = x_self + a * ENC_XS + cidx;
                else if (cidx < P.self_dim + P.nbr_dim * NB) {
                    const int q = cidx - P.self_dim, nb = div_by(q, mN), j = q - nb * P.nbr_dim;
                    dst = mode == ENC_NBR_MLP ? x_nbr + a * ENC_XW + q : x_nbr + (nb * ENC_TA + a) * ENC_XS + j;
                } else dst = x_obst + a * ENC_XS + (cidx - P.self_dim - P.nbr_dim * NB);
                put1<SP>(dst, v[it]);
            }
        }
    }
    __syncthreads();

    ENC_STAMP(1);
    mlp2_one_tile<SP>(P.s1, P.s2, mt0, x_self, ENC_XS, buf_b, cat, ENC_CS, 0);                  // self encoder -> cat[:, 0:256]
    ENC_STAMP(2);
    if (P.obst_dim > 0) {
        __syncthreads();
        mlp2_one_tile<SP>(P.o1, P.o2, mt0, x_obst, ENC_XS, buf_b, cat, ENC_CS, col_obst);       // obstacle encoder -> cat[:, 512:768]
    }
    __syncthreads();

    ENC_STAMP(3);
    // ---- neighbour encoder -> cat[:, 256:512] ----
    if (nbr_enc && mode == ENC_NBR_MLP) {
        // mlp neighbour encoder (:104-122): three layers on the concatenated neighbour observations of the agent
        f32x4 acc[ENC_MT][1];
        init_bias<ENC_MT, 1>(P.n1, mt0, acc);
        gemm_tiles<ENC_MT, 1, SP>(P.n1, mt0, x_nbr, ENC_XW, acc);
        store_tanh<ENC_MT, 1, SP>(acc, mt0, buf_a, ENC_YS);
        __syncthreads();
        mlp2_one_tile<SP>(P.n2, P.n3, mt0, buf_a, ENC_YS, buf_a + ENC_TA * ENC_YS, cat, ENC_CS, col_nbr);
    } else if (nbr_enc) {
        // mean_embed (:22-43) in passes of up to ENC_NH neighbour tiles: the hidden layer of the neighbour MLP is the largest LDS
        // buffer, and at half its size two workgroups fit one CU (the layer chain of one workgroup is latency-bound, a second overlaps it)
        f32x4 mean[ENC_MT];
#pragma unroll
        for (int mt = 0; mt < ENC_MT; ++mt) mean[mt] = (f32x4){0, 0, 0, 0};
        for (int t0 = 0; t0 < NB; t0 += ENC_NH) {
#define ENC_CALL(n) mean_pass<n, SP>(P, t0, x_nbr, buf_a, mean)
            ENC_DISPATCH_NT(NB - t0, ENC_NH, ENC_CALL)
#undef ENC_CALL
        }
        const float inv = 1.0f / (float)NB;   // torch.mean(neighbor_embeds, dim=1) (:41-42)
#pragma unroll
        for (int mt = 0; mt < ENC_MT; ++mt) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = mean[mt][r] * inv;
            put4<SP>(cat + (lane & 15) * ENC_CS + col_nbr + (mt0 + mt) * 16 + (lane >> 4) * 4, v);
        }
    }
    __syncthreads();

    ENC_STAMP(7);
    feed_forward<ENC_MTF, SP>(P, cat, a0, B, out, (float *)buf_a);
    ENC_STAMP(9);
}
extern "C" __global__ void __launch_bounds__(64 * ENC_WAVES, ENC_OCC) qs_encoder_kernel(const float *__restrict__ obs, int B,
    EncParams P, float *__restrict__ out) {
    main_body<false>(obs, B, P, out);
}
extern "C" __global__ void __launch_bounds__(64 * ENC_WAVES, 2) qs_encoder_split_kernel(const float *__restrict__ obs, int B,
    EncParams P, float *__restrict__ out) {
    main_body<true>(obs, B, P, out);
}


// ================================================================================================
// Wide variants: 32 agents per workgroup, one workgroup per CU (batches of >= ENC_WIDE_MIN agents).
//
// Phase stamps (tools/enc_stamps.py, profiles/r02_encoder_*): one 16-agent workgroup ALONE on the GPU needs 89 % of the time 512
// of them need - the kernel is the latency of one workgroup's chain of dependent layers, and a third of that chain is exposed L2
// latency: every layer starts with a load of its bias and first weight fragments (~650 ticks of a 36 k-tick chain each), a
// 256-wide layer waits a second time for the K-steps beyond the four-deep ring, the 512-wide feed-forward four times.  Here
//   * the weight ring holds a whole 256-wide layer (ENC_WPD = 8 K-steps) and is carried ACROSS layers: the slot an MFMA group of
//     the last ENC_WPD K-steps has consumed is refilled with the NEXT layer's fragment, so those loads fly during the tail of the
//     K loop, the tanh epilogue and the barrier, and the next layer starts on weights that are already in registers;
//   * the bias is added after the K loop instead of seeding the accumulators (its load is issued before the loop and first
//     needed behind it);
//   * 32 agents per workgroup halve the weight stream per agent and the barriers per agent; the register file of the lone
//     workgroup (2 waves per SIMD, 256 VGPRs) holds the deeper ring and the wider accumulator tiles.
// ================================================================================================
#define ENC_AT 2                    // agent tiles per workgroup
#define ENC_WA (16 * ENC_AT)        // agents per workgroup
#define ENC_WPD 8                   // weight ring depth (K-steps)
#define ENC_WSLOTS (ENC_MAX_NBR + 1)   // neighbour slots in the staging rows: ceil(8 / 3) * 3
struct WRing { bf16x8 a[ENC_WPD][ENC_MT]; };

// a layer without weights (w == nullptr, M == 0): every load is out of range and returns zero
__device__ __forceinline__ __amdgpu_buffer_rsrc_t layer_rsrc(const EncLayer &L) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)L.w, 0, L.M * L.K * 2, 0x00020000);
}
#define ENC_RFRAG(rs, mtile, kst, mt, ks) ENC_WLOAD(__builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (((mtile) + (mt)) * (kst) + (ks)) * 1024, 0)))
// a ring slot past the layer's K-steps (the 32-wide input layers fill one of the eight): through a resource of zero records - the load
// returns zero and moves no data (in range it would read the next feature tiles' fragments: 14 KiB per wave and layer of traffic on
// the CU's 64 B / clock L2 port that nobody multiplies - a third more than the network's weights, ahead of the observation loads)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t layer_rsrc_k(const EncLayer &L, int ks) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)L.w, 0, ks < (L.K >> 5) ? L.M * L.K * 2 : 0, 0x00020000);
}
#define ENC_RFRAG_K(L, mtile, kst, mt, ks) ENC_RFRAG(layer_rsrc_k(L, ks), mtile, kst, mt, ks)

__device__ __forceinline__ void ring_fill(WRing &R, const EncLayer &L, int mtile0) {
    const uint32_t voff = (threadIdx.x & 63) * 16;
    const int kst = L.K >> 5;
    const __amdgpu_buffer_rsrc_t rs = layer_rsrc(L);
#pragma unroll
    for (int s = 0; s < ENC_WPD; ++s)
#pragma unroll
        for (int mt = 0; mt < ENC_MT; ++mt) R.a[s][mt] = ENC_RFRAG_K(L, mtile0, kst, mt, s);
}

// acc (+)= L[features of (wave, mt)] x X[row tiles]; the ring holds L's first ENC_WPD K-steps on entry and Ln's on exit.
// KS = K / 32 of L is a template parameter: a run-time K-step count puts branches and a loop around the loads, behind which the
// compiler no longer knows how many are in flight and waits for ALL of them (s_waitcnt vmcnt(0)) at the next use of any loaded
// value - i.e. for the whole prefetched next layer at the end of every layer.
template <int NT, int KS>
__device__ __forceinline__ void gemm_ring(WRing &R, const EncLayer &L, int mtile0, const EncLayer &Ln, int mtile0n, const uint16_t *X,
    int xstride,
                                          f32x4 (&acc)[ENC_MT][NT]) {
    const int lane = threadIdx.x & 63, kn = Ln.K >> 5;
    const uint16_t *xrow = X + (lane & 15) * xstride + 8 * (lane >> 4);
    const uint32_t voff = lane * 16;
    const __amdgpu_buffer_rsrc_t rs = layer_rsrc(L), rsn = layer_rsrc(Ln);
    bf16x8 b[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = ENC_XFRAG(nt, 0);
    if constexpr (KS < ENC_WPD) {   // the 32- and 64-wide input layers
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                mfma_tile<ENC_MT, NT>(R.a[s], b[nt], acc, nt);
                if (s + 1 < KS) b[nt] = ENC_XFRAG(nt, s + 1);
            }
#pragma unroll
        for (int s = 0; s < ENC_WPD; ++s)
#pragma unroll
            for (int mt = 0; mt < ENC_MT; ++mt) R.a[s][mt] = ENC_RFRAG_K(Ln, mtile0n, kn, mt, s);
    } else {
        static_assert(KS % ENC_WPD == 0, "K-steps of a hidden layer: a multiple of the ring depth");
#pragma unroll
        for (int ks0 = 0; ks0 + ENC_WPD < KS; ks0 += ENC_WPD) {   // K > 256: the ring is refilled with this layer's next K-steps
#pragma unroll
            for (int s = 0; s < ENC_WPD; ++s) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    mfma_tile<ENC_MT, NT>(R.a[s], b[nt], acc, nt);
                    b[nt] = ENC_XFRAG(nt, ks0 + s + 1);
                }
#pragma unroll
                for (int mt = 0; mt < ENC_MT; ++mt) R.a[s][mt] = ENC_RFRAG(rs, mtile0, KS, mt, ks0 + s + ENC_WPD);
                // keep the K-steps in program order: hoisted LDS reads of later K-steps cost 4 VGPRs per tile each
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int s = 0; s < ENC_WPD; ++s) {   // the last ENC_WPD K-steps: each consumed slot takes the next layer's fragment
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                mfma_tile<ENC_MT, NT>(R.a[s], b[nt], acc, nt);
                if (s + 1 < ENC_WPD) b[nt] = ENC_XFRAG(nt, KS - ENC_WPD + s + 1);
            }
#pragma unroll
            for (int mt = 0; mt < ENC_MT; ++mt) R.a[s][mt] = ENC_RFRAG_K(Ln, mtile0n, kn, mt, s);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

struct Bias { f32x4 v[ENC_MT]; };
__device__ __forceinline__ Bias load_bias(const EncLayer &L, int mtile0) {
    Bias b;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int mt = 0; mt < ENC_MT; ++mt) b.v[mt] = *(const f32x4 *)(L.b + (mtile0 + mt) * 16 + (lane >> 4) * 4);
    return b;
}
template <int NT>
__device__ __forceinline__ void add_bias(f32x4 (&acc)[ENC_MT][NT], const Bias &b) {
#pragma unroll
    for (int mt = 0; mt < ENC_MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mt][nt][r] += b.v[mt][r];
}
// one layer of the chain: acc = L X + b (fp32, before the non-linearity)
template <int NT, int KS>
__device__ __forceinline__ void layer_ring(WRing &R, const EncLayer &L, int mtile0, const EncLayer &Ln, int mtile0n, const uint16_t *X,
    int xstride,
                                           f32x4 (&acc)[ENC_MT][NT]) {
    const Bias b = load_bias(L, mtile0);
    zero_acc<ENC_MT, NT>(acc);
    gemm_ring<NT, KS>(R, L, mtile0, Ln, mtile0n, X, xstride, acc);
    add_bias<NT>(acc, b);
}

// the same without the bias: acc = L X, bc = b * ENC_TANH_C for the tanh4_bias epilogues (mean_embed: wide_body, pp_body)
struct BiasC { f32x4 v[ENC_MT]; };
template <int NT, int KS>
__device__ __forceinline__ void layer_ring_raw(WRing &R, const EncLayer &L, int mtile0, const EncLayer &Ln, int mtile0n, const uint16_t *X,
    int xstride,
                                               f32x4 (&acc)[ENC_MT][NT], BiasC &bc) {
    const Bias b = load_bias(L, mtile0);
    zero_acc<ENC_MT, NT>(acc);
    gemm_ring<NT, KS>(R, L, mtile0, Ln, mtile0n, X, xstride, acc);
#pragma unroll
    for (int mt = 0; mt < ENC_MT; ++mt) bc.v[mt] = b.v[mt] * ENC_TANH_C;
}

// A 512-wide layer (16 K-steps) on TWO rings: R holds its K-steps 0-7 and R2 its K-steps 8-15 on entry, the next layer's (or feature
// half's) on exit.  The feed-forward layer's 512 KB are half of the network's weights and four times its MFMA time on the CU's 64 B / clock
// port; with its first feature half resident when the layer starts (R through the ring as always, R2 filled while the neighbour MLP - which
// leaves the port three quarters idle - still runs), only the second half streams under the first half's K loop and epilogue.
template <int NT>
__device__ __forceinline__ void gemm_ring2(WRing &R, WRing &R2, const EncLayer &Ln, int mtile0n, const uint16_t *X, int xstride,
                                           f32x4 (&acc)[ENC_MT][NT]) {
    const int lane = threadIdx.x & 63, kn = Ln.K >> 5;
    const uint16_t *xrow = X + (lane & 15) * xstride + 8 * (lane >> 4);
    const uint32_t voff = lane * 16;
    bf16x8 b[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = ENC_XFRAG(nt, 0);
#pragma unroll
    for (int s = 0; s < 2 * ENC_WPD; ++s) {
        WRing &Q = s < ENC_WPD ? R : R2;
        const int q = s & (ENC_WPD - 1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            mfma_tile<ENC_MT, NT>(Q.a[q], b[nt], acc, nt);
            if (s + 1 < 2 * ENC_WPD) b[nt] = ENC_XFRAG(nt, s + 1);
        }
#pragma unroll
        for (int mt = 0; mt < ENC_MT; ++mt) Q.a[q][mt] = ENC_RFRAG_K(Ln, mtile0n, kn, mt, s);
        __builtin_amdgcn_sched_barrier(0);
    }
}
__device__ __forceinline__ void ring2_fill(WRing &R2, const EncLayer &L, int mtile0) {   // K-steps 8-15 of L
    const uint32_t voff = (threadIdx.x & 63) * 16;
    const int kst = L.K >> 5;
#pragma unroll
    for (int s = 0; s < ENC_WPD; ++s)
#pragma unroll
        for (int mt = 0; mt < ENC_MT; ++mt) R2.a[s][mt] = ENC_RFRAG_K(L, mtile0, kst, mt, ENC_WPD + s);
}
template <int NT>
__device__ __forceinline__ void layer_ring2_raw(WRing &R, WRing &R2, const EncLayer &L, int mtile0, const EncLayer &Ln, int mtile0n,
                                                const uint16_t *X, int xstride, f32x4 (&acc)[ENC_MT][NT], BiasC &bc) {
    const Bias b = load_bias(L, mtile0);
    zero_acc<ENC_MT, NT>(acc);
    gemm_ring2<NT>(R, R2, Ln, mtile0n, X, xstride, acc);
#pragma unroll
    for (int mt = 0; mt < ENC_MT; ++mt) bc.v[mt] = b.v[mt] * ENC_TANH_C;
}

// epilogues of the wide kernels: one row tile at a time (a scheduling fence after each - interleaving a dozen tanh chains costs
// more registers than it hides latency, and the weight ring has to stay resident through them)
template <int NT>
__device__ __forceinline__ void store_tanh_wide(const f32x4 (&acc)[ENC_MT][NT], int mtile0, uint16_t *Y, int ystride, int col0 = 0) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int mt = 0; mt < ENC_MT; ++mt) {
            const f32x4 t = tanh4(acc[mt][nt]);
            bf16x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (__bf16)t[r];
            *(bf16x4 *)(Y + (nt * 16 + (lane & 15)) * ystride + col0 + (mtile0 + mt) * 16 + (lane >> 4) * 4) = v;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int NT>
__device__ __forceinline__ void store_tanh_wide_b(const f32x4 (&acc)[ENC_MT][NT], const BiasC &bc, int mtile0, uint16_t *Y, int ystride, int col0 = 0) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int mt = 0; mt < ENC_MT; ++mt) {
            const f32x4 t = tanh4_bias(acc[mt][nt], bc.v[mt]);
            bf16x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (__bf16)t[r];
            *(bf16x4 *)(Y + (nt * 16 + (lane & 15)) * ystride + col0 + (mtile0 + mt) * 16 + (lane >> 4) * 4) = v;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// observation rows of the workgroup's ENC_WA agents -> bf16 staging rows (self [WA][XS] | neighbours [(k*WA + a)][XS] | obstacles [WA][XS]; the
// three are contiguous).  One lane = one 8-column chunk of one staging row: its eight observation elements through the buffer resource
// (a column past the row's width, a neighbour slot past the count, an agent past the batch: out of range, reads as zero - the padding
// needs no separate clearing), four v_cvt_pk_bf16_f32, one ds_write_b128.  Chunk-major order over rows padded to 6 waves: the chunk
// index is wave-uniform, and a chunk past every input width is written as zeros without loads.  No division, no lane-divergent branch,
// every LDS element written once (no barrier inside): ~40 instructions per lane and iteration where the element-wise version spent
// ~50 per ELEMENT on index arithmetic under exec masks - in a kernel that issues one instruction per ~5 ticks per wave.
__device__ __forceinline__ void stage_obs_wide(const float *__restrict__ obs, int B, const EncParams &P, int a0, uint16_t *x_self,
    uint16_t *x_nbr, uint16_t *x_obst) {
    (void)x_nbr; (void)x_obst;
    constexpr int ROWS = (2 + ENC_WSLOTS) * ENC_WA, ROWS_P = (ROWS + 63) / 64 * 64, ITERS = (4 * ROWS_P + 64 * ENC_WAVES - 1) / (64 * ENC_WAVES);
    static_assert(ENC_WA == 32 && ENC_XS % 8 == 0, "row decoding by shifts; 16-byte aligned chunks");
    const int tid = threadIdx.x, D = P.obs_dim, NB = P.num_nbr;
    const __amdgpu_buffer_rsrc_t ors = obs_rsrc(obs, B, D);
    const int maxdim = max(P.self_dim, max(P.nbr_dim, P.obst_dim));
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int idx = tid + it * 64 * ENC_WAVES;
        const int ch = __builtin_amdgcn_readfirstlane(idx / ROWS_P), row = idx - ch * ROWS_P, col0 = ch * 8;   // ROWS_P: a multiple of 64
        if (ch >= 4) break;
        const bool is_self = row < ENC_WA, is_obst = row >= (1 + ENC_WSLOTS) * ENC_WA;
        const int r = row - ENC_WA, nb = r >> 5;
        const int a = is_self ? row : (is_obst ? row - (1 + ENC_WSLOTS) * ENC_WA : (r & (ENC_WA - 1)));
        const int dim = is_self ? P.self_dim : (is_obst ? P.obst_dim : (nb < NB ? P.nbr_dim : 0));
        const int cbase = is_self ? 0 : (is_obst ? P.self_dim + P.nbr_dim * NB : P.self_dim + nb * P.nbr_dim);
        const int ga = a0 + a;
        const uint32_t first = (uint32_t)ga * (uint32_t)D + (uint32_t)(cbase + col0);
        const int left = (ga < B && row < ROWS) ? dim - col0 : 0;   // valid elements of this chunk
        bf16x8 h;
        if (col0 < maxdim) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = obs_at(ors, u < left, first + u);
#pragma unroll
            for (int u = 0; u < 8; ++u) h[u] = (__bf16)v[u];
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) h[u] = (__bf16)0.0f;
        }
        if (row < ROWS) *(bf16x8 *)(x_self + row * ENC_XS + col0) = h;
    }
}

// linear head on the features of the wide kernels (see feed_forward): red = [8 waves][8 heads][ENC_WA] floats; acc = the wave's tanh'd
// feed-forward outputs
__device__ __forceinline__ void wide_head(const EncParams &P, const f32x4 (&acc)[2][ENC_MT][ENC_AT], int a0, int B, float *red) {
    const int wave = wave_id(), lane = threadIdx.x & 63, mf0 = wave * ENC_MTF;
    {
        for (int hd = 0; hd < P.head_dim; ++hd) {
            float sp[ENC_AT];
#pragma unroll
            for (int h = 0; h < ENC_AT; ++h) sp[h] = 0.0f;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int mt = 0; mt < ENC_MT; ++mt) {
                    const f32x4 w = *(const f32x4 *)(P.head_w + hd * (2 * ENC_H) + (mf0 + hf * ENC_MT + mt) * 16 + (lane >> 4) * 4);
#pragma unroll
                    for (int h = 0; h < ENC_AT; ++h)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sp[h] += acc[hf][mt][h][r] * w[r];
                }
#pragma unroll
            for (int h = 0; h < ENC_AT; ++h) {
                const float t = lane_groups_sum(sp[h]);
                if (lane < 16) red[(wave * 8 + hd) * ENC_WA + h * 16 + lane] = t;
            }
        }
        __syncthreads();
        const int tid = threadIdx.x, hd = tid / ENC_WA, row = tid % ENC_WA;
        if (hd < P.head_dim && a0 + row < B) {
            float t = P.head_b[hd];
#pragma unroll
            for (int w = 0; w < ENC_WAVES; ++w) t += red[(w * 8 + hd) * ENC_WA + row];
            P.head_out[(size_t)(a0 + row) * P.head_dim + hd] = t;
            if (P.sample_log_std) P.act_out[(size_t)(a0 + row) * P.head_dim + hd] = sample_action(P, a0 + row, hd, t);
        }
    }
}

// feed forward on the ring: the wave's 64 output features as two 32-feature halves over the same `cat` rows
template <int KS>   // K-steps of the feed-forward layer: 16 ([self | neighbourhood]) or 24 (with obstacles)
__device__ __forceinline__ void feed_forward_wide(WRing &R, const EncParams &P, const uint16_t *cat, int a0, int B,
    float *__restrict__ out, float *red) {
    const int wave = wave_id(), lane = threadIdx.x & 63, mf0 = wave * ENC_MTF;
    const EncLayer none = {nullptr, nullptr, 0, 0};
    f32x4 acc[2][ENC_MT][ENC_AT];
    BiasC bc[2];
    layer_ring_raw<ENC_AT, KS>(R, P.f, mf0, P.f, mf0 + ENC_MT, cat, ENC_CS, acc[0], bc[0]);
    layer_ring_raw<ENC_AT, KS>(R, P.f, mf0 + ENC_MT, none, 0, cat, ENC_CS, acc[1], bc[1]);
    ENC_STAMP(8);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int mt = 0; mt < ENC_MT; ++mt)
#pragma unroll
            for (int h = 0; h < ENC_AT; ++h) {
                const int ga = a0 + h * 16 + (lane & 15);
                acc[hf][mt][h] = tanh4_bias(acc[hf][mt][h], bc[hf].v[mt]);
                if (out && ga < B) *(f32x4 *)(out + (size_t)ga * (2 * ENC_H) + (mf0 + hf * ENC_MT + mt) * 16 + (lane >> 4) * 4) = acc[hf][mt][h];
            }
    if (P.head_dim > 0) wide_head(P, acc, a0, B, red);
}

// mean += tanh(acc + b) of the row tiles of neighbours t0.. (a neighbour slot past the count: masked out)
template <int NT>
__device__ __forceinline__ void tanh_into_mean(const f32x4 (&acc)[ENC_MT][NT], const BiasC &bc, int t0, int num_nbr, f32x4 (&mean)[ENC_MT][ENC_AT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float keep = t0 + nt / ENC_AT < num_nbr ? 1.0f : 0.0f;
#pragma unroll
        for (int mt = 0; mt < ENC_MT; ++mt) mean[mt][nt % ENC_AT] += keep * tanh4_bias(acc[mt][nt], bc.v[mt]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// mean_embed, wide: per-neighbour MLP in passes of WNP neighbours (WNP * ENC_AT row tiles, tile = neighbour * ENC_AT + agent half).
// WNP is a template parameter picked per neighbour count at launch (one pass body, no run-time tile counts); a last pass that
// runs past the neighbour count works on zero rows and is masked out of the mean.
template <int WNP>
__device__ __forceinline__ void mean_pass_wide(WRing &R, const EncParams &P, int t0, const EncLayer &after, int mt_after,
    const uint16_t *x_nbr, uint16_t *buf_a,
                                               f32x4 (&mean)[ENC_MT][ENC_AT]) {
    constexpr int NT = WNP * ENC_AT;
    const int wave = wave_id(), mt0 = wave * ENC_MT;
    f32x4 acc[ENC_MT][NT];
    BiasC bc;
    layer_ring_raw<NT, 1>(R, P.n1, mt0, P.n2, mt0, x_nbr + t0 * ENC_WA * ENC_XS, ENC_XS, acc, bc);
    ENC_STAMP(4);
    if (t0) __syncthreads();   // the previous pass's second layer is done reading buf_a
    store_tanh_wide_b<NT>(acc, bc, mt0, buf_a, ENC_YS);
    __syncthreads();
    ENC_STAMP(5);
    layer_ring_raw<NT, 8>(R, P.n2, mt0, after, mt_after, buf_a, ENC_YS, acc, bc);
    ENC_STAMP(6);
    tanh_into_mean<NT>(acc, bc, t0, P.num_nbr, mean);
}

template <int WNP>
__device__ __forceinline__ void wide_body(const float *__restrict__ obs, int B, const EncParams &P, float *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint16_t *x_self = (uint16_t *)smem;                              // [WA][XS]
    uint16_t *x_nbr = x_self + ENC_WA * ENC_XS;                       // [WSLOTS*WA][XS]
    uint16_t *x_obst = x_nbr + ENC_WSLOTS * ENC_WA * ENC_XS;          // [WA][XS]
    uint16_t *buf_a = x_obst + ENC_WA * ENC_XS;                       // [3*WA][YS]   hidden layer of the neighbour MLP (one pass at a time)
    uint16_t *buf_b = buf_a + 3 * ENC_WA * ENC_YS;                    // [WA][YS]     hidden layer of the self / obstacle MLPs
    uint16_t *cat = buf_b + ENC_WA * ENC_YS;                          // [WA][CS]: self | neighbourhood | obstacles
    const int wave = wave_id(), lane = threadIdx.x & 63, a0 = blockIdx.x * ENC_WA, mt0 = wave * ENC_MT, NB = P.num_nbr;
    const bool obst = P.obst_dim > 0;
    const int col_nbr = ENC_H, col_obst = 2 * ENC_H;

    ENC_STAMP(0);
    WRing R;
    ring_fill(R, P.s1, mt0);   // in flight while the observations are staged
    traj_copy(P, a0, ENC_WA, B);
    stage_obs_wide(obs, B, P, a0, x_self, x_nbr, x_obst);
    __syncthreads();
    ENC_STAMP(1);
    {
        f32x4 acc[ENC_MT][ENC_AT];
        BiasC bc;
        layer_ring_raw<ENC_AT, 1>(R, P.s1, mt0, P.s2, mt0, x_self, ENC_XS, acc, bc);
        store_tanh_wide_b<ENC_AT>(acc, bc, mt0, buf_b, ENC_YS);
        __syncthreads();
        layer_ring_raw<ENC_AT, 8>(R, P.s2, mt0, obst ? P.o1 : P.n1, mt0, buf_b, ENC_YS, acc, bc);
        store_tanh_wide_b<ENC_AT>(acc, bc, mt0, cat, ENC_CS, 0);                              // self encoder -> cat[:, 0:256]
        ENC_STAMP(2);
        if (obst) {
            layer_ring_raw<ENC_AT, 1>(R, P.o1, mt0, P.o2, mt0, x_obst, ENC_XS, acc, bc);
            __syncthreads();   // the self encoder's second layer is done reading buf_b
            store_tanh_wide_b<ENC_AT>(acc, bc, mt0, buf_b, ENC_YS);
            __syncthreads();
            layer_ring_raw<ENC_AT, 8>(R, P.o2, mt0, P.n1, mt0, buf_b, ENC_YS, acc, bc);
            store_tanh_wide_b<ENC_AT>(acc, bc, mt0, cat, ENC_CS, col_obst);                   // obstacle encoder -> cat[:, 512:768]
        }
    }
    ENC_STAMP(3);
    f32x4 mean[ENC_MT][ENC_AT];
#pragma unroll
    for (int mt = 0; mt < ENC_MT; ++mt)
#pragma unroll
        for (int h = 0; h < ENC_AT; ++h) mean[mt][h] = (f32x4){0, 0, 0, 0};
#pragma unroll 1
    for (int t0 = 0; t0 < NB; t0 += WNP) {
        const bool last = t0 + WNP >= NB;
        mean_pass_wide<WNP>(R, P, t0, last ? P.f : P.n1, last ? wave * ENC_MTF : mt0, x_nbr, buf_a, mean);
    }
    const float inv = 1.0f / (float)NB;   // torch.mean(neighbor_embeds, dim=1) (:41-42)
#pragma unroll
    for (int mt = 0; mt < ENC_MT; ++mt)
#pragma unroll
        for (int h = 0; h < ENC_AT; ++h) {
            bf16x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (__bf16)(mean[mt][h][r] * inv);
            *(bf16x4 *)(cat + (h * 16 + (lane & 15)) * ENC_CS + col_nbr + (mt0 + mt) * 16 + (lane >> 4) * 4) = v;
        }
    __syncthreads();
    ENC_STAMP(7);
    if (obst) feed_forward_wide<24>(R, P, cat, a0, B, out, (float *)buf_a);
    else feed_forward_wide<16>(R, P, cat, a0, B, out, (float *)buf_a);
    ENC_STAMP(9);
}
#define ENC_WIDE_KERNEL(n)                                                                                                                          \
    extern "C" __global__ void __launch_bounds__(64 * ENC_WAVES, 2) qs_encoder_wide##n##_kernel(const float *__restrict__ obs, int B, EncParams P, \
                                                                                                 float *__restrict__ out) {                         \
        wide_body<n>(obs, B, P, out);                                                                                                               \
    }
ENC_WIDE_KERNEL(1) ENC_WIDE_KERNEL(2) ENC_WIDE_KERNEL(3)

// ------------------------------------------------------------------------------------------------
// mean_embed, 32 agents per workgroup, the two waves of every SIMD half a layer apart ("ping-pong").
//
// In wide_body all eight waves run the same phase at the same time: the two waves of a SIMD queue for its matrix pipe through every
// K loop and for its VALU through every tanh epilogue, and each of the two units idles through the other's phase.  Here the work is cut
// into JOBS - one layer on one group of row tiles: G (K loop: MFMA + LDS fragment reads + weight stream) then T (bias, tanh, bf16, LDS
// store) - and waves 4-7 (the second wave of SIMD 0-3) run the same job list ONE SLOT behind waves 0-3: while one wave of a SIMD is in a
// G the other is in a T, matrix pipe beside VALU (MI355X_MICROARCH.md, two waves per SIMD).  One s_barrier per slot keeps the two halves
// in that pairing.  A job that reads what job j wrote has to be at least two jobs behind j (the late half's T(j) ends one slot after
// the early half's); the list is ordered for that, with one empty job in front of the feed-forward layer:
//     n1(A) s1 n2(A) n1(B) s2 n2(B) [o1 - o2] - f(lo) f(hi)          A / B: the first / second WNP neighbours, 2 * WNP row tiles each
// LDS buffers as in wide_body; the single hidden buffer of the neighbour MLP is rewritten by T(n1(B)) two slots after the last G(n2(A))
// has read it.  Same MFMA order per output, same epilogues: the features are those of wide_body bit for bit.
// ------------------------------------------------------------------------------------------------
#define ENC_SLOT() __syncthreads()   // end of a slot: LDS writes of this wave's T visible, every wave of both halves has arrived
template <int WNP, bool OBST>
__device__ __forceinline__ void pp_body(const float *__restrict__ obs, int B, const EncParams &P, float *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint16_t *x_self = (uint16_t *)smem;                              // [WA][XS]
    uint16_t *x_nbr = x_self + ENC_WA * ENC_XS;                       // [WSLOTS*WA][XS]
    uint16_t *x_obst = x_nbr + ENC_WSLOTS * ENC_WA * ENC_XS;          // [WA][XS]
    uint16_t *buf_a = x_obst + ENC_WA * ENC_XS;                       // [3*WA][YS]   hidden layer of the neighbour MLP (one group at a time)
    uint16_t *buf_b = buf_a + 3 * ENC_WA * ENC_YS;                    // [WA][YS]     hidden layer of the self / obstacle MLPs
    uint16_t *cat = buf_b + ENC_WA * ENC_YS;                          // [WA][CS]: self | neighbourhood | obstacles
    constexpr int NT = WNP * ENC_AT, KSF = OBST ? 24 : 16;
    const int wave = wave_id(), lane = threadIdx.x & 63, a0 = blockIdx.x * ENC_WA, mt0 = wave * ENC_MT, mf0 = wave * ENC_MTF, NB = P.num_nbr;
    const bool late = wave >= ENC_WAVES / 2;
    const EncLayer none = {nullptr, nullptr, 0, 0};

    ENC_STAMP(0);
    WRing R;
    ring_fill(R, P.n1, mt0);   // in flight while the observations are staged
    traj_copy(P, a0, ENC_WA, B);
    stage_obs_wide(obs, B, P, a0, x_self, x_nbr, x_obst);
    __syncthreads();
    ENC_STAMP(1);
    if (late) ENC_SLOT();
    f32x4 accn[ENC_MT][NT], accs[ENC_MT][ENC_AT], mean[ENC_MT][ENC_AT];
    BiasC bcn, bcs;
#pragma unroll
    for (int mt = 0; mt < ENC_MT; ++mt)
#pragma unroll
        for (int h = 0; h < ENC_AT; ++h) mean[mt][h] = (f32x4){0, 0, 0, 0};
    // n1(A)
    layer_ring_raw<NT, 1>(R, P.n1, mt0, P.s1, mt0, x_nbr, ENC_XS, accn, bcn);
    ENC_SLOT();
    store_tanh_wide_b<NT>(accn, bcn, mt0, buf_a, ENC_YS);
    ENC_SLOT();
    // s1
    layer_ring_raw<ENC_AT, 1>(R, P.s1, mt0, P.n2, mt0, x_self, ENC_XS, accs, bcs);
    ENC_SLOT();
    store_tanh_wide_b<ENC_AT>(accs, bcs, mt0, buf_b, ENC_YS);
    ENC_SLOT();
    ENC_STAMP(2);
    // n2(A)
    layer_ring_raw<NT, 8>(R, P.n2, mt0, P.n1, mt0, buf_a, ENC_YS, accn, bcn);
    ENC_SLOT();
    tanh_into_mean<NT>(accn, bcn, 0, NB, mean);
    ENC_SLOT();
    ENC_STAMP(3);
    // n1(B)
    layer_ring_raw<NT, 1>(R, P.n1, mt0, P.s2, mt0, x_nbr + WNP * ENC_WA * ENC_XS, ENC_XS, accn, bcn);
    ENC_SLOT();
    store_tanh_wide_b<NT>(accn, bcn, mt0, buf_a, ENC_YS);
    ENC_SLOT();
    ENC_STAMP(4);
    // s2
    layer_ring_raw<ENC_AT, 8>(R, P.s2, mt0, P.n2, mt0, buf_b, ENC_YS, accs, bcs);
    ENC_SLOT();
    store_tanh_wide_b<ENC_AT>(accs, bcs, mt0, cat, ENC_CS, 0);                                       // self encoder -> cat[:, 0:256]
    ENC_SLOT();
    ENC_STAMP(5);
    // n2(B)
    WRing R2;
    if constexpr (!OBST) ring2_fill(R2, P.f, mf0);   // the feed-forward layer's K-steps 8-15 (gemm_ring2): in flight from here on
    layer_ring_raw<NT, 8>(R, P.n2, mt0, OBST ? P.o1 : P.f, OBST ? mt0 : mf0, buf_a, ENC_YS, accn, bcn);
    ENC_SLOT();
    tanh_into_mean<NT>(accn, bcn, WNP, NB, mean);
    {
        const float inv = 1.0f / (float)NB;   // torch.mean(neighbor_embeds, dim=1) (:41-42)
#pragma unroll
        for (int mt = 0; mt < ENC_MT; ++mt)
#pragma unroll
            for (int h = 0; h < ENC_AT; ++h) {
                bf16x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (__bf16)(mean[mt][h][r] * inv);
                *(bf16x4 *)(cat + (h * 16 + (lane & 15)) * ENC_CS + ENC_H + (mt0 + mt) * 16 + (lane >> 4) * 4) = v;
            }
    }
    ENC_SLOT();
    ENC_STAMP(6);
    if constexpr (OBST) {
        // o1 (buf_b: the last G(s2) read it two slots ago)
        layer_ring_raw<ENC_AT, 1>(R, P.o1, mt0, P.o2, mt0, x_obst, ENC_XS, accs, bcs);
        ENC_SLOT();
        store_tanh_wide_b<ENC_AT>(accs, bcs, mt0, buf_b, ENC_YS);
        ENC_SLOT();
        ENC_SLOT(); ENC_SLOT();   // (empty job)
        // o2
        layer_ring_raw<ENC_AT, 8>(R, P.o2, mt0, P.f, mf0, buf_b, ENC_YS, accs, bcs);
        ENC_SLOT();
        store_tanh_wide_b<ENC_AT>(accs, bcs, mt0, cat, ENC_CS, 2 * ENC_H);                           // obstacle encoder -> cat[:, 512:768]
        ENC_SLOT();
    }
    ENC_SLOT(); ENC_SLOT();   // (empty job: the feed-forward layer reads what the late half's T of the job before wrote)
    ENC_STAMP(7);
    // f: the wave's 64 output features as two 32-feature halves over the same `cat` rows
    f32x4 acc[2][ENC_MT][ENC_AT];
    BiasC bcf[2];
    if constexpr (OBST) layer_ring_raw<ENC_AT, KSF>(R, P.f, mf0, P.f, mf0 + ENC_MT, cat, ENC_CS, acc[0], bcf[0]);
    else layer_ring2_raw<ENC_AT>(R, R2, P.f, mf0, P.f, mf0 + ENC_MT, cat, ENC_CS, acc[0], bcf[0]);
    ENC_SLOT();
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        if (hf == 1) {
            ENC_SLOT();
            if constexpr (OBST) layer_ring_raw<ENC_AT, KSF>(R, P.f, mf0 + ENC_MT, none, 0, cat, ENC_CS, acc[1], bcf[1]);
            else layer_ring2_raw<ENC_AT>(R, R2, P.f, mf0 + ENC_MT, none, 0, cat, ENC_CS, acc[1], bcf[1]);
            ENC_SLOT();
            ENC_STAMP(8);
        }
#pragma unroll
        for (int mt = 0; mt < ENC_MT; ++mt)
#pragma unroll
            for (int h = 0; h < ENC_AT; ++h) {
                const int ga = a0 + h * 16 + (lane & 15);
                acc[hf][mt][h] = tanh4_bias(acc[hf][mt][h], bcf[hf].v[mt]);
                if (out && ga < B) *(f32x4 *)(out + (size_t)ga * (2 * ENC_H) + (mf0 + hf * ENC_MT + mt) * 16 + (lane >> 4) * 4) = acc[hf][mt][h];
            }
    }
    ENC_SLOT();
    if (!late) ENC_SLOT();
    if (P.head_dim > 0) wide_head(P, acc, a0, B, (float *)buf_a);
    ENC_STAMP(9);
}
#define ENC_PP_KERNEL(n)                                                                                                                           \
    extern "C" __global__ void __launch_bounds__(64 * ENC_WAVES, 2) qs_encoder_pp##n##_kernel(const float *__restrict__ obs, int B, EncParams P,  \
                                                                                               float *__restrict__ out) {                         \
        pp_body<n, false>(obs, B, P, out);                                                                                                          \
    }                                                                                                                                               \
    extern "C" __global__ void __launch_bounds__(64 * ENC_WAVES, 2) qs_encoder_pp##n##o_kernel(const float *__restrict__ obs, int B, EncParams P, \
                                                                                                float *__restrict__ out) {                         \
        pp_body<n, true>(obs, B, P, out);                                                                                                           \
    }
ENC_PP_KERNEL(1) ENC_PP_KERNEL(2) ENC_PP_KERNEL(3)

// ------------------------------------------------------------------------------------------------
// attention, wide.  Launch 1: e_i -> ebuf, g = W_m e_mean -> gbuf (see qs_encoder_embed_kernel).
// ------------------------------------------------------------------------------------------------
template <int WNP>
__device__ __forceinline__ void embed_wide_body(const float *__restrict__ obs, int B, const EncParams &P) {
    constexpr int NT = WNP * ENC_AT;
    extern __shared__ __align__(16) unsigned char smem[];
    uint16_t *x_in = (uint16_t *)smem;                                // [WSLOTS*WA][XS]
    uint16_t *buf_a = x_in + ENC_WSLOTS * ENC_WA * ENC_XS;            // [3*WA][YS]
    uint16_t *emean = buf_a + 3 * ENC_WA * ENC_YS;                    // [WA][YS]
    const int tid = threadIdx.x, wave = wave_id(), lane = tid & 63, a0 = blockIdx.x * ENC_WA;
    const int NB = P.num_nbr, D = P.obs_dim, mt0 = wave * ENC_MT;
    const EncLayer none = {nullptr, nullptr, 0, 0};
    WRing R;
    ring_fill(R, P.n1, mt0);
    traj_copy(P, a0, ENC_WA, B);
    {
        const float invB = 1.0f / (float)B;
        const __amdgpu_buffer_rsrc_t ors = obs_rsrc(obs, B, D);
#pragma unroll 6
        // 18 iterations; neighbour slots past NB are zero rows
        for (int idx = tid; idx < ENC_WSLOTS * ENC_WA * 32; idx += 64 * ENC_WAVES) {
            const int row = idx >> 5, c = idx & 31, k = row / ENC_WA, a = row % ENC_WA, ga = a0 + a;
            // self_obs.repeat(K, 1)  (:84)
            const uint32_t i_self = mod_batch((uint32_t)ga * (uint32_t)NB + (uint32_t)k, (uint32_t)B, invB) * (uint32_t)D + c;
            const uint32_t i_nbr = (uint32_t)ga * (uint32_t)D + P.self_dim + k * P.nbr_dim + (c - P.self_dim);
            const float v = obs_at(ors, ga < B && k < NB && c < P.self_dim + P.nbr_dim, c < P.self_dim ? i_self : i_nbr);
            x_in[row * ENC_XS + c] = __builtin_bit_cast(uint16_t, (__bf16)v);
        }
    }
    __syncthreads();
    f32x4 mean[ENC_MT][ENC_AT];
#pragma unroll
    for (int mt = 0; mt < ENC_MT; ++mt)
#pragma unroll
        for (int h = 0; h < ENC_AT; ++h) mean[mt][h] = (f32x4){0, 0, 0, 0};
#pragma unroll 1
    for (int t0 = 0; t0 < NB; t0 += WNP) {
        const bool last = t0 + WNP >= NB;
        f32x4 acc[ENC_MT][NT];
        layer_ring<NT, 1>(R, P.n1, mt0, P.n2, mt0, x_in + t0 * ENC_WA * ENC_XS, ENC_XS, acc);
        if (t0) __syncthreads();   // the previous pass is done reading buf_a
        store_tanh_wide<NT>(acc, mt0, buf_a, ENC_YS);
        __syncthreads();
        layer_ring<NT, 8>(R, P.n2, mt0, last ? P.a1m : P.n1, mt0, buf_a, ENC_YS, acc);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int k = t0 + nt / ENC_AT, ga = a0 + (nt % ENC_AT) * 16 + (lane & 15);
            const bool live = k < NB;
#pragma unroll
            for (int mt = 0; mt < ENC_MT; ++mt) {
                bf16x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float e = fast_tanh(acc[mt][nt][r]); mean[mt][nt % ENC_AT][r] += live ? e : 0.0f;
                    v[r] = (__bf16)e; }
                if (live && ga < B) *(bf16x4 *)(P.ebuf + ((size_t)ga * NB + k) * ENC_H + (mt0 + mt) * 16 + (lane >> 4) * 4) = v;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const float inv = 1.0f / (float)NB;   // e_mean (:90-91), then its half of the score MLP's first layer once per agent
#pragma unroll
    for (int mt = 0; mt < ENC_MT; ++mt)
#pragma unroll
        for (int h = 0; h < ENC_AT; ++h) {
            bf16x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (__bf16)(mean[mt][h][r] * inv);
            *(bf16x4 *)(emean + (h * 16 + (lane & 15)) * ENC_YS + (mt0 + mt) * 16 + (lane >> 4) * 4) = v;
        }
    __syncthreads();
    f32x4 g[ENC_MT][ENC_AT];
    zero_acc<ENC_MT, ENC_AT>(g);
    gemm_ring<ENC_AT, 8>(R, P.a1m, mt0, none, 0, emean, ENC_YS, g);
#pragma unroll
    for (int h = 0; h < ENC_AT; ++h) {
        const int ga = a0 + h * 16 + (lane & 15);
        if (ga < B) {
#pragma unroll
            for (int mt = 0; mt < ENC_MT; ++mt) *(f32x4 *)(P.gbuf + (size_t)ga * ENC_H + (mt0 + mt) * 16 + (lane >> 4) * 4) = g[mt][h];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// attention, wide.  Launch 2: groups of WNP neighbours (WNP * ENC_AT row tiles): score MLP, value MLP, online softmax (see attn_pass).
// ------------------------------------------------------------------------------------------------
struct AttnStateWide { f32x4 o[ENC_MT][ENC_AT]; float mx[ENC_AT], den[ENC_AT]; };

template <int WNP>
__device__ __forceinline__ void attn_wide_body(const float *__restrict__ obs, int B, const EncParams &P, float *__restrict__ out) {
    constexpr int NT = WNP * ENC_AT;
    extern __shared__ __align__(16) unsigned char smem[];
    uint16_t *x_self = (uint16_t *)smem;                              // [WA][XS]
    uint16_t *x_obst = x_self + ENC_WA * ENC_XS;                      // [WA][XS]
    uint16_t *buf_a = x_obst + ENC_WA * ENC_XS;                       // [3*WA][YS]  e_i of the group
    uint16_t *buf_h = buf_a + 3 * ENC_WA * ENC_YS;                    // [3*WA][YS]  hidden layers; first the self / obstacle MLPs'
    uint16_t *cat = buf_h + 3 * ENC_WA * ENC_YS;                      // [WA][CS]: self | neighbourhood | obstacles
    float *s_alpha = (float *)(cat + ENC_WA * ENC_CS);                // [8 waves][3*AT tiles][16] partial scores of the group
    const int tid = threadIdx.x, wave = wave_id(), lane = tid & 63, a0 = blockIdx.x * ENC_WA;
    const int NB = P.num_nbr, D = P.obs_dim, mt0 = wave * ENC_MT;
    const bool obst = P.obst_dim > 0;
    const int col_nbr = ENC_H, col_obst = 2 * ENC_H;
    const float invB = 1.0f / (float)B;

    WRing R;
    ring_fill(R, P.s1, mt0);
    {
        const __amdgpu_buffer_rsrc_t ors = obs_rsrc(obs, B, D);
#pragma unroll
        for (int idx = tid; idx < 2 * ENC_WA * 32; idx += 64 * ENC_WAVES) {   // self and obstacle columns as bf16, zero padded to K = 32
            const int which = idx / (ENC_WA * 32), a = (idx >> 5) % ENC_WA, c = idx & 31, ga = a0 + a;
            const int dim = which ? P.obst_dim : P.self_dim, col = which ? P.self_dim + P.nbr_dim * NB : 0;
            const float v = obs_at(ors, ga < B && c < dim, (uint32_t)ga * (uint32_t)D + col + c);
            (which ? x_obst : x_self)[a * ENC_XS + c] = __builtin_bit_cast(uint16_t, (__bf16)v);
        }
    }
    __syncthreads();
    {
        f32x4 acc[ENC_MT][ENC_AT];
        layer_ring<ENC_AT, 1>(R, P.s1, mt0, P.s2, mt0, x_self, ENC_XS, acc);
        store_tanh_wide<ENC_AT>(acc, mt0, buf_h, ENC_YS);
        __syncthreads();
        layer_ring<ENC_AT, 8>(R, P.s2, mt0, obst ? P.o1 : P.a1e, mt0, buf_h, ENC_YS, acc);
        store_tanh_wide<ENC_AT>(acc, mt0, cat, ENC_CS, 0);
        if (obst) {
            layer_ring<ENC_AT, 1>(R, P.o1, mt0, P.o2, mt0, x_obst, ENC_XS, acc);
            __syncthreads();   // the self encoder's second layer is done reading buf_h
            store_tanh_wide<ENC_AT>(acc, mt0, buf_h, ENC_YS);
            __syncthreads();
            layer_ring<ENC_AT, 8>(R, P.o2, mt0, P.a1e, mt0, buf_h, ENC_YS, acc);
            store_tanh_wide<ENC_AT>(acc, mt0, cat, ENC_CS, col_obst);
        }
    }
    AttnStateWide st;
#pragma unroll
    for (int h = 0; h < ENC_AT; ++h) {
        st.mx[h] = -3.0e38f; st.den[h] = 0.0f;
#pragma unroll
        for (int mt = 0; mt < ENC_MT; ++mt) st.o[mt][h] = (f32x4){0, 0, 0, 0};
    }
    const __amdgpu_buffer_rsrc_t ers = __builtin_amdgcn_make_buffer_rsrc((void *)P.ebuf, 0, (uint32_t)B * (uint32_t)NB * (ENC_H * 2),
        0x00020000);
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)P.gbuf, 0, (uint32_t)B * (ENC_H * 4), 0x00020000);
#pragma unroll 1
    for (int t0 = 0; t0 < NB; t0 += WNP) {
        const bool last = t0 + WNP >= NB;
        f32x4 acc[ENC_MT][NT];
        // score MLP, first layer on [e_i | e_mean.repeat(K, 1)]: W_e e_i + b + g[(a*K + k) mod B]   (:92-94): g seeds the accumulators
        // (issued first; it has landed by the time the e_i tile has made its round trip through the registers into LDS)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int k = t0 + nt / ENC_AT, ga = a0 + (nt % ENC_AT) * 16 + (lane & 15);
            const uint32_t j = mod_batch((uint32_t)ga * (uint32_t)NB + (uint32_t)k, (uint32_t)B, invB);
            // padding rows: out of range, reads zero
            const uint32_t off = (ga < B && k < NB) ? j * (ENC_H * 4) + (lane >> 4) * 16 : 0xffffffffu;
#pragma unroll
            for (int mt = 0; mt < ENC_MT; ++mt) acc[mt][nt] = __builtin_bit_cast(f32x4,
                __builtin_amdgcn_raw_buffer_load_b128(grs, off, (mt0 + mt) * 64, 0));
        }
        {   // e_i rows of the group in 16-byte chunks, coalesced
            constexpr int PER = NT * 16 * (ENC_H / 8) / (64 * ENC_WAVES);
            bf16x8 ev[PER];
#pragma unroll
            for (int it = 0; it < PER; ++it) {
                const int idx = tid + it * 64 * ENC_WAVES, row = idx >> 5, ch = idx & 31, k = t0 + row / ENC_WA, ra = a0 + row % ENC_WA;
                const uint32_t off = (ra < B && k < NB) ? ((uint32_t)ra * (uint32_t)NB + (uint32_t)k) * (ENC_H * 2) + ch * 16 : 0xffffffffu;
                ev[it] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ers, off, 0, 0));
            }
            if (t0) __syncthreads();   // the previous group's value layers are done with buf_a / buf_h
#pragma unroll
            for (int it = 0; it < PER; ++it) {
                const int idx = tid + it * 64 * ENC_WAVES, row = idx >> 5, ch = idx & 31;
                *(bf16x8 *)(buf_a + row * ENC_YS + ch * 8) = ev[it];
            }
        }
        const Bias b1 = load_bias(P.a1e, mt0);
        __syncthreads();   // e_i is in buf_a
        gemm_ring<NT, 8>(R, P.a1e, mt0, P.a2, mt0, buf_a, ENC_YS, acc);
        add_bias<NT>(acc, b1);
        store_tanh_wide<NT>(acc, mt0, buf_h, ENC_YS);
        __syncthreads();
        layer_ring<NT, 8>(R, P.a2, mt0, P.v1, mt0, buf_h, ENC_YS, acc);
        // last score layer 256 -> 1 straight from the accumulators (see attn_pass)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float sp = 0.0f;
#pragma unroll
            for (int mt = 0; mt < ENC_MT; ++mt) {
                const f32x4 w = *(const f32x4 *)(P.a3w + (mt0 + mt) * 16 + (lane >> 4) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) sp += fast_tanh(acc[mt][nt][r]) * w[r];
            }
            sp = lane_groups_sum(sp);
            if (lane < 16) s_alpha[(wave * (3 * ENC_AT) + nt) * 16 + lane] = sp;
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();   // partial scores visible; every wave is done reading buf_h (second score layer)
        layer_ring<NT, 8>(R, P.v1, mt0, P.v2, mt0, buf_a, ENC_YS, acc);
        store_tanh_wide<NT>(acc, mt0, buf_h, ENC_YS);
        __syncthreads();
        layer_ring<NT, 8>(R, P.v2, mt0, last ? P.f : P.a1e, last ? wave * ENC_MTF : mt0, buf_h, ENC_YS, acc);
        // online softmax over the neighbours of agent (h, lane & 15)   (:95-100)
#pragma unroll
        for (int h = 0; h < ENC_AT; ++h) {
            float al[WNP], mx = st.mx[h];
#pragma unroll
            for (int j = 0; j < WNP; ++j) {
                al[j] = P.a3b;
#pragma unroll
                for (int w = 0; w < ENC_WAVES; ++w) al[j] += s_alpha[(w * (3 * ENC_AT) + j * ENC_AT + h) * 16 + (lane & 15)];
                if (t0 + j >= NB) al[j] = -3.0e38f;   // padded neighbour slot of the last group
                mx = fmaxf(mx, al[j]);
            }
            const float scale = __expf(st.mx[h] - mx);
            st.den[h] *= scale;
#pragma unroll
            for (int mt = 0; mt < ENC_MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) st.o[mt][h][r] *= scale;
#pragma unroll
            for (int j = 0; j < WNP; ++j) {
                const float e = t0 + j < NB ? __expf(al[j] - mx) : 0.0f;
                st.den[h] += e;
#pragma unroll
                for (int mt = 0; mt < ENC_MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) st.o[mt][h][r] += e * fast_tanh(acc[mt][j * ENC_AT + h][r]);
                __builtin_amdgcn_sched_barrier(0);
            }
            st.mx[h] = mx;
        }
    }
#pragma unroll
    for (int h = 0; h < ENC_AT; ++h) {
        const float rden = 1.0f / st.den[h];
#pragma unroll
        for (int mt = 0; mt < ENC_MT; ++mt) {
            bf16x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (__bf16)(st.o[mt][h][r] * rden);
            *(bf16x4 *)(cat + (h * 16 + (lane & 15)) * ENC_CS + col_nbr + (mt0 + mt) * 16 + (lane >> 4) * 4) = v;
        }
    }
    __syncthreads();
    if (obst) feed_forward_wide<24>(R, P, cat, a0, B, out, (float *)buf_a);
    else feed_forward_wide<16>(R, P, cat, a0, B, out, (float *)buf_a);
}
#define ENC_WIDE_ATT_KERNELS(n)                                                                                                                        \
    extern "C" __global__ void __launch_bounds__(64 * ENC_WAVES, 2) qs_encoder_embed_wide##n##_kernel(const float *__restrict__ obs, int B, EncParams P) { \
        embed_wide_body<n>(obs, B, P);                                                                                                                 \
    }                                                                                                                                                  \
    extern "C" __global__ void __launch_bounds__(64 * ENC_WAVES, 2) qs_encoder_attn_wide##n##_kernel(const float *__restrict__ obs, int B, EncParams P,   \
                                                                                                      float *__restrict__ out) {                       \
        attn_wide_body<n>(obs, B, P, out);                                                                                                             \
    }
ENC_WIDE_ATT_KERNELS(1) ENC_WIDE_ATT_KERNELS(2) ENC_WIDE_ATT_KERNELS(3)


// ------------------------------------------------------------------------------------------------
// Closed-loop glue (quad-swarm-rl_amd/rollout.py): what sits between the encoder and the environment step in a rollout segment,
// as ONE launch before the step (trajectory copy of the observations + Gaussian sampling of the actions from the head's mean)
// and ONE after it (trajectory copies of rewards / dones) instead of eight small framework kernels (copy, randn, exp, mul, add,
// copy, copy, copy) at 1.5 - 2 us each inside a HIP graph.  The noise is Philox4x32-10 keyed (seed, launch counter, agent); the
// counter lives in device memory and is advanced by the second launch, so a captured graph draws fresh noise on every replay.
// ------------------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(256) qs_rollout_pre_kernel(const float *__restrict__ obs, float *__restrict__ obs_out,
    int n_obs, const float *__restrict__ mean,
                                                                        const float *__restrict__ log_std, float *__restrict__ act_out,
                                                                            int A, uint32_t seed_lo,
                                                                        uint32_t seed_hi, const uint32_t *__restrict__ counter) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
    // 16-byte copies when both rows are 16-byte aligned (a trajectory slot obs[t] of A * D floats is only when A * D % 4 == 0)
    const int n4 = (((size_t)obs | (size_t)obs_out) & 15) == 0 ? n_obs >> 2 : 0;
    for (int i = tid; i < n4; i += nthreads) ((f32x4 *)obs_out)[i] = ((const f32x4 *)obs)[i];
    for (int i = (n4 << 2) + tid; i < n_obs; i += nthreads) obs_out[i] = obs[i];
    const bool act16 = ((((size_t)mean | (size_t)act_out)) & 15) == 0;
    for (int a = tid; a < A; a += nthreads) {
        f32x4 m;
        if (act16) m = *(const f32x4 *)(mean + (size_t)a * 4);
        else { m[0] = mean[(size_t)a * 4]; m[1] = mean[(size_t)a * 4 + 1]; m[2] = mean[(size_t)a * 4 + 2]; m[3] = mean[(size_t)a * 4 + 3]; }
        if (log_std) {   // action = mean + exp(log_std) * N(0, 1): two Box-Muller pairs from one Philox group
            uint32_t w[4];
            glue_philox((uint32_t)a, *counter, 0x51u, 0u, seed_lo, seed_hi, w);
            const float u0 = ((float)(w[0] >> 9) + 0.5f) * (1.0f / 8388608.0f), u1 = ((float)(w[1] >> 9) + 0.5f) * (1.0f / 8388608.0f);
            const float u2 = ((float)(w[2] >> 9) + 0.5f) * (1.0f / 8388608.0f), u3 = ((float)(w[3] >> 9) + 0.5f) * (1.0f / 8388608.0f);
            const float r0 = sqrtf(-2.0f * __logf(u0)), r1 = sqrtf(-2.0f * __logf(u2));
            float s0, c0, s1, c1;
            __sincosf(6.283185307179586f * u1, &s0, &c0);
            __sincosf(6.283185307179586f * u3, &s1, &c1);
            m[0] += __expf(log_std[0]) * r0 * c0; m[1] += __expf(log_std[1]) * r0 * s0;
            m[2] += __expf(log_std[2]) * r1 * c1; m[3] += __expf(log_std[3]) * r1 * s1;
        }
        if (act16) *(f32x4 *)(act_out + (size_t)a * 4) = m;
        else { act_out[(size_t)a * 4] = m[0]; act_out[(size_t)a * 4 + 1] = m[1]; act_out[(size_t)a * 4 + 2] = m[2];
            act_out[(size_t)a * 4 + 3] = m[3]; }
    }
}
extern "C" __global__ void __launch_bounds__(256) qs_rollout_post_kernel(const float *__restrict__ rew, float *__restrict__ rew_out,
    const uint8_t *__restrict__ done,
                                                                         uint8_t *__restrict__ done_out, int A,
                                                                             uint32_t *__restrict__ counter) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
    for (int a = tid; a < A; a += nthreads) { rew_out[a] = rew[a]; done_out[a] = done[a]; }
    if (tid == 0) *counter += 1u;
}

// ------------------------------------------------------------------------------------------------
// C ABI (include/quadswarm_encoder.h)
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_enc_error;
extern "C" {

const char *qs_enc_last_error(void) { return g_enc_error.c_str(); }
size_t qs_enc_sizeof_params(void) { return sizeof(EncParams); }

static size_t lds_main(int attention) {
    if (attention) return sizeof(uint16_t) * (ENC_TA * ENC_XS * 2 + 2 * ENC_ANH * ENC_TA * ENC_YS + ENC_TA * ENC_CS) + sizeof(float) * ENC_WAVES * ENC_ANH * 16;
    return sizeof(uint16_t) * (ENC_TA * ENC_XS * 2 + ENC_MAX_NBR * ENC_TA * ENC_XS + ENC_NH * ENC_TA * ENC_YS + ENC_TA * ENC_YS + ENC_TA * ENC_CS);
}
// reference precision: both planes of the largest 16-agent layout (+ the attention kernel's partial scores behind them)
static size_t lds_split(int attention) { return sizeof(uint16_t) * 2 * ENC_SPLANE + (attention ? sizeof(float) * ENC_WAVES * ENC_ANH * 16 : 0); }
static_assert(sizeof(uint16_t) * (ENC_TA * ENC_XS * 2 + ENC_MAX_NBR * ENC_TA * ENC_XS + ENC_NH * ENC_TA * ENC_YS + ENC_TA * ENC_YS + ENC_TA * ENC_CS) <=
                  sizeof(uint16_t) * ENC_SPLANE &&
              sizeof(uint16_t) * (ENC_TA * ENC_XS * 2 + 2 * ENC_ANH * ENC_TA * ENC_YS + ENC_TA * ENC_CS) <= sizeof(uint16_t) * ENC_SPLANE,
              "ENC_SPLANE holds one plane of every 16-agent layout");
static_assert(sizeof(uint16_t) * 2 * ENC_SPLANE + sizeof(float) * ENC_WAVES * ENC_ANH * 16 <= 160 * 1024, "two planes fit a CU's LDS");
static size_t lds_mha(void) {
    return sizeof(uint16_t) * (2 * ENC_TA * ENC_XS + ENC_TA * ENC_XW + 3 * ENC_TA * ENC_YS + 2 * ENC_TA * ENC_OS + ENC_TA * ENC_CS) + sizeof(float) * (4 * 2 * 4 * 16 + ENC_WAVES * 2 * 2 * 16);
}
static size_t lds_wide(void) { return sizeof(uint16_t) * ((2 + ENC_WSLOTS) * ENC_WA * ENC_XS + 3 * ENC_WA * ENC_YS + ENC_WA * ENC_YS + ENC_WA * ENC_CS); }
static size_t lds_embed_wide(void) { return sizeof(uint16_t) * (ENC_WSLOTS * ENC_WA * ENC_XS + 3 * ENC_WA * ENC_YS + ENC_WA * ENC_YS); }
static size_t lds_attn_wide(void) { return sizeof(uint16_t) * (2 * ENC_WA * ENC_XS + 6 * ENC_WA * ENC_YS + ENC_WA * ENC_CS) + sizeof(float) * ENC_WAVES * 3 * ENC_AT * 16; }
// Batches from this many agents on take the 32-agent workgroups (mean_embed, attention).  Default (-1): more agents than one
// 16-agent workgroup per CU can hold - up to there every 16-agent workgroup has a CU to itself and its shorter chain wins (measured
// 8192 / 4096 agents, us: mean_embed 28.4 / 18.9 narrow vs 25.0 / 22.3 wide, attention 76.5 / 51.0 vs 70.1 / 68.2;
// tools/enc_threshold.sh).  0 = never; QS_ENC_WIDE_MIN / qs_enc_set_wide_min override.
static int g_wide_min = [] { const char *e = getenv("QS_ENC_WIDE_MIN"); return e ? atoi(e) : -1; }();
static int wide_min_agents(int dev) {
    if (g_wide_min >= 0) return g_wide_min;
    static int cus[64] = {0};
    if (dev < 0 || dev >= 64) return 4097;
    if (!cus[dev]) { int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError();
        n = 256; } cus[dev] = n; }
    return ENC_TA * cus[dev] + 1;
}
// mean_embed on 32-agent workgroups: the ping-pong schedule (pp_body) or, QS_ENC_PP=0 / qs_enc_set_pingpong(0), the lock-step one (wide_body)
static int g_pp = [] { const char *e = getenv("QS_ENC_PP"); return e ? atoi(e) : 1; }();
int32_t qs_enc_set_pingpong(int32_t on) { const int prev = g_pp; if (on >= 0) g_pp = on != 0; return prev; }
// -1: the default rule; < -1: read only
int32_t qs_enc_set_wide_min(int32_t agents) { const int prev = g_wide_min; if (agents >= -1) g_wide_min = agents; return prev; }
static size_t lds_embed(void) { return sizeof(uint16_t) * (ENC_MAX_NBR * ENC_TA * ENC_XS + ENC_NH * ENC_TA * ENC_YS + ENC_TA * ENC_YS); }
size_t qs_enc_lds_bytes(void) { return lds_main(0); }
// LDS request of the kernel that serves `model` (QS_ENC_NBR_* / QS_ENC_MODEL_*): the multi-head and Sim2Real kernels ask for more than
// half of a CU's 160 KiB ON PURPOSE - one workgroup per CU by construction (an experiment with two co-resident
// workgroups of this body was not run-to-run deterministic and was never shipped); tests/test_c_abi.py pins that.
size_t qs_enc_lds_bytes_of(int32_t model) {
    if (model == ENC_MODEL_MHA || model == ENC_MODEL_S2R) return lds_mha();
    if (model == ENC_NBR_ATTENTION) return lds_embed() > lds_main(0) ? lds_embed() : lds_main(0);
    return lds_main(0);
}
size_t qs_enc_lds_bytes_split(int32_t attention) { return lds_split(attention); }

// out[B, 512] (ENC_MODEL_S2R: [B, 256]) = encoder(obs[B, obs_dim]); all pointers (obs, out, the weights / biases inside `params`) are
// device pointers
int qs_enc_forward(const float *obs, int32_t B, const EncParams *params, float *out, void *stream) {
    if (!obs || !params || B < 0) { g_enc_error = "bad argument"; return -1; }
    const EncParams &P = *params;
    if (P.head_dim < 0 || P.head_dim > 8 || (P.head_dim > 0 && (!P.head_w || !P.head_b || !P.head_out)) || (!out && P.head_dim == 0) ||
        (P.sample_log_std && (P.head_dim == 0 || !P.act_out || !P.sample_counter))) {
        g_enc_error = "bad argument";   // neither the features nor a head output requested, or an incomplete head
        return -1;
    }
    const bool att = P.nbr_encoder == ENC_NBR_ATTENTION && P.num_nbr > 0, s2r = P.nbr_encoder == ENC_MODEL_S2R,
        mha = P.nbr_encoder == ENC_MODEL_MHA || s2r, sp = P.precision == 1;
    if (P.precision != 0 && P.precision != 1) { g_enc_error = "precision: 0 (bf16) or 1 (reference precision, fp16 pairs)"; return -1; }
    if (P.num_nbr > ENC_MAX_NBR || P.self_dim > 32 || P.obst_dim > 32 || P.nbr_dim > 32 || P.nbr_encoder < 0
        || P.nbr_encoder > ENC_MODEL_S2R ||
        (att && P.self_dim + P.nbr_dim > 32) || ((P.nbr_encoder == ENC_NBR_MLP || mha) && P.nbr_dim * P.num_nbr > 64) ||
        (mha && (P.num_nbr < 1 || P.obst_dim < 1 || !P.ln_w || !P.ln_b))) {
        g_enc_error = "unsupported encoder shape (inputs wider than 32 - 64 for the mlp neighbour encoder - or more than 8 neighbours)";
        return -4;
    }
    if (att && !P.a3w) { g_enc_error = "the attention neighbour encoder needs the last score layer's weight row (a3w)"; return -1; }
    if (att && (!P.ebuf || !P.gbuf)) { g_enc_error = "the attention neighbour encoder needs the ebuf / gbuf scratch buffers"; return -1; }
    if (att && (int64_t)B * P.num_nbr * (ENC_H * 2) * (sp ? 2 : 1) > 0x7fffffffll) { g_enc_error = "attention: batch x neighbours too large for 32-bit scratch offsets"; return -4; }
    if ((int64_t)B * P.obs_dim * 4 > 0xffffffffll) { g_enc_error = "batch x obs_dim too large for 32-bit observation offsets"; return -4; }
    if (B == 0) return 0;
    // launch on the device that owns `obs` (a process may drive several GPUs); the > 64 KB dynamic-LDS attribute is per device
    int dev = 0;
    {
        hipPointerAttribute_t pa;
        if (hipPointerGetAttributes(&pa, obs) == hipSuccess) dev = pa.device; else { (void)hipGetLastError(); (void)hipGetDevice(&dev); }
        if (hipSetDevice(dev) != hipSuccess) { g_enc_error = "hipSetDevice failed"; return -2; }
    }
    const size_t lds = lds_main(att);
    {
        static std::mutex attr_mutex;
        static uint64_t attr_set = 0;   // bit d: attributes set on device d
        std::lock_guard<std::mutex> lock(attr_mutex);
        if (dev < 0 || dev >= 64) { g_enc_error = "device index out of range"; return -2; }
        if (!(attr_set >> dev & 1)) {
            if (hipFuncSetAttribute((const void *)qs_encoder_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                (int)lds_main(0)) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                    (int)lds_main(1)) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_mha_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                    (int)lds_mha()) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_s2r_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                    (int)lds_mha()) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_embed_wide1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                    (int)lds_embed_wide()) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_embed_wide2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                    (int)lds_embed_wide()) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_embed_wide3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                    (int)lds_embed_wide()) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_attn_wide1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                    (int)lds_attn_wide()) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_attn_wide2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                    (int)lds_attn_wide()) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_attn_wide3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                    (int)lds_attn_wide()) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_wide1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                    (int)lds_wide()) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_wide2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                    (int)lds_wide()) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_wide3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                    (int)lds_wide()) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_pp1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_wide()) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_pp2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_wide()) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_pp3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_wide()) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_pp1o_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_wide()) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_pp2o_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_wide()) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_pp3o_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_wide()) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                    (int)lds_split(0)) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_embed_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                    (int)lds_split(0)) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_attn_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                    (int)lds_split(1)) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_mha_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                    (int)lds_split(0)) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_s2r_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                    (int)lds_split(0)) != hipSuccess ||
                hipFuncSetAttribute((const void *)qs_encoder_embed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                    (int)lds_embed()) != hipSuccess) {
                g_enc_error = "cannot raise the dynamic LDS limit";
                return -2;
            }
            attr_set |= 1ull << dev;
        }
    }
    const int wmin = wide_min_agents(dev);
    const bool wide = !sp && wmin > 0 && B >= wmin && P.num_nbr > 0 && (P.nbr_encoder == ENC_NBR_MEAN_EMBED || att);
    if (sp) {   // reference precision: the 16-agent bodies on fp16 pairs, one workgroup per CU (two LDS planes)
        const dim3 grid((B + ENC_TA - 1) / ENC_TA), block(64 * ENC_WAVES);
        if (s2r) hipLaunchKernelGGL(qs_encoder_s2r_split_kernel, grid, block, lds_split(0), (hipStream_t)stream, obs, B, P, out);
        else if (mha) hipLaunchKernelGGL(qs_encoder_mha_split_kernel, grid, block, lds_split(0), (hipStream_t)stream, obs, B, P, out);
        else if (att) {
            hipLaunchKernelGGL(qs_encoder_embed_split_kernel, grid, block, lds_split(0), (hipStream_t)stream, obs, B, P);
            hipLaunchKernelGGL(qs_encoder_attn_split_kernel, grid, block, lds_split(1), (hipStream_t)stream, obs, B, P, out);
        } else
            hipLaunchKernelGGL(qs_encoder_split_kernel, grid, block, lds_split(0), (hipStream_t)stream, obs, B, P, out);
    } else if (wide) {
        // neighbours per pass: the fewest padded neighbour slots, then the fewest passes (1 -> 1; 2, 4 -> 2; 3, 5, 6, 7, 8 -> 3)
        const dim3 grid((B + ENC_WA - 1) / ENC_WA), block(64 * ENC_WAVES);
        const int wnp = P.num_nbr == 1 ? 1 : (P.num_nbr == 2 || P.num_nbr == 4) ? 2 : 3;
        hipStream_t st = (hipStream_t)stream;
        if (att) {
            if (wnp == 1) {
                hipLaunchKernelGGL(qs_encoder_embed_wide1_kernel, grid, block, lds_embed_wide(), st, obs, B, P);
                hipLaunchKernelGGL(qs_encoder_attn_wide1_kernel, grid, block, lds_attn_wide(), st, obs, B, P, out);
            } else if (wnp == 2) {
                hipLaunchKernelGGL(qs_encoder_embed_wide2_kernel, grid, block, lds_embed_wide(), st, obs, B, P);
                hipLaunchKernelGGL(qs_encoder_attn_wide2_kernel, grid, block, lds_attn_wide(), st, obs, B, P, out);
            } else {
                hipLaunchKernelGGL(qs_encoder_embed_wide3_kernel, grid, block, lds_embed_wide(), st, obs, B, P);
                hipLaunchKernelGGL(qs_encoder_attn_wide3_kernel, grid, block, lds_attn_wide(), st, obs, B, P, out);
            }
        } else if (g_pp && (P.num_nbr == 2 || (P.num_nbr >= 4 && P.num_nbr <= 6))) {
            // two groups of ceil(K / 2) neighbours, the two waves of a SIMD half a layer apart (pp_body)
            const int half = (P.num_nbr + 1) / 2;
            const bool ob = P.obst_dim > 0;
            if (half == 1) hipLaunchKernelGGL(ob ? qs_encoder_pp1o_kernel : qs_encoder_pp1_kernel, grid, block, lds_wide(), st, obs, B, P, out);
            else if (half == 2) hipLaunchKernelGGL(ob ? qs_encoder_pp2o_kernel : qs_encoder_pp2_kernel, grid, block, lds_wide(), st, obs, B, P, out);
            else hipLaunchKernelGGL(ob ? qs_encoder_pp3o_kernel : qs_encoder_pp3_kernel, grid, block, lds_wide(), st, obs, B, P, out);
        } else if (wnp == 1)
            hipLaunchKernelGGL(qs_encoder_wide1_kernel, grid, block, lds_wide(), st, obs, B, P, out);
        else if (wnp == 2)
            hipLaunchKernelGGL(qs_encoder_wide2_kernel, grid, block, lds_wide(), st, obs, B, P, out);
        else
            hipLaunchKernelGGL(qs_encoder_wide3_kernel, grid, block, lds_wide(), st, obs, B, P, out);
    } else if (s2r)
        hipLaunchKernelGGL(qs_encoder_s2r_kernel, dim3((B + ENC_TA - 1) / ENC_TA), dim3(64 * ENC_WAVES), lds_mha(), (hipStream_t)stream,
            obs, B, P, out);
    else if (mha)
        hipLaunchKernelGGL(qs_encoder_mha_kernel, dim3((B + ENC_TA - 1) / ENC_TA), dim3(64 * ENC_WAVES), lds_mha(), (hipStream_t)stream,
            obs, B, P, out);
    else if (att) {
        hipLaunchKernelGGL(qs_encoder_embed_kernel, dim3((B + ENC_TA - 1) / ENC_TA), dim3(64 * ENC_WAVES), lds_embed(),
            (hipStream_t)stream, obs, B, P);
        hipLaunchKernelGGL(qs_encoder_attn_kernel, dim3((B + ENC_TA - 1) / ENC_TA), dim3(64 * ENC_WAVES), lds, (hipStream_t)stream, obs,
            B, P, out);
    } else
        hipLaunchKernelGGL(qs_encoder_kernel, dim3((B + ENC_TA - 1) / ENC_TA), dim3(64 * ENC_WAVES), lds, (hipStream_t)stream, obs, B, P,
            out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_enc_error = hipGetErrorString(e); return -2; }
    return 0;
}

// rollout glue, see the kernels above: obs[n_obs] -> obs_out, act_out[A, 4] = mean[A, 4] (+ exp(log_std[4]) * N(0, 1) if log_std != NULL)
int qs_rollout_pre(const float *obs, float *obs_out, int32_t n_obs, const float *mean, const float *log_std, float *act_out, int32_t A,
    uint64_t seed,
                   const uint32_t *counter, void *stream) {
    if ((n_obs > 0 && (!obs || !obs_out)) || !mean || !act_out || !counter || n_obs < 0 || A < 0) { g_enc_error = "bad argument";
        return -1; }
    if (A == 0 && n_obs == 0) return 0;
    const int work = (n_obs >> 2) > A ? (n_obs >> 2) : A, blocks = (work + 255) / 256 < 2048 ? (work + 255) / 256 : 2048;
    hipLaunchKernelGGL(qs_rollout_pre_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, (hipStream_t)stream, obs, obs_out, n_obs,
        mean, log_std, act_out, A,
                       (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), counter);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_enc_error = hipGetErrorString(e); return -2; }
    return 0;
}
// rew[A] -> rew_out, done[A] -> done_out, *counter += 1 (the next qs_rollout_pre draws new noise)
int qs_rollout_post(const float *rew, float *rew_out, const uint8_t *done, uint8_t *done_out, int32_t A, uint32_t *counter, void *stream) {
    if (!rew || !rew_out || !done || !done_out || !counter || A < 0) { g_enc_error = "bad argument"; return -1; }
    hipLaunchKernelGGL(qs_rollout_post_kernel, dim3((A + 255) / 256 > 0 ? ((A + 255) / 256 < 2048 ? (A + 255) / 256 : 2048) : 1),
        dim3(256), 0, (hipStream_t)stream, rew, rew_out, done, done_out, A, counter);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_enc_error = hipGetErrorString(e); return -2; }
    return 0;
}

#ifdef ENC_TIMING
int qs_enc_stamps(unsigned long long *out16) { return hipMemcpyFromSymbol(out16, HIP_SYMBOL(enc_stamps),
    sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -2; }
int qs_enc_wg_times(unsigned long long *out,
    int n) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(enc_wg_times), sizeof(unsigned long long) * 2 * n) == hipSuccess ? 0 : -2; }
#endif
// `iters` back-to-back forward passes timed with HIP events on `stream` (no host work in between): average ms per pass
int qs_enc_benchmark(const float *obs, int32_t B, const EncParams *params, float *out, void *stream, int32_t iters, double *avg_ms) {
    if (iters < 1 || !avg_ms) { g_enc_error = "bad argument"; return -1; }
    int rc = qs_enc_forward(obs, B, params, out, stream);   // warm-up (+ sets the LDS attribute)
    if (rc != 0) return rc;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { g_enc_error = "hipEventCreate failed"; return -2; }
    (void)hipEventRecord(e0, (hipStream_t)stream);
    for (int i = 0; i < iters && rc == 0; ++i) rc = qs_enc_forward(obs, B, params, out, stream);
    (void)hipEventRecord(e1, (hipStream_t)stream);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_ms = (double)ms / iters;
    return rc;
}
}
