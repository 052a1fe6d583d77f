/*
 * quadswarm_encoder.h - C ABI of the fused policy-encoder forward pass (MI355X / gfx950 matrix cores).
 *
 * Replaces, for inference during rollouts, the forward() of the reference's QuadMultiEncoder
 * (swarm_rl/models/quad_multi_model.py:250-350): self MLP, neighbour encoder, optional obstacle MLP, feed-forward; tanh;
 * hidden size 256; output [B, 512] fp32.  All four --quads_neighbor_encoder_type choices: `mean_embed` (:22-43, per-neighbour
 * MLP + mean), `attention` (:46-101; two launches, needs the two scratch buffers below), `mlp` (:104-122) and `no_encoder`
 * (:289-291: the neighbour columns are part of the row but are not read).  QS_ENC_MODEL_MHA selects the other encoder class
 * of that file, QuadMultiHeadAttentionEncoder (:124-196); QS_ENC_MODEL_S2R its --quads_sim2real subclass
 * QuadSingleHeadAttentionEncoder_Sim2Real (:199-248; output [B, 256]).
 * bf16 weights / activations, fp32 accumulation (v_mfma_f32_16x16x32_bf16).  Weights are handed over pre-packed:
 *   layer with torch weight W[M_real, K_real], bias b[M_real]  ->  M = ceil16(M_real), K = ceil32(K_real), zero padded,
 *   w[((mt * (K/32) + ks) * 64 + lane) * 8 + j] = bf16(W[mt*16 + (lane & 15)][ks*32 + 8*(lane >> 4) + j]),  b as fp32[M].
 * (quad-swarm-rl_amd/policy.py does the packing from a torch module.)  All pointers are device pointers.
 *
 * Reference precision (qs_enc_params.precision = 1).  The reference's modules run in fp32; with bf16 operands the features come out
 * ~1e-2 away from them.  For a sampler whose learner is fp32 every GEMM operand - weight or activation - can instead be a PAIR of fp16
 * numbers, x = h + l / 2048 (h = fp16(x), 0 below 2^-14; l = fp16((x - h) * 2048)): three v_mfma_f32_16x16x32_f16 per product, fp32
 * accumulation, ~2^-22 relative per product; features within 1e-5 of the fp32 module (tests/test_policy_encoder_gpu.py).  Weights are then
 * packed as two 1 KiB planes per fragment:
 *   w[(((mt * (K/32) + ks) * 2 + plane) * 64 + lane) * 8 + j] = (plane ? l : h)(W[mt*16 + (lane & 15)][ks*32 + 8*(lane >> 4) + j]),
 * `ebuf` rows are the two planes [2][256], and the 16-agent kernels run with one workgroup per CU (two LDS planes).  Built for every
 * encoder class: QuadMultiEncoder's four neighbour encoders and the two multi-head classes (embeddings, q / k / v, output projection and
 * feed-forward layer on fp16 pairs; scores, softmax, residual and LayerNorm in fp32; the value projection runs once per query token).
 */
#ifndef QUADSWARM_ENCODER_H
#define QUADSWARM_ENCODER_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { QS_ENC_NBR_MEAN_EMBED = 0, QS_ENC_NBR_ATTENTION = 1, QS_ENC_NBR_MLP = 2, QS_ENC_NBR_NONE = 3,
       QS_ENC_MODEL_MHA = 4 /* QuadMultiHeadAttentionEncoder instead of QuadMultiEncoder */,
       QS_ENC_MODEL_S2R = 5 /* QuadSingleHeadAttentionEncoder_Sim2Real */ };

typedef struct qs_enc_layer { const uint16_t *w; const float *b; int32_t M, K; } qs_enc_layer;

typedef struct qs_enc_params {
    int32_t self_dim, nbr_dim, num_nbr, obst_dim, obs_dim;   /* obs row = [self | num_nbr x nbr_dim | obst] */
    int32_t nbr_encoder;   /* QS_ENC_NBR_*; attention needs self_dim + nbr_dim <= 32, mlp needs num_nbr * nbr_dim <= 64 */
    qs_enc_layer s1, s2;   /* self encoder      (quad_multi_model.py:303-309) */
    qs_enc_layer n1, n2, n3; /* mean_embed: neighbour MLP (:29-34) n1, n2, then the mean over neighbours (:41-42);
                              attention: embedding_mlp (:52-57) n1, n2, input [self obs | neighbour obs];
                              mlp: neighbor_mlp (:110-117) n1, n2, n3, input = all num_nbr x nbr_dim neighbour columns */
    qs_enc_layer o1, o2;   /* obstacle encoder  (:315-322), unused when obst_dim == 0 */
    qs_enc_layer v1, v2;   /* attention: neighbor_value_mlp (:60-65) */
    qs_enc_layer a1e, a1m; /* attention: attention_mlp[0] (:69) split by input columns: W[:, 0:256] with the bias (e_i half),
                              W[:, 256:512] (e_mean half; its bias pointer is not read) */
    qs_enc_layer a2;       /* attention: attention_mlp[2] */
    const float *a3w;      /* attention: attention_mlp[4] (256 -> 1): fp32 weight row [256] ... */
    float a3b;             /* ... and bias */
    int32_t precision;     /* 0: bf16 operands; 1: reference precision (fp16 pairs, see the top of this file) */
    uint16_t *ebuf;        /* attention scratch, device, bf16 [B * num_nbr, 256]; precision 1: fp16 [B * num_nbr, 2, 256] */
    float *gbuf;           /* attention scratch, device, fp32 [B, 256] */
    qs_enc_layer f;        /* feed forward      (:329-332): K = 256 * (1 + (neighbour encoder present) + (obst_dim > 0)), M = 512 */
    /* QS_ENC_MODEL_MHA only (quad_multi_model.py:124-196, --quads_encoder_type=attention; needs num_nbr >= 1, obst_dim >= 1):
       s1/s2 = self_embed_layer, n1/n2 = neighbor_embed_layer (input = all neighbour columns, <= 64), o1/o2 = obstacle_embed_layer,
       f = feed_forward (K = 768); MultiHeadAttention(4, 256, 256, 256) of swarm_rl/models/attention_layer.py:12-56: */
    qs_enc_layer mq, mk, mv; /* w_qs, w_ks, w_vs: M = 1024 (head-major), K = 256; no bias (b is not read) */
    qs_enc_layer mfc;        /* fc: M = 256, K = 1024; no bias */
    const float *ln_w, *ln_b;/* layer_norm weight / bias, fp32 [256] */
    /* QS_ENC_MODEL_S2R (:199-248): one-layer embeddings s1, n1, o1 (s2, n2, o2 are not read), OneHeadAttention(256)
       (attention_layer.py:56-97): mq, mk, mv, mfc all M = 256, K = 256; f: K = 768, M = 256; head_w is [head_dim, 256] */
    /* optional linear head on the encoder output, fused into the last kernel's epilogue (Sample Factory's action-parameter or
       value layer): head_out[B, head_dim] = features . head_w^T + head_b.  With a head, `out` of qs_enc_forward may be NULL and
       the [B, 512] features are then never written. */
    const float *head_w, *head_b;   /* fp32 [head_dim, 512], [head_dim] */
    float *head_out;                /* fp32 [B, head_dim] */
    int32_t head_dim;               /* 0: none; <= 8 */
    /* optional Gaussian sampling on the head's output, for rollout segments (SF's continuous action parameterisation with a
       state-independent std): act_out[B, head_dim] = head_out + exp(sample_log_std[h]) * N(0, 1).  Philox4x32-10 keyed
       (seed, *sample_counter + sample_step, agent): the same draws as qs_rollout_pre with that counter value.  `sample_step` is a
       launch argument (a captured graph gives every step its own), `sample_counter` lives in device memory (advance it once per
       replay and the graph draws fresh noise).  sample_log_std == NULL: no sampling. */
    uint32_t sample_step;
    const float *sample_log_std;    /* fp32 [head_dim] */
    float *act_out;                 /* fp32 [B, head_dim] */
    const uint32_t *sample_counter;
    uint32_t sample_seed_lo, sample_seed_hi;
    /* optional trajectory copy, for rollout segments: the FIRST kernel of the forward pass also copies, for every agent, the reward and the done
       flag of the environment step that produced the observations it is reading (traj_rew_src[B] -> traj_rew_dst[B], traj_done_src[B] ->
       traj_done_dst[B]) - the reward / done copy of step t rides on the policy's launch of step t + 1 instead of a launch of its own
       (qs_rollout_post).  traj_rew_dst == NULL: no copy. */
    const float *traj_rew_src;
    float *traj_rew_dst;
    const uint8_t *traj_done_src;
    uint8_t *traj_done_dst;
} qs_enc_params;

size_t qs_enc_sizeof_params(void);
size_t qs_enc_lds_bytes(void);
/* dynamic LDS the kernel of `model` (QS_ENC_NBR_* / QS_ENC_MODEL_*) requests.  QS_ENC_MODEL_MHA / _S2R: more than half of a CU's
 * 160 KiB, i.e. one workgroup per CU by construction. */
size_t qs_enc_lds_bytes_of(int32_t model);
/* ... of the reference-precision kernels (attention != 0: the second launch of the attention encoder). */
size_t qs_enc_lds_bytes_split(int32_t attention);
const char *qs_enc_last_error(void);

/* out[B, 512] (QS_ENC_MODEL_S2R: [B, 256]) = encoder(obs[B, obs_dim]) on `stream` (out may be NULL when params->head_dim > 0).  0 on success, < 0 on error
 * (qs_enc_last_error()). */
int qs_enc_forward(const float *obs, int32_t B, const qs_enc_params *params, float *out, void *stream);

/* The mean_embed and attention encoders have a second set of kernels with 32 agents per workgroup (half the weight stream per
 * agent), taken for batches of at least this many agents.  Default (-1): more agents than 16 x (number of CUs), i.e. as soon as
 * the 16-agent workgroups would have to share CUs (4097 on MI355X); 0 = never; environment QS_ENC_WIDE_MIN.  Returns the previous
 * value; an argument below -1 only reads. */
int32_t qs_enc_set_wide_min(int32_t agents);

/* mean_embed on the 32-agent workgroups (2, 4, 5 or 6 neighbours): 1 (default) = the two waves of every SIMD run the layer list half
 * a layer apart - one in a K loop on the matrix pipe while the other does a tanh epilogue on the VALU (qs_policy_encoder.hip pp_body);
 * 0 = all eight waves in the same phase (wide_body).  Same features bit for bit.  Environment QS_ENC_PP.  Returns the previous value; a
 * negative argument only reads. */
int32_t qs_enc_set_pingpong(int32_t on);

/* Closed-loop glue of a rollout segment (quad-swarm-rl_amd/rollout.py): one launch before the environment step - trajectory copy of
 * the observations, act_out[A, 4] = mean[A, 4] + exp(log_std[4]) * N(0, 1) (log_std NULL: the mean itself; Philox4x32-10 keyed by
 * seed, the device counter and the agent) - and one after it - trajectory copies of rewards / dones, counter += 1.  They replace
 * eight framework kernels per control step (reference side: Sample Factory's sampler loop around env.step, swarm_rl/train.py).
 * Alignment: any 4-byte aligned pointers work; 16-byte aligned obs / obs_out (and mean / act_out) take the 16-byte path. */
int qs_rollout_pre(const float *obs, float *obs_out, int32_t n_obs, const float *mean, const float *log_std, float *act_out, int32_t A, uint64_t seed,
                   const uint32_t *counter, void *stream);
int qs_rollout_post(const float *rew, float *rew_out, const uint8_t *done, uint8_t *done_out, int32_t A, uint32_t *counter, void *stream);

/* `iters` back-to-back forward passes timed with HIP events on `stream` (no host work in between): average ms per pass. */
int qs_enc_benchmark(const float *obs, int32_t B, const qs_enc_params *params, float *out, void *stream, int32_t iters, double *avg_ms);

#ifdef __cplusplus
}
#endif
#endif
