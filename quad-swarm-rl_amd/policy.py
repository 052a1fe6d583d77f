"""Policy-encoder inference next to the stepper.

`QuadMultiEncoderRef` is a plain-PyTorch restatement of the reference's QuadMultiEncoder with the `mean_embed` or the
`attention` neighbour encoder (swarm_rl/models/quad_multi_model.py:22-43, :46-101, :250-350; Sample Factory's `fc_layer` is `nn.Linear`, `nonlinearity` is
tanh in the reference's runs) - it is the fp32 reference the fused kernel is tested against and the source of its weights.
`FusedQuadEncoder` packs those weights for `csrc/qs_policy_encoder.hip` (include/quadswarm_encoder.h) and runs the forward
pass as ONE kernel (two for `attention`) that reads the stepper's observation buffer.  No CPU fallback: without the extension or a GPU it raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import native

ENC_LIB_PATH = os.environ.get("QS_ENC_LIB", os.path.join(native.CSRC, "libquadswarm_encoder.so"))
ENC_SOURCE = os.path.join(native.CSRC, "qs_policy_encoder.hip")
HIDDEN = 256


def build(force=False, verbose=False):
    if not force and os.path.exists(ENC_LIB_PATH) and os.path.getmtime(ENC_LIB_PATH) >= os.path.getmtime(ENC_SOURCE):
        return ENC_LIB_PATH
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", ENC_LIB_PATH, ENC_SOURCE]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    native.finish_library(ENC_LIB_PATH, verbose)   # the code-object check every library goes through
    return ENC_LIB_PATH


class EncLayer(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("M", C.c_int32), ("K", C.c_int32)]


class EncParams(C.Structure):
    _fields_ = [("self_dim", C.c_int32), ("nbr_dim", C.c_int32), ("num_nbr", C.c_int32), ("obst_dim", C.c_int32), ("obs_dim", C.c_int32),
                ("nbr_encoder", C.c_int32),
                ("s1", EncLayer), ("s2", EncLayer), ("n1", EncLayer), ("n2", EncLayer), ("n3", EncLayer), ("o1", EncLayer), ("o2", EncLayer),
                ("v1", EncLayer), ("v2", EncLayer), ("a1e", EncLayer), ("a1m", EncLayer), ("a2", EncLayer), ("a3w", C.c_void_p), ("a3b", C.c_float), ("precision", C.c_int32),
                ("ebuf", C.c_void_p), ("gbuf", C.c_void_p), ("f", EncLayer),
                ("mq", EncLayer), ("mk", EncLayer), ("mv", EncLayer), ("mfc", EncLayer), ("ln_w", C.c_void_p), ("ln_b", C.c_void_p),
                ("head_w", C.c_void_p), ("head_b", C.c_void_p), ("head_out", C.c_void_p), ("head_dim", C.c_int32),
                ("sample_step", C.c_uint32), ("sample_log_std", C.c_void_p), ("act_out", C.c_void_p), ("sample_counter", C.c_void_p),
                ("sample_seed_lo", C.c_uint32), ("sample_seed_hi", C.c_uint32),
                ("traj_rew_src", C.c_void_p), ("traj_rew_dst", C.c_void_p), ("traj_done_src", C.c_void_p), ("traj_done_dst", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ENC_LIB_PATH):
            build()
        native._preload_torch_hip_runtime()
        L = C.CDLL(ENC_LIB_PATH)
        L.qs_enc_last_error.restype = C.c_char_p
        L.qs_enc_sizeof_params.restype = C.c_size_t
        L.qs_enc_lds_bytes.restype = C.c_size_t
        L.qs_enc_lds_bytes_split.restype = C.c_size_t
        L.qs_enc_lds_bytes_split.argtypes = [C.c_int32]
        L.qs_enc_set_wide_min.argtypes = [C.c_int32]
        L.qs_enc_set_wide_min.restype = C.c_int32
        L.qs_enc_set_pingpong.argtypes = [C.c_int32]
        L.qs_enc_set_pingpong.restype = C.c_int32
        L.qs_rollout_pre.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p]
        L.qs_rollout_post.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.qs_enc_forward.argtypes = [C.c_void_p, C.c_int32, C.POINTER(EncParams), C.c_void_p, C.c_void_p]
        L.qs_enc_benchmark.argtypes = [C.c_void_p, C.c_int32, C.POINTER(EncParams), C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_double)]
        if L.qs_enc_sizeof_params() != C.sizeof(EncParams):
            raise RuntimeError("qs_enc_params layout mismatch between policy.py and libquadswarm_encoder.so")
        _lib = L
    return _lib


MODELS = ("mean_embed", "attention", "mlp", "no_encoder", "multi_head_attention", "single_head_sim2real")   # qs_enc_params.nbr_encoder
NBR_ENCODERS = ("mean_embed", "attention", "mlp", "no_encoder")   # --quads_neighbor_encoder_type (quadrotor_params.py:38-40); index = QS_ENC_NBR_*


NONLINEARITIES = ("tanh", "elu", "relu")   # sample_factory.model.model_utils.nonlinearity(cfg): --nonlinearity


def make_reference_encoder(self_dim=18, nbr_dim=6, num_nbr=6, obst_dim=0, hidden=HIDDEN, seed=0, attention=False, nbr_encoder=None,
                           nbr_hidden=None, obst_hidden=None, nonlinearity="tanh"):
    """QuadMultiEncoder as a torch module; random init (there are no checkpoints in this image).
    nbr_encoder: one of NBR_ENCODERS (default mean_embed; attention=True is shorthand for "attention").
    hidden = cfg.rnn_size (self encoder, feed forward), nbr_hidden = cfg.quads_neighbor_hidden_size (every layer of the neighbour
    encoder, :29-34,:52-75,:110-117), obst_hidden = cfg.quads_obst_hidden_size (:315-322); both default to `hidden`, which is what
    the published runs use.  nonlinearity: what SF's nonlinearity(cfg) returns; the feed-forward layer is always tanh (:329-332)."""
    nbr_encoder = nbr_encoder or ("attention" if attention else "mean_embed")
    if nbr_encoder not in NBR_ENCODERS:
        raise NotImplementedError(nbr_encoder)                                  # :292-293
    if nonlinearity not in NONLINEARITIES:
        raise NotImplementedError(f"nonlinearity {nonlinearity!r}")
    import torch
    from torch import nn
    nh = hidden if nbr_hidden is None else nbr_hidden
    oh = hidden if obst_hidden is None else obst_hidden
    act = {"tanh": nn.Tanh, "elu": lambda: nn.ELU(inplace=True), "relu": lambda: nn.ReLU(inplace=True)}[nonlinearity]

    class QuadMultiEncoderRef(nn.Module):
        def __init__(self):
            super().__init__()
            self.self_dim, self.nbr_dim, self.num_nbr, self.obst_dim = self_dim, nbr_dim, num_nbr, obst_dim
            self.hidden, self.nbr_hidden, self.obst_hidden, self.nonlinearity = hidden, nh, oh, nonlinearity
            mlp = lambda i, h=hidden: nn.Sequential(nn.Linear(i, h), act(), nn.Linear(h, h), act())
            # Parameters are created in the reference's order (neighbour encoder :279-293, self encoder :300-309, obstacle encoder
            # :313-322, feed forward :329-332), so that the same torch seed gives the same weights as the reference class - that is
            # what lets tests/golden/encoder_*.npz hold a seed instead of megabytes of weights.
            self.nbr_encoder = nbr_encoder if num_nbr > 0 else "no_encoder"
            self.attention = self.nbr_encoder == "attention"
            if self.nbr_encoder == "mlp":                                       # :110-117
                self.neighbor_encoder = nn.Sequential(nn.Linear(nbr_dim * num_nbr, nh), act(), nn.Linear(nh, nh), act(),
                                                      nn.Linear(nh, nh), act())
            elif self.nbr_encoder == "no_encoder":                              # :289-291 "blind agent"
                self.neighbor_encoder = None
            else:                                                               # :29-34 / :52-57
                self.neighbor_encoder = mlp(self_dim + nbr_dim if self.attention else nbr_dim, nh)
            if self.attention:
                self.neighbor_value_mlp = mlp(nh, nh)                           # :60-65
                self.attention_mlp = nn.Sequential(nn.Linear(2 * nh, nh), act(), nn.Linear(nh, nh), act(),
                                                   nn.Linear(nh, 1))            # :68-75
            self.self_encoder = mlp(self_dim)                                   # :303-309
            self.obstacle_encoder = mlp(obst_dim, oh) if obst_dim > 0 else None # :315-322
            total = hidden + (nh if self.neighbor_encoder is not None else 0) + (oh if obst_dim > 0 else 0)   # :324
            self.feed_forward = nn.Sequential(nn.Linear(total, 2 * hidden), nn.Tanh())   # :329-332

        def forward(self, obs):
            B = obs.shape[0]
            emb = [self.self_encoder(obs[:, :self.self_dim])]
            nb = self.nbr_dim * self.num_nbr
            if self.attention:                                                  # :77-101, including what .repeat() pairs up
                K = self.num_nbr
                nbrs = obs[:, self.self_dim:self.self_dim + nb].reshape(-1, self.nbr_dim)
                e = self.neighbor_encoder(torch.cat((obs[:, :self.self_dim].repeat(K, 1), nbrs), dim=1))
                h = self.neighbor_value_mlp(e)
                e_mean = e.reshape(B, -1, e.shape[-1]).mean(dim=1)
                alpha = self.attention_mlp(torch.cat((e, e_mean.repeat(K, 1)), dim=1)).view(B, -1)
                w = torch.softmax(alpha, dim=1).view(-1, 1)
                emb.append((w * h).view(B, -1, h.shape[-1]).sum(dim=1))
            elif self.nbr_encoder == "mlp":                                     # :119-122
                emb.append(self.neighbor_encoder(obs[:, self.self_dim:self.self_dim + nb]))
            elif self.neighbor_encoder is not None:                             # :36-43
                e = self.neighbor_encoder(obs[:, self.self_dim:self.self_dim + nb].reshape(-1, self.nbr_dim))
                emb.append(e.reshape(B, -1, e.shape[-1]).mean(dim=1))
            if self.obstacle_encoder is not None:
                emb.append(self.obstacle_encoder(obs[:, self.self_dim + nb:]))
            return self.feed_forward(torch.cat(emb, dim=1))                     # :334-350

    if seed is not None:   # None: the caller's generator state (Sample Factory seeds once, globally)
        torch.manual_seed(seed)
    return QuadMultiEncoderRef()


def make_reference_mha_encoder(self_dim=19, nbr_dim=6, num_nbr=2, obst_dim=9, hidden=HIDDEN, seed=0, n_head=4):
    """QuadMultiHeadAttentionEncoder (quad_multi_model.py:124-196, --quads_encoder_type=attention) with the MultiHeadAttention block
    of swarm_rl/models/attention_layer.py:12-56 as a plain torch module; random init."""
    import torch
    from torch import nn

    class MultiHeadAttentionRef(nn.Module):                                     # attention_layer.py:12-56
        def __init__(self):
            super().__init__()
            self.w_qs = nn.Linear(hidden, n_head * hidden, bias=False)
            self.w_ks = nn.Linear(hidden, n_head * hidden, bias=False)
            self.w_vs = nn.Linear(hidden, n_head * hidden, bias=False)
            self.fc = nn.Linear(n_head * hidden, hidden, bias=False)
            self.layer_norm = nn.LayerNorm(hidden, eps=1e-6)

        def forward(self, x):                                                   # q = k = v = x: [B, L, hidden]
            Bq, L = x.shape[0], x.shape[1]
            q = self.w_qs(x).view(Bq, L, n_head, hidden).transpose(1, 2)
            k = self.w_ks(x).view(Bq, L, n_head, hidden).transpose(1, 2)
            v = self.w_vs(x).view(Bq, L, n_head, hidden).transpose(1, 2)
            attn = torch.softmax(torch.matmul(q / hidden ** 0.5, k.transpose(2, 3)), dim=-1)   # :118-125
            o = torch.matmul(attn, v).transpose(1, 2).contiguous().view(Bq, L, -1)
            return self.layer_norm(self.fc(o) + x)

    class QuadMultiHeadAttentionEncoderRef(nn.Module):
        def __init__(self):
            super().__init__()
            self.self_dim, self.nbr_dim, self.num_nbr, self.obst_dim = self_dim, nbr_dim, num_nbr, obst_dim
            self.nbr_encoder = "multi_head_attention"
            mlp = lambda i: nn.Sequential(nn.Linear(i, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh())
            self.self_encoder = mlp(self_dim)                                   # self_embed_layer      :148-153
            self.neighbor_encoder = mlp(nbr_dim * num_nbr)                      # neighbor_embed_layer  :154-159
            self.obstacle_encoder = mlp(obst_dim)                               # obstacle_embed_layer  :161-166
            self.attention_layer = MultiHeadAttentionRef()                      # :169
            self.feed_forward = nn.Sequential(nn.Linear(3 * hidden, 2 * hidden), nn.Tanh())   # :173-174

        def forward(self, obs):                                                 # :176-196
            nb = self.nbr_dim * self.num_nbr
            s = self.self_encoder(obs[:, :self.self_dim])
            n = self.neighbor_encoder(obs[:, self.self_dim:self.self_dim + nb])
            o = self.obstacle_encoder(obs[:, self.self_dim + nb:])
            tokens = self.attention_layer(torch.stack((n, o), dim=1))
            return self.feed_forward(torch.cat((s, tokens.reshape(obs.shape[0], -1)), dim=1))

    if seed is None:   # the Sim2Real subclass: construct (and drop) the parent's layers on the current generator state
        return QuadMultiHeadAttentionEncoderRef()
    torch.manual_seed(seed)
    return QuadMultiHeadAttentionEncoderRef()


def make_reference_sim2real_encoder(self_dim=19, nbr_dim=6, num_nbr=2, obst_dim=9, hidden=HIDDEN, seed=0):
    """QuadSingleHeadAttentionEncoder_Sim2Real (quad_multi_model.py:199-248, --quads_encoder_type=attention --quads_sim2real=True):
    one-layer embeddings, OneHeadAttention (attention_layer.py:56-97) over the token pair [neighbour embedding, obstacle
    embedding], feed forward 3*hidden -> hidden.  Random init in the reference's order: the class first builds everything its
    parent (QuadMultiHeadAttentionEncoder) builds - consuming the generator - and then replaces the layers."""
    import torch
    from torch import nn

    class OneHeadAttentionRef(nn.Module):                                       # attention_layer.py:56-97
        def __init__(self):
            super().__init__()
            self.w_qs = nn.Linear(hidden, hidden, bias=False)
            self.w_ks = nn.Linear(hidden, hidden, bias=False)
            self.w_vs = nn.Linear(hidden, hidden, bias=False)
            self.fc = nn.Linear(hidden, hidden, bias=False)
            self.layer_norm = nn.LayerNorm(hidden, eps=1e-6)

        def forward(self, x):                                                   # q = k = v = x: [B, L, hidden]
            q, k, v = self.w_qs(x), self.w_ks(x), self.w_vs(x)
            attn = torch.softmax(torch.matmul(q / hidden ** 0.5, k.transpose(-1, -2)), dim=-1)   # :83-85
            return self.layer_norm(self.fc(torch.matmul(attn, v)) + x)

    class QuadSingleHeadAttentionEncoderSim2RealRef(nn.Module):
        def __init__(self):
            super().__init__()
            self.self_dim, self.nbr_dim, self.num_nbr, self.obst_dim = self_dim, nbr_dim, num_nbr, obst_dim
            self.nbr_encoder = "single_head_sim2real"
            make_reference_mha_encoder(self_dim, nbr_dim, num_nbr, obst_dim, hidden, seed=None)   # the parent's draws (:201)
            emb = lambda i: nn.Sequential(nn.Linear(i, hidden), nn.Tanh())
            self.self_encoder = emb(self_dim)                                   # self_embed_layer      :222-225
            self.neighbor_encoder = emb(nbr_dim * num_nbr)                      # neighbor_embed_layer  :226-229
            self.obstacle_encoder = emb(obst_dim)                               # obstacle_embed_layer  :231-234
            self.attention_layer = OneHeadAttentionRef()                        # :237
            self.feed_forward = nn.Sequential(nn.Linear(3 * hidden, hidden), nn.Tanh())   # :240-242

        def forward(self, obs):                                                 # the parent's forward (:176-196)
            nb = self.nbr_dim * self.num_nbr
            s = self.self_encoder(obs[:, :self.self_dim])
            n = self.neighbor_encoder(obs[:, self.self_dim:self.self_dim + nb])
            o = self.obstacle_encoder(obs[:, self.self_dim + nb:])
            tokens = self.attention_layer(torch.stack((n, o), dim=1))
            return self.feed_forward(torch.cat((s, tokens.reshape(obs.shape[0], -1)), dim=1))

    if seed is not None:
        torch.manual_seed(seed)
    return QuadSingleHeadAttentionEncoderSim2RealRef()


OBS_REPR_DIMS = {"xyz_vxyz_R_omega": 18, "xyz_vxyz_R_omega_floor": 19, "xyz_vxyz_R_omega_wall": 24}   # quad_utils.py:30-34


def encoder_from_cfg(cfg, seed=None):
    """The module swarm_rl/models/quad_multi_model.py:355-370 `make_quadmulti_encoder(cfg, obs_space)` builds, from the same flags:
    --quads_encoder_type / --quads_sim2real pick the class, --rnn_size / --quads_neighbor_hidden_size / --quads_obst_hidden_size the
    widths, --nonlinearity the activation of QuadMultiEncoder's MLPs.  Combinations the reference itself cannot run raise here
    instead of silently building something else.  seed=None: the current torch generator state (what SF does)."""
    import torch
    self_dim = OBS_REPR_DIMS[cfg.quads_obs_repr]
    if cfg.quads_neighbor_obs_type == "none":
        num_nbr = 0
    else:
        num_nbr = cfg.quads_num_agents - 1 if cfg.quads_neighbor_visible_num == -1 else cfg.quads_neighbor_visible_num
    hidden = cfg.rnn_size
    if cfg.quads_encoder_type == "attention":                                   # :356-362
        # QuadMultiHeadAttentionEncoder (:124-196) always embeds 6*K neighbour columns and 9 obstacle columns with rnn_size-wide tanh
        # layers, whatever the other flags say
        if not cfg.quads_use_obstacles or num_nbr == 0:
            raise NotImplementedError("--quads_encoder_type=attention needs neighbour and obstacle observations: the reference's "
                                      "QuadMultiHeadAttentionEncoder slices both out of the observation row (quad_multi_model.py:176-196)")
        make = make_reference_sim2real_encoder if getattr(cfg, "quads_sim2real", False) else make_reference_mha_encoder
        return make(self_dim=self_dim, num_nbr=num_nbr, obst_dim=9, hidden=hidden, seed=seed)
    return make_reference_encoder(seed=seed, self_dim=self_dim, num_nbr=num_nbr, obst_dim=9 if cfg.quads_use_obstacles else 0, hidden=hidden,
                                  nbr_encoder=cfg.quads_neighbor_encoder_type, nbr_hidden=getattr(cfg, "quads_neighbor_hidden_size", hidden),
                                  obst_hidden=getattr(cfg, "quads_obst_hidden_size", hidden), nonlinearity=getattr(cfg, "nonlinearity", "tanh"))


# reference / Sample Factory parameter names -> the names of the restatements above (the encoder sits under "encoder." in an SF
# actor-critic checkpoint; any such prefix is stripped)
_KEYMAP_MULTI = (("neighbor_encoder.embedding_mlp.", "neighbor_encoder."), ("neighbor_encoder.neighbor_mlp.", "neighbor_encoder."),
                 ("neighbor_encoder.neighbor_value_mlp.", "neighbor_value_mlp."), ("neighbor_encoder.attention_mlp.", "attention_mlp."))
_KEYMAP_MHA = (("self_embed_layer.", "self_encoder."), ("neighbor_embed_layer.", "neighbor_encoder."), ("obstacle_embed_layer.", "obstacle_encoder."))


def _strip_prefix(sd, probe):
    keys = [k for k in sd if k.endswith(probe)]
    if not keys:
        raise KeyError(f"state dict has no '{probe}': not a QuadMultiEncoder / QuadMultiHeadAttentionEncoder checkpoint")
    prefix = keys[0][:-len(probe)]
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def encoder_from_state_dict(sd, num_nbr, nbr_dim=6):
    """A restatement module (make_reference_encoder / make_reference_mha_encoder) carrying the weights of a reference checkpoint:
    `sd` is the state_dict of the reference's QuadMultiEncoder (swarm_rl/models/quad_multi_model.py:250-350) or
    QuadMultiHeadAttentionEncoder (:124-196), or of a whole Sample Factory actor-critic (keys under "...encoder.").  Shapes (self /
    obstacle observation widths, hidden size) and the neighbour encoder type are read off the tensors; `num_nbr` (neighbours per
    row) is not recoverable from mean_embed / attention weights and must be given.  FusedQuadEncoder(encoder_from_state_dict(...))
    runs a trained reference policy's encoder on the fused kernel."""
    import torch
    mha = any(k.endswith("self_embed_layer.0.weight") for k in sd)
    sd = _strip_prefix(sd, "self_embed_layer.0.weight" if mha else "self_encoder.0.weight")
    out = {}
    for k, v in sd.items():
        for a, b in (_KEYMAP_MHA if mha else _KEYMAP_MULTI):
            if k.startswith(a):
                k = b + k[len(a):]
                break
        out[k] = v
    if mha:
        hidden, self_dim = out["self_encoder.0.weight"].shape
        obst_dim = out["obstacle_encoder.0.weight"].shape[1]
        make = make_reference_mha_encoder if "self_encoder.2.weight" in out else make_reference_sim2real_encoder   # one-layer embeddings: Sim2Real
        module = make(self_dim=self_dim, nbr_dim=nbr_dim, num_nbr=out["neighbor_encoder.0.weight"].shape[1] // nbr_dim, obst_dim=obst_dim, hidden=hidden)
        out = {k: v for k, v in out.items() if not k.startswith("attention_layer.attention")}
    else:
        hidden, self_dim = out["self_encoder.0.weight"].shape
        obst_dim = out["obstacle_encoder.0.weight"].shape[1] if "obstacle_encoder.0.weight" in out else 0
        if "attention_mlp.0.weight" in out:
            kind = "attention"
        elif "neighbor_encoder.4.weight" in out:
            kind, num_nbr = "mlp", out["neighbor_encoder.0.weight"].shape[1] // nbr_dim
        elif "neighbor_encoder.0.weight" in out:
            kind = "mean_embed"
        else:
            kind = "no_encoder"
        module = make_reference_encoder(self_dim=self_dim, nbr_dim=nbr_dim, num_nbr=num_nbr, obst_dim=obst_dim, hidden=hidden, nbr_encoder=kind)
    module.load_state_dict({k: torch.as_tensor(v) for k, v in out.items()}, strict=True)
    return module


SPLIT_SCALE = 2048.0          # qs_policy_encoder.hip ENC_SPLIT_SCALE
FP16_MIN_NORMAL = 2.0 ** -14


def split_fp16(x):
    """float32 array -> (h, l) float16 with x = h + l / 2048 to ~2^-22 relative (the operand format of the reference-precision kernels,
    qs_policy_encoder.hip split2): h = fp16(x), 0 below the smallest normal; l = fp16((x - h) * 2048)."""
    x = np.clip(np.asarray(x, dtype=np.float32), -65504.0, 65504.0)
    h = np.where(np.abs(x) < FP16_MIN_NORMAL, np.float32(0), x).astype(np.float16)
    l = ((x - h.astype(np.float32)) * np.float32(SPLIT_SCALE)).astype(np.float16)
    return h, l


def pack_linear(linear, device, cols=None, split=False):
    """nn.Linear (or its input columns cols[0]:cols[1]) -> (packed bf16 weights in MFMA A-fragment order, padded fp32 bias, M, K);
    split: every fragment as the two fp16 planes of split_fp16, [M/16, K/32, 2, 64, 8].  See include/quadswarm_encoder.h."""
    import torch
    W = linear.weight.detach().float().cpu().numpy()
    if cols is not None:
        W = W[:, cols[0]:cols[1]]
    b = linear.bias.detach().float().cpu().numpy() if linear.bias is not None else np.zeros(W.shape[0], dtype=np.float32)
    m_real, k_real = W.shape
    M, K = -(-m_real // 16) * 16, -(-k_real // 32) * 32
    Wp = np.zeros((M, K), dtype=np.float32)
    Wp[:m_real, :k_real] = W
    bp = np.zeros(M, dtype=np.float32)
    bp[:m_real] = b
    lane = np.arange(64)
    rows = (lane & 15)[None, None, :, None] + 16 * np.arange(M // 16)[:, None, None, None]
    cols = (8 * (lane >> 4))[None, None, :, None] + np.arange(8)[None, None, None, :] + 32 * np.arange(K // 32)[None, :, None, None]
    packed = Wp[rows, cols]                                                     # [M/16, K/32, 64, 8]
    if split:
        w = torch.from_numpy(np.ascontiguousarray(np.stack(split_fp16(packed), axis=2))).to(device).contiguous()
    else:
        w = torch.from_numpy(np.ascontiguousarray(packed)).to(device).to(torch.bfloat16).contiguous()
    return w, torch.from_numpy(bp).to(device), M, K


class FusedQuadEncoder:
    """forward(obs[B, D] float32 on the GPU) -> [B, 512] float32 ([B, 256] for the Sim2Real encoder), one kernel launch.
    precision "bf16": bf16 operands, fp32 accumulation (features within ~1e-2 of the fp32 module).  precision "fp32": reference precision for a
    sampler whose learner is fp32 - every operand as a pair of fp16 numbers, three MFMAs per product (qs_enc_params.precision = 1; features
    within 1e-5 of the fp32 module; QuadMultiEncoder's four neighbour encoders, 16-agent kernels)."""

    def __init__(self, module, device=0, precision="bf16"):
        import torch
        self._torch = torch
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision: 'bf16' or 'fp32'")
        self.precision = precision
        split = precision == "fp32"
        self.device = torch.device("cuda", device)
        if not torch.cuda.is_available():
            raise native.QsError("FusedQuadEncoder needs a GPU: there is no CPU fallback")
        self._keep = []
        # the kernels are built for the published architecture: every hidden width 256, tanh (HIDDEN); a module with other widths or
        # another --nonlinearity runs as the torch module it is (sf_models.QuadEncoder), not here
        widths = {getattr(module, a, HIDDEN) for a in ("hidden", "nbr_hidden", "obst_hidden")} | {module.self_encoder[0].out_features}
        if widths != {HIDDEN} or getattr(module, "nonlinearity", "tanh") != "tanh":
            raise NotImplementedError(f"the fused encoder kernels cover rnn_size = neighbor / obstacle hidden size = {HIDDEN} with tanh; got widths "
                                      f"{sorted(widths)}, nonlinearity {getattr(module, 'nonlinearity', 'tanh')!r}")
        P = EncParams()
        P.self_dim, P.nbr_dim, P.num_nbr, P.obst_dim = module.self_dim, module.nbr_dim, module.num_nbr, module.obst_dim
        P.obs_dim = module.self_dim + module.nbr_dim * module.num_nbr + module.obst_dim

        self._packed, self._raw = [], []   # what refresh() re-reads: (nn.Linear, cols, packed weights, bias) / (source getter, device tensor)

        def layer(linear, cols=None):
            w, b, M, K = pack_linear(linear, self.device, cols, split)
            self._keep += [w, b]
            self._packed.append((linear, cols, w, b))
            return EncLayer(w.data_ptr(), b.data_ptr(), M, K)

        P.nbr_encoder = MODELS.index(getattr(module, "nbr_encoder", "mean_embed"))
        P.precision = int(split)
        self._split = split
        s2r = P.nbr_encoder == 5   # one-layer embeddings, one head, 256 outputs
        P.s1 = layer(module.self_encoder[0])
        if module.neighbor_encoder is not None:
            P.n1 = layer(module.neighbor_encoder[0])
        if module.obstacle_encoder is not None:
            P.o1 = layer(module.obstacle_encoder[0])
        if not s2r:
            P.s2 = layer(module.self_encoder[2])
            if module.neighbor_encoder is not None:
                P.n2 = layer(module.neighbor_encoder[2])
            if module.obstacle_encoder is not None:
                P.o2 = layer(module.obstacle_encoder[2])
        if P.nbr_encoder in (4, 5):
            if module.nbr_dim * module.num_nbr > 64:
                raise ValueError("multi-head attention encoder: num_nbr * nbr_dim must fit two 32-wide K steps")
            att = module.attention_layer
            P.mq, P.mk, P.mv, P.mfc = layer(att.w_qs), layer(att.w_ks), layer(att.w_vs), layer(att.fc)
            ln = [att.layer_norm.weight.detach().float().to(self.device).contiguous(), att.layer_norm.bias.detach().float().to(self.device).contiguous()]
            self._keep += ln
            self._raw += [(lambda: att.layer_norm.weight, ln[0]), (lambda: att.layer_norm.bias, ln[1])]
            P.ln_w, P.ln_b = ln[0].data_ptr(), ln[1].data_ptr()
        self.attention = P.nbr_encoder == 1
        if P.nbr_encoder == 2:
            if module.nbr_dim * module.num_nbr > 64:
                raise ValueError("mlp neighbour encoder: num_nbr * nbr_dim must fit two 32-wide K steps")
            P.n3 = layer(module.neighbor_encoder[4])
        if self.attention:
            if module.self_dim + module.nbr_dim > 32:
                raise ValueError("attention encoder: self_dim + nbr_dim must fit one 32-wide K step")
            P.v1, P.v2 = layer(module.neighbor_value_mlp[0]), layer(module.neighbor_value_mlp[2])
            P.a1e, P.a1m = layer(module.attention_mlp[0], (0, HIDDEN)), layer(module.attention_mlp[0], (HIDDEN, 2 * HIDDEN))
            P.a2 = layer(module.attention_mlp[2])
            a3 = module.attention_mlp[4]   # 256 -> 1: reduced from the second score layer's accumulators, fp32 weights
            a3w = a3.weight.detach().float().reshape(-1).to(self.device).contiguous()
            self._keep.append(a3w)
            self._raw.append((lambda: a3.weight.reshape(-1), a3w))
            self._a3 = a3
            P.a3w, P.a3b = a3w.data_ptr(), float(a3.bias.detach().float().item())
        self._scratch_rows = 0
        P.f = layer(module.feed_forward[0])
        self.out_dim = HIDDEN if s2r else 2 * HIDDEN
        if P.f.M != self.out_dim or P.s1.M != HIDDEN:
            raise ValueError("the fused encoder is built for hidden size 256")
        self.params = P
        lib()

    def refresh(self):
        """Re-read the module's weights (an optimiser moved them) INTO the buffers the kernels - and any captured HIP graph - already point at.
        One attention-score bias lives in the parameter struct (a3b): a graph that was captured before keeps the old value of that one scalar
        until it is recaptured; everything else is picked up by replays."""
        for linear, cols, w, b in self._packed:
            nw, nb, _, _ = pack_linear(linear, self.device, cols, self._split)
            w.copy_(nw)
            b.copy_(nb)
        for src, dst in self._raw:
            dst.copy_(src().detach().float().reshape(dst.shape))
        if getattr(self, "_a3", None) is not None:
            self.params.a3b = float(self._a3.bias.detach().float().item())
        if getattr(self, "_head_src", None) is not None:
            self._head[0].copy_(self._head_src[0].detach().float())
            self._head[1].copy_(self._head_src[1].detach().float())

    def forward(self, obs, out=None):
        torch = self._torch
        assert obs.is_cuda and obs.dtype == torch.float32 and obs.is_contiguous() and obs.shape[1] == self.params.obs_dim
        B = obs.shape[0]
        if out is None:
            out = torch.empty((B, self.out_dim), device=obs.device, dtype=torch.float32)
        self._scratch(B)
        self.params.head_dim = 0
        rc = lib().qs_enc_forward(obs.data_ptr(), B, C.byref(self.params), out.data_ptr(), C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream))
        if rc != 0:
            raise native.QsError(f"qs_enc_forward failed ({rc}): {lib().qs_enc_last_error().decode()}")
        return out

    def set_head(self, weight, bias):
        """Linear head on the 512 features (SF's action-parameter / value layer), evaluated in the encoder's epilogue by forward_head."""
        torch = self._torch
        w = weight.detach().to(self.device, torch.float32).clone().contiguous()   # (own copies: refresh() writes into them)
        b = bias.detach().to(self.device, torch.float32).clone().contiguous()
        if w.dim() != 2 or w.shape[1] != self.out_dim or not 1 <= w.shape[0] <= 8 or b.shape != (w.shape[0],):
            raise ValueError(f"head: weight [h, {self.out_dim}] with 1 <= h <= 8, bias [h]")
        self._head, self._head_src = (w, b), (weight, bias)

    def forward_head(self, obs, head_out=None, features=None, sample=None, traj=None):
        """head(encoder(obs)) -> [B, h] float32; the [B, 512] features are written only if a `features` tensor is passed.
        sample = (log_std [h], act_out [B, h], counter (int32 device tensor), step, seed): the epilogue also writes
        act_out = head + exp(log_std) * N(0, 1), Philox keyed (seed, counter + step, agent) - qs_enc_params.sample_*.
        traj = (rew_src [B] float32, rew_dst [B], done_src [B] uint8, done_dst [B]): the first kernel of the pass also copies the reward / done
        flags of the step that produced `obs` into a trajectory slot (qs_enc_params.traj_*)."""
        torch = self._torch
        assert obs.is_cuda and obs.dtype == torch.float32 and obs.is_contiguous() and obs.shape[1] == self.params.obs_dim
        if getattr(self, "_head", None) is None:
            raise native.QsError("forward_head: call set_head(weight, bias) first")
        w, b = self._head
        B = obs.shape[0]
        if head_out is None:
            head_out = torch.empty((B, w.shape[0]), device=obs.device, dtype=torch.float32)
        assert head_out.is_contiguous() and head_out.shape == (B, w.shape[0]) and head_out.dtype == torch.float32
        self._scratch(B)
        P = self.params
        P.head_w, P.head_b, P.head_out, P.head_dim = w.data_ptr(), b.data_ptr(), head_out.data_ptr(), w.shape[0]
        if sample is not None:
            log_std, act_out, counter, step, seed = sample
            assert act_out.is_contiguous() and act_out.shape == head_out.shape and act_out.dtype == torch.float32 and log_std.numel() == w.shape[0]
            P.sample_log_std, P.act_out, P.sample_counter = log_std.data_ptr(), act_out.data_ptr(), counter.data_ptr()
            P.sample_step, P.sample_seed_lo, P.sample_seed_hi = int(step) & 0xffffffff, int(seed) & 0xffffffff, (int(seed) >> 32) & 0xffffffff
        if traj is not None:
            rs, rd, ds, dd = traj
            assert rs.dtype == torch.float32 and rd.dtype == torch.float32 and ds.dtype == torch.uint8 and dd.dtype == torch.uint8
            assert all(x.is_contiguous() and x.numel() == B for x in traj)
            P.traj_rew_src, P.traj_rew_dst, P.traj_done_src, P.traj_done_dst = rs.data_ptr(), rd.data_ptr(), ds.data_ptr(), dd.data_ptr()
        rc = lib().qs_enc_forward(obs.data_ptr(), B, C.byref(P), C.c_void_p(features.data_ptr() if features is not None else None),
                                  C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream))
        P.head_dim = 0
        P.sample_log_std = None
        P.traj_rew_dst = None
        if rc != 0:
            raise native.QsError(f"qs_enc_forward failed ({rc}): {lib().qs_enc_last_error().decode()}")
        return head_out

    __call__ = forward

    def _scratch(self, B):
        """attention only: e_i [B*K, 256] bf16 (reference precision: the two fp16 planes, [B*K, 2, 256]) and W_m e_mean [B, 256] fp32, handed
        from the first launch to the second"""
        if self.attention and B > self._scratch_rows:
            torch = self._torch
            self._ebuf = torch.empty((B * self.params.num_nbr, HIDDEN * (2 if self._split else 1)), device=self.device, dtype=torch.bfloat16)
            self._gbuf = torch.empty((B, HIDDEN), device=self.device, dtype=torch.float32)
            self.params.ebuf, self.params.gbuf = self._ebuf.data_ptr(), self._gbuf.data_ptr()
            self._scratch_rows = B

    def benchmark(self, obs, out, iters=200):
        """Average seconds per forward pass over `iters` back-to-back launches (HIP events, no host work in between)."""
        torch = self._torch
        ms = C.c_double(0)
        self.params.head_dim = 0
        self._scratch(obs.shape[0])
        rc = lib().qs_enc_benchmark(obs.data_ptr(), obs.shape[0], C.byref(self.params), out.data_ptr(),
                                    C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream), iters, C.byref(ms))
        if rc != 0:
            raise native.QsError(f"qs_enc_benchmark failed ({rc}): {lib().qs_enc_last_error().decode()}")
        return ms.value * 1e-3
