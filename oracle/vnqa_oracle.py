"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

A plain PyTorch-CPU / fp32 restatement of the reference's video-question
fusion path (catalina17/VideoNavQA).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this module; the product (`videonavqa_amd/`) never does.

Parity status
-------------
* Pinned against golden vectors captured from the reference itself by
  `tools/capture_goldens.py` (see tests/golden/*.npz, tests/test_oracle_golden.py):
  ObjDetectCNN, FiLMAttnPretrainedStem, FiLMGlobalPoolingPretrainedStem,
  TimeMultiHopFiLMPretrainedStem, QOnlyLSTM, VideoOnlyCNN3D, MACNetwork, and the loss/clip/Adam step.
* `vgg_front` (VGG-16 features[0:10]) restates a THIRD-PARTY dependency that
  is not vendored in the reference (`demo.get_frcnn_feature_extractor`,
  catalina17/faster-rcnn.pytorch, no pinned version; call sites
  eval/q_and_v_eval.py:17,106,308): **parity unpinned** for that function.
  Its definition here is the shape-derived one: the standard
  VGG-16 "D" front: conv3x3(3,64) ReLU conv3x3(64,64) ReLU pool2
  conv3x3(64,128) ReLU conv3x3(128,128) ReLU pool2.

All functions are functional: weights come in a dict keyed by the
reference's GPU-flavour state_dict names, plus `conv1x1_layers.<k>.*` for the
unregistered frozen 1x1 convs (film_attn_pt_stem.py:44,101-104).
Every function cites the reference file:line it follows.
"""
import math

import torch
import torch.nn.functional as F

NEG_MASK = float(-(1 << 31))  # film_attn_pt_stem.py:251
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------
# Stem
# --------------------------------------------------------------------------
VGG_FRONT_CFG = [(3, 64), (64, 64), "M", (64, 128), (128, 128), "M"]


def vgg_front(x, W):
    """VGG-16 features[0:10] on [N,3,H,W] -> [N,128,H/4,W/4].  parity unpinned (see header).
    Weights: `features.{0,2,5,7}.{weight,bias}` (torchvision VGG-16 'D' numbering)."""
    idx = 0
    for item in VGG_FRONT_CFG:
        if item == "M":
            x = F.max_pool2d(x, 2, 2)
            idx += 1
        else:
            x = F.relu(F.conv2d(x, W["features.%d.weight" % idx], W["features.%d.bias" % idx], padding=1))
            idx += 2
    return x


def _bn_eval(x, W, name):
    return F.batch_norm(x, W[name + ".running_mean"], W[name + ".running_var"],
                        W[name + ".weight"], W[name + ".bias"], False, 0.0, BN_EPS)


def obj_detect_cnn(x, W):
    """ObjDetectCNN.forward with pretrained_features=True, eval mode
    (models/obj_detector.py:69-86; eval() at eval/utils.py:50)."""
    x = _bn_eval(x, W, "bn_input")                                        # :70
    h = F.conv2d(F.conv2d(x, W["conv11.weight"], W["conv11.bias"], padding=1),
                 W["conv12.weight"], W["conv12.bias"], padding=1)          # :72 (no ReLU between)
    h = F.max_pool2d(F.relu(_bn_eval(h, W, "bn1")), 2, 2)                  # :73-75
    h = F.conv2d(F.conv2d(h, W["conv21.weight"], W["conv21.bias"], padding=1),
                 W["conv22.weight"], W["conv22.bias"], padding=1)          # :77
    h = F.max_pool2d(F.relu(_bn_eval(h, W, "bn2")), 2, 2)                  # :78-80
    h = F.conv2d(F.conv2d(h, W["conv31.weight"], W["conv31.bias"], padding=1),
                 W["conv32.weight"], W["conv32.bias"], padding=1)          # :82
    return F.relu(_bn_eval(h, W, "bn3"))                                   # :83-86


def stem_forward(v_inputs, W_vgg, W_od, frames=None):
    """Per-frame frozen stem loop (eval/q_and_v_eval.py:102-110):
    [B,3,H,W,T] -> [B,512,h,w,T].  `frames` restricts the loop (bench sampling)."""
    T = v_inputs.shape[-1]
    feats = []
    with torch.no_grad():
        for j in (range(T) if frames is None else frames):
            f = vgg_front(v_inputs[:, :, :, :, j], W_vgg)                  # :106
            feats.append(obj_detect_cnn(f, W_od))                          # :108
    return torch.stack(feats).permute(1, 2, 3, 4, 0)                       # :110


def sort_batch(v_inputs, q_inputs, v_lens, q_lens, ys):
    """Sort a minibatch by descending video length (eval/q_and_v_eval.py:113-116)."""
    v_lens_s, perm = v_lens.sort(0, descending=True)
    return v_inputs[perm], q_inputs[perm], v_lens_s, q_lens[perm], ys[perm], perm


# --------------------------------------------------------------------------
# Question LSTM (packed, carried state)
# --------------------------------------------------------------------------
def lstm_cell(x_gates, h, c, w_hh, b_hh):
    """One LSTM step given the precomputed input projection; gate order i,f,g,o."""
    g = x_gates + h @ w_hh.t() + b_hh
    H = h.shape[1]
    i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return h2, c2


def lstm_packed(x_emb, q_lens, w_ih, w_hh, b_ih, b_hh, h0, c0):
    """Semantics of sort -> pack_padded_sequence -> nn.LSTM(packed, hidden) -> pad -> unsort
    (film_attn_pt_stem.py:150-166): every sample b advances exactly q_lens[b] steps from its
    own carried (h0[b], c0[b]); outputs past a sample's length are zero.  The reference keeps
    the hidden state in length-sorted order; since the same q_lens (hence the same sort) is
    used at every call inside a forward, that is per-sample carry.
    x_emb [B,L,E]; returns out [B,Lmax,H] (Lmax=max(q_lens)), (hN, cN) [B,H]."""
    B, L, _ = x_emb.shape
    Lmax = int(q_lens.max())
    xg = x_emb @ w_ih.t() + b_ih
    h, c = h0, c0
    outs = []
    for t in range(Lmax):
        h2, c2 = lstm_cell(xg[:, t], h, c, w_hh, b_hh)
        m = (q_lens > t).to(h.dtype).unsqueeze(1)
        h = m * h2 + (1 - m) * h
        c = m * c2 + (1 - m) * c
        outs.append(m * h2)
    return torch.stack(outs, 1), (h, c)


def gather_last(out, q_lens):
    """lstm_op_out.gather(1, q_len-1) (film_attn_pt_stem.py:168-171)."""
    B, _, H = out.shape
    idx = (q_lens.view(B, 1, 1) - 1).expand(B, 1, H)
    return out.gather(1, idx).view(B, H)


# --------------------------------------------------------------------------
# Shared FiLM trunk pieces
# --------------------------------------------------------------------------
def ct_batch_sizes(v_lens, num_frames):
    """Effective batch size per frame (film_attn_pt_stem.py:201-208): number of videos with
    v_len >= i+1, given v_lens sorted descending; frames after the longest video are skipped."""
    out = []
    for i in range(num_frames):
        ct = int((v_lens >= (i + 1)).sum())
        if ct == 0:
            break
        out.append(ct)
    return out


def bn_train_frame(x, gamma, beta, state):
    """Train-mode BatchNorm2d on one frame's [ct_B,C,h,w] (film_attn_pt_stem.py:211), batch
    statistics over ct_B*h*w, running stats updated every frame (momentum 0.1, unbiased var)."""
    if state.get("training", True):
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        n = x.numel() // x.shape[1]
        with torch.no_grad():
            state["running_mean"] = (1 - BN_MOMENTUM) * state["running_mean"] + BN_MOMENTUM * mean
            state["running_var"] = (1 - BN_MOMENTUM) * state["running_var"] + \
                BN_MOMENTUM * var * (n / max(n - 1, 1))
            state["num_batches_tracked"] = state["num_batches_tracked"] + 1
    else:
        mean, var = state["running_mean"], state["running_var"]
    xh = (x - mean.view(1, -1, 1, 1)) * torch.rsqrt(var.view(1, -1, 1, 1) + BN_EPS)
    return xh * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)


def film_res_block(x, film_values, start_idx, W, k):
    """One FiLM residual block (film_attn_pt_stem.py:217-241)."""
    res_x = F.relu(F.conv2d(x, W["conv1x1_layers.%d.weight" % k], W["conv1x1_layers.%d.bias" % k]))
    y = F.conv2d(res_x, W["film_pipeline.%d.weight" % k], W["film_pipeline.%d.bias" % k], padding=1)
    C = y.shape[1]
    alphas = film_values[:, start_idx:start_idx + C].view(-1, C, 1, 1)
    betas = film_values[:, start_idx + C:start_idx + 2 * C].view(-1, C, 1, 1)
    return F.relu(alphas * y + betas) + res_x, start_idx + 2 * C


def _num_blocks(W):
    k = 0
    while "film_pipeline.%d.weight" % k in W:
        k += 1
    return k


def _bn_state(W, training):
    return {"running_mean": W["bn_init.running_mean"].clone(),
            "running_var": W["bn_init.running_var"].clone(),
            "num_batches_tracked": W["bn_init.num_batches_tracked"].clone(),
            "training": training}


def _film_values_lstm(W, q_input, q_lens, hidden, ct_B):
    """compute_film_values, LSTM encoder (film_attn_pt_stem.py:144-181)."""
    x = F.embedding(q_input, W["embed.weight"],
                    padding_idx=W.get("_embed_padding_idx", None))
    out, hidden = lstm_packed(x, q_lens, W["film_layer.0.weight_ih_l0"], W["film_layer.0.weight_hh_l0"],
                              W["film_layer.0.bias_ih_l0"], W["film_layer.0.bias_hh_l0"], *hidden)
    last = gather_last(out, q_lens)[:ct_B]
    return F.relu(last @ W["film_layer.1.weight"].t() + W["film_layer.1.bias"]), hidden


def _film_values_bow(W, q_input, ct_B):
    """compute_film_values, bag-of-words encoder (film_attn_pt_stem.py:145,171-181; the pooling model's is the same,
    film_global_pooling_pt_stem.py:138,164-174): Linear over every token position of the UNSORTED padded question, summed over
    positions (padding tokens included); the division by the question length at :175-176 discards its result, so the
    sum stands.  No carried state."""
    x = F.embedding(q_input, W["embed.weight"], padding_idx=W.get("_embed_padding_idx", None))
    x = (x @ W["film_layer.0.weight"].t() + W["film_layer.0.bias"]).sum(dim=1)[:ct_B]
    return F.relu(x @ W["film_layer.1.weight"].t() + W["film_layer.1.bias"])


def _is_bow(W):
    return "film_layer.0.weight" in W


def _film_values(W, q_input, q_lens, hidden, ct_B):
    if _is_bow(W):
        return _film_values_bow(W, q_input, ct_B), hidden
    return _film_values_lstm(W, q_input, q_lens, hidden, ct_B)


# --------------------------------------------------------------------------
# FiLMAttnPretrainedStem.forward
# --------------------------------------------------------------------------
def film_attn_forward(W, v_input, q_input, v_lens, q_lens, training=True, aux=None):
    """FiLMAttnPretrainedStem.forward after init_hidden() (film_attn_pt_stem.py:133-138,188-301).
    v_input [B,C_in,h,w,T]; v_lens sorted descending.  Returns logits [B,nb_classes].
    `aux` (dict) receives BN running stats, the carried LSTM state and intermediates."""
    B, T = v_input.shape[0], v_input.shape[-1]
    Hq = W["film_layer.1.weight"].shape[1]
    at = W["fc_attn_1.weight"].shape[1]
    nblocks = _num_blocks(W)
    hidden = (v_input.new_zeros(B, Hq), v_input.new_zeros(B, Hq))           # init_hidden :133-138
    bn = _bn_state(W, training)
    cts = ct_batch_sizes(v_lens, T)
    masks = v_input.new_zeros(B, T, 1)                                      # :194
    all_features = []
    film_per_frame = []
    for i, ct in enumerate(cts):                                            # :201
        x = v_input[:ct, :, :, :, i]                                        # :210
        x = F.relu(F.conv2d(x, W["conv_init.weight"], W["conv_init.bias"], padding=1))
        x = bn_train_frame(x, W["bn_init.weight"], W["bn_init.bias"], bn)   # :211
        film_values, hidden = _film_values(W, q_input, q_lens, hidden, ct)  # :213
        film_per_frame.append(film_values)
        s = 0
        for k in range(nblocks):                                            # :217-241
            x, s = film_res_block(x, film_values, s, W, k)
        f = x.reshape(ct, -1) @ W["fc_embed_attn.weight"].t() + W["fc_embed_attn.bias"]  # :244
        all_features.append(F.pad(f, (0, 0, 0, B - ct)))                    # :245-247
        masks[ct:, i, 0] = NEG_MASK                                         # :251
    all_features = torch.stack(all_features, 0).permute(1, 0, 2)            # :253-254
    all_features = F.pad(all_features, (0, 0, 0, T - all_features.shape[1]))  # :255-256
    # fc_attn_1 on the valid (frame, sample) entries only; zeros elsewhere (:268-281)
    valid = v_input.new_zeros(B, T, 1)
    for i, ct in enumerate(cts):
        valid[:ct, i, 0] = 1.0
    features = (all_features @ W["fc_attn_1.weight"].t() + W["fc_attn_1.bias"]) * valid
    h = v_input.new_zeros(B, at)
    cell = v_input.new_zeros(B, at)
    hs = []
    coefs = None
    for i in range(T):                                                      # :283
        v_i = (h @ W["fc_hidden_attn.weight"].t() + W["fc_hidden_attn.bias"]).view(B, 1, 1)  # :285
        coefs = torch.softmax(v_i + features + masks, dim=1)                # :288
        ctxt = torch.bmm(coefs.permute(0, 2, 1), all_features).view(B, -1)  # :290
        g = ctxt @ W["lstm_attn.weight_ih"].t() + W["lstm_attn.bias_ih"]
        h, cell = lstm_cell(g, h, cell, W["lstm_attn.weight_hh"], W["lstm_attn.bias_hh"])  # :293
        hs.append(h)
    hs = torch.stack(hs, 1).reshape(B, -1)                                  # :294-298
    logits = hs @ W["out_linear.weight"].t() + W["out_linear.bias"]         # :301
    if aux is not None:
        aux.update(bn=bn, hidden=hidden, all_features=all_features, coefs=coefs,
                   film_values=film_per_frame, cts=cts)
    return logits


# --------------------------------------------------------------------------
# FiLMGlobalPoolingPretrainedStem.forward
# --------------------------------------------------------------------------
def _gp_tail(W, feats_per_frame, B):
    """Global temporal max pooling + classifier (film_global_pooling_pt_stem.py:228-238)."""
    padded = []
    for x in feats_per_frame:
        ct = x.shape[0]
        t = F.relu(F.conv2d(x, W["c1x1_tail.weight"], W["c1x1_tail.bias"]))  # :228
        padded.append(F.pad(t.reshape(ct, -1), (0, 0, 0, B - ct)))         # :230-232
    pooled = torch.stack(padded, 0).max(dim=0)[0]                          # :235
    return pooled @ W["out_linear.weight"].t() + W["out_linear.bias"]      # :238


def film_gp_forward(W, v_input, q_input, v_lens, q_lens, training=True, aux=None):
    """FiLMGlobalPoolingPretrainedStem.forward (film_global_pooling_pt_stem.py:180-238);
    embedding has padding_idx=0 (:34) which only matters for the gradient."""
    B, T = v_input.shape[0], v_input.shape[-1]
    Hq = W["film_layer.1.weight"].shape[1]
    nblocks = _num_blocks(W)
    W = dict(W)
    W["_embed_padding_idx"] = 0
    hidden = (v_input.new_zeros(B, Hq), v_input.new_zeros(B, Hq))
    bn = _bn_state(W, training)
    feats = []
    for i, ct in enumerate(ct_batch_sizes(v_lens, T)):
        x = v_input[:ct, :, :, :, i]
        x = F.relu(F.conv2d(x, W["conv_init.weight"], W["conv_init.bias"], padding=1))
        x = bn_train_frame(x, W["bn_init.weight"], W["bn_init.bias"], bn)   # :196
        film_values, hidden = _film_values(W, q_input, q_lens, hidden, ct)  # :198
        s = 0
        for k in range(nblocks):
            x, s = film_res_block(x, film_values, s, W, k)
        feats.append(x)
    if aux is not None:
        aux.update(bn=bn, hidden=hidden)
    return _gp_tail(W, feats, B)


# --------------------------------------------------------------------------
# TimeMultiHopFiLMPretrainedStem.forward
# --------------------------------------------------------------------------
def tmh_forward(W, v_input, q_input, v_lens, q_lens, training=True, aux=None):
    """TimeMultiHopFiLMPretrainedStem.forward (time_multi_hop_pt_stem.py:124-184,191-250)."""
    B, T = v_input.shape[0], v_input.shape[-1]
    Hq = W["q_encoder.weight_hh_l0"].shape[1]
    nblocks = _num_blocks(W)
    hidden = (v_input.new_zeros(B, Hq), v_input.new_zeros(B, Hq))           # init_hidden :111-116
    bn = _bn_state(W, training)
    feats = []
    emb = F.embedding(q_input, W["embed.weight"], padding_idx=0)            # :30,:126
    for i, ct in enumerate(ct_batch_sizes(v_lens, T)):
        x = v_input[:ct, :, :, :, i]
        x = F.relu(F.conv2d(x, W["conv_init.weight"], W["conv_init.bias"], padding=1))
        x = bn_train_frame(x, W["bn_init.weight"], W["bn_init.bias"], bn)   # :207
        # compute_film_encoding (:124-158): carried LSTM state, context reset every frame
        rnn_states, hidden = lstm_packed(emb, q_lens, W["q_encoder.weight_ih_l0"],
                                         W["q_encoder.weight_hh_l0"], W["q_encoder.bias_ih_l0"],
                                         W["q_encoder.bias_hh_l0"], *hidden)      # :135
        enc = gather_last(rnn_states, q_lens)[:ct]                          # :143-147
        enc = F.layer_norm(enc, (Hq,), W["encoder_norm.weight"], W["encoder_norm.bias"])  # :148
        nw = rnn_states.shape[1]
        hctx = enc.view(ct, 1, Hq).repeat(1, nw, 1)                         # :157-158
        states = rnn_states[:ct]                                            # :167
        s = 0
        for k in range(nblocks):
            # decode_to_film_values (:165-184): unmasked softmax over words
            prod = hctx * states                                            # :170
            coefs = torch.softmax(prod @ W["fc_hidden_attn.weight"].t() + W["fc_hidden_attn.bias"],
                                  dim=1).view(ct, 1, nw)                    # :171-172
            hv = torch.bmm(coefs, prod).view(ct, Hq)                        # :175-176
            cond = hv @ W["fc_attn_out.weight"].t() + W["fc_attn_out.bias"]  # :179
            hctx = hv.view(ct, 1, Hq).repeat(1, nw, 1)                      # :180-181
            film_values = F.layer_norm(cond, (cond.shape[1],), W["decoder_norm.weight"],
                                       W["decoder_norm.bias"])              # :184
            x, s = film_res_block(x, film_values, s, W, k)                  # :215-238
        feats.append(x)
    if aux is not None:
        aux.update(bn=bn, hidden=hidden)
    return _gp_tail(W, feats, B)                                            # :240-250


# --------------------------------------------------------------------------
# QOnlyLSTM.forward (config 1 plumbing)
# --------------------------------------------------------------------------
def q_only_lstm_forward(W, q_input, q_lens, h0, c0):
    """QOnlyLSTM.forward (models/q_only_lstm.py:57-69); caller pre-sorts by q_len."""
    x = F.embedding(q_input, W["embed.weight"], padding_idx=0)
    out, hidden = lstm_packed(x, q_lens, W["lstm.weight_ih_l0"], W["lstm.weight_hh_l0"],
                              W["lstm.bias_ih_l0"], W["lstm.bias_hh_l0"], h0, c0)
    last = gather_last(out, q_lens)
    return last @ W["out_linear.weight"].t() + W["out_linear.bias"], hidden


# --------------------------------------------------------------------------
# VideoOnlyCNN3D.forward (config 2: 3-D conv bring-up)
# --------------------------------------------------------------------------
def _bn(x, W, name, training):
    return F.batch_norm(x, W[name + ".running_mean"].clone(), W[name + ".running_var"].clone(),
                        W[name + ".weight"], W[name + ".bias"], training, BN_MOMENTUM, BN_EPS)


def video_only_cnn3d_features(W, x, training=False):
    """Conv trunk of VideoOnlyCNN3D.forward (models/v_only_cnn3d.py:59-72): x [B,3,D,H,W]."""
    h = _bn(x, W, "bn_input", training)                                                    # :60
    h = F.max_pool3d(F.relu(F.conv3d(h, W["conv1.weight"], W["conv1.bias"], padding=1)), (1, 2, 2), (1, 2, 2))
    h = _bn(h, W, "bn1", training)                                                         # :62-64
    h = F.max_pool3d(F.relu(F.conv3d(h, W["conv2.weight"], W["conv2.bias"], padding=1)), 4, 4)
    h = _bn(h, W, "bn2", training)                                                         # :66-68
    h = F.max_pool3d(F.relu(F.conv3d(h, W["conv3a.weight"], W["conv3a.bias"], padding=1)), 4, 4)
    return _bn(h, W, "bn3", training)                                                      # :70-72


def video_only_cnn3d_forward(W, x, training=False):
    """VideoOnlyCNN3D.forward (models/v_only_cnn3d.py:59-81)."""
    h = video_only_cnn3d_features(W, x, training)
    h = h.reshape(h.shape[0], -1)                                                          # :74
    h = _bn(F.relu(h @ W["fc6.weight"].t() + W["fc6.bias"]), W, "bn6", training)           # :76-77
    h = _bn(F.relu(h @ W["fc7.weight"].t() + W["fc7.bias"]), W, "bn7", training)           # :78-79
    return h @ W["fc8.weight"].t() + W["fc8.bias"]                                         # :81


# --------------------------------------------------------------------------
# MACNetwork.forward (models/mac.py) — the fourth stem-consuming model of eval/q_and_v_eval.py
# --------------------------------------------------------------------------
def _lin(W, name, x):
    y = x @ W[name + ".weight"].t()
    return y + W[name + ".bias"] if (name + ".bias") in W else y


def _lstm_dir(xg, lens, w_hh, b_hh, reverse):
    """One direction of a packed nn.LSTM from a zero state.  xg [B,Lmax,4H] (input projection incl. b_ih).
    Sample b walks t = 0..len-1 (or len-1..0 when reverse); outputs past its length are zero.
    Returns out [B,Lmax,H] and the state after each sample's last processed step."""
    B, Lmax, H4 = xg.shape
    H = H4 // 4
    h = xg.new_zeros(B, H)
    c = xg.new_zeros(B, H)
    outs = [None] * Lmax
    order = range(Lmax - 1, -1, -1) if reverse else range(Lmax)
    for t in order:
        h2, c2 = lstm_cell(xg[:, t], h, c, w_hh, b_hh)
        m = (lens > t).to(h.dtype).unsqueeze(1)
        h = m * h2 + (1 - m) * h
        c = m * c2 + (1 - m) * c
        outs[t] = m * h2
    return torch.stack(outs, 1), h


def mac_question(W, question, q_lens):
    """Question side of MACNetwork.forward (models/mac.py:203-221): embed (padding_idx 0) -> sort by
    length -> packed bidirectional LSTM -> pad -> UNSORT lstm_out only -> lstm_proj.
    `h` (final hidden of both directions) is returned in the SORTED order, exactly as upstream leaves it
    (:221 never applies invperm_idx to h) — row s of h belongs to the sample at sorted position s."""
    emb = F.embedding(question, W["embed.weight"], padding_idx=0)                       # :206
    lens_sorted, perm = q_lens.sort(0, descending=True)                                 # :208
    emb = emb[perm]
    Lmax = int(lens_sorted[0])
    outs, hs = [], []
    for sfx, rev in (("", False), ("_reverse", True)):
        xg = emb[:, :Lmax] @ W["lstm.weight_ih_l0" + sfx].t() + W["lstm.bias_ih_l0" + sfx]
        o, hN = _lstm_dir(xg, lens_sorted, W["lstm.weight_hh_l0" + sfx], W["lstm.bias_hh_l0" + sfx], rev)
        outs.append(o)
        hs.append(hN)
    lstm_out = torch.cat(outs, 2)                                                       # [B,Lmax,2*dim], zero past len
    inv = perm.sort(0)[1]                                                               # :217
    lstm_out = lstm_out[inv]
    context = _lin(W, "lstm_proj", lstm_out)                                            # :220 (pads become the bias)
    h = torch.cat(hs, 1)                                                                # :221, sorted order
    return context, h


def mac_unit(W, context, question, know, max_step, self_attention, memory_gate, masks=None):
    """MACUnit.forward (models/mac.py:131-155) with its Control/Read/Write units (:15-107).
    context [b,L,dim], question [b,2dim], know [b,dim,S].  masks = (control_mask, memory_mask) reproduces the
    train-mode variational dropout (:137-141,146-147,152-153); None = no dropout."""
    b = question.shape[0]
    dim = W["mac.mem_0"].shape[1]
    control = W["mac.control_0"].expand(b, dim)
    memory = W["mac.mem_0"].expand(b, dim)
    if masks is not None:
        control = control * masks[0]
        memory = memory * masks[1]
    controls, memories = [control], [memory]
    for i in range(max_step):
        # ControlUnit (:28-42)
        pa = _lin(W, "mac.control.position_aware.%d" % i, question)
        cq = _lin(W, "mac.control.control_question", torch.cat([control, pa], 1)).unsqueeze(1)
        aw = _lin(W, "mac.control.attn", cq * context)                                  # [b,L,1]
        control = (F.softmax(aw, 1) * context).sum(1)
        if masks is not None:
            control = control * masks[0]
        controls.append(control)
        # ReadUnit (:53-62)
        mem = _lin(W, "mac.read.mem", memories[-1]).unsqueeze(2)                        # [b,dim,1]
        cc = _lin(W, "mac.read.concat", torch.cat([mem * know, know], 1).permute(0, 2, 1))   # [b,S,dim]
        ra = _lin(W, "mac.read.attn", cc * controls[-1].unsqueeze(1)).squeeze(2)        # [b,S]
        ra = F.softmax(ra, 1).unsqueeze(1)
        read = (ra * know).sum(2)                                                       # [b,dim]
        # WriteUnit (:82-105)
        prev = memories[-1]
        concat = _lin(W, "mac.write.concat", torch.cat([read, prev], 1))
        nxt = concat
        if self_attention:
            ccat = torch.stack(controls[:-1], 2)                                        # [b,dim,i+1]
            sa = controls[-1].unsqueeze(2) * ccat
            sa = _lin(W, "mac.write.attn", sa.permute(0, 2, 1))                         # [b,i+1,1]
            sa = F.softmax(sa, 1).permute(0, 2, 1)
            mcat = torch.stack(memories, 2)
            nxt = _lin(W, "mac.write.mem", (sa * mcat).sum(2)) + concat
        if memory_gate:
            gate = torch.sigmoid(_lin(W, "mac.write.control", controls[-1]))
            nxt = gate * prev + (1 - gate) * nxt
        memory = nxt
        if masks is not None:
            memory = memory * masks[1]
        memories.append(memory)
    return memory


def mac_forward(W, images, question, v_lens, q_lens, max_step, max_num_frames=35, self_attention=False,
                memory_gate=False, masks=None):
    """MACNetwork.forward (models/mac.py:199-257).  images [B,512,h,w,T]; v_lens sorted descending.
    masks: optional list, one (control_mask, memory_mask) pair per processed frame."""
    B = images.shape[0]
    dim = W["mac.mem_0"].shape[1]
    context, h = mac_question(W, question[:B], q_lens)
    cts = ct_batch_sizes(v_lens, int(v_lens[0]))                                        # :226-233
    outs = []
    for i, ct in enumerate(cts):
        img = images[:ct, :, :, :, i]
        for k in (0, 2, 4):                                                             # :177-182 conv+ELU x3
            img = F.elu(F.conv2d(img, W["conv.%d.weight" % k], W["conv.%d.bias" % k], padding=1))
        know = img.reshape(ct, dim, -1)
        mem = mac_unit(W, context[:ct], h[:ct], know, max_step, self_attention, memory_gate,
                       None if masks is None else masks[i])
        out = torch.cat([mem, h[:ct]], 1)                                               # :240
        outs.append(F.pad(out, (0, 0, 0, B - ct)).unsqueeze(1))                         # :242-243
    outs = torch.cat(outs, 1)                                                           # [B,n_frames,3dim]
    # packed lstm_tail from a zero state, output at each sample's last frame (:249-255); the zero padding
    # up to max_num_frames (:246-247) is never read by a packed sequence
    xg = outs @ W["lstm_tail.weight_ih_l0"].t() + W["lstm_tail.bias_ih_l0"]
    tail, _ = _lstm_dir(xg, v_lens, W["lstm_tail.weight_hh_l0"], W["lstm_tail.bias_hh_l0"], False)
    last = gather_last(tail, v_lens)
    y = F.elu(_lin(W, "classifier.0", last))                                            # :187-189
    return _lin(W, "classifier.2", y)


def mac_train_step(W, images, question, v_lens, q_lens, ys, adam, lr, max_step, clip=1.0, grad_clamp=1.0, **kw):
    """One optimisation step of `--model mac`: per-parameter gradient clamp to [-1,1] by tensor hooks
    (eval/q_and_v_eval.py:348-351), then clip_grad_norm and Adam (:137-138).  Dropout masks via kw['masks']."""
    names = [k for k in W if W[k].is_floating_point()]
    for k in names:
        W[k] = W[k].detach().requires_grad_(True)
    logits = mac_forward(W, images, question, v_lens, q_lens, max_step, **kw)
    loss = cross_entropy_sum(logits, ys)
    gl = torch.autograd.grad(loss, [W[k] for k in names], allow_unused=True)
    grads = {k: (None if g is None else g.clamp(-grad_clamp, grad_clamp)) for k, g in zip(names, gl)}
    for k in names:
        W[k] = W[k].detach()
    clip_and_adam(W, grads, adam, lr, clip)
    return float(loss.detach()), logits.detach(), grads


FORWARDS = {"film_attn_pt": film_attn_forward, "film_gp_pt": film_gp_forward,
            "time_multi_hop": tmh_forward}


# --------------------------------------------------------------------------
# Loss / clip / Adam step (eval/q_and_v_eval.py:124-139, :321, :333)
# --------------------------------------------------------------------------
def cross_entropy_sum(logits, ys):
    """nn.CrossEntropyLoss(reduction='sum') (eval/q_and_v_eval.py:321, eval.sh:16)."""
    lse = torch.logsumexp(logits, dim=1)
    return (lse - logits.gather(1, ys.view(-1, 1)).squeeze(1)).sum()


def is_trainable(name, frozen_prefixes=("conv1x1_layers.",)):
    """Registered parameters only: conv1x1_layers are a plain list (§0.5), BN buffers excluded.
    On a CUDA-less box the reference ALSO leaves film_layer unregistered
    (film_attn_pt_stem.py:84-86): pass frozen_prefixes=("conv1x1_layers.", "film_layer.") to
    restate that flavour (the goldens were captured on such a box)."""
    return not (name.startswith(tuple(frozen_prefixes)) or name.startswith("_") or
                name.endswith("running_mean") or name.endswith("running_var") or
                name.endswith("num_batches_tracked"))


class AdamState(object):
    """torch.optim.Adam defaults (betas .9/.999, eps 1e-8, no weight decay)."""
    def __init__(self, names):
        self.m = {k: None for k in names}
        self.v = {k: None for k in names}
        self.t = 0


def clip_and_adam(W, grads, state, lr, clip=1.0):
    """clip_grad_norm(params, clip) then Adam.step() (eval/q_and_v_eval.py:137-138)."""
    names = [k for k in grads if grads[k] is not None]
    total = math.sqrt(sum(float((grads[k].double() ** 2).sum()) for k in names))
    coef = min(1.0, clip / (total + 1e-6))
    state.t += 1
    b1, b2, eps = 0.9, 0.999, 1e-8
    for k in names:
        g = grads[k] * coef
        if state.m[k] is None:
            state.m[k] = torch.zeros_like(g)
            state.v[k] = torch.zeros_like(g)
        state.m[k] = b1 * state.m[k] + (1 - b1) * g
        state.v[k] = b2 * state.v[k] + (1 - b2) * g * g
        bc1 = 1 - b1 ** state.t
        bc2 = 1 - b2 ** state.t
        denom = state.v[k].sqrt() / math.sqrt(bc2) + eps
        W[k] = (W[k] - (lr / bc1) * state.m[k] / denom).detach()
    return total


def train_step(model, W, v_input, q_input, v_lens, q_lens, ys, adam, lr, clip=1.0,
               frozen_prefixes=("conv1x1_layers.",)):
    """One optimisation step on pre-extracted, pre-sorted features; returns (loss, logits, grads).
    Mutates W (weights and BN running stats) in place."""
    names = [k for k in W if is_trainable(k, frozen_prefixes) and W[k].is_floating_point()]
    for k in names:
        W[k] = W[k].detach().requires_grad_(True)
    aux = {}
    logits = FORWARDS[model](W, v_input, q_input, v_lens, q_lens, training=True, aux=aux)
    loss = cross_entropy_sum(logits, ys)
    gl = torch.autograd.grad(loss, [W[k] for k in names], allow_unused=True)
    grads = {k: g for k, g in zip(names, gl)}
    for k in names:
        W[k] = W[k].detach()
    clip_and_adam(W, grads, adam, lr, clip)
    for key in ("running_mean", "running_var", "num_batches_tracked"):
        W["bn_init." + key] = aux["bn"][key]
    return float(loss.detach()), logits.detach(), grads
