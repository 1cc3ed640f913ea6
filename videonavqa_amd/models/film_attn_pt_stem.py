"""FiLMAttnPretrainedStem — drop-in for models/film_attn_pt_stem.py of the reference."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from .. import ops
from .common import (FiLMTrunkBase, NativeFeatures, compute_dtype, grad_scale_of, repeated_question_lstm)

NEG_MASK = float(-(1 << 31))  # film_attn_pt_stem.py:251


class FiLMAttnPretrainedStem(FiLMTrunkBase):
    """Same positional signature and defaults as the reference (film_attn_pt_stem.py:14-25).
    Extra keyword-only options: spatial_size (the reference hard-codes 130 = 10x13, :56) and
    precision ('bf16' MFMA fast path | 'fp32' exact-f32 MFMA path)."""

    def __init__(self, batch_size, q_embedding_size, nb_classes, num_input_channels=512,
                 num_res_block_channels=512, num_res_blocks=1, hidden_size=128, at_hidden_size=128,
                 max_num_frames=35, q_encoder='lstm', vocab_size=134, *, spatial_size=130,
                 precision='fp16h'):
        super(FiLMAttnPretrainedStem, self).__init__()
        assert q_encoder.lower() in ['lstm', 'bow'], "Invalid question encoder! ('lstm', 'bow')"
        self.q_encoder = q_encoder
        self.nb_classes = nb_classes
        self.batch_size = batch_size
        self.q_embedding_size = q_embedding_size
        self.at_hidden_size = at_hidden_size
        self.hidden_size = hidden_size
        self.spatial_size = spatial_size
        self.compute_dtype = compute_dtype(precision)
        self.hyb = precision == "fp16h"      # fp16 storage; split features, conv_init as three products, 1x1 / fc with split weights (common.compute_dtype)

        self.embed = nn.Embedding(vocab_size, q_embedding_size)                       # :37
        self._build_trunk_head(num_input_channels, num_res_block_channels)             # :39-44
        total_out_size = 2 * num_res_block_channels * num_res_blocks
        encoder = nn.LSTM(q_embedding_size, hidden_size) if q_encoder == 'lstm' else \
            nn.Linear(q_embedding_size, hidden_size)                                   # :75-77 (compared as given, like upstream)
        self.film_layer = nn.ModuleList([encoder,                                      # :51,75-87 (GPU flavour)
                                         nn.Linear(hidden_size, total_out_size),
                                         nn.ReLU(inplace=True)])
        self._build_film_pipeline(num_res_block_channels, num_res_blocks)              # :52,93-108
        self.fc_embed_attn = nn.Linear(spatial_size * num_res_block_channels, at_hidden_size)  # :56-57
        self.fc_attn_1 = nn.Linear(at_hidden_size, 1)                                 # :58
        self.fc_hidden_attn = nn.Linear(at_hidden_size, 1)                            # :60
        self.lstm_attn = nn.LSTMCell(at_hidden_size, at_hidden_size)                  # :62
        self.out_linear = nn.Linear(max_num_frames * at_hidden_size, nb_classes)      # :65
        for module in self.modules():                                                 # :67-68
            self.weights_init(module)
        self.film_hidden = None
        self.init_hidden()

    def init_hidden(self):
        """film_attn_pt_stem.py:133-138 (the bag-of-words encoder carries no state)."""
        if self.q_encoder == 'lstm':
            self.film_hidden = self._zero_hidden(self.batch_size, self.hidden_size, self.embed.weight.device)

    def forward(self, v_input, q_input, v_lens, q_lens):
        """v_input: fp32 [B, C_in, h, w, T] (reference layout) or NativeFeatures from the stem;
        v_lens sorted descending.  Returns fp32 logits [batch_size, nb_classes] (:188-301)."""
        x, lay, h, w = self._prepare_input(v_input, v_lens)
        assert lay.B == self.batch_size, "B must equal batch_size (film_attn_pt_stem.py:168)"
        assert h * w == self.spatial_size, "spatial size %dx%d != spatial_size=%d" % (h, w, self.spatial_size)
        dev = x.device
        B, T = lay.B, lay.T
        C = self.num_res_block_channels

        fused = self._use_fused_trunk()
        # fp16 storage: the tail's backward emits d f times 2^10, FcNativeFn / FilmTrunkFn divide their fp32 results by it
        gscale = grad_scale_of(self.compute_dtype) if fused else 1.0
        self._trunk_grad_scale = gscale
        bow = not isinstance(self.film_layer[0], nn.LSTM)                    # :158
        if fused:       # train mode: generator and conv trunk on fused HIP ops (one autograd node for the trunk)
            # the FiLM generator (question LSTM re-run per frame, :213) on the side stream, joined after conv_init + BatchNorm
            film_img, join = self._fork_generator(
                lambda: self.bow_film_values(self.film_layer[0], self.film_layer[1], q_input, lay) if bow else
                self.question_film_values(self.film_layer[0], self.film_layer[1], q_input, q_lens, lay), n_img=lay.n_img)
            x = self._trunk_fused(x, lay, [(film_img, 2 * C * k) for k in range(self.num_res_blocks)], join)   # :229-233
        else:
            # FiLM generator: question LSTM re-run per processed frame with carried state (:213) — on the side stream
            def generator():
                if bow:
                    return self.bow_film_values(self.film_layer[0], self.film_layer[1], q_input, lay)
                emb = self.embed(q_input)
                h0, c0 = self._question_state(B, self.hidden_size, q_lens, dev)
                h_last, _, (hn, cn) = repeated_question_lstm(self.film_layer[0], emb, q_lens, lay.n_frames, h0, c0,
                                                              wgrad_dtype=self._lstm_wgrad_dtype())
                self._store_question_state(hn, cn, q_lens)
                film = F.relu(self.film_layer[1](h_last))                   # [B, n_frames, 2*C*blocks] (:179)
                return film[lay.sample_of, lay.frame_of]                    # [n_img, 2*C*blocks]

            film_img, join = self._fork_generator(generator, n_img=lay.n_img)
            x = self._trunk_head(x, lay)
            join()

            def film_fn(k):
                s = 2 * C * k
                return film_img[:, s:s + C], film_img[:, s + C:s + 2 * C]   # :229-233

            x = self._trunk_blocks(x, lay, film_fn)

        # fc_embed_attn over the flattened map (:244) as one split-K GEMM for all images
        n_img, hp, wp, c_pad = x.shape
        at = self.at_hidden_size
        at_pad = L.round_up(at, 64)
        f = ops.fc_native(x.view(n_img, -1), self.fc_embed_attn.weight, self.fc_embed_attn.bias, C, h, w, at_pad, gscale,
                          split_weights=self.hyb)
        if fused:
            # temporal attention (:245-290) straight from the packed GEMM output: the zero-padded [B,T,at] tensor, the
            # validity grid and the -(1<<31) masks are formed inside the kernel.  v_i = fc_hidden_attn(h) is constant along
            # the frame axis and softmax is shift invariant, so coefs/ctxt are identical at every step of the
            # :283 loop and are computed once (fc_hidden_attn consequently receives a zero gradient).
            ctxt, coefs = ops.temporal_attention_packed(f, lay.frame_off_i32, lay.n_frames, B, T, at,
                                                        self.fc_attn_1.weight, self.fc_attn_1.bias, gscale)
            # the 35-step LSTMCell chain on a constant input = the persistent LSTM kernel with one "token" repeated T
            # times (:283,293-298); input projection and classifier (:301) on the fp32 HIP GEMM
            gi = ops.linear(ctxt, self.lstm_attn.weight_ih, self.lstm_attn.bias_ih + self.lstm_attn.bias_hh)
            z = self._zero_hidden(B, at, dev)[0][0] if at == self.hidden_size else torch.zeros(B, at, device=dev)
            ones = self.__dict__.get("_ones_i32")
            if ones is None or ones.numel() != B or ones.device != dev:
                ones = self.__dict__["_ones_i32"] = torch.ones(B, dtype=torch.int32, device=dev)
            hs, _, _ = ops.lstm_seq(gi.unsqueeze(1), self.lstm_attn.weight_hh, z, z, ones, T, T, self._lstm_wgrad_dtype())
            return ops.linear(hs.view(B, T * at), self.out_linear.weight, self.out_linear.bias)      # :301
        f = f[:, :at].float()
        all_features = torch.zeros(B, T, at, device=dev).index_put((lay.sample_of, lay.frame_of), f)  # :245-256
        valid = torch.zeros(B, T, 1, device=dev).index_put(
            (lay.sample_of, lay.frame_of), torch.ones(n_img, 1, device=dev))
        # masks: -(1<<31) where a frame was processed without this sample (:251); frames past the
        # longest video keep mask 0 and feature 0 (un-masked on purpose)
        processed = torch.zeros(1, T, 1, device=dev)
        processed[:, :lay.n_frames] = 1
        masks = (processed - valid) * NEG_MASK
        # temporal attention (:268-290) as ONE fused HIP kernel: fc_attn_1 scores on the valid entries,
        # masked softmax over frames, weighted sum.
        ctxt, coefs = ops.temporal_attention(all_features, valid.squeeze(2), masks.squeeze(2),
                                             self.fc_attn_1.weight, self.fc_attn_1.bias)
        gi = F.linear(ctxt, self.lstm_attn.weight_ih, self.lstm_attn.bias_ih + self.lstm_attn.bias_hh)
        zeros = torch.zeros(B, at, device=dev)
        ones = torch.ones(B, dtype=torch.int32, device=dev)
        hs, _, _ = ops.lstm_seq(gi.unsqueeze(1), self.lstm_attn.weight_hh, zeros, zeros, ones, T, T, self.compute_dtype)
        hs = hs.reshape(B, T * at)
        return self.out_linear(hs)                                      # :301
