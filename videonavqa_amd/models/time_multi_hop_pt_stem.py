"""TimeMultiHopFiLMPretrainedStem — drop-in for models/time_multi_hop_pt_stem.py."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from .. import ops
from .common import FiLMTrunkBase, compute_dtype, grad_scale_of


class TimeMultiHopFiLMPretrainedStem(FiLMTrunkBase):
    """Signature/defaults: time_multi_hop_pt_stem.py:13-22; extra keyword-only spatial_size, precision."""

    def __init__(self, batch_size, q_embedding_size, nb_classes, num_input_channels=512,
                 num_res_block_channels=512, num_res_blocks=1, num_tail_channels=32, hidden_size=128,
                 vocab_size=134, *, spatial_size=130, precision='fp16h'):
        super(TimeMultiHopFiLMPretrainedStem, self).__init__()
        self.nb_classes = nb_classes
        self.batch_size = batch_size
        self.q_embedding_size = q_embedding_size
        self.hidden_size = hidden_size
        self.spatial_size = spatial_size
        self.compute_dtype = compute_dtype(precision)
        self.hyb = precision == "fp16h"      # fp16 storage; split features, conv_init as three products, 1x1 / fc with split weights (common.compute_dtype)

        self.embed = nn.Embedding(vocab_size, q_embedding_size, padding_idx=0)        # :30
        self._build_trunk_head(num_input_channels, num_res_block_channels)             # :32-36
        total_out_size = 2 * num_res_block_channels * num_res_blocks
        self.q_encoder = nn.LSTM(q_embedding_size, hidden_size)                       # :45
        self.encoder_norm = nn.LayerNorm(hidden_size)                                 # :46
        self.h = None
        self.fc_hidden_attn = nn.Linear(hidden_size, 1)                               # :49
        self.fc_attn_out = nn.Linear(hidden_size, total_out_size)                     # :50
        self.decoder_norm = nn.LayerNorm(total_out_size)                              # :51
        self._build_film_pipeline(num_res_block_channels, num_res_blocks)              # :53
        self.c1x1_tail = nn.Conv2d(num_res_block_channels, num_tail_channels, kernel_size=1)  # :56
        self.out_linear = nn.Linear(spatial_size * num_tail_channels, nb_classes)     # :60
        for module in self.modules():                                                 # :62-65
            self.weights_init(module)
        for module in self.conv1x1_layers:
            self.weights_init(module)
        self.film_hidden = None
        self.init_hidden()

    def init_hidden(self):
        """time_multi_hop_pt_stem.py:111-116."""
        self.film_hidden = self._zero_hidden(self.batch_size, self.hidden_size, self.embed.weight.device)
        self.h = None

    def _generator_hip(self, q_input, q_lens, lay):
        """compute_film_encoding + decode_to_film_values (:124-184) for all frames at once on HIP kernels: embedding + input
        projection (one launch), the question LSTM re-run per frame with carried state as ONE persistent chain, then the
        multi-hop attention read straight from the chain's output rows (ops.MultiHopGenFn)."""
        B, H, dev = q_input.shape[0], self.hidden_size, q_input.device
        ql_cpu = q_lens.detach().cpu().long()
        Lmax = int(ql_cpu.max())
        S = Lmax * lay.n_frames
        ql = [int(v) for v in ql_cpu]
        pairs = [(t, b) for t, ct in enumerate(lay.cts) for b in range(ct)]
        base = np.asarray([b * S + t * ql[b] for t, b in pairs], np.int32)
        qimg = np.asarray([ql[b] for _, b in pairs], np.int32)
        tbl = np.concatenate([ql_cpu.numpy().astype(np.int32), lay.last_token_rows(ql_cpu, S), base, qimg])
        tbl = L.to_device_async(torch.from_numpy(tbl), dev)          # ONE pinned upload for all tables
        n = lay.n_img
        ql_i32, last_rows, base_row, qlen = tbl[:B], tbl[B:B + n], tbl[B + n:B + 2 * n], tbl[B + 2 * n:]
        lstm = self.q_encoder
        xg = ops.embed_proj(q_input, self.embed.weight, lstm.weight_ih_l0, lstm.bias_ih_l0, lstm.bias_hh_l0, 0)
        h0, c0 = self._question_state(B, H, q_lens, dev)
        hs, hn, cn = ops.lstm_seq(xg, lstm.weight_hh_l0, h0, c0, ql_i32, lay.n_frames, S, self._lstm_wgrad_dtype())
        self._store_question_state(hn, cn, q_lens)
        return list(ops.multi_hop_generator(hs.view(B * S, H), (last_rows, base_row, qlen), Lmax, self.num_res_blocks,
                                            self.encoder_norm, self.fc_hidden_attn, self.fc_attn_out, self.decoder_norm))

    def forward(self, v_input, q_input, v_lens, q_lens):
        """time_multi_hop_pt_stem.py:191-250 with compute_film_encoding (:124-158) and
        decode_to_film_values (:165-184) evaluated for all frames at once."""
        x, lay, h, w = self._prepare_input(v_input, v_lens)
        assert lay.B == self.batch_size
        assert h * w == self.spatial_size
        B, Fn, Hq = lay.B, lay.n_frames, self.hidden_size
        C = self.num_res_block_channels
        fused = self._use_fused_trunk()
        gen = self._generator_hip      # (its op-by-op torch form is test infrastructure: tests/torch_partners.py)
        join = None
        if fused:       # the generator (question LSTM chain + hop attention) on the side stream, joined after conv_init + BatchNorm
            film_per_block, join = self._fork_generator(lambda: tuple(gen(q_input, q_lens, lay)), n_img=lay.n_img)
        else:
            film_per_block = gen(q_input, q_lens, lay)

        def film_fn(k):
            s = 2 * C * k
            fv = film_per_block[k]
            return fv[:, s:s + C], fv[:, s + C:s + 2 * C]                 # :228-230

        self._trunk_grad_scale = grad_scale_of(self.compute_dtype) if fused else 1.0
        if fused:       # train mode: the conv trunk on fused conv epilogues (two autograd nodes around the generator's join)
            x = self._trunk_fused(x, lay, [(film_per_block[k], 2 * C * k) for k in range(self.num_res_blocks)], join)
        else:
            x = self._trunk(x, lay, film_fn)
        return self._gp_tail(x, lay, h, w)                                # :240-250
