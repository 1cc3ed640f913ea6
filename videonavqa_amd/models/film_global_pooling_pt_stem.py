"""FiLMGlobalPoolingPretrainedStem — drop-in for models/film_global_pooling_pt_stem.py."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .common import FiLMTrunkBase, compute_dtype, grad_scale_of, repeated_question_lstm


class FiLMGlobalPoolingPretrainedStem(FiLMTrunkBase):
    """Signature/defaults: film_global_pooling_pt_stem.py:13-23; extra keyword-only spatial_size, precision."""

    # A pooling head hands single-frame values to its classifier (no averaging over frames: 2.5 - 3 x the attention head's sensitivity to the
    # stem's roundings): its stem keeps conv22's / conv31's outputs and the features as split tensors on top of the mean-shifted storage
    # (FrozenStem(split_depth=3): 832 instead of 874 clips/s).  Measured at 8 x 35 x 224^2, precision fp16h vs fp32: 0.68 - 0.82e-3 instead of
    # 0.61 - 1.34e-3 (profiles/r06_pooling_heads.txt; the multi-hop model's T = 70 reads 0.44 - 0.65e-3 either way and keeps the default
    # plan).  Read by bench.build and the CLIs when they build the model's FrozenStem.
    stem_split_depth = 3

    def __init__(self, batch_size, q_embedding_size, nb_classes, num_input_channels=512,
                 num_res_block_channels=512, num_tail_channels=16, num_res_blocks=1, hidden_size=128,
                 q_encoder='lstm', vocab_size=134, *, spatial_size=130, precision='fp16h'):
        super(FiLMGlobalPoolingPretrainedStem, self).__init__()
        assert q_encoder.lower() in ['lstm', 'bow'], "Invalid question encoder! ('lstm', 'bow')"
        self.q_encoder = q_encoder
        self.nb_classes = nb_classes
        self.batch_size = batch_size
        self.q_embedding_size = q_embedding_size
        self.hidden_size = hidden_size
        self.spatial_size = spatial_size
        self.compute_dtype = compute_dtype(precision)
        self.hyb = precision == "fp16h"      # fp16 storage; split features, conv_init as three products, 1x1 / fc with split weights (common.compute_dtype)

        self.embed = nn.Embedding(vocab_size, q_embedding_size, padding_idx=0)        # :34
        self._build_trunk_head(num_input_channels, num_res_block_channels)             # :38-41
        total_out_size = 2 * num_res_block_channels * num_res_blocks
        encoder = nn.LSTM(q_embedding_size, hidden_size) if q_encoder == 'lstm' else \
            nn.Linear(q_embedding_size, hidden_size)                                   # :69-71
        self.film_layer = nn.ModuleList([encoder,                                      # :48
                                         nn.Linear(hidden_size, total_out_size),
                                         nn.ReLU(inplace=True)])
        self._build_film_pipeline(num_res_block_channels, num_res_blocks)              # :49
        self.c1x1_tail = nn.Conv2d(num_res_block_channels, num_tail_channels, kernel_size=1)  # :52
        self.out_linear = nn.Linear(spatial_size * num_tail_channels, nb_classes)     # :56
        for module in self.modules():                                                 # :58-59
            self.weights_init(module)
        for module in self.conv1x1_layers:                                            # :60-61
            self.weights_init(module)
        self.film_hidden = None
        self.init_hidden()

    def init_hidden(self):
        if self.q_encoder == 'lstm':          # :125-130 (the bag-of-words encoder carries no state)
            self.film_hidden = self._zero_hidden(self.batch_size, self.hidden_size, self.embed.weight.device)

    def forward(self, v_input, q_input, v_lens, q_lens):
        """film_global_pooling_pt_stem.py:180-238."""
        x, lay, h, w = self._prepare_input(v_input, v_lens)
        assert lay.B == self.batch_size
        assert h * w == self.spatial_size
        C = self.num_res_block_channels
        dev = x.device

        bow = not isinstance(self.film_layer[0], nn.LSTM)                    # :150

        def generator():     # question LSTM + FiLM projection on the side stream (common.FiLMTrunkBase._fork_generator)
            if bow:
                return self.bow_film_values(self.film_layer[0], self.film_layer[1], q_input, lay)
            emb = self.embed(q_input)
            h0, c0 = self._question_state(lay.B, self.hidden_size, q_lens, dev)
            h_last, _, (hn, cn) = repeated_question_lstm(self.film_layer[0], emb, q_lens, lay.n_frames, h0, c0,
                                                          wgrad_dtype=self._lstm_wgrad_dtype())
            self._store_question_state(hn, cn, q_lens)
            film = F.relu(self.film_layer[1](h_last))
            return film[lay.sample_of, lay.frame_of]

        self._trunk_grad_scale = grad_scale_of(self.compute_dtype) if self._use_fused_trunk() else 1.0
        if self._use_fused_trunk():       # train mode: generator and conv trunk on fused HIP ops
            film_img, join = self._fork_generator(
                lambda: self.bow_film_values(self.film_layer[0], self.film_layer[1], q_input, lay) if bow else
                self.question_film_values(self.film_layer[0], self.film_layer[1], q_input, q_lens, lay, padding_idx=0), n_img=lay.n_img)
            x = self._trunk_fused(x, lay, [(film_img, 2 * C * k) for k in range(self.num_res_blocks)], join)
        else:
            film_img, join = self._fork_generator(generator, n_img=lay.n_img)
            x = self._trunk_head(x, lay)
            join()

            def film_fn(k):
                s = 2 * C * k
                return film_img[:, s:s + C], film_img[:, s + C:s + 2 * C]

            x = self._trunk_blocks(x, lay, film_fn)
        return self._gp_tail(x, lay, h, w)
