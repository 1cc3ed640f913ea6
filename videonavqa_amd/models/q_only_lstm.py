"""QOnlyLSTM — the question-only baseline (config-1 plumbing; drop-in for models/q_only_lstm.py) with its
recurrence on the persistent HIP LSTM kernel."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .common import reference_init_


class QOnlyLSTM(nn.Module):
    """Signature of the reference (q_only_lstm.py:11).  Embedding(padding_idx 0) -> LSTM whose initial state
    is drawn from N(0,1) by init_hidden() and otherwise CARRIED from the previous call (q_only_lstm.py:50-54,
    q_only_eval.py:80-82) -> output at each question's last token -> Linear."""

    def __init__(self, batch_size, embedding_size, hidden_size, nb_classes, vocab_size):
        super(QOnlyLSTM, self).__init__()
        self.batch_size, self.hidden_size, self.nb_classes = batch_size, hidden_size, nb_classes
        self.embed = nn.Embedding(vocab_size, embedding_size, padding_idx=0)
        self.lstm = nn.LSTM(embedding_size, hidden_size)
        self.out_linear = nn.Linear(hidden_size, nb_classes)
        self.apply(reference_init_)
        self.hidden_1 = None
        self.init_hidden()

    def init_hidden(self):
        shape = (1, self.batch_size, self.hidden_size)
        dev = self.embed.weight.device
        self.hidden_1 = (torch.randn(shape, device=dev), torch.randn(shape, device=dev))

    def forward(self, q_input, q_lens):
        """q_input int64 [B, L] (caller pre-sorted by length, q_only_eval.py:76-77), q_lens [B] -> logits [B, K]."""
        dev = self.embed.weight.device
        lens = q_lens.detach().cpu().long()
        B = q_input.shape[0]
        tokens = self.embed(q_input.to(dev))
        gates_in = F.linear(tokens, self.lstm.weight_ih_l0, self.lstm.bias_ih_l0 + self.lstm.bias_hh_l0)
        h0, c0 = (t[0].to(dev) for t in self.hidden_1)
        hs, h_n, c_n = ops.lstm_seq(gates_in, self.lstm.weight_hh_l0, h0, c0, lens.to(torch.int32).to(dev),
                                    1, int(lens.max()))
        self.hidden_1 = (h_n.detach().unsqueeze(0), c_n.detach().unsqueeze(0))
        last = hs[torch.arange(B, device=dev), lens.to(dev) - 1]
        return self.out_linear(last)
