"""QOnlyLSTM — drop-in for models/q_only_lstm.py (config-1 plumbing model) on the persistent LSTM kernel."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class QOnlyLSTM(nn.Module):
    """Same signature as the reference (q_only_lstm.py:11): Embedding(pad 0) -> LSTM with a RANDOM initial
    state that is carried between calls unless init_hidden() (:50-54) -> last valid step -> Linear."""

    def __init__(self, batch_size, embedding_size, hidden_size, nb_classes, vocab_size):
        super(QOnlyLSTM, self).__init__()
        self.nb_classes, self.batch_size, self.hidden_size = nb_classes, batch_size, hidden_size
        self.embed = nn.Embedding(vocab_size, embedding_size, padding_idx=0)
        self.lstm = nn.LSTM(embedding_size, hidden_size)
        self.hidden_1 = None
        self.out_linear = nn.Linear(hidden_size, nb_classes)
        for m in self.modules():                                      # weights_init (:28-44)
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                nn.init.xavier_uniform_(m.weight.data)
                m.bias.data.fill_(0.0)
            if isinstance(m, nn.LSTM):
                nn.init.xavier_uniform_(m.weight_ih_l0)
                nn.init.orthogonal_(m.weight_hh_l0)
                for names in m._all_weights:
                    for name in filter(lambda n: "bias" in n, names):
                        bias = getattr(m, name)
                        n = bias.size(0)
                        bias.data[n // 4:n // 2].fill_(1.0)
                m.bias_ih_l0.data.fill_(0.0)
        self.init_hidden()

    def init_hidden(self):
        dev = self.embed.weight.device
        self.hidden_1 = (torch.randn(1, self.batch_size, self.hidden_size, device=dev),
                         torch.randn(1, self.batch_size, self.hidden_size, device=dev))

    def forward(self, q_input, q_lens):
        """q_only_lstm.py:57-69; the caller pre-sorts by q_len (q_only_eval.py:76-77)."""
        dev = self.embed.weight.device
        emb = self.embed(q_input.to(dev))
        ql_cpu = q_lens.detach().cpu().long()
        B = emb.shape[0]
        xg = F.linear(emb, self.lstm.weight_ih_l0, self.lstm.bias_ih_l0 + self.lstm.bias_hh_l0)
        h0, c0 = self.hidden_1[0][0].to(dev), self.hidden_1[1][0].to(dev)
        S = int(ql_cpu.max())
        out, hn, cn = ops.lstm_seq(xg, self.lstm.weight_hh_l0, h0, c0, ql_cpu.to(torch.int32).to(dev), 1, S)
        self.hidden_1 = (hn.unsqueeze(0), cn.unsqueeze(0))           # carried WITH its graph upstream (:62); detached here
        self.hidden_1 = (self.hidden_1[0].detach(), self.hidden_1[1].detach())
        idx = (ql_cpu.to(dev).view(B, 1, 1) - 1).expand(B, 1, self.hidden_size)
        return self.out_linear(out.gather(1, idx).view(B, self.hidden_size))
