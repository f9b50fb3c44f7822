"""Hop-combine modules: same constructor signatures, attributes and state_dict keys as the
reference's layers/combine.py (AttentionCombine :8-27, GeometricCombine :30-58).

GeometricCombine is normally not executed as a module at all: the conv layers hand its theta to the
aggregation kernel, which applies it in the epilogue (include/kpgnn.h, `theta`).  Calling the module
directly still works (used for KP-GIN, whose combine sits behind the per-hop MLP)."""
import torch
import torch.nn as nn


class AttentionCombine(nn.Module):
    """softmax_k( sum_c biLSTM(x)[:,k,c] ) weighted sum over the K hop slots.
    Args (reference order): hidden_size, K."""

    def __init__(self, hidden_size, K):
        super().__init__()
        self.attention_lstm = nn.LSTM(hidden_size, K, 1, batch_first=True, bidirectional=True, dropout=0.)

    def reset_parameters(self):
        self.attention_lstm.reset_parameters()

    def forward(self, x):
        from ..ops_combine import attention_combine
        return attention_combine(x, self.attention_lstm)


class GeometricCombine(nn.Module):
    """theta[k,:] = softmax_k( a (1-a)^k ), a = sigmoid(alphas).  Args (reference order): K, hidden_size."""

    def __init__(self, K, hidden_size):
        super().__init__()
        self.alphas = nn.Parameter(torch.zeros(hidden_size))
        self.K = K
        self.hidden_size = hidden_size

    def reset_parameters(self):
        nn.init.zeros_(self.alphas)

    def geometric_distribution(self):
        a = torch.sigmoid(self.alphas)
        powers = torch.arange(self.K, device=a.device, dtype=a.dtype).unsqueeze(-1)  # K,1
        thetas = a.unsqueeze(0) * (1 - a).unsqueeze(0) ** powers                       # K,D
        return torch.softmax(thetas, dim=0).unsqueeze(0)                              # 1,K,D

    def theta(self):
        """[K, D] contiguous weights for the fused kernel epilogues (one HIP launch; fp32 device parameters)."""
        if self.alphas.is_cuda and self.alphas.dtype == torch.float32:
            from ..ops_dense import geo_theta
            return geo_theta(self.alphas, self.K)
        return self.geometric_distribution().squeeze(0).contiguous()

    def forward(self, x):
        if self.alphas.is_cuda and self.alphas.dtype == torch.float32:
            return torch.sum(x * self.theta().unsqueeze(0), dim=-2)
        return torch.sum(x * self.geometric_distribution(), dim=-2)


def make_combine(combine, K, width):
    if combine == "attention":
        return AttentionCombine(width, K)
    if combine == "geometric":
        return GeometricCombine(K, width)
    raise ValueError("Not implemented combine function")
