"""Attention hop-combine operator (reference layers/combine.py:22-27)."""
import torch


def attention_combine(x, lstm):
    """x [N,K,D] -> [N,D].  v1: the bi-LSTM runs through torch's GPU LSTM; the softmax / weighted sum follow."""
    lstm.flatten_parameters()
    score, _ = lstm(x)
    score = torch.softmax(score.sum(-1), dim=1).unsqueeze(-1)
    return (x * score).sum(1)
