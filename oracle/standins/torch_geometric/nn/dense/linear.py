import torch


class Linear(torch.nn.Linear):
    def __init__(self, in_channels, out_channels, bias=True, weight_initializer=None, bias_initializer=None):
        super().__init__(in_channels, out_channels, bias=bias)
