"""The reference's "type-1" GRU cell (models/GRU_cell.py:7-31) as a parameter container plus a
forward that runs the fused HIP decay+GRU kernel (TEMP_GRU_TYPE1): gates r,z come from the hidden
state only, W_ih is (H,I) and feeds the new gate only, h' = n + z*(h - n)."""
import torch
from torch.nn import Module, Parameter

from . import functional as TF


class GRUCell(Module):
    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.weight_ih = Parameter(torch.randn(hidden_size, input_size))
        self.weight_hh = Parameter(torch.randn(3 * hidden_size, hidden_size))
        self.bias_ih = Parameter(torch.randn(hidden_size))
        self.bias_hh = Parameter(torch.randn(3 * hidden_size))

    def forward(self, input, hidden):
        """Reference call shape: input (1,n,I), hidden (1,n,H) -> (None, h'[None]).
        (The reference's `.squeeze()` breaks for n == 1, models/GRU_cell.py:19-20; this does not.)"""
        x = input.reshape(-1, self.input_size)
        h = hidden.reshape(-1, self.hidden_size)
        dt = torch.zeros(x.shape[0], dtype=x.dtype, device=x.device)      # no decay: exp(0) = 1
        hy = TF.gru_step(x, h, dt, self.weight_ih, self.weight_hh, self.bias_ih, self.bias_hh, 0.0, None, None, type1=True)
        return None, hy.unsqueeze(0)
