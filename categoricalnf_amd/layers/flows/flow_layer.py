"""FlowLayer protocol — same surface as layers/flows/flow_layer.py:5-31 of the reference."""
import torch.nn as nn


class FlowLayer(nn.Module):
    """forward(z, ldj=None, reverse=False, **kwargs) -> (z, ldj) or (z, ldj, detail)."""

    def __init__(self):
        super().__init__()

    def forward(self, z, ldj=None, reverse=False, **kwargs):
        raise NotImplementedError

    def reverse(self, z, ldj=None, **kwargs):
        return self.forward(z, ldj, reverse=True, **kwargs)

    def need_data_init(self):
        """True if the layer wants a data-dependent initialisation pass (ActNorm)."""
        return False

    def data_init_forward(self, input_data, **kwargs):
        raise NotImplementedError

    def info(self):
        raise NotImplementedError
