"""Model assembly for language modelling: categorical encoder (linear flows) +
n x (ActNorm, [invertible 1x1 conv], autoregressive mixture-CDF coupling with an LSTM sub-network).

Same structure, constructor arguments and parameter names as the reference's
experiments/language_modeling/flow_model.py (FlowLanguageModeling :16-79), so a reference checkpoint loads
unchanged.  Forward (likelihood / training) only: the reference's autoregressive coupling has no inverse
(layers/flows/autoregressive_coupling.py:42).  The LSTM sub-network is PyTorch-ROCm; every flow layer around it
runs on the HIP kernels (all channels transformed, K = 51 in the published configuration: the
four-lanes-per-item path of the mixture kernel)."""
import torch
import torch.nn as nn

from ..host_utils import create_channel_mask, create_transformer_mask
from ..layers.categorical_encoding.mutils import create_encoding
from ..layers.flows.activation_normalization import ActNormFlow
from ..layers.flows.autoregressive_coupling import AutoregressiveMixtureCDFCoupling
from ..layers.flows.flow_model import FlowModel
from ..layers.flows.permutation_layers import InvertibleConv
from ..layers.networks.autoregressive_layers import AutoregressiveLSTMModel


class FlowLanguageModeling(FlowModel):

    def __init__(self, model_params, dataset_class, vocab_size, vocab):
        super().__init__(layers=None, name="Language Modeling Flow")
        self.model_params = model_params
        self.dataset_class = dataset_class
        self.max_seq_len = self.model_params["max_seq_len"]
        self.vocab_size = vocab_size
        self.vocab = vocab
        self._create_layers()
        self.print_overview()

    def _create_layers(self):
        p = self.model_params
        self.latent_dim = p["categ_encoding"]["num_dimensions"]
        model_func = lambda c_out: AutoregressiveLSTMModel(c_in=self.latent_dim, c_out=c_out, max_seq_len=self.max_seq_len,
                                                           num_layers=p["coupling_hidden_layers"],
                                                           hidden_size=p["coupling_hidden_size"],
                                                           dp_rate=p["coupling_dropout"],
                                                           input_dp_rate=p["coupling_input_dropout"])
        p["categ_encoding"]["flow_config"]["model_func"] = model_func
        self.encoding_layer = create_encoding(p["categ_encoding"], dataset_class=self.dataset_class,
                                              vocab_size=self.vocab_size, vocab=self.vocab)
        layers = []
        for i in range(p["coupling_num_flows"]):
            layers.append(ActNormFlow(self.latent_dim))
            if i > 0:
                layers.append(InvertibleConv(self.latent_dim))
            layers.append(AutoregressiveMixtureCDFCoupling(c_in=self.latent_dim, model_func=model_func,
                                                           block_type="LSTM model",
                                                           num_mixtures=p["coupling_num_mixtures"]))
        self.flow_layers = nn.ModuleList([self.encoding_layer] + layers)

    def forward(self, z, ldj=None, reverse=False, length=None, **kwargs):
        if length is not None:
            # max_len = the padded length of this batch (== length.max() for every batch the reference can
            # process) without the host sync
            kwargs["src_key_padding_mask"] = create_transformer_mask(length, max_len=z.size(1))
            kwargs["channel_padding_mask"] = create_channel_mask(length, max_len=z.size(1))
        return super().forward(z, ldj=ldj, reverse=reverse, length=length, **kwargs)

    def initialize_data_dependent(self, batch_list):
        print("Initializing data dependent...")
        with torch.no_grad():
            for _, kwargs in batch_list:
                kwargs["src_key_padding_mask"] = create_transformer_mask(kwargs["length"])
                kwargs["channel_padding_mask"] = create_channel_mask(kwargs["length"])
            for layer in self.flow_layers:
                batch_list = FlowModel.run_data_init_layer(batch_list, layer)
