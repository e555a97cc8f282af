"""CLI flags and the encoder factory (layers/categorical_encoding/mutils.py:14-72)."""
from .linear_encoding import LinearCategoricalEncoding
from .variational_dequantization import VariationalDequantization


def add_encoding_parameters(parser, postfix=""):
    parser.add_argument("--encoding_dim" + postfix, help="Dimensionality of the embeddings.", type=int, default=4)
    parser.add_argument("--encoding_dequantization" + postfix, action="store_true",
                        help="If selected, variational dequantization is used for encoding categorical data.")
    parser.add_argument("--encoding_variational" + postfix, action="store_true",
                        help="If selected, the encoder distribution is joint over categorical variables.")
    parser.add_argument("--encoding_num_flows" + postfix, type=int, default=0,
                        help="Number of flows used in the embedding layer.")
    parser.add_argument("--encoding_hidden_layers" + postfix, type=int, default=2,
                        help="Number of hidden layers of flows used in the parallel embedding layer.")
    parser.add_argument("--encoding_hidden_size" + postfix, type=int, default=128,
                        help="Hidden size of flows used in the parallel embedding layer.")
    parser.add_argument("--encoding_num_mixtures" + postfix, type=int, default=8,
                        help="Number of mixtures used in the coupling layers (if applicable).")
    parser.add_argument("--encoding_use_decoder" + postfix, action="store_true",
                        help="If selected, we use a decoder instead of calculating the likelihood by inverting all flows.")
    parser.add_argument("--encoding_dec_num_layers" + postfix, type=int, default=1,
                        help="Number of hidden layers used in the decoder of the parallel embedding layer.")
    parser.add_argument("--encoding_dec_hidden_size" + postfix, type=int, default=64,
                        help="Hidden size used in the decoder of the parallel embedding layer.")


def encoding_args_to_params(args, postfix=""):
    g = lambda name: getattr(args, name + postfix)
    return {
        "use_dequantization": g("encoding_dequantization"),
        "use_variational": g("encoding_variational"),
        "use_decoder": g("encoding_use_decoder"),
        "num_dimensions": g("encoding_dim"),
        "flow_config": {"num_flows": g("encoding_num_flows"), "hidden_layers": g("encoding_hidden_layers"),
                        "hidden_size": g("encoding_hidden_size")},
        "decoder_config": {"num_layers": g("encoding_dec_num_layers"), "hidden_size": g("encoding_dec_hidden_size")},
    }


def create_encoding(encoding_params, dataset_class, vocab=None, vocab_size=-1, category_prior=None):
    """Pops `use_dequantization` / `use_variational` from the dict (observable, like the reference :53-54)."""
    assert not (vocab is None and vocab_size <= 0), \
        "[!] ERROR: When creating the encoding, either a torchtext vocabulary or the vocabulary size needs to be passed."
    use_dequantization = encoding_params.pop("use_dequantization")
    use_variational = encoding_params.pop("use_variational")
    if use_dequantization and "model_func" not in encoding_params["flow_config"]:
        print("[#] WARNING: For using variational dequantization as encoding scheme, a model function needs to be specified"
              " in the encoding parameters, key \"flow_config\" which was missing here. Will deactivate dequantization...")
        use_dequantization = False
    if use_dequantization:
        cls = VariationalDequantization
    elif use_variational:
        # the reference's VariationalCategoricalEncoding raises in forward under every configuration
        # (SURVEY.md §2 row 13), so there is no behaviour to reproduce
        raise NotImplementedError("variational encoding is non-functional in the reference and is not provided")
    else:
        cls = LinearCategoricalEncoding
    return cls(dataset_class=dataset_class, vocab=vocab, vocab_size=vocab_size, category_prior=category_prior,
               **encoding_params)
