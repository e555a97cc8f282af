"""The encoder factory (layers/categorical_encoding/mutils.py:51-72)."""
from .linear_encoding import LinearCategoricalEncoding
from .variational_dequantization import VariationalDequantization


# the CLI flag builders (:14-48) are host-side code of the reference's training template: served from the user's checkout
# when it is on sys.path, by NAME — any other missing attribute is a plain AttributeError
_FROM_REFERENCE = ("add_encoding_parameters", "encoding_args_to_params")


def __getattr__(name):
    if name in _FROM_REFERENCE:
        from ... import compat
        return compat.fall_through("layers.categorical_encoding.mutils", name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))


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
