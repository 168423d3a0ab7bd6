"""The producers of the training set's feature files: CN-HuBERT content features
(4-cnhubert/<name>.pt) and Chinese-RoBERTa phone-level features (3-bert/<name>.pt), on the library's kernels."""
from .cnhubert import CNHubert, HubertModel  # noqa: F401
from .roberta import BertFeatures  # noqa: F401
