"""MI355X-native PQ / IVFPQ nearest-neighbour engine: drop-in for the search path of
MKLab-ITI/multimedia-indexing (gr.iti.mklab.visual.datastructures.{PQ, IVFPQ}).

The directory name carries a hyphen (it follows the upstream project name), so import it with
    import importlib; mi = importlib.import_module("multimedia-indexing_amd")
"""
from . import _native  # noqa: F401
from ._native import MmidxError, build, lib  # noqa: F401
from .index import IVFPQ, PQ, Linear, AbstractSearchStructure, Answer, TransformationType, read_quantizer  # noqa: F401
from .frontend import PCA, VladAggregator, VladAggregatorMultipleVocabularies  # noqa: F401
from . import quantization  # noqa: F401
