"""pytrimal_amd -- MI355X-native MSA column-statistics engine behind pytrimal's Trimmer API.

Drop-in for the statistics path of `pytrimal <https://github.com/althonos/pytrimal>`_:
`Alignment`, `TrimmedAlignment`, `AutomaticTrimmer`, `ManualTrimmer`, `OverlapTrimmer`,
`RepresentativeTrimmer` and `SimilarityMatrix` keep the reference's names, arguments and error
behaviour (``/root/reference/src/pytrimal/__init__.py:3-28``); the gap / similarity / identity /
overlap statistics run as hand-written HIP kernels for gfx950 through ``libmsastat_hip.so``
(C ABI: ``include/msastat.h``).  There is no CPU fallback.
"""
from .alignment import Alignment, AlignmentResidues, AlignmentSequences, TrimmedAlignment
from .matrix import SimilarityMatrix
from .trimmer import (
    AutomaticTrimmer,
    BaseTrimmer,
    ManualTrimmer,
    OverlapTrimmer,
    RepresentativeTrimmer,
)

__version__ = "0.1.0"
__all__ = [
    "Alignment", "AlignmentResidues", "AlignmentSequences", "TrimmedAlignment", "SimilarityMatrix",
    "BaseTrimmer", "AutomaticTrimmer", "ManualTrimmer", "OverlapTrimmer", "RepresentativeTrimmer",
]
