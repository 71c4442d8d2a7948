"""Import-compatibility alias of the reference's extension module (``pytrimal._trimal``): the same public names
plus the runtime-support flags its test-suite inspects (``tests/test_automatic_trimmer.py:98-108``).  The CPU SIMD
backends do not exist here; the only compute platform is ``"hip"``."""
from .alignment import Alignment, AlignmentResidues, AlignmentSequences, TrimmedAlignment
from .matrix import SimilarityMatrix
from . import trimmer as _trimmer
from .trimmer import (
    AutomaticTrimmer,
    BaseTrimmer,
    ManualTrimmer,
    OverlapTrimmer,
    RepresentativeTrimmer,
)

_SSE2_RUNTIME_SUPPORT = False
_AVX2_RUNTIME_SUPPORT = False
_NEON_RUNTIME_SUPPORT = False
_SSE2_BUILD_SUPPORT = False
_AVX2_BUILD_SUPPORT = False
_NEON_BUILD_SUPPORT = False
_HIP_BUILD_SUPPORT = True


def __getattr__(name):
    if name == "_HIP_RUNTIME_SUPPORT":  # counted on first use, not at import (see trimmer._best_platform)
        return _trimmer._best_platform() == "hip"
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")

__all__ = [
    "Alignment", "AlignmentResidues", "AlignmentSequences", "TrimmedAlignment", "SimilarityMatrix",
    "BaseTrimmer", "AutomaticTrimmer", "ManualTrimmer", "OverlapTrimmer", "RepresentativeTrimmer",
]
