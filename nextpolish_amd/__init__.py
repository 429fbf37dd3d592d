"""MI355X-native short-read polishing core of NextPolish (score_chain hot path).

Python layer = thin host-side mirror of the reference's caller (source/lib/nextpolish1.py) above the
C ABI in include/nextpolish1.h; all compute runs in the in-tree HIP library (nextpolish_amd/lib)."""
from . import _native  # noqa: F401
