"""MI355X-native drop-in for the ``kosmosx`` package of kyegomez/Kosmos-X
(/root/reference/kosmosx/__init__.py:1-4)."""
from kosmosx.model import KosmosTokenizer, Kosmos, KosmosLanguage

__all__ = ["KosmosTokenizer", "Kosmos", "KosmosLanguage"]
