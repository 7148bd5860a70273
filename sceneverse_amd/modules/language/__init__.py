from .bert import BERTLanguageEncoder  # noqa: F401
from .clip import CLIPLanguageEncoder  # noqa: F401
