from .minimal import Minimal  # noqa: F401
from .explorer import Explorer  # noqa: F401
from .deathmatch import Deathmatch  # noqa: F401
