"""Drop-in `models` package: importing it registers CDM and CMDM in `models.base.Model`
(the reference's models/__init__.py does the same through star-imports)."""
from .cdm import *    # noqa: F401,F403
from .cmdm import *   # noqa: F401,F403
