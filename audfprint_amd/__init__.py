"""audfprint_amd: the precompute / ingest hot path of dpwe/audfprint on MI355X.
Importing the package changes nothing in the process; configure_runtime() is the one explicit knob (audfprint_amd/_lib.py)."""
from ._lib import configure_runtime, runtime_info          # noqa: F401
