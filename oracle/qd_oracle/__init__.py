"""qd_oracle -- TEST INFRASTRUCTURE ONLY (CPU oracle; see oracle/README.md).

Parity status: PINNED against the reference itself, imported in the authoring
container (oracle/gen_golden.py, outputs committed under tests/golden/), and
against the known-answer values of SURVEY.md Appendix A.  The reference's own
test-suite holds no vectors for this path.
"""
from .params import defaults, is_set            # noqa: F401
from .grid import Grid                          # noqa: F401
from .atmos import AtmosOracle                  # noqa: F401
from .ocean import OceanOracle                  # noqa: F401
from .forcing import Forcing, Orbit             # noqa: F401
