"""The module path the reference's trainer and tester import (``models/trans_sr_trainer.py:3``,
``models/trans_sr_tester.py:3``).  In the reference this file is byte-identical to
``rdst_variations.py`` from ``class DenseSTLayer`` on; here it simply re-exports."""
from .rdst_variations import *  # noqa: F401,F403
from .rdst_variations import DenseSTLayer, RDSTB, RDSTSR, make_RDSTSR  # noqa: F401
