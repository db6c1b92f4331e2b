"""Re-export of ``rdst_amd.networks.common`` under the reference's module path."""
from rdst_amd.networks.common import *  # noqa: F401,F403
from rdst_amd.networks import common as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
