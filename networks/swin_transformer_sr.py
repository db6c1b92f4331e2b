"""Re-export of ``rdst_amd.networks.swin_transformer_sr`` under the reference's module path."""
from rdst_amd.networks.swin_transformer_sr import *  # noqa: F401,F403
from rdst_amd.networks import swin_transformer_sr as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
