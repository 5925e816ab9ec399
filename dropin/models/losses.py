"""Drop-in shim: resolves `models.losses` of a reference checkout to the MI355X implementation."""
from vrdone_amd.models.losses import *  # noqa: F401,F403
from vrdone_amd.models import losses as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
