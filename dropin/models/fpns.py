"""Drop-in shim: resolves `models.fpns` of a reference checkout to the MI355X implementation."""
from vrdone_amd.models.fpns import *  # noqa: F401,F403
from vrdone_amd.models import fpns as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
