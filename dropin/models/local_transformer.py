"""Drop-in shim: resolves `models.local_transformer` of a reference checkout to the MI355X implementation."""
from vrdone_amd.models.local_transformer import *  # noqa: F401,F403
from vrdone_amd.models import local_transformer as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
