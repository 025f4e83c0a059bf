"""Public names of the package."""
from . import pod  # noqa: F401
from .backend import EXPORTS, HIP_LIB, BackendError, HipBackend, hip_lib  # noqa: F401
from .scene import HOST_LIB, Scene, into_device_material  # noqa: F401
from . import dist  # noqa: F401
