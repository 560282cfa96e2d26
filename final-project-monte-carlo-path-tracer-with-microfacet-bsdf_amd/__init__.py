"""MI355X-native wavefront path-tracing hot path (package directory name is fixed by the build contract;
import it through `mcpt_loader.load()` because the directory name is not a Python identifier)."""
from . import scenes, pngio, hip_backend, build  # noqa: F401
from .hip_backend import HipGroup, HipScene, McptError  # noqa: F401
