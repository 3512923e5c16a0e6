"""MI355X-native (gfx950) hot path of LI-Yiquan/3DPointCloudAttack: the adversarial-attack iteration.

The directory name is not a Python identifier; load it with::

    import importlib; pc3d = importlib.import_module("3dpointcloudattack_amd")

Sub-packages ``attack/``, ``model/`` and ``utils/`` mirror the reference's import paths
(``attack.CW.CW_attack.CW`` ...). ``install_dropin()`` aliases them into ``sys.modules`` under the
reference's top-level names so reference-style driver code runs unchanged.
"""
import importlib
import importlib.abc
import importlib.machinery
import sys

from . import _lib  # noqa: F401
from ._lib import LIB_PATH, Pc3dError, load  # noqa: F401

__version__ = "0.1.0"

_MIRRORS = ("attack", "model", "utils", "dataset")


class _MirrorFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Resolves ``attack.*`` / ``model.*`` / ``utils.*`` to the SAME module objects as
    ``3dpointcloudattack_amd.attack.*`` ... (no second copy, relative imports inside keep working)."""

    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".", 1)[0] not in _MIRRORS:
            return None
        try:
            real = importlib.import_module(f"{__name__}.{fullname}")
        except ModuleNotFoundError:
            return None
        spec = importlib.machinery.ModuleSpec(fullname, self, is_package=hasattr(real, "__path__"))
        spec._pc3d_real = real
        return spec

    def create_module(self, spec):
        return spec._pc3d_real

    def exec_module(self, module):
        pass


_finder = None


def install_dropin(force=False):
    """Make the reference's import paths (``attack.CW.CW_attack``, ``model.pointnet``, ``utils.dis_utils_numpy``
    ...) resolve to this package's MI355X mirrors, so reference-style driver code runs unchanged.
    Refuses to shadow foreign modules already imported under those names unless force=True."""
    global _finder
    for name in _MIRRORS:
        mod = sys.modules.get(name)
        if mod is not None and not getattr(mod, "__name__", "").startswith(__name__):
            if not force:
                raise ImportError(f"a foreign module named {name!r} is already imported; pass force=True")
            for k in [k for k in sys.modules if k == name or k.startswith(name + ".")]:
                del sys.modules[k]
    if _finder is None:
        _finder = _MirrorFinder()
        sys.meta_path.insert(0, _finder)


def uninstall_dropin():
    global _finder
    if _finder is not None:
        sys.meta_path.remove(_finder)
        _finder = None
    for k in [k for k in sys.modules if k.split(".", 1)[0] in _MIRRORS
              and getattr(sys.modules[k], "__name__", "").startswith(__name__)]:
        del sys.modules[k]
