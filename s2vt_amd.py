"""Import alias: ``import s2vt_amd`` == the package directory multitask-end-to-end-video-captioning_amd/ (whose name is
not a Python identifier).  Submodules resolve to the SAME module objects under both names (``s2vt_amd.ops is
importlib.import_module("multitask-end-to-end-video-captioning_amd.ops")``): one library handle, one set of exception
classes, whichever way a caller spells the import."""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_REAL = "multitask-end-to-end-video-captioning_amd"
_ALIAS = __name__
_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.startswith(_ALIAS + "."):
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(_ALIAS):])     # the one real module object

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[__name__] = _pkg
