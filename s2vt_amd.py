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

    # `python -m s2vt_amd.train_rl` (runpy): the code of the real module, run as __main__ with __package__ = "s2vt_amd" -- its relative
    # imports come back through this finder and resolve to the one real module object each
    def _real_spec(self, fullname):
        return importlib.util.find_spec(_REAL + fullname[len(_ALIAS):])

    def get_code(self, fullname):
        spec = self._real_spec(fullname)
        return spec.loader.get_code(spec.name)

    def get_source(self, fullname):
        spec = self._real_spec(fullname)
        return spec.loader.get_source(spec.name)

    def get_filename(self, fullname):
        return self._real_spec(fullname).origin

    def is_package(self, fullname):
        return self._real_spec(fullname).submodule_search_locations is not None


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[__name__] = _pkg
