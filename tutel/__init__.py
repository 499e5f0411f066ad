"""Drop-in alias: `import tutel`, `from tutel import moe, net, system`,
`import tutel.impls.fast_dispatch` ... all resolve to the MI355X-native package `tutel_amd`.
Existing Tutel user code runs against the HIP implementation without edits."""
import importlib
import importlib.abc
import importlib.util
import sys

import tutel_amd

_PREFIX = __name__ + "."


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        real = "tutel_amd." + fullname[len(_PREFIX):]
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except ModuleNotFoundError:
            return None
        return importlib.util.spec_from_loader(fullname, self)

    def create_module(self, spec):
        return importlib.import_module("tutel_amd." + spec.name[len(_PREFIX):])

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _AliasFinder())
__version__ = tutel_amd.__version__


def __getattr__(name):
    try:
        return importlib.import_module(_PREFIX + name)
    except ModuleNotFoundError as ex:
        raise AttributeError(name) from ex
