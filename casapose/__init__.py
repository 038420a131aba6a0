"""`casapose` -- the reference's import surface, served by the MI355X implementation.

The reference's scripts and downstream code import `casapose.pose_models.tfkeras.Classifiers` (tfkeras.py:17),
`casapose.pose_estimation.voting_layers_2d.CoordLSVotingWeighted`, `casapose.pose_estimation.ransac_voting`,
`casapose.pose_estimation.pose_evaluation`, `casapose.utils.config_parser.parse_config`, `casapose.utils.io_utils.write_poses`,
`casapose.utils.learning_rate_schedules`, `casapose.data_handler.vectorfield_dataset` ...  This package makes those imports resolve
UNCHANGED: `casapose.<x>` IS the module object `casapose_amd.<x>` (one copy -- the alias is registered in sys.modules, nothing is
loaded twice, so the HIP library handle, launch plans and caches are shared whichever name was used).  A module the reference has
and this implementation does not (the TensorFlow data-augmentation layers, drawing utilities) raises ModuleNotFoundError naming
the casapose_amd module that is missing.
"""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import sys

import casapose_amd as _impl

_PREFIX, _REAL = __name__ + ".", _impl.__name__ + "."


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        real = _REAL + fullname[len(_PREFIX):]
        try:
            spec = importlib.util.find_spec(real)
        except ModuleNotFoundError:
            spec = None
        if spec is None:
            raise ModuleNotFoundError("No module named %r (the MI355X implementation has no %r)" % (fullname, real), name=fullname)
        return importlib.machinery.ModuleSpec(fullname, self, is_package=spec.submodule_search_locations is not None)

    def create_module(self, spec):
        module = importlib.import_module(_REAL + spec.name[len(_PREFIX):])  # the SAME module object under a second name
        self._real_spec[id(module)] = module.__spec__
        return module

    def exec_module(self, module):
        # importlib has just overwritten module.__spec__ with the ALIAS spec (it does so unconditionally): put the real one back, so that
        # importlib.reload() re-executes the real source and relative imports inside the module see __package__ == __spec__.parent
        real = self._real_spec.pop(id(module), None)
        if real is not None:
            module.__spec__ = real

    _real_spec = {}


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
__path__ = []  # a package whose submodules come from the finder above only
__version__ = getattr(_impl, "__version__", "0")
