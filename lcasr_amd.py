"""Import shim: the product package lives in ``long-context-asr_amd/`` (not a valid Python identifier),
so ``import lcasr_amd`` loads that directory as the package ``lcasr_amd``."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'long-context-asr_amd')
_spec = _u.spec_from_file_location('lcasr_amd', _os.path.join(_dir, '__init__.py'), submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules['lcasr_amd'] = _mod
_spec.loader.exec_module(_mod)
