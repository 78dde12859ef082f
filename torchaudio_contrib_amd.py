"""Import shim: ``import torchaudio_contrib_amd`` loads the package that lives in the directory
``torchaudio-contrib_amd/`` (a hyphen is not importable as-is)."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'torchaudio-contrib_amd')
_spec = importlib.util.spec_from_file_location(
    'torchaudio_contrib_amd', os.path.join(_pkg_dir, '__init__.py'),
    submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['torchaudio_contrib_amd'] = _mod
_spec.loader.exec_module(_mod)
