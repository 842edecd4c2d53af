"""Import alias: `import f5_tts_amd` loads the package that lives in ./korean-f5-tts_amd/ (a directory name Python's
import statement cannot spell).  The module object registered under this name IS that package."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "korean-f5-tts_amd")
_spec = _ilu.spec_from_file_location(__name__, _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
