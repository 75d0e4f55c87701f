"""Import shim: `import rust_snappy_amd` loads the package that lives in the
directory `rust-snappy_amd/` (a hyphen is not importable as a module name)."""
import importlib.util
import sys
from pathlib import Path

_dir = Path(__file__).resolve().parent / "rust-snappy_amd"
_spec = importlib.util.spec_from_file_location(
    "rust_snappy_amd", _dir / "__init__.py",
    submodule_search_locations=[str(_dir)])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rust_snappy_amd"] = _mod
_spec.loader.exec_module(_mod)
