"""Import alias: `cpu-vision_amd/` (the directory name the project layout prescribes) is not a valid Python
identifier, so `import cpu_vision_amd` resolves here and executes the real package from that directory."""
from pathlib import Path as _Path

_real = _Path(__file__).resolve().parent.parent / "cpu-vision_amd"
__path__ = [str(_real)]
__file__ = str(_real / "__init__.py")
exec(compile((_real / "__init__.py").read_text(), __file__, "exec"))
