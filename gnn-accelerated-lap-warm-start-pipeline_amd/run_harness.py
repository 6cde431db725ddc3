"""Run one of the reference's harness scripts, unedited, on the MI355X packages.

    python /path/to/gnn-accelerated-lap-warm-start-pipeline_amd/run_harness.py scripts/gnn_benchmark.py [its args]

Why a launcher and not PYTHONPATH: the harness scripts put the reference's repository root at
sys.path[0] (scripts/gnn_benchmark.py:21-22), and the reference's `solvers/__init__.py:22-25` puts
its `LAP/` directory in front as well, so the reference's own `gnn/`, `solvers/` and `lap/` win
over anything on PYTHONPATH.  `sys.modules` is consulted before `sys.path`, so this launcher
imports the MI355X `lap`, `solvers` and `gnn` packages FIRST and then runs the script in this
process; the script's `from solvers import ...` / `from gnn import ...` lines then bind to them.
"""
from __future__ import annotations

import importlib
import runpy
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
PACKAGES = ("lap", "solvers", "gnn")


def preload() -> dict:
    """Import the three packages from this directory and pin them in sys.modules."""
    sys.path.insert(0, str(HERE))
    try:
        loaded = {}
        for name in PACKAGES:
            stale = sys.modules.get(name)
            if stale is not None and not str(getattr(stale, "__file__", "")).startswith(str(HERE)):
                raise RuntimeError(f"'{name}' is already imported from {stale.__file__}; start the launcher "
                                   "in a fresh interpreter")
            mod = importlib.import_module(name)
            if not str(mod.__file__).startswith(str(HERE)):
                raise RuntimeError(f"'{name}' resolved to {mod.__file__}, not to {HERE}")
            loaded[name] = mod
        return loaded
    finally:
        sys.path.remove(str(HERE))


def main(argv=None) -> None:
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit(__doc__)
    preload()
    script = Path(argv[0]).resolve()
    sys.argv = [str(script)] + argv[1:]
    runpy.run_path(str(script), run_name="__main__")


if __name__ == "__main__":
    main()
