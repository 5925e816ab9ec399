"""Launcher that runs a script of a reference checkout (eval.py, train.py) with `models.*` resolved to the MI355X
implementation:

    cd /path/to/vrdone
    python /path/to/this/repo/dropin/run.py eval.py --cfg_path configs/vidvrd.yaml ...

Why a launcher: `python eval.py` puts the script's directory at sys.path[0], ahead of PYTHONPATH, so the checkout's own
`models/` package would win however PYTHONPATH is set (and this image's Python 3.10 has no -P / PYTHONSAFEPATH).  Here
dropin/ (the `models` shims) and the repo root (`vrdone_amd`) go in front, the script's directory right behind them (for
the reference's `utils`, `dataloaders`, ...), and the script runs as __main__.  Nothing here touches the GPU.
"""
import os
import runpy
import sys


def main(argv):
    if len(argv) < 2:
        sys.exit("usage: python dropin/run.py <script of the reference checkout> [its arguments ...]")
    here = os.path.dirname(os.path.abspath(__file__))
    repo = os.path.dirname(here)
    script = os.path.abspath(argv[1])
    script_dir = os.path.dirname(script)
    # python put dropin/ (this file's directory) at sys.path[0] already; make the order explicit and complete
    front = [here, repo]
    sys.path[:] = front + [script_dir] + [p for p in sys.path if os.path.abspath(p or os.getcwd()) not in front + [script_dir]]
    sys.argv = [script] + argv[2:]
    import models.maskvrd                   # fail here, loudly, if the shims do not resolve
    assert models.maskvrd.MaskVRD.__module__.startswith("vrdone_amd."), models.maskvrd.__file__
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main(sys.argv)
