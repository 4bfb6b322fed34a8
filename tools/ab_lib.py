"""A/B helper: run a script (bench.py, tools/step_without_input_stage.py ...) against ANOTHER build of libwwhip.so in the same
gpurun call, e.g. one compiled with an experimental macro:  WW_AB_LIB=wakeword_trainer_home_amd/csrc/libwwhip_x.so python
tools/ab_lib.py bench.py --no-cpu-baseline   (same-box pairs are the only comparisons finer than the +-1.5 % box-to-box spread)."""
import os, runpy, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from wakeword_trainer_home_amd import _native
if os.environ.get("WW_AB_LIB"):
    _native._LIB_PATH = Path(os.environ["WW_AB_LIB"]).resolve()
script = sys.argv[1]
sys.argv = sys.argv[1:]
runpy.run_path(script, run_name="__main__")
