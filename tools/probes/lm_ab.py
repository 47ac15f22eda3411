"""C2 / C3 fits (batched LM of independent frames) with the library BODYFIT_LIB names: seconds of 5 repeats each."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fit_bench
api = importlib.import_module("3dbodyanimation_amd.api"); synth = importlib.import_module("3dbodyanimation_amd.synth")
model = synth.make_model(0); gm = api.Model(model)
c3 = fit_bench.fit_c3(api, synth, model, gm, repeats=7)
c2 = fit_bench.fit_c2(api, synth, model, gm, repeats=7)
print(os.path.basename(os.environ.get("BODYFIT_LIB", "libbodyfit.so")), "c3 ms", [round(x * 1e3, 3) for x in sorted(c3["seconds_all"])], "us/it", c3["us_per_iteration"], "final", c3["final_cost"],
      "| c2 ms", [round(x * 1e3, 4) for x in sorted(c2["seconds_all"])], "final", c2["final_cost"], flush=True)
