"""A/B of the one-launch sweep's tuning knobs (BODYFIT_MESH_PRIO, BODYFIT_TRICKLE_START, BODYFIT_TRICKLE_SLEEP: read once per
process): one line per frame count with the wall time per step of back-to-back sweeps and the dispatch's own duration."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ab_sweep

tag = {k: os.environ.get(k) for k in ("BODYFIT_MESH_PRIO", "BODYFIT_TRICKLE_START", "BODYFIT_TRICKLE_SLEEP", "BODYFIT_J_SCOPE") if os.environ.get(k)}
for F in [int(a) for a in sys.argv[1:]] or [256]:
    r = ab_sweep.run(F, True, iters=300)
    print(json.dumps(dict(tune=tag, F=F, us_per_step=r["us_per_step"], kernel_us=r["sweep_roles"])), flush=True)
