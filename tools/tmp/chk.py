import sys, os; sys.path.insert(0, os.getcwd())
import importlib, numpy as np
api = importlib.import_module("3dbodyanimation_amd.api"); synth = importlib.import_module("3dbodyanimation_amd.synth")
m = synth.make_model(0); gm = api.Model(m)
F=40
seq = synth.make_sequence(m, F, seed=4)
w, mu, cov = synth.make_gmm(0)
prob = api.Problem.from_sequence(gm, seq, n_cols=86, use_shape=True, beta_per_frame=True, beta_pose=20.0, gmm=api.Gmm(w, mu, cov), beta_shape=30.0)
rng=np.random.default_rng(0)
x = seq.init_params + 0.05*rng.standard_normal(seq.init_params.shape); b = 0.3*rng.standard_normal((F,10))
r1,J1,c1 = prob.evaluate(x,b,True)
r0,_,c0 = prob.evaluate(x,b,False)
print("r bit-identical:", np.array_equal(r1,r0), np.abs(r1-r0).max(), np.array_equal(c0,c1))
import os
for it in (5,10,20,30,40,50,60):
    res=[]
    for plain in ("0","1"):
        os.environ["BODYFIT_LM_PLAIN"]=plain
        xs,bs,ss = prob.solve(seq.init_params, np.zeros((F,10)), independent=True, max_iters=it)
        res.append((xs,bs,ss))
    d=np.abs(res[0][0]-res[1][0]).max(axis=1)
    print(it, "max dx", d.max(), "frame", int(d.argmax()), "f39 dx", d[39], [ (s.iterations,s.n_successful,s.n_unsuccessful) for s in (res[0][2][39],res[1][2][39])])
