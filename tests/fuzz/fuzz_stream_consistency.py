"""Randomised consistency check of the MFMA streaming kernel (run on an MI355X): at a FIXED iteration count (tolerances 0, so no
early-exit decisions can differ) its x, u, d, p must agree with the row kernels (fma arithmetic for nx + nu <= 16, the exact wave
kernel for nx = 32) to rounding level, for random classes, ragged batches, settings and warm starts.
      python tests/fuzz/fuzz_stream_consistency.py [seconds]"""
import sys, time; sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import numpy as np
import accelerated_tinympc_amd as T
pr=T.problems
rng=np.random.default_rng(5); t_end=time.time()+float(sys.argv[1]) if len(sys.argv) > 1 else 60.0; rounds=0; worst=0.0
while time.time()<t_end:
    kind,N=[("quad",30),("quad",13),("cartpole",10),("odd",7),("rand32",50)][rng.integers(5)]
    prob={"quad":lambda:pr.quadrotor(20,N),"cartpole":lambda:pr.cartpole(N),"odd":lambda:pr.random_system(8,3,N,seed=99),"rand32":lambda:pr.random_system(32,16,N)}[kind]()
    nx,nu=prob["nx"],prob["nu"]; B=int(rng.choice([1,15,16,17,100]))
    settings=dict(abs_pri_tol=0.0,abs_dua_tol=0.0,max_iter=int(rng.choice([1,5,12])),check_termination=1,en_state_bound=int(rng.integers(2)),en_input_bound=1)
    x0=rng.uniform(-0.3,0.3,size=(B,nx)).astype(np.float32); xref=(rng.standard_normal((B,N,nx))*0.1).astype(np.float32)
    res=[]
    for v in ((1,2) if kind=="rand32" else (1,3)):
        s=T.TinyBatchSolver(prob,B,settings=settings); s.select_kernel(v); s.set_bounds(*pr.bounds_arrays(prob)); s.set_xref(xref); s.set_x0(x0)
        s.solve(); s.reset_dual_variables(); s.solve(); res.append(s.get_state()); s.close()
    for k in ("u","x","d","p"):
        sc=max(float(np.abs(res[1][k]).max()),1e-2); e=float(np.abs(res[0][k].astype(np.float64)-res[1][k]).max())/sc; worst=max(worst,e)
        assert np.all(np.isfinite(res[0][k])) and e<2e-3,(kind,N,B,settings,k,e)
    rounds+=1
print(f"fuzz ok: {rounds} rounds, MFMA streaming kernel within {worst:.2e} (relative) of the row kernels at fixed iteration counts")
