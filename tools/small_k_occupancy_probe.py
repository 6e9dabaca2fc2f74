import os, sys, time
import numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tools'); sys.path.insert(0,'tests')
import switches
from pilot_amd import engine
from pilot_amd.synthetic import make_problem
from conftest import GOLDEN_REAL, load_golden
g=load_golden(GOLDEN_REAL)
cases=[("kidney 634x14", g["proportions"], g["cost"]/g["cost"].max())]
for K in ([int(a) for a in sys.argv[1:]] or (2,8,16,24,32)): cases.append(("600x%d"%K,)+make_problem(600,K,8,seed=K,cells_per_patient=200))
for name,P,M in cases:
    plan=engine.DevicePlan(np.ascontiguousarray(P),np.ascontiguousarray(M)); plan.enable_timing(True)
    out=[]
    for wg in (0,2,3,4,6):
        switches.set("PILOT_OT_DEBUG", str(wg<<4) if wg else None)
        for _ in range(4): plan.run(0.1)
        plan.sync()
        for _ in range(20): plan.run(0.1)
        plan.sync()
        m,tr=plan.kernel_times_ms(20)
        out.append("%s: %.3f"%(wg or "default", m.mean()))
    switches.set("PILOT_OT_DEBUG", None)
    print(name, "main kernel ms by resident workgroups per CU ->", ", ".join(out), flush=True)
    plan.close()
