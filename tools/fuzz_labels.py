"""Label columns -> first-appearance codes (tl._first_appearance_codes: the native pass of pilot_ot_label_codes, or its pandas fallback) against
pd.factorize on strings, integers, floats, booleans, mixed objects, Categoricals (unused categories, NaN) and strided / shuffled views.  CPU only.
usage: python tools/fuzz_labels.py [seed]"""
import sys, numpy as np, pandas as pd
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from pilot_amd import tl
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
bad=0
for case in range(300):
    n=int(rng.choice([0,1,2,5,63,64,65,1000,4097,200000]))
    k=int(rng.choice([1,2,3,50,300,5000]))
    kind=rng.choice(["str","int","float","mixed","cat","cat_unused","cat_nan","obj_nan","bool"])
    base=rng.integers(0,k,n)
    if kind=="str": s=pd.Series(np.array(["t%d"%v for v in base],dtype=object))
    elif kind=="int": s=pd.Series(base.astype(np.int64))
    elif kind=="float": s=pd.Series(base.astype(np.float64)/3)
    elif kind=="bool": s=pd.Series(base%2==0)
    elif kind=="mixed": s=pd.Series(np.array([("t%d"%v if v%2 else int(v)) for v in base],dtype=object))
    elif kind=="cat": s=pd.Series(pd.Categorical(["t%d"%v for v in base]))
    elif kind=="cat_unused": s=pd.Series(pd.Categorical(["t%d"%v for v in base],categories=["zz","t0"]+["t%d"%v for v in range(1,k+3)]))
    elif kind=="cat_nan":
        vals=np.array(["t%d"%v for v in base],dtype=object)
        if n: vals[rng.random(n)<0.1]=np.nan
        s=pd.Series(pd.Categorical(vals))
    else:
        vals=np.array(["t%d"%v for v in base],dtype=object)
        if n: vals[rng.random(n)<0.1]=None
        s=pd.Series(vals)
    if rng.random()<0.3 and n>3: s=s.iloc[::2]            # a strided view
    if rng.random()<0.3 and n>3: s=s.sample(frac=1.0,random_state=int(rng.integers(1<<30)))
    try:
        codes,uniq=tl._first_appearance_codes(s)
    except Exception as e:
        print("EXC",kind,n,k,repr(e)); bad+=1; continue
    want_u=s.unique()
    wc,wu=pd.factorize(s, use_na_sentinel=True)
    wu=np.asarray(wu)
    ok = len(uniq)==len(wu) and np.array_equal(np.asarray(codes),wc) and all((a==b) or (a!=a and b!=b) for a,b in zip(list(uniq),list(wu)))
    if not ok:
        bad+=1; print("FAIL",kind,n,k,len(uniq),len(wu),np.asarray(codes)[:8],wc[:8])
print(bad,"of 300 failed")
