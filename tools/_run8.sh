python3 - <<'PY'
import os, sys, json
os.environ["FORA_HIP_LIB"] = os.path.abspath("variants/lib_levels.so")
sys.path.insert(0, ".")
import numpy as np, fora_amd
from fora_amd import synth
n, m, row_ptr, col = synth.preset("webstanford")
eng = fora_amd.Engine(0); eng.set_graph(n, m, row_ptr, col); eng.set_params(alpha=0.2, epsilon=0.5, seed=1)
srcs = synth.query_set(n, 1000, 20261001)
eng.push(srcs, want=False)
st0 = eng.stamps().astype(np.int64)
eng.push(srcs, want=False)
st = eng.stamps().astype(np.int64) - st0
print("Mcyc per level (sum over 256 WGs):", [round(int(x) / 1e6) for x in st])
print("total", round(st.sum() / 1e6))
PY
