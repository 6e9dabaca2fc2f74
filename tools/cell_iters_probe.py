import sys; sys.path.insert(0, ".")
import numpy as np
from pilot_amd import engine
from pilot_amd.synthetic import make_cell_clouds
for cells in (1000, 2000, 5000):
    for reg in (0.5, 0.2, 0.1):
        X, offs, scale = make_cell_clouds(3, cells, 30, seed=cells + int(100 * reg))
        W, inf = engine.cell_w2_grid(X, offs, scale, reg, return_info=True)
        print(cells, reg, inf["iters"].tolist(), inf["err"].max())
