"""SpMV of ONE rank's share of BASELINE config 5 (400^3 cube cut into 8 z-slabs; here slab 3: 48 M elements, 7.96 M owned +
159 201 ghost rows, z-neighbour 159 201 rows away) alone on the GPU: the relative-row-group form with 32-bit gaps
against the int32 row form it replaces there.   python tools/probe_slab_spmv.py > out.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pfemfort_amd as pf   # noqa: E402
from pfemfort_amd import host as H   # noqa: E402

n, parts, part = 400, 8, 3
sz = H.box_slab_sizes(n, n, n, 0, 1, parts, part)
s = pf.PetscSolver().initialise(sz["size_local"], sz["size_global"], row_start=sz["row_start"])
s.generateBoxMesh(pf.POISSON_TET, -1.0, 1.0, n, -1.0, 1.0, n, -1.0, 1.0, n, bc_mode=0, nparts=parts, part=part)
s.buildPattern()
s.assemble(H.POISSON_ELEMDATA, H.TIMEDATA)
info = s.matrixInfo()
alg = 12 * info["nnz"] + 20 * info["n_local"]
out = {"slab": f"{part} of {parts} of the {n}^3 cube", "rows_local": info["n_local"], "rows_owned": info["n_owned"], "nnz_local": info["nnz"],
       "algorithmic_bytes": alg, "forms": {}}
for fmt in ("auto", "int32", "auto", "int32"):
    s.setSpmvFormat(fmt)
    ms = s.benchSpmv(50)
    out["forms"].setdefault(fmt, []).append({"rows_per_lane": s.spmvRowGroup(), "column_bits": s.spmvColumnBits(), "us": round(ms * 1e3, 1),
                                             "effective_GBps": round(alg / ms / 1e6), "form_bytes": s.spmvFormatBytes(),
                                             "form_GBps": round(s.spmvFormatBytes() / ms / 1e6)})
print(json.dumps(out))
