#!/usr/bin/env python3
"""Golden vectors for VolatilityKernel.forward(..., last_dim_is_batch=True) with and without ``diag``
(voltron/kernels/VolKernel.py:24-26,35-40 -- the branch the reference marks "TODO: check this"), made by executing the
reference's own file behind make_golden.py's stand-ins.  Writes fill_ldb.npz.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ldb.py"""
import importlib.util
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(OUT, "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)


def main():
    mg._install_standins()
    VK = mg._load("ref_volkernel", "kernels/VolKernel.py")
    kern = VK.VolatilityKernel()
    g = torch.Generator().manual_seed(404)
    out = {}
    for tag, (n, d) in {"n12d3": (12, 3), "n9d9": (9, 9), "n5d8": (5, 8), "n130d2": (130, 2)}.items():
        x = torch.arange(n) / 252.
        vol = torch.rand(n, d, generator=g) * 0.3 + 0.1                 # [N, D]: the LAST dim is the batch
        out[f"{tag}_x"], out[f"{tag}_vol"] = x.numpy(), vol.numpy()
        out[f"{tag}_K"] = kern.forward(x, vol, last_dim_is_batch=True).numpy()                   # [N, N, D]
        out[f"{tag}_diag"] = kern.forward(x, vol, diag=True, last_dim_is_batch=True).numpy()     # [N, min(N, D)]
    np.savez_compressed(os.path.join(OUT, "fill_ldb.npz"), **out)
    print("fill_ldb.npz", {k: v.shape for k, v in out.items() if k.endswith(("_K", "_diag"))})


if __name__ == "__main__":
    main()
