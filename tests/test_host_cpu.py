"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/volt_hip.h
declares, argument validation is reachable without a device, host logic (sharding, synthetic data,
parameter order) behaves, and the product path refuses to run without the GPU."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from volt_amd import _lib
    from volt_amd.build import build_lib
    build_lib()                                             # hipcc cross-compiles gfx950 without a GPU
    handle = _lib.lib()
    header = open(os.path.join(ROOT, "include", "volt_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(volt_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in volt_hip.h but not exported"
        assert name in _lib.EXPORTS, f"{name} has no ctypes signature in volt_amd/_lib.py"
    assert handle.volt_abi_version() == _lib.ABI_VERSION
    assert handle.volt_padded_n(1) == 128 and handle.volt_padded_n(4096) == 4096 and handle.volt_padded_n(4097) == 4224


def test_argument_validation_without_a_device():
    from volt_amd import _lib
    L = _lib.lib()
    assert L.volt_fill_f32(None, None, 1, 8, 8, 64, None) == -1
    assert L.volt_cumtrapz_f32(1, 8, 1, 0, 1, 1, 1, 1, None) == -7            # N < 2: x[1]-x[0] undefined
    assert L.volt_potrf_f32(1, 1, 1, 1, 100, None) == -5                      # Np not a multiple of 128
    assert L.volt_mll_workspace_bytes(64, 4096, 1) > L.volt_mll_workspace_bytes(64, 4096, 0) > 0
    assert L.volt_rollout_scratch_bytes(2, 3, 4) == 2 * 3 * 16 * 4
    with pytest.raises(_lib.VoltHipError):
        _lib.check(-3, "x")


def test_product_path_has_no_cpu_fallback():
    from volt_amd import ops
    from volt_amd._lib import VoltHipError
    with pytest.raises(VoltHipError):
        ops.fill(torch.zeros(8))
    with pytest.raises(VoltHipError):
        ops.cumtrapz(torch.ones(8), torch.arange(8.))
    from volt_amd.kernels import VolatilityKernel
    with pytest.raises(VoltHipError):
        VolatilityKernel().forward(torch.arange(8.).unsqueeze(-1), torch.ones(8, 1))


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "volt_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle/", "").lower() or f == "synthetic.py" or \
                    "import oracle" not in src and "from oracle" not in src, f
                assert "from oracle" not in src and "import oracle" not in src, f"{f} imports the oracle"


def test_shard_range_partitions_contiguously():
    from volt_amd.distributed import shard_range
    for total, world in ((256, 8), (64, 8), (10, 4), (3, 8)):
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def test_synthetic_series_are_reproducible_and_rank_shardable():
    from volt_amd.synthetic import sde_batch, sde_series
    x, F, V = sde_batch(4, 64, seed=2019)
    _, F2, V2 = sde_batch(2, 64, seed=2019, first=2)
    assert np.array_equal(F[2:], F2) and np.array_equal(V[2:], V2)
    f, v = sde_series(64, 2019)
    assert np.array_equal(f, F[0]) and (F > 0).all() and (V >= 1e-3).all()
    assert x.dtype == np.float32 and abs(x[1] - 1 / 252.) < 1e-9


def test_parameter_order_matches_positional_grad_flags():
    """train_utils.py:226-227 freezes parameters by POSITION: the likelihood noise must come first, then
    the mean parameters, then the vol model's."""
    from volt_amd import gp
    from volt_amd.means import LogLinearMean

    class M(gp.ExactGP):
        def __init__(self):
            super().__init__(torch.arange(4.), torch.zeros(4), gp.GaussianLikelihood())
            self.mean_module = gp.ConstantMean()
            self.vol_lh = gp.GaussianLikelihood()
    m = M()
    names = [n for n, _ in m.named_parameters()]
    assert names[0] == "likelihood.noise_covar.raw_noise" and names[1] == "mean_module.constant"
    m.mean_module = LogLinearMean(1)
    names = [n for n, _ in m.named_parameters()]
    assert names[:3] == ["likelihood.noise_covar.raw_noise", "mean_module.weights", "mean_module.bias"]
    lh = gp.GaussianLikelihood()
    lh.raw_noise.data = torch.tensor([1e-5])
    assert abs(float(lh.noise) - 0.6932522) < 1e-6
    lh.noise = 0.25
    assert abs(float(lh.noise) - 0.25) < 1e-6
    assert m.train_inputs[0].shape == (4, 1)                     # 1-D inputs stored as [N,1] like gpytorch


def test_install_as_voltron_alias():
    import importlib
    import sys
    import volt_amd
    saved = {k: v for k, v in sys.modules.items() if k == "voltron" or k.startswith("voltron.")}
    try:
        volt_amd.install_as_voltron()
        assert importlib.import_module("voltron.kernels").VolatilityKernel is volt_amd.VolatilityKernel
        assert importlib.import_module("voltron.rollout_utils").Rollouts is volt_amd.Rollouts
    finally:
        for k in [k for k in sys.modules if k == "voltron" or k.startswith("voltron.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_gpcv_stage_refuses_cpu_tensors_and_validates_arguments():
    """LearnGPCV has no CPU path either; its C entry validates before touching the device."""
    from volt_amd import _lib
    from volt_amd.train_utils import LearnGPCV
    L = _lib.lib()
    assert L.volt_gpcv_workspace_bytes(2, 300, 1) > L.volt_gpcv_workspace_bytes(2, 300, 0) > L.volt_mll_workspace_bytes(2, 300, 1)
    assert L.volt_gpcv_step_f32(None, 8, 64, 1e-3, *([None] * 6), 75, 1e-6, 1e-3, 1.0, 1.0, *([None] * 7), 1, 8, None) == -1
    assert L.volt_gemm_nt_f32(256, 128, 0, 0, 256, 128, 0, 0, 256, 128, 0, 0, 1.0, 0.0, 1, 100, 128, 128, None) == -16
    assert L.volt_mll_grad_k_f32(None, None, None, None, 1, 8, None) == -1
    assert L.volt_rollout_shared_f32(*([None] * 8), 1, 1, 4, 5, 0, 0.5, None) == -1
    x = torch.arange(50, dtype=torch.float32) / 252
    with pytest.raises(_lib.VoltHipError):
        LearnGPCV(x, torch.rand(51) + 1.0, train_iters=1)
    from volt_amd.train_utils import TrainBasicModel
    with pytest.raises(_lib.VoltHipError):
        TrainBasicModel(x, torch.rand(50) + 1.0, train_iters=1)
