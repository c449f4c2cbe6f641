"""Baseline GPs with stationary kernels (Matern / RBF / spectral mixture) and their training loop -- NOT part of the
product package.  SURVEY 2 rows 7, 10 and 13 mark voltron/models/BasicGPModels.py, TrainBasicModel and these kernels out
of scope; what IS in scope is ``nonvol_rollouts``, which needs some GP with a sample-independent
kernel to roll out.  These classes are that GP for the tests (tests/test_gpu_baselines.py); like everything else they
compute on libvolt_hip.so through volt_amd.ops / volt_amd.gp."""
